// FRAME_RESIDENT_HIP -- ONE step for the whole hot path of a frame: what MATCH_SIFT, CLUSTER, POSE, FILTER, POSE2
// and FILTER2 do one after the other on a FrameData (the loop body of MopedPimpl::processImages, src/moped.cpp:183-191,
// with the step list of src/config.hpp:83-120) in one call of the C ABI, mh_frame_run_host: the frame's features go
// up once, the chain of kernels runs stream-ordered on the device, the objects come back -- instead of six steps that
// each carry their inputs and outputs over PCIe and synchronise.  A maintainer who does not need the intermediate
// lists on the host replaces the six addAlg lines by
//     pipeline.addAlg( "MATCH_SIFT", new FRAME_RESIDENT_HIP( 128, "SIFT", 0.8,  200, 20, 7, 100,
//                                                            1024, 4, 5, 6, 10,  5, 4096., 2,
//                                                            1024, 4, 6, 8, 5,   7, 4096., 3 ) );
// (the constructor takes the six reference constructors' arguments in pipeline order) and keeps the step-by-step
// plugins (MATCH_BRUTE_HIP .. FILTER_PROJECTION_HIP) for pipelines that read matches / clusters in between.
// Contract kept: reads detectedFeatures[DescriptorType] (all images of the frame), L2-normalises the query descriptors
// in place (MATCH_ANN_CPU.hpp:157), appends the frame's final objects {model, pose, score} to *frameData.objects in
// FILTER2's list order; frameData.matches is sized to models->size() and left empty unless FillMatches is set (setConfig:
// the frame's match lists are then copied back for a display / bookkeeping step behind this one), clusters stay empty
// (they never leave the device).  capable = false without a gfx950 device, as for every HIP step.
#pragma once
#include "hip_session.hpp"

#ifndef MH_PACK_THREADS
#define MH_PACK_THREADS 8   // threads of the loops that pack / unpack the frame's descriptors (OpenMP builds)
#endif

namespace MopedNS {

class FRAME_RESIDENT_HIP : public MopedAlg {
  int DescriptorSize;
  string DescriptorType;
  mh_frame_params prm;
  bool skipCalculation;
  unsigned long frameCounter;
  int FillMatches;
  // the frame's descriptors / keypoints / image indices as the C ABI takes them: ONE page-locked block (mh_host_alloc),
  // grown when a frame has more features -- packing into pageable vectors cost the frame two staged 1.5 MB copies
  void* pinBlock;
  size_t pinBytes;
  float *packed, *uv;
  int32_t* imageOf;
  vector<int32_t> matchQuery, matchModel;

  bool pinFor(mh_ctx* ctx, int Q) {
    const size_t need = (size_t)Q * (MH_DESC_DIM + 2) * sizeof(float) + (size_t)Q * sizeof(int32_t);
    if (need > pinBytes) {
      if (pinBlock) mh_host_free(ctx, pinBlock);
      pinBlock = NULL;
      pinBytes = 0;
      const size_t want = need + need / 4;
      if (mh_host_alloc(ctx, want, &pinBlock) != MH_OK) return false;
      pinBytes = want;
    }
    const size_t cap = pinBytes / ((MH_DESC_DIM + 2) * sizeof(float) + sizeof(int32_t));
    packed = (float*)pinBlock;
    uv = packed + cap * MH_DESC_DIM;
    imageOf = (int32_t*)(uv + cap * 2);
    return true;
  }

  void Update() {
    skipCalculation = true;
    mh_ctx* ctx = HipSession::get();
    size_t n = 0;
    for (size_t m = 0; m < models->size(); ++m) n += (*models)[m]->IPs[DescriptorType].size();
    vector<float> desc(n * MH_DESC_DIM), xyz(n * 3);
    vector<int32_t> owner(n);
    size_t x = 0;
    for (size_t m = 0; m < models->size(); ++m) {
      vector<Model::IP>& ips = (*models)[m]->IPs[DescriptorType];
      for (size_t f = 0; f < ips.size(); ++f, ++x) {
        owner[x] = (int32_t)m;
        for (int i = 0; i < MH_DESC_DIM; ++i) desc[x * MH_DESC_DIM + i] = ips[f].descriptor[i];
        for (int i = 0; i < 3; ++i) xyz[x * 3 + i] = ips[f].coord3D[i];
      }
    }
    if (n > 1) {
      // Update() normalises the model descriptors in place (MATCH_ANN_CPU.hpp:94)
      if (mh_normalize(ctx, &desc[0], (int)n) != MH_OK) { HipSession::warn("mh_normalize"); return; }
      x = 0;
      for (size_t m = 0; m < models->size(); ++m) {
        vector<Model::IP>& ips = (*models)[m]->IPs[DescriptorType];
        for (size_t f = 0; f < ips.size(); ++f, ++x)
          for (int i = 0; i < MH_DESC_DIM; ++i) ips[f].descriptor[i] = desc[x * MH_DESC_DIM + i];
      }
      if (mh_db_upload(ctx, &desc[0], &owner[0], &xyz[0], (int)n, (int)models->size(), 0) != MH_OK) {
        HipSession::warn("mh_db_upload");
        return;
      }
      skipCalculation = false;
    }
    configUpdated = false;
  }

 public:
  FRAME_RESIDENT_HIP(int DescriptorSize, string DescriptorType, Float Ratio,                                  // MATCH
                     Float Radius, Float Merge, int MinPts, int MaxIterations,                               // CLUSTER
                     int NHyp1, int MaxObj1, int NPtsAlign1, int MinNPts1, Float ErrorThreshold1,            // POSE
                     int MinPoints1, Float FeatureDistance1, Float MinScore1,                                // FILTER
                     int NHyp2, int MaxObj2, int NPtsAlign2, int MinNPts2, Float ErrorThreshold2,            // POSE2
                     int MinPoints2, Float FeatureDistance2, Float MinScore2)                                // FILTER2
      : DescriptorSize(DescriptorSize), DescriptorType(DescriptorType), skipCalculation(true), frameCounter(0), FillMatches(0),
        pinBlock(NULL), pinBytes(0), packed(NULL), uv(NULL), imageOf(NULL) {
    mh_frame_default_params(&prm);
    prm.ratio = (float)Ratio;
    prm.ms_radius = (float)Radius;
    prm.ms_merge = (float)Merge;
    prm.ms_min_pts = MinPts;
    prm.ms_max_iter = MaxIterations;
    prm.pose1.n_hypotheses = NHyp1;
    prm.pose1.max_objects_per_cluster = MaxObj1;
    prm.pose1.n_pts_align = NPtsAlign1;
    prm.pose1.min_n_pts_object = MinNPts1;
    prm.pose1.error_threshold = (float)ErrorThreshold1;
    prm.f1_min_points = MinPoints1;
    prm.f1_feature_distance = (float)FeatureDistance1;
    prm.f1_min_score = (float)MinScore1;
    prm.pose2.n_hypotheses = NHyp2;
    prm.pose2.max_objects_per_cluster = MaxObj2;
    prm.pose2.n_pts_align = NPtsAlign2;
    prm.pose2.min_n_pts_object = MinNPts2;
    prm.pose2.error_threshold = (float)ErrorThreshold2;
    prm.f2_min_points = MinPoints2;
    prm.f2_feature_distance = (float)FeatureDistance2;
    prm.f2_min_score = (float)MinScore2;
    prm.run_stage2 = 1;
    capable = (DescriptorSize == MH_DESC_DIM) && HipSession::get() != 0;
  }

  // Every constant the six reference steps publish (GET_CONFIG in MATCH_ANN_CPU.hpp:122-125, CLUSTER_MEAN_SHIFT_CPU.hpp:168-171,
  // POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:248-252, FILTER_PROJECTION_CPU.hpp:68-70), under the reference's names; the second
  // POSE / FILTER stage's carry a 2.  NHypotheses stands where the reference has MaxRANSACTests / MaxLMTests.
  // FillMatches = 1: frameData.matches is filled from the device's match lists after the frame (one more copy; for a
  // STATUS_DISPLAY-like step behind this one), 0 (default): sized and left empty.
#define MH_FR_CONFIG(OP)                                                          \
  OP("Ratio", prm.ratio)                                                          \
  OP("Radius", prm.ms_radius) OP("Merge", prm.ms_merge) OP("MinPts", prm.ms_min_pts) OP("MaxIterations", prm.ms_max_iter) \
  OP("NHypotheses", prm.pose1.n_hypotheses) OP("MaxObjectsPerCluster", prm.pose1.max_objects_per_cluster)                 \
  OP("NPtsAlign", prm.pose1.n_pts_align) OP("MinNPtsObject", prm.pose1.min_n_pts_object)                                  \
  OP("ErrorThreshold", prm.pose1.error_threshold)                                                                         \
  OP("MinPoints", prm.f1_min_points) OP("FeatureDistance", prm.f1_feature_distance) OP("MinScore", prm.f1_min_score)      \
  OP("NHypotheses2", prm.pose2.n_hypotheses) OP("MaxObjectsPerCluster2", prm.pose2.max_objects_per_cluster)               \
  OP("NPtsAlign2", prm.pose2.n_pts_align) OP("MinNPtsObject2", prm.pose2.min_n_pts_object)                                \
  OP("ErrorThreshold2", prm.pose2.error_threshold)                                                                        \
  OP("MinPoints2", prm.f2_min_points) OP("FeatureDistance2", prm.f2_feature_distance) OP("MinScore2", prm.f2_min_score)   \
  OP("FillMatches", FillMatches)
  void getConfig(map<string, string>& config) const {
    hipGetConfig(config, _stepName, _alg, "FRAME_RESIDENT_HIP", "DescriptorType", DescriptorType);
    hipGetConfig(config, _stepName, _alg, "FRAME_RESIDENT_HIP", "DescriptorSize", DescriptorSize);
#define MH_FR_GET(NAME, VAR) hipGetConfig(config, _stepName, _alg, "FRAME_RESIDENT_HIP", NAME, VAR);
    MH_FR_CONFIG(MH_FR_GET)
#undef MH_FR_GET
  }
  void setConfig(map<string, string>& config) {
#define MH_FR_SET(NAME, VAR) hipSetConfig(config, _stepName, _alg, "FRAME_RESIDENT_HIP", NAME, VAR);
    MH_FR_CONFIG(MH_FR_SET)
#undef MH_FR_SET
  }
#undef MH_FR_CONFIG

  void process(FrameData& frameData) {
    if (configUpdated) Update();
    ++frameCounter;
    if (skipCalculation) return;
    vector<FrameData::DetectedFeature>& feats = frameData.detectedFeatures[DescriptorType];
    if (feats.empty()) return;
    frameData.matches.resize(models->size());
    const int Q = (int)feats.size();
    // the cameras the features refer to, renumbered in image order (as HipCameraTable does for matches)
    vector<int> local(frameData.images.size(), -1);
    for (int i = 0; i < Q; ++i) {
      const int im = feats[i].imageIdx;
      if (im < 0 || im >= (int)local.size()) {
        std::clog << "[moped_hip] FRAME_RESIDENT_HIP: a feature refers to an image outside FrameData::images: frame skipped" << std::endl;
        return;
      }
      local[im] = 0;
    }
    vector<mh_cam> cams;
    for (size_t i = 0; i < local.size(); ++i) {
      if (local[i] < 0) continue;
      local[i] = (int)cams.size();
      const Image& im = *frameData.images[i];
      mh_cam c;
      for (int j = 0; j < 4; ++j) c.K[j] = im.intrinsicLinearCalibration[j];
      for (int j = 0; j < 4; ++j) c.cam[j] = im.cameraPose.rotation[j];
      for (int j = 0; j < 3; ++j) c.cam[4 + j] = im.cameraPose.translation[j];
      cams.push_back(c);
    }
    if (cams.empty() || (int)cams.size() > MH_MAX_IMAGES) {
      std::clog << "[moped_hip] FRAME_RESIDENT_HIP: the frame's features refer to " << cams.size() << " images (1.."
                << MH_MAX_IMAGES << " supported): frame skipped" << std::endl;
      return;
    }
    mh_ctx* ctx = HipSession::get();
    if (!pinFor(ctx, Q)) { HipSession::warn("mh_host_alloc"); return; }
    // (3 000 features = 3 000 separately allocated descriptors: ~0.1 ms on one thread -- the reference's steps use OpenMP
    //  for their own loops over the features, MATCH_ANN_CPU.hpp:160)
    #pragma omp parallel for num_threads(MH_PACK_THREADS) schedule(static)
    for (int i = 0; i < Q; ++i) {
      for (int j = 0; j < MH_DESC_DIM; ++j) packed[(size_t)i * MH_DESC_DIM + j] = feats[i].descriptor[j];
      uv[2 * i] = feats[i].coord2D[0];
      uv[2 * i + 1] = feats[i].coord2D[1];
      imageOf[i] = local[feats[i].imageIdx];
    }
    vector<mh_object> out(256);
    int32_t n = 0, counts[4];
    // upload + the whole chain of launches; the normalised descriptors are back right after the frame's first kernel and go
    // into detectedFeatures (MATCH_ANN_CPU.hpp:157 normalises them in place) WHILE the device clusters and poses
    int rc = mh_frame_run_host_begin(ctx, &packed[0], &uv[0], &imageOf[0], Q, &cams[0], (int)cams.size(), &prm,
                                     (uint64_t)frameCounter * 2654435761ul + _alg, 1);
    if (rc == MH_OK) rc = mh_frame_wait_descriptors(ctx);
    if (rc != MH_OK) { HipSession::warn("mh_frame_run_host_begin"); return; }
    #pragma omp parallel for num_threads(MH_PACK_THREADS) schedule(static)
    for (int i = 0; i < Q; ++i)
      for (int j = 0; j < MH_DESC_DIM; ++j) feats[i].descriptor[j] = packed[(size_t)i * MH_DESC_DIM + j];
    rc = mh_frame_fetch(ctx, &out[0], (int)out.size(), &n, counts);
    if (rc == MH_OK && n > (int)out.size()) {   // more objects than the first guess: fetch again into a block that holds them
      out.resize(n);
      rc = mh_frame_fetch(ctx, &out[0], (int)out.size(), &n, counts);
    }
    if (rc != MH_OK) { HipSession::warn("mh_frame_fetch"); return; }
    if (FillMatches) {
      // frameData.matches as MATCH_ANN_CPU::process leaves it (MATCH_ANN_CPU.hpp:165-176): per model, in ascending
      // query order, {imageIdx, coord2D, coord3D of the nearest model point} -- from the device's lists of the frame
      matchQuery.resize(Q);
      matchModel.resize(Q);
      int32_t nm = 0;
      vector<mh_corr> pts(Q);
      if (mh_frame_fetch_matches(ctx, &matchQuery[0], &matchModel[0], Q, &nm) == MH_OK &&
          mh_frame_fetch_match_points(ctx, &pts[0], Q, &nm) == MH_OK) {
        for (int k = 0; k < nm && k < Q; ++k) {
          const int q = matchQuery[k], m = matchModel[k];
          if (q < 0 || q >= Q || m < 0 || m >= (int)models->size()) continue;
          FrameData::Match mt;
          mt.imageIdx = feats[q].imageIdx;
          mt.coord2D = feats[q].coord2D;
          mt.coord3D.init(pts[k].x, pts[k].y, pts[k].z);
          frameData.matches[m].push_back(mt);
        }
      }
    }
    for (int o = 0; o < n && o < (int)out.size(); ++o) {
      if (out[o].model < 0 || out[o].model >= (int)models->size()) continue;
      SP_Object obj(new Object);
      frameData.objects->push_back(obj);
      obj->pose.rotation.init(out[o].pose[0], out[o].pose[1], out[o].pose[2], out[o].pose[3]);
      obj->pose.translation.init(out[o].pose[4], out[o].pose[5], out[o].pose[6]);
      obj->model = (*models)[out[o].model];
      obj->score = out[o].score;
    }
  }
};

}  // namespace MopedNS
