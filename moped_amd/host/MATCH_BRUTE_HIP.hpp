// MATCH_BRUTE_HIP -- drop-in for MATCH_ANN_CPU / MATCH_FLANN_CPU
// (src/match/MATCH_ANN_CPU.hpp, src/match/MATCH_FLANN_CPU.hpp): exact 2-NN +
// ratio test on the GPU.  Wire it BEFORE the CPU matcher under the same step name:
//     pipeline.addAlg( "MATCH_SIFT", new MATCH_BRUTE_HIP( 128, "SIFT", 0.8 ) );
//     pipeline.addAlg( "MATCH_SIFT", new MATCH_ANN_CPU( 128, "SIFT", 5., 0.8 ) );   // fallback
// Contract kept: reads detectedFeatures[DescriptorType], L2-normalises model and
// query descriptors IN PLACE (MATCH_ANN_CPU.hpp:94,157), resizes matches to
// models->size() and appends Match{imageIdx, coord2D, coord3D} per accepted query
// in ascending query order (:165-176).  Silent return on empty input (:140,143).
//
// Round 5: the frame's features are packed into ONE page-locked block (mh_host_alloc) on MH_PACK_THREADS threads and --
// where the frame has one camera and matches is empty before the step -- the search runs as the first slot of the
// device-resident hand-over (mh_step_match): the normalised descriptors come back on a stream of their own while the
// device still searches, the match lists stay on the device for CLUSTER_MEAN_SHIFT_HIP (HipHandover, hip_session.hpp)
// and come back once for frameData.matches.
#pragma once
#include "hip_session.hpp"

#ifndef MH_PACK_THREADS
#define MH_PACK_THREADS 8   // threads of the loops that pack / unpack the frame's descriptors (OpenMP builds)
#endif

namespace MopedNS {

class MATCH_BRUTE_HIP : public MopedAlg {
  int DescriptorSize;
  string DescriptorType;
  Float Ratio;
  bool skipCalculation;
  vector<int> correspModel;
  vector<Pt<3>*> correspFeat;
  vector<float> packed;
  // the frame's descriptors and keypoints as the C ABI takes them: one page-locked block, grown on demand
  void* pinBlock;
  size_t pinBytes;
  float *pinDesc, *pinUv;

  bool pinFor(mh_ctx* ctx, int Q) {
    const size_t need = (size_t)Q * (MH_DESC_DIM + 2) * sizeof(float);
    if (need > pinBytes) {
      if (pinBlock) mh_host_free(ctx, pinBlock);
      pinBlock = NULL;
      pinBytes = 0;
      const size_t want = need + need / 4;
      if (mh_host_alloc(ctx, want, &pinBlock) != MH_OK) return false;
      pinBytes = want;
    }
    const size_t cap = pinBytes / ((MH_DESC_DIM + 2) * sizeof(float));
    pinDesc = (float*)pinBlock;
    pinUv = pinDesc + cap * MH_DESC_DIM;
    return true;
  }

  // MATCH as the first slot of the resident frame; false = not taken (the caller runs the upload path)
  bool processResident(FrameData& frameData, vector<FrameData::DetectedFeature>& corresp, mh_ctx* ctx) {
    const int Q = (int)corresp.size();
    vector<vector<FrameData::Match> >& matches = frameData.matches;
    if (!HipHandover::enabled()) return false;
    const int img = corresp[0].imageIdx;
    if (img < 0 || img >= (int)frameData.images.size()) return false;
    for (int i = 1; i < Q; ++i)
      if (corresp[i].imageIdx != img) return false;       // several cameras: the upload path
    for (size_t m = 0; m < matches.size(); ++m)
      if (!matches[m].empty()) return false;              // another MATCH step has filled lists already
    if (!pinFor(ctx, Q)) return false;
    #pragma omp parallel for num_threads(MH_PACK_THREADS) schedule(static)
    for (int i = 0; i < Q; ++i) {
      for (int j = 0; j < MH_DESC_DIM; ++j) pinDesc[(size_t)i * MH_DESC_DIM + j] = corresp[i].descriptor[j];
      pinUv[2 * i] = corresp[i].coord2D[0];
      pinUv[2 * i + 1] = corresp[i].coord2D[1];
    }
    const Image& im = *frameData.images[img];
    mh_cam cam;
    for (int j = 0; j < 4; ++j) cam.K[j] = im.intrinsicLinearCalibration[j];
    for (int j = 0; j < 4; ++j) cam.cam[j] = im.cameraPose.rotation[j];
    for (int j = 0; j < 3; ++j) cam.cam[4 + j] = im.cameraPose.translation[j];
    if (mh_step_match(ctx, pinDesc, pinUv, Q, &cam, Ratio, 1) != MH_OK || mh_frame_wait_descriptors(ctx) != MH_OK) {
      HipSession::warn("mh_step_match");
      return false;
    }
    // the reference normalises the query descriptors in place (:157): written back while the device still searches
    #pragma omp parallel for num_threads(MH_PACK_THREADS) schedule(static)
    for (int i = 0; i < Q; ++i)
      for (int j = 0; j < MH_DESC_DIM; ++j) corresp[i].descriptor[j] = pinDesc[(size_t)i * MH_DESC_DIM + j];
    const int nm = (int)models->size();
    vector<int32_t> off(nm + 1), mq(Q);
    vector<mh_corr> pts(Q);
    int32_t M = 0;
    if (mh_step_match_fetch(ctx, &off[0], &mq[0], &pts[0], Q, &M) != MH_OK) {
      HipSession::warn("mh_step_match_fetch");
      return true;   // (the descriptors are normalised: searching again would not see the frame MATCH_ANN_CPU would; no matches)
    }
    // the slot's contract (:165-176): matches[model] in ascending query order -- the device's lists as they lie
    for (int m = 0; m < nm; ++m) {
      matches[m].resize(off[m + 1] - off[m]);
      for (int k = off[m]; k < off[m + 1]; ++k) {
        FrameData::Match& out = matches[m][k - off[m]];
        out.imageIdx = corresp[mq[k]].imageIdx;
        out.coord2D = corresp[mq[k]].coord2D;
        out.coord3D.init(pts[k].x, pts[k].y, pts[k].z);
      }
    }
    HipHandover& ho = HipHandover::get();
    ho.stage = 0;
    ++ho.taken;
    ho.frame = &frameData;
    ho.matchesTag = HipHandover::tagMatches(frameData);
    return true;
  }

  void Update() {
    skipCalculation = true;
    mh_ctx* ctx = HipSession::get();
    size_t n = 0;
    for (size_t m = 0; m < models->size(); ++m) n += (*models)[m]->IPs[DescriptorType].size();
    correspModel.resize(n);
    correspFeat.resize(n);
    packed.resize(n * MH_DESC_DIM);
    vector<float> xyz(n * 3);
    vector<int32_t> owner(n);
    size_t x = 0;
    for (size_t m = 0; m < models->size(); ++m) {
      vector<Model::IP>& ips = (*models)[m]->IPs[DescriptorType];
      for (size_t f = 0; f < ips.size(); ++f, ++x) {
        correspModel[x] = (int)m;
        correspFeat[x] = &ips[f].coord3D;
        owner[x] = (int32_t)m;
        for (int i = 0; i < MH_DESC_DIM; ++i) packed[x * MH_DESC_DIM + i] = ips[f].descriptor[i];
        for (int i = 0; i < 3; ++i) xyz[x * 3 + i] = ips[f].coord3D[i];
      }
    }
    if (n > 1) {
      // normalise on the device (bit-identical to norm(), :54-57) and write back,
      // as Update() mutates the model descriptors (:94)
      if (mh_normalize(ctx, &packed[0], (int)n) != MH_OK) { HipSession::warn("mh_normalize"); return; }
      x = 0;
      for (size_t m = 0; m < models->size(); ++m) {
        vector<Model::IP>& ips = (*models)[m]->IPs[DescriptorType];
        for (size_t f = 0; f < ips.size(); ++f, ++x)
          for (int i = 0; i < MH_DESC_DIM; ++i) ips[f].descriptor[i] = packed[x * MH_DESC_DIM + i];
      }
      if (mh_db_upload(ctx, &packed[0], &owner[0], &xyz[0], (int)n, (int)models->size(), 0) != MH_OK) {
        HipSession::warn("mh_db_upload");
        return;
      }
      skipCalculation = false;
    }
    configUpdated = false;
  }

 public:
  MATCH_BRUTE_HIP(int DescriptorSize, string DescriptorType, Float Ratio)
      : DescriptorSize(DescriptorSize), DescriptorType(DescriptorType), Ratio(Ratio), skipCalculation(true),
        pinBlock(NULL), pinBytes(0), pinDesc(NULL), pinUv(NULL) {
    capable = (DescriptorSize == MH_DESC_DIM) && HipSession::get() != 0;
  }

  void getConfig(map<string, string>& config) const {
    hipGetConfig(config, _stepName, _alg, "MATCH_BRUTE_HIP", "DescriptorType", DescriptorType);
    hipGetConfig(config, _stepName, _alg, "MATCH_BRUTE_HIP", "DescriptorSize", DescriptorSize);
    hipGetConfig(config, _stepName, _alg, "MATCH_BRUTE_HIP", "Ratio", Ratio);
  }
  void setConfig(map<string, string>&) {}  // a no-op in the reference as well

  void process(FrameData& frameData) {
    HipHandover::get().drop();   // MATCH begins a frame: whatever is resident belongs to an earlier one
    if (configUpdated) Update();
    if (skipCalculation) return;
    vector<FrameData::DetectedFeature>& corresp = frameData.detectedFeatures[DescriptorType];
    if (corresp.empty()) return;
    vector<vector<FrameData::Match> >& matches = frameData.matches;
    matches.resize(models->size());
    const int Q = (int)corresp.size();
    mh_ctx* ctx = HipSession::get();
    if (processResident(frameData, corresp, ctx)) return;
    // the upload path: norm() + the search in one call, the nearest rows back, the lists built here
    float* buf = pinFor(ctx, Q) ? pinDesc : (packed.resize((size_t)Q * MH_DESC_DIM), &packed[0]);
    #pragma omp parallel for num_threads(MH_PACK_THREADS) schedule(static)
    for (int i = 0; i < Q; ++i)
      for (int j = 0; j < MH_DESC_DIM; ++j) buf[(size_t)i * MH_DESC_DIM + j] = corresp[i].descriptor[j];
    vector<int32_t> nn(Q);
    if (mh_normalize_match(ctx, buf, Q, Ratio, &nn[0], 0, 0, 0) != MH_OK) { HipSession::warn("mh_normalize_match"); return; }
    #pragma omp parallel for num_threads(MH_PACK_THREADS) schedule(static)
    for (int i = 0; i < Q; ++i)  // the reference normalises the query descriptors in place (:157)
      for (int j = 0; j < MH_DESC_DIM; ++j) corresp[i].descriptor[j] = buf[(size_t)i * MH_DESC_DIM + j];
    // the slot's contract (:165-176): matches[model] in ascending query order.  Two passes:
    // count per model, size each list once, then fill.
    vector<size_t> fill(models->size(), 0);
    for (int i = 0; i < Q; ++i)
      if (nn[i] >= 0) ++fill[correspModel[nn[i]]];
    for (size_t m = 0; m < fill.size(); ++m) {
      const size_t had = matches[m].size();
      matches[m].resize(had + fill[m]);
      fill[m] = had;
    }
    for (int i = 0; i < Q; ++i) {
      if (nn[i] < 0) continue;
      FrameData::Match& out = matches[correspModel[nn[i]]][fill[correspModel[nn[i]]]++];
      out.imageIdx = corresp[i].imageIdx;
      out.coord2D = corresp[i].coord2D;
      out.coord3D = *correspFeat[nn[i]];
    }
  }
};

}  // namespace MopedNS
