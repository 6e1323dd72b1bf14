// POSE_RANSAC_P3P_HIP -- drop-in for POSE_RANSAC_LM_DIFF_REPROJECTION_CPU
// (src/pose/POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp), for both the POSE and the
// POSE2 slot:
//     pipeline.addAlg( "POSE",  new POSE_RANSAC_P3P_HIP( 1024, 4, 5, 6, 10 ) );   // was (600, 200, 4, 5, 6, 10)
//     pipeline.addAlg( "POSE2", new POSE_RANSAC_P3P_HIP( 1024, 4, 6, 8, 5 ) );    // was (100, 500, 4, 6, 8, 5)
// Arguments: NHypotheses (P3P hypotheses per replica; replaces MaxRANSACTests /
// MaxLMTests, which have no meaning for a best-of-N search), then the reference's
// MaxObjectsPerCluster, NPtsAlign, MinNPtsObject, ErrorThreshold.  Appends one
// Object{model, pose} per successful (cluster, replica) to *frameData.objects in
// task order (:275-303) and sets oldObjects when named "POSE" (:306).
#pragma once
#include "hip_session.hpp"

namespace MopedNS {

class POSE_RANSAC_P3P_HIP : public MopedAlg {
  int NHypotheses;
  int MaxObjectsPerCluster;
  int NPtsAlign;
  int MinNPtsObject;
  Float ErrorThreshold;
  unsigned long frameCounter;

 public:
  POSE_RANSAC_P3P_HIP(int NHypotheses, int MaxObjectsPerCluster, int NPtsAlign, int MinNPtsObject,
                      Float ErrorThreshold)
      : NHypotheses(NHypotheses), MaxObjectsPerCluster(MaxObjectsPerCluster), NPtsAlign(NPtsAlign),
        MinNPtsObject(MinNPtsObject), ErrorThreshold(ErrorThreshold), frameCounter(0) {
    capable = HipSession::get() != 0;
  }

  void getConfig(map<string, string>& config) const {
    hipGetConfig(config, _stepName, _alg, "POSE_RANSAC_P3P_HIP", "NHypotheses", NHypotheses);
    hipGetConfig(config, _stepName, _alg, "POSE_RANSAC_P3P_HIP", "NPtsAlign", NPtsAlign);
    hipGetConfig(config, _stepName, _alg, "POSE_RANSAC_P3P_HIP", "MinNPtsObject", MinNPtsObject);
    hipGetConfig(config, _stepName, _alg, "POSE_RANSAC_P3P_HIP", "ErrorThreshold", ErrorThreshold);
  }
  void setConfig(map<string, string>&) {}

  // POSE / POSE2 on the clusters the HIP step before left on the device (HipHandover): after CLUSTER_MEAN_SHIFT_HIP the
  // first stage, after FILTER_PROJECTION_HIP the second (its random streams keyed like the one-call frame's:
  // mh_frame_run_host, seed ^ 0x5DEECE66D).  false = not taken.
  bool processResident(FrameData& frameData, mh_ctx* ctx, const mh_pose_params& prm, uint64_t seed) {
    HipHandover& ho = HipHandover::get();
    int which = 0;
    if (ho.at(1, frameData) && frameData.objects->empty()) which = 1;
    else if (ho.at(3, frameData) && ho.objectsTag == HipHandover::tagObjects(frameData)) which = 2;
    if (!which || MaxObjectsPerCluster < 1) return false;
    if (ho.matchesTag != HipHandover::tagMatches(frameData) || ho.clustersTag != HipHandover::tagClusters(frameData)) return false;
    size_t ncl = 0;
    for (size_t m = 0; m < frameData.clusters.size(); ++m) ncl += frameData.clusters[m].size();
    const int cap = (int)ncl * MaxObjectsPerCluster + 1;
    vector<mh_step_object> out(cap);
    int32_t nout = 0;
    if (mh_step_pose(ctx, which, &prm, which == 1 ? seed : seed ^ 0x5DEECE66Dull, &out[0], cap, &nout) != MH_OK) {
      HipSession::warn("mh_step_pose");
      return false;
    }
    for (int o = 0; o < nout; ++o) {
      if (out[o].model < 0 || out[o].model >= (int)models->size()) continue;
      SP_Object obj(new Object);
      frameData.objects->push_back(obj);
      obj->pose.rotation.init(out[o].pose[0], out[o].pose[1], out[o].pose[2], out[o].pose[3]);
      obj->pose.translation.init(out[o].pose[4], out[o].pose[5], out[o].pose[6]);
      obj->model = (*models)[out[o].model];
      obj->score = 0;
    }
    ho.stage = which == 1 ? 2 : 4;
    ++ho.taken;
    ho.objectsTag = HipHandover::tagObjects(frameData);
    return true;
  }

  void process(FrameData& frameData) {
    mh_ctx* ctx = HipSession::get();
    ++frameCounter;
    mh_pose_params prm;
    prm.n_hypotheses = NHypotheses;
    prm.max_objects_per_cluster = MaxObjectsPerCluster;
    prm.n_pts_align = NPtsAlign;
    prm.min_n_pts_object = MinNPtsObject;
    prm.error_threshold = ErrorThreshold;
    prm.lm_iters_l2 = 2;
    prm.lm_iters_l4 = 10;
    if (processResident(frameData, ctx, prm, (uint64_t)frameCounter * 2654435761ul + _alg)) {
      if (_stepName == "POSE") frameData.oldObjects = *frameData.objects;
      return;
    }
    HipHandover::get().drop();
    // every cluster of the frame in ONE launch, in the reference's task order (model, cluster) (:275-303); each
    // correspondence carries its own image (LmData::image, :228-237): CLUSTER's clusters live in one image,
    // FILTER's (the input of POSE2) mix images
    const HipCameraTable table(frameData);
    if (table.ok) {
      vector<mh_corr> corr;
      vector<int32_t> imageOf;
      vector<int32_t> off(1, 0);
      vector<int> clModel;
      for (int model = 0; model < (int)frameData.clusters.size(); ++model)
        for (int c = 0; c < (int)frameData.clusters[model].size(); ++c) {
          const FrameData::Cluster& cl = frameData.clusters[model][c];
          if (cl.empty()) continue;
          for (FrameData::Cluster::const_iterator it = cl.begin(); it != cl.end(); ++it) {
            const FrameData::Match& m = frameData.matches[model][*it];
            mh_corr k;
            k.u = m.coord2D[0]; k.v = m.coord2D[1];
            k.x = m.coord3D[0]; k.y = m.coord3D[1]; k.z = m.coord3D[2];
            corr.push_back(k);
            imageOf.push_back(table.local[m.imageIdx]);
          }
          off.push_back((int32_t)corr.size());
          clModel.push_back(model);
        }
      const int ncl = (int)clModel.size();
      if (ncl > 0) {
        vector<mh_pose_out> out((size_t)ncl * MaxObjectsPerCluster);
        int32_t nout = 0;
        if (mh_pose_ransac_images(ctx, &corr[0], &imageOf[0], &off[0], ncl, &table.cams[0], (int)table.cams.size(),
                                  &prm, (uint64_t)frameCounter * 2654435761ul + _alg, &out[0], &nout) != MH_OK) {
          HipSession::warn("mh_pose_ransac_images");
          nout = 0;
        }
        for (int o = 0; o < nout; ++o) {
          SP_Object obj(new Object);
          frameData.objects->push_back(obj);
          obj->pose.rotation.init(out[o].pose[0], out[o].pose[1], out[o].pose[2], out[o].pose[3]);
          obj->pose.translation.init(out[o].pose[4], out[o].pose[5], out[o].pose[6]);
          obj->model = (*models)[clModel[out[o].cluster]];
          obj->score = 0;
        }
      }
    }
    if (_stepName == "POSE") frameData.oldObjects = *frameData.objects;
  }
};

}  // namespace MopedNS
