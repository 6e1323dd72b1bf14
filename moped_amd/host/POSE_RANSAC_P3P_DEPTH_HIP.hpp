// POSE_RANSAC_P3P_DEPTH_HIP -- moped3d only: drop-in for
// POSE_RANSAC_LM_DIFF_BACKPROJECTION_DEPTH_CPU (the moped3d default, config.hpp:46,48)
// and POSE_RANSAC_LM_DIFF_REPROJECTION_DEPTH_CPU
// (moped3d/libmoped/src/pose/POSE_RANSAC_LM_DIFF_{BACKPROJECTION,REPROJECTION}_DEPTH_CPU.hpp):
//     pipeline.addAlg( "POSE",  new POSE_RANSAC_P3P_DEPTH_HIP( 1024, 4, 5, 6, 8, 0.5 ) );   // was (192, 100, 4, 5, 6, 8, 0.5)
//     pipeline.addAlg( "POSE2", new POSE_RANSAC_P3P_DEPTH_HIP( 1024, 4, 6, 8, 5, 0.5 ) );   // was (64, 250, 4, 6, 8, 5, 0.5)
// Arguments: NHypotheses, then the reference's MaxObjectsPerCluster, NPtsAlign,
// MinNPtsObject, ErrorThreshold, Alpha; optional Kind (MH_DEPTH_BACKPROJECTION default).
// Reads Match.depthData (filled by DEPTHMAP_PROP_CPU, which stays on the host):
// world3D = depthData.coord3D, cauchyWeight = 1/(1+(fillDistance/scale)^2) with the
// class's scale (0.1 back-projection, 25 reprojection; ...BACKPROJECTION_DEPTH_CPU.hpp:66,194-197).
#pragma once
#include "hip_session.hpp"

namespace MopedNS {

class POSE_RANSAC_P3P_DEPTH_HIP : public MopedAlg {
  int NHypotheses;
  int MaxObjectsPerCluster;
  int NPtsAlign;
  int MinNPtsObject;
  Float ErrorThreshold;
  Float Alpha;
  int Kind;
  unsigned long frameCounter;

 public:
  POSE_RANSAC_P3P_DEPTH_HIP(int NHypotheses, int MaxObjectsPerCluster, int NPtsAlign, int MinNPtsObject,
                            Float ErrorThreshold, Float Alpha, int Kind = MH_DEPTH_BACKPROJECTION)
      : NHypotheses(NHypotheses), MaxObjectsPerCluster(MaxObjectsPerCluster), NPtsAlign(NPtsAlign),
        MinNPtsObject(MinNPtsObject), ErrorThreshold(ErrorThreshold), Alpha(Alpha), Kind(Kind), frameCounter(0) {
    capable = HipSession::get() != 0;
  }

  void getConfig(map<string, string>& config) const {
    hipGetConfig(config, _stepName, _alg, "POSE_RANSAC_P3P_DEPTH_HIP", "NHypotheses", NHypotheses);
    hipGetConfig(config, _stepName, _alg, "POSE_RANSAC_P3P_DEPTH_HIP", "NPtsAlign", NPtsAlign);
    hipGetConfig(config, _stepName, _alg, "POSE_RANSAC_P3P_DEPTH_HIP", "MinNPtsObject", MinNPtsObject);
    hipGetConfig(config, _stepName, _alg, "POSE_RANSAC_P3P_DEPTH_HIP", "ErrorThreshold", ErrorThreshold);
  }
  void setConfig(map<string, string>&) {}

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
    const Float scale = (Kind == MH_DEPTH_BACKPROJECTION) ? 0.100 : 25.0;  // FillInCauchyScale of the class
    for (int img = 0; img < (int)frameData.images.size(); ++img) {
      vector<mh_corr> corr;
      vector<mh_depth> depth;
      vector<int32_t> off(1, 0);
      vector<int> clModel;
      for (int model = 0; model < (int)frameData.clusters.size(); ++model)
        for (int c = 0; c < (int)frameData.clusters[model].size(); ++c) {
          const FrameData::Cluster& cl = frameData.clusters[model][c];
          if (cl.empty() || frameData.matches[model][cl.front()].imageIdx != img) continue;
          for (FrameData::Cluster::const_iterator it = cl.begin(); it != cl.end(); ++it) {
            const FrameData::Match& m = frameData.matches[model][*it];
            mh_corr k;
            k.u = m.coord2D[0]; k.v = m.coord2D[1];
            k.x = m.coord3D[0]; k.y = m.coord3D[1]; k.z = m.coord3D[2];
            corr.push_back(k);
            mh_depth d;
            d.wx = m.depthData.coord3D[0]; d.wy = m.depthData.coord3D[1]; d.wz = m.depthData.coord3D[2];
            const Float factor = m.depthData.fillDistance / scale;   // getCauchyWeight
            d.w = 1.0 / (1 + factor * factor);
            depth.push_back(d);
          }
          off.push_back((int32_t)corr.size());
          clModel.push_back(model);
        }
      const int ncl = (int)clModel.size();
      if (ncl == 0) continue;
      const Image& im = *frameData.images[img];
      mh_cam cam;
      for (int i = 0; i < 4; ++i) cam.K[i] = im.intrinsicLinearCalibration[i];
      for (int i = 0; i < 4; ++i) cam.cam[i] = im.cameraPose.rotation[i];
      for (int i = 0; i < 3; ++i) cam.cam[4 + i] = im.cameraPose.translation[i];
      vector<mh_pose_out> out((size_t)ncl * MaxObjectsPerCluster);
      int32_t nout = 0;
      if (mh_pose_ransac_depth(ctx, &corr[0], &depth[0], &off[0], ncl, &cam, &prm, Kind, Alpha,
                               (uint64_t)frameCounter * 2654435761ul + _alg, &out[0], &nout) != MH_OK) {
        HipSession::warn("mh_pose_ransac_depth");
        continue;
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
    if (_stepName == "POSE") frameData.oldObjects = *frameData.objects;
  }
};

}  // namespace MopedNS
