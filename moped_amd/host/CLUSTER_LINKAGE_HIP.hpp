// CLUSTER_LINKAGE_HIP -- moped3d only: drop-in for CLUSTER_LINKAGE_CPU
// (moped3d/libmoped/src/cluster/CLUSTER_LINKAGE_CPU.hpp, config.hpp:45):
//     pipeline.addAlg( "CLUSTER", new CLUSTER_LINKAGE_HIP( 0.1, 7, 2, 1, 0.0, 1, -1, -1 ) );
//     pipeline.addAlg( "CLUSTER", new CLUSTER_LINKAGE_CPU( 0.1, 7, 2, 1, 0.0, 1, -1, -1 ) );   // fallback
// Same constructor arguments (WeightGamma and Alpha are unused by the reference as well: its
// adaptiveWeightSum call passes 0.5 and 25, :697).  LinkageType 0 (minimum), 1 (average: the shipped
// configuration) and 2 (maximum) as the reference's update loop has them (:506-526).
// Reads matches[model] (coord2D, coord3D, depthData.coord3D), the depth map and its ".distance"
// map (:577-593); writes clusters[model] in the reference's cluster and member order; sets
// oldClusters when the step is named "CLUSTER" (:703).
#pragma once
#include "hip_session.hpp"

namespace MopedNS {

class CLUSTER_LINKAGE_HIP : public MopedAlg {
  Float Cutoff;
  int MinPts;
  int Use3DFilter;
  Float WeightGamma;
  Float Alpha;
  int LinkageType;
  Float Sigma2D;
  Float Sigma3D;

 public:
  CLUSTER_LINKAGE_HIP(Float Cutoff, int MinPts, int Use3DFilter, Float WeightGamma, Float Alpha, int LinkageType,
                      Float Sigma2D, Float Sigma3D)
      : Cutoff(Cutoff), MinPts(MinPts), Use3DFilter(Use3DFilter), WeightGamma(WeightGamma), Alpha(Alpha),
        LinkageType(LinkageType), Sigma2D(Sigma2D), Sigma3D(Sigma3D) {
    capable = LinkageType >= 0 && LinkageType <= 2 && HipSession::get() != 0;
  }

  void getConfig(map<string, string>& config) const {
    hipGetConfig(config, _stepName, _alg, "CLUSTER_LINKAGE_HIP", "Cutoff", Cutoff);
    hipGetConfig(config, _stepName, _alg, "CLUSTER_LINKAGE_HIP", "MinPts", MinPts);
    hipGetConfig(config, _stepName, _alg, "CLUSTER_LINKAGE_HIP", "Use3DFilter", Use3DFilter);
    hipGetConfig(config, _stepName, _alg, "CLUSTER_LINKAGE_HIP", "WeightGamma", WeightGamma);
    hipGetConfig(config, _stepName, _alg, "CLUSTER_LINKAGE_HIP", "Alpha", Alpha);
    hipGetConfig(config, _stepName, _alg, "CLUSTER_LINKAGE_HIP", "LinkageType", LinkageType);
  }
  void setConfig(map<string, string>&) {}

  void process(FrameData& frameData) {
    frameData.clusters.resize(models->size());
    mh_ctx* ctx = HipSession::get();
    Image* depthmap = 0;
    Image* distanceMap = 0;
    for (size_t i = 0; i < frameData.images.size(); ++i)
      if (frameData.images[i]->imageType == IMAGE_TYPE_DEPTH_MAP) { depthmap = frameData.images[i].get(); break; }
    if (!depthmap) return;
    for (size_t i = 0; i < frameData.images.size(); ++i)
      if (frameData.images[i]->imageType == IMAGE_TYPE_PROB_MAP && frameData.images[i]->name == depthmap->name + ".distance") {
        distanceMap = frameData.images[i].get();
        break;
      }
    if (mh_frame_set_depth_image_host(ctx, (const float*)&depthmap->data[0], distanceMap ? (const float*)&distanceMap->data[0] : 0,
                                      depthmap->width, depthmap->height, MH_DEPTH_BACKPROJECTION, 0.5f, 0.1f) != MH_OK) {
      HipSession::warn("mh_frame_set_depth_image_host");
      return;
    }
    // every model's matches in ONE call
    vector<mh_corr> corr;
    vector<mh_depth> depth;
    vector<int32_t> off(1, 0);
    for (int model = 0; model < (int)frameData.matches.size(); ++model) {
      const vector<FrameData::Match>& mm = frameData.matches[model];
      for (int k = 0; k < (int)mm.size(); ++k) {
        mh_corr c;
        c.u = mm[k].coord2D[0]; c.v = mm[k].coord2D[1];
        c.x = mm[k].coord3D[0]; c.y = mm[k].coord3D[1]; c.z = mm[k].coord3D[2];
        mh_depth d;
        d.wx = mm[k].depthData.coord3D[0]; d.wy = mm[k].depthData.coord3D[1]; d.wz = mm[k].depthData.coord3D[2];
        d.w = 1.f;
        corr.push_back(c);
        depth.push_back(d);
      }
      off.push_back((int32_t)corr.size());
    }
    const int n_problems = (int)frameData.matches.size(), total = off[n_problems];
    if (total > 0) {
      mh_linkage_params prm;
      prm.cutoff = Cutoff;
      prm.min_pts = MinPts;
      prm.use3d_filter = Use3DFilter;
      prm.sigma2d = Sigma2D;
      prm.sigma3d = Sigma3D;
      prm.linkage_type = LinkageType;
      vector<int32_t> label(total), order(total), ncl(n_problems);
      if (mh_cluster_linkage(ctx, &corr[0], &depth[0], &off[0], n_problems, &prm, &label[0], &order[0], &ncl[0]) != MH_OK) {
        HipSession::warn("mh_cluster_linkage");
      } else {
        for (int p = 0; p < n_problems; ++p) {
          const int b = off[p], n = off[p + 1] - b;
          int pos = 0;
          for (int c = 0; c < ncl[p]; ++c) {
            frameData.clusters[p].resize(frameData.clusters[p].size() + 1);
            FrameData::Cluster& cl = frameData.clusters[p].back();
            while (pos < n && order[b + pos] >= 0 && label[b + order[b + pos]] == c) cl.push_back(order[b + pos++]);
          }
        }
      }
    }
    if (_stepName == "CLUSTER") frameData.oldClusters = frameData.clusters;
  }
};

}  // namespace MopedNS
