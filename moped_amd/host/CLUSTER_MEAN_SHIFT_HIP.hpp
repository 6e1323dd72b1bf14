// CLUSTER_MEAN_SHIFT_HIP -- drop-in for CLUSTER_MEAN_SHIFT_CPU
// (src/cluster/CLUSTER_MEAN_SHIFT_CPU.hpp):
//     pipeline.addAlg( "CLUSTER", new CLUSTER_MEAN_SHIFT_HIP( 200, 20, 7, 100 ) );
// Same constructor arguments and the same result: per model, per image, the
// clusters of match indices in the reference's emission and splice order (:80-158,
// :182-199); sets oldClusters when the step is named "CLUSTER" (:198).
#pragma once
#include "hip_session.hpp"

namespace MopedNS {

class CLUSTER_MEAN_SHIFT_HIP : public MopedAlg {
  float Radius;
  float Merge;
  int MinPts;
  int MaxIterations;

 public:
  CLUSTER_MEAN_SHIFT_HIP(float Radius, float Merge, unsigned int MinPts, unsigned int MaxIterations)
      : Radius(Radius), Merge(Merge), MinPts(MinPts), MaxIterations(MaxIterations) {
    capable = HipSession::get() != 0;
  }

  void getConfig(map<string, string>& config) const {
    hipGetConfig(config, _stepName, _alg, "CLUSTER_MEAN_SHIFT_HIP", "Radius", Radius);
    hipGetConfig(config, _stepName, _alg, "CLUSTER_MEAN_SHIFT_HIP", "Merge", Merge);
    hipGetConfig(config, _stepName, _alg, "CLUSTER_MEAN_SHIFT_HIP", "MinPts", MinPts);
    hipGetConfig(config, _stepName, _alg, "CLUSTER_MEAN_SHIFT_HIP", "MaxIterations", MaxIterations);
  }
  void setConfig(map<string, string>&) {}

  // CLUSTER on the lists MATCH_BRUTE_HIP left on the device (HipHandover); false = not taken
  bool processResident(FrameData& frameData, mh_ctx* ctx) {
    HipHandover& ho = HipHandover::get();
    if (!ho.at(0, frameData) || frameData.matches.size() != models->size()) return false;
    for (size_t m = 0; m < frameData.clusters.size(); ++m)
      if (!frameData.clusters[m].empty()) return false;
    if (ho.matchesTag != HipHandover::tagMatches(frameData)) return false;   // a step in between changed the lists
    size_t total = 0;
    for (size_t m = 0; m < frameData.matches.size(); ++m) total += frameData.matches[m].size();
    const int cap = (int)total + 1;
    vector<int32_t> clModel(cap), clOff(cap + 1), members(cap);
    int32_t ncl = 0;
    if (mh_step_cluster(ctx, Radius, Merge, MinPts, MaxIterations, &clModel[0], &clOff[0], &members[0], cap, cap, &ncl) != MH_OK) {
      HipSession::warn("mh_step_cluster");
      return false;
    }
    for (int c = 0; c < ncl; ++c) {
      vector<FrameData::Cluster>& dst = frameData.clusters[clModel[c]];
      dst.resize(dst.size() + 1);
      for (int j = clOff[c]; j < clOff[c + 1]; ++j) dst.back().push_back(members[j]);
    }
    ho.stage = 1;
    ++ho.taken;
    ho.clustersTag = HipHandover::tagClusters(frameData);
    return true;
  }

  void process(FrameData& frameData) {
    frameData.clusters.resize(models->size());
    mh_ctx* ctx = HipSession::get();
    if (processResident(frameData, ctx)) {
      if (_stepName == "CLUSTER") frameData.oldClusters = frameData.clusters;
      return;
    }
    HipHandover::get().drop();
    // every (model, image) point set of the frame goes to the device in ONE call
    vector<float> pts;
    vector<int> matchIdx;
    vector<int32_t> off(1, 0), owner;
    for (int model = 0; model < (int)frameData.matches.size(); ++model) {
      const vector<FrameData::Match>& mm = frameData.matches[model];
      for (int img = 0; img < (int)frameData.images.size(); ++img) {
        const size_t before = matchIdx.size();
        for (int k = 0; k < (int)mm.size(); ++k)
          if (mm[k].imageIdx == img) {
            pts.push_back(mm[k].coord2D[0]);
            pts.push_back(mm[k].coord2D[1]);
            matchIdx.push_back(k);
          }
        if (matchIdx.size() == before) continue;
        off.push_back((int32_t)matchIdx.size());
        owner.push_back(model);
      }
    }
    const int n_problems = (int)owner.size();
    if (n_problems > 0) {
      const int total = off[n_problems];
      vector<int32_t> label(total), order(total), ncl(n_problems);
      if (mh_meanshift_batch(ctx, &pts[0], &off[0], n_problems, 2, Radius, Merge, MinPts, MaxIterations,
                             &label[0], &order[0], &ncl[0]) != MH_OK) {
        HipSession::warn("mh_meanshift_batch");
      } else {
        for (int p = 0; p < n_problems; ++p) {
          const int b = off[p], n = off[p + 1] - b;
          int pos = 0;
          for (int c = 0; c < ncl[p]; ++c) {
            frameData.clusters[owner[p]].resize(frameData.clusters[owner[p]].size() + 1);
            FrameData::Cluster& cl = frameData.clusters[owner[p]].back();
            while (pos < n && order[b + pos] >= 0 && label[b + order[b + pos]] == c)
              cl.push_back(matchIdx[b + order[b + pos++]]);
          }
        }
      }
    }
    if (_stepName == "CLUSTER") frameData.oldClusters = frameData.clusters;
  }
};

}  // namespace MopedNS
