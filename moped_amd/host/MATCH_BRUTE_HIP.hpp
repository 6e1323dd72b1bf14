// MATCH_BRUTE_HIP -- drop-in for MATCH_ANN_CPU / MATCH_FLANN_CPU
// (src/match/MATCH_ANN_CPU.hpp, src/match/MATCH_FLANN_CPU.hpp): exact 2-NN +
// ratio test on the GPU.  Wire it BEFORE the CPU matcher under the same step name:
//     pipeline.addAlg( "MATCH_SIFT", new MATCH_BRUTE_HIP( 128, "SIFT", 0.8 ) );
//     pipeline.addAlg( "MATCH_SIFT", new MATCH_ANN_CPU( 128, "SIFT", 5., 0.8 ) );   // fallback
// Contract kept: reads detectedFeatures[DescriptorType], L2-normalises model and
// query descriptors IN PLACE (MATCH_ANN_CPU.hpp:94,157), resizes matches to
// models->size() and appends Match{imageIdx, coord2D, coord3D} per accepted query
// in ascending query order (:165-176).  Silent return on empty input (:140,143).
#pragma once
#include "hip_session.hpp"

namespace MopedNS {

class MATCH_BRUTE_HIP : public MopedAlg {
  int DescriptorSize;
  string DescriptorType;
  Float Ratio;
  bool skipCalculation;
  vector<int> correspModel;
  vector<Pt<3>*> correspFeat;
  vector<float> packed;

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
      : DescriptorSize(DescriptorSize), DescriptorType(DescriptorType), Ratio(Ratio), skipCalculation(true) {
    capable = (DescriptorSize == MH_DESC_DIM) && HipSession::get() != 0;
  }

  void getConfig(map<string, string>& config) const {
    hipGetConfig(config, _stepName, _alg, "MATCH_BRUTE_HIP", "DescriptorType", DescriptorType);
    hipGetConfig(config, _stepName, _alg, "MATCH_BRUTE_HIP", "DescriptorSize", DescriptorSize);
    hipGetConfig(config, _stepName, _alg, "MATCH_BRUTE_HIP", "Ratio", Ratio);
  }
  void setConfig(map<string, string>&) {}  // a no-op in the reference as well

  void process(FrameData& frameData) {
    if (configUpdated) Update();
    if (skipCalculation) return;
    vector<FrameData::DetectedFeature>& corresp = frameData.detectedFeatures[DescriptorType];
    if (corresp.empty()) return;
    vector<vector<FrameData::Match> >& matches = frameData.matches;
    matches.resize(models->size());
    const int Q = (int)corresp.size();
    packed.resize((size_t)Q * MH_DESC_DIM);
    for (int i = 0; i < Q; ++i)
      for (int j = 0; j < MH_DESC_DIM; ++j) packed[(size_t)i * MH_DESC_DIM + j] = corresp[i].descriptor[j];
    mh_ctx* ctx = HipSession::get();
    vector<int32_t> nn(Q);
    // norm() + the search in one call: one upload, one synchronisation
    if (mh_normalize_match(ctx, &packed[0], Q, Ratio, &nn[0], 0, 0, 0) != MH_OK) { HipSession::warn("mh_normalize_match"); return; }
    for (int i = 0; i < Q; ++i)  // the reference normalises the query descriptors in place (:157)
      for (int j = 0; j < MH_DESC_DIM; ++j) corresp[i].descriptor[j] = packed[(size_t)i * MH_DESC_DIM + j];
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
