// MATCH_ADAPTIVE_BRUTE_HIP -- moped3d only: drop-in for MATCH_ADAPTIVE_FLANN_CPU
// (moped3d/libmoped/src/match/MATCH_ADAPTIVE_FLANN_CPU.hpp, config.hpp:43):
//     pipeline.addAlg( "MATCH_SIFT", new MATCH_ADAPTIVE_BRUTE_HIP( 128, "SIFT", 0.6, 0.75, 0.65, 0.8, 150, 50 ) );
//     pipeline.addAlg( "MATCH_SIFT", new MATCH_ADAPTIVE_FLANN_CPU( 128, "SIFT", 8, 0.6, 0.75, 0.65, 0.8, 150, 50 ) );  // fallback
// (NumTrees has no meaning for the exact search and is dropped.)  The exact 2-NN search runs on
// the GPU (mh_match with the ratio test off); the ratio each feature has to pass depends on the
// depth under it and on the model of its nearest neighbour, as in the reference:
//   Update  (:100-177)  per model: the depths at which its bounding box projects to DimensionPeak
//                       / DimensionFade pixels (binary search, :318-357) and the ratio bounds from
//                       its feature count (sigmoid, :165-171)
//   process (:447-470)  features deeper than MaximumDepth are skipped; ratio = Cauchy-weighted mix
//                       of getRatio(depth) and getRatio(DefaultDepth) (:361-376)
// The device-resident frame does the same inside its group kernel (mh_frame_set_depth_rules with
// this class's table(), which is what Update computes).
#pragma once
#include <cmath>

#include "hip_session.hpp"

namespace MopedNS {

class MATCH_ADAPTIVE_BRUTE_HIP : public MopedAlg {
  int DescriptorSize;
  string DescriptorType;
  Float MinRatioMin, MinRatioMax, MaxRatioMin, MaxRatioMax;
  Float DimensionPeak, DimensionFade;
  Float MaximumDepth, DefaultDepth, CauchyScale;
  bool skipCalculation;
  vector<int> correspModel;
  vector<Pt<3>*> correspFeat;
  vector<float> packed;
  vector<float> controlPoints;   // per model: maxRatioDepth, minRatioDepth, ratioLow, ratioHigh

  // sqrt of the image area the largest face of the (centred) bounding box covers at `depth` (:262-312)
  static Float projectedLength(const Pt<3> box[2], const Pt<4>& k, Float depth) {
    const Float xr = box[1][0] - box[0][0], yr = box[1][1] - box[0][1], zr = box[1][2] - box[0][2];
    const Float lo[3] = {-xr / 2, -yr / 2, -zr / 2}, hi[3] = {xr / 2, yr / 2, zr / 2};
    const Float zc = (hi[0] - lo[0]) * (hi[1] - lo[1]), yc = (hi[0] - lo[0]) * (hi[2] - lo[2]),
                xc = (hi[1] - lo[1]) * (hi[2] - lo[2]);
    int a = 1, b = 2;                                   // face with X constant
    if (zc >= xc && zc >= yc) { a = 0; b = 1; }         // Z constant
    else if (yc >= xc && yc >= zc) { a = 0; b = 2; }    // Y constant
    const Float cx[4] = {lo[a], lo[a], hi[a], hi[a]}, cy[4] = {lo[b], hi[b], hi[b], lo[b]};
    Float umin = 0, umax = 0, vmin = 0, vmax = 0;
    for (int i = 0; i < 4; ++i) {                       // getProjectedArea (:238-256)
      const Float u = (k[0] * cx[i] + k[2] * depth) / depth, v = (k[1] * cy[i] + k[3] * depth) / depth;
      if (i == 0 || u < umin) umin = u;
      if (i == 0 || u > umax) umax = u;
      if (i == 0 || v < vmin) vmin = v;
      if (i == 0 || v > vmax) vmax = v;
    }
    return std::sqrt((umax - umin) * (vmax - vmin));
  }
  // depth at which the box is about targetLength pixels long (:318-357; one shared iteration budget)
  static Float solveDepth(const Pt<3> box[2], const Pt<4>& k, Float targetLength, int iters, Float tolerance) {
    Float left = 0, right = 2;
    int iter = 0;
    while (iter < iters) {
      ++iter;
      if (projectedLength(box, k, right) > targetLength) right *= 2; else break;
    }
    const Float maxError = targetLength * tolerance;
    while (iter < iters) {
      ++iter;
      const Float middle = (left + right) / 2, length = projectedLength(box, k, middle);
      if (std::fabs(length - targetLength) < maxError) return middle;
      if (length > targetLength) left = middle; else right = middle;
    }
    return (left + right) / 2;
  }
  Float getRatio(Float depth, int m) const {             // :193-215
    if (depth > MaximumDepth) return 0;
    const Float maxRatioDepth = controlPoints[4 * m], minRatioDepth = controlPoints[4 * m + 1],
                ratioLow = controlPoints[4 * m + 2], ratioHigh = controlPoints[4 * m + 3];
    if (depth < maxRatioDepth) return ratioLow + (depth / maxRatioDepth) * (ratioHigh - ratioLow);
    if (depth < minRatioDepth) return ratioHigh;
    if (depth < minRatioDepth * 2) return ((minRatioDepth * 2 - depth) / minRatioDepth) * ratioHigh;
    return 0;
  }

  void Update(FrameData& frameData) {
    skipCalculation = true;
    MaximumDepth = 4.0;
    DefaultDepth = 1.0;
    CauchyScale = 0.1;
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
      if (mh_normalize(ctx, &packed[0], (int)n) != MH_OK) { HipSession::warn("mh_normalize"); return; }
      x = 0;
      for (size_t m = 0; m < models->size(); ++m) {       // Update() normalises the model descriptors in place (:122)
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
    // intrinsics of the gray image (:146-153)
    Pt<4> k;
    k.init(0.f, 0.f, 0.f, 0.f);
    for (size_t i = 0; i < frameData.images.size(); ++i)
      if (frameData.images[i]->imageType == IMAGE_TYPE_GRAY_IMAGE) k = frameData.images[i]->intrinsicLinearCalibration;
    controlPoints.assign(4 * models->size(), 0.f);
    const Float sigmoidTranslate = 1750, sigmoidScale = 250;
    for (size_t m = 0; m < models->size(); ++m) {
      Model& model = *(*models)[m];
      const Float featureCount = (int)model.IPs[DescriptorType].size();
      const Float densityAdjust = 1.0 / (1.0 + std::exp(-1.0 * ((sigmoidTranslate - featureCount) / sigmoidScale)));
      controlPoints[4 * m] = solveDepth(model.boundingBox, k, DimensionPeak, 100, 0.01);
      controlPoints[4 * m + 1] = solveDepth(model.boundingBox, k, DimensionFade, 100, 0.01);
      controlPoints[4 * m + 2] = MinRatioMin + densityAdjust * (MinRatioMax - MinRatioMin);
      controlPoints[4 * m + 3] = MaxRatioMin + densityAdjust * (MaxRatioMax - MaxRatioMin);
    }
    configUpdated = false;
  }

 public:
  MATCH_ADAPTIVE_BRUTE_HIP(int DescriptorSize, string DescriptorType, Float MinRatioMin, Float MinRatioMax,
                           Float MaxRatioMin, Float MaxRatioMax, Float DimensionPeak, Float DimensionFade)
      : DescriptorSize(DescriptorSize), DescriptorType(DescriptorType), MinRatioMin(MinRatioMin),
        MinRatioMax(MinRatioMax), MaxRatioMin(MaxRatioMin), MaxRatioMax(MaxRatioMax), DimensionPeak(DimensionPeak),
        DimensionFade(DimensionFade), MaximumDepth(4.0), DefaultDepth(1.0), CauchyScale(0.1), skipCalculation(true) {
    capable = (DescriptorSize == MH_DESC_DIM) && HipSession::get() != 0;
  }

  // what mh_frame_set_depth_rules takes as ratio_table (valid after the first frame)
  const vector<float>& table() const { return controlPoints; }

  void getConfig(map<string, string>& config) const {
    hipGetConfig(config, _stepName, _alg, "MATCH_ADAPTIVE_BRUTE_HIP", "MinRatioMin", MinRatioMin);
    hipGetConfig(config, _stepName, _alg, "MATCH_ADAPTIVE_BRUTE_HIP", "MinRatioMax", MinRatioMax);
    hipGetConfig(config, _stepName, _alg, "MATCH_ADAPTIVE_BRUTE_HIP", "MaxRatioMin", MaxRatioMin);
    hipGetConfig(config, _stepName, _alg, "MATCH_ADAPTIVE_BRUTE_HIP", "MaxRatioMax", MaxRatioMax);
    hipGetConfig(config, _stepName, _alg, "MATCH_ADAPTIVE_BRUTE_HIP", "DimensionPeak", DimensionPeak);
    hipGetConfig(config, _stepName, _alg, "MATCH_ADAPTIVE_BRUTE_HIP", "DimensionFade", DimensionFade);
    hipGetConfig(config, _stepName, _alg, "MATCH_ADAPTIVE_BRUTE_HIP", "DescriptorType", DescriptorType);
    hipGetConfig(config, _stepName, _alg, "MATCH_ADAPTIVE_BRUTE_HIP", "DescriptorSize", DescriptorSize);
  }
  void setConfig(map<string, string>&) {}

  void process(FrameData& frameData) {
    if (configUpdated) Update(frameData);
    if (skipCalculation) return;
    vector<FrameData::DetectedFeature>& corresp = frameData.detectedFeatures[DescriptorType];
    if (corresp.empty()) return;
    // the depth map and its ".distance" map (:425-441)
    Image* depthmap = 0;
    Image* distanceMap = 0;
    for (size_t i = 0; i < frameData.images.size(); ++i)
      if (frameData.images[i]->imageType == IMAGE_TYPE_DEPTH_MAP) depthmap = frameData.images[i].get();
    if (!depthmap) return;
    for (size_t i = 0; i < frameData.images.size(); ++i)
      if (frameData.images[i]->imageType == IMAGE_TYPE_PROB_MAP && frameData.images[i]->name == depthmap->name + ".distance") {
        distanceMap = frameData.images[i].get();
        break;
      }
    vector<vector<FrameData::Match> >& matches = frameData.matches;
    matches.resize(models->size());
    const int Q = (int)corresp.size();
    packed.resize((size_t)Q * MH_DESC_DIM);
    for (int i = 0; i < Q; ++i)
      for (int j = 0; j < MH_DESC_DIM; ++j) packed[(size_t)i * MH_DESC_DIM + j] = corresp[i].descriptor[j];
    mh_ctx* ctx = HipSession::get();
    vector<int32_t> acc(Q), nn(Q);
    vector<float> d1(Q), d2(Q);
    // norm() + the search in one call; ratio 2: every query with a neighbour passes, the decision is taken below per feature
    if (mh_normalize_match(ctx, &packed[0], Q, 2.f, &acc[0], &nn[0], &d1[0], &d2[0]) != MH_OK) { HipSession::warn("mh_normalize_match"); return; }
    for (int i = 0; i < Q; ++i)   // the reference normalises the query descriptors in place (:453)
      for (int j = 0; j < MH_DESC_DIM; ++j) corresp[i].descriptor[j] = packed[(size_t)i * MH_DESC_DIM + j];
    for (int i = 0; i < Q; ++i) {
      if (nn[i] < 0) continue;
      int x = (int)corresp[i].coord2D[0], y = (int)corresp[i].coord2D[1];
      x = x < 0 ? 0 : (x >= depthmap->width ? depthmap->width - 1 : x);    // the reference clamps to [0, width]
      y = y < 0 ? 0 : (y >= depthmap->height ? depthmap->height - 1 : y);
      const Float depth = depthmap->getDepth(x, y);
      if (depth > MaximumDepth) continue;                                   // :457-460
      const int m = correspModel[nn[i]];
      const Float weightTerm = (distanceMap ? distanceMap->getProb(x, y) : (Float)0) / CauchyScale;   // :362-364
      const Float weight = 1.0 / (1.0 + weightTerm * weightTerm);
      const Float Ratio = weight * getRatio(depth, m) + (1.0 - weight) * getRatio(DefaultDepth, m);
      if (d1[i] / d2[i] < Ratio) {
        if (matches[m].capacity() < 1000) matches[m].reserve(1000);
        matches[m].resize(matches[m].size() + 1);
        FrameData::Match& match = matches[m].back();
        match.imageIdx = corresp[i].imageIdx;
        match.coord2D = corresp[i].coord2D;
        match.coord3D = *correspFeat[nn[i]];
      }
    }
  }
};

}  // namespace MopedNS
