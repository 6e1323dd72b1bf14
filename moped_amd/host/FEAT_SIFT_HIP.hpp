// FEAT_SIFT_HIP -- drop-in for FEAT_SIFT_CPU (src/feat/FEAT_SIFT_CPU.hpp:54-113; the
// reference's own GPU variant is FEAT_SIFT_GPU over SiftGPU/GLSL).  Wire it BEFORE the
// CPU extractor under the same step name (config.hpp:69):
//     pipeline.addAlg( "SIFT", new FEAT_SIFT_HIP( "-1" ) );
//     pipeline.addAlg( "SIFT", new FEAT_SIFT_CPU( "-1" ) );   // fallback
// Contract kept: for every image of the frame, appends DetectedFeature{imageIdx,
// coord2D = (col, row), descriptor[128]} to detectedFeatures[_stepName] in
// libsiftfast's keypoint-list order (:95-108).  ScaleOrigin "-1" doubles the image
// first (DoubleImSize, :70-75).  The reference's constructor leaves DoubleImSize at
// libsiftfast's default (1) until setConfig runs; so does this one.
#pragma once
#include "hip_session.hpp"

namespace MopedNS {

class FEAT_SIFT_HIP : public MopedAlg {
  string ScaleOrigin;
  int DoubleImSize;
  int Capacity;
  vector<float> xy, desc;

 public:
  FEAT_SIFT_HIP(string ScaleOrigin) : ScaleOrigin(ScaleOrigin), DoubleImSize(1), Capacity(8192) {
    capable = HipSession::get() != 0;
  }

  void getConfig(map<string, string>& config) const {
    hipGetConfig(config, _stepName, _alg, "FEAT_SIFT_HIP", "ScaleOrigin", ScaleOrigin);
  }
  void setConfig(map<string, string>& config) {
    map<string, string>::iterator it =
        config.find(_stepName + ":" + toString(_alg) + ":FEAT_SIFT_HIP/ScaleOrigin");
    if (it != config.end()) ScaleOrigin = it->second;
    DoubleImSize = (ScaleOrigin == "-1") ? 1 : 0;   // FEAT_SIFT_CPU.hpp:72-75
  }

  void process(FrameData& frameData) {
    mh_ctx* ctx = HipSession::get();
    for (int i = 0; i < (int)frameData.images.size(); i++) {
      Image* img = frameData.images[i].get();
      if (img->width <= 0 || img->height <= 0 || (int)img->data.size() < img->width * img->height) continue;
      vector<FrameData::DetectedFeature>& detectedFeatures = frameData.detectedFeatures[_stepName];
      int32_t n = 0;
      int rc;
      for (;;) {
        xy.resize((size_t)Capacity * 2);
        desc.resize((size_t)Capacity * MH_DESC_DIM);
        rc = mh_sift_extract(ctx, &img->data[0], img->width, img->height, DoubleImSize, &xy[0], 0, &desc[0],
                             Capacity, &n);
        if (rc != MH_ERR_CAPACITY || Capacity >= (1 << 20)) break;
        Capacity *= 2;   // more keypoints than room: grow and run the image again
      }
      if (rc != MH_OK) {
        HipSession::warn("mh_sift_extract");
        continue;
      }
      const size_t base = detectedFeatures.size();
      detectedFeatures.resize(base + n);
      for (int k = 0; k < n; ++k) {
        FrameData::DetectedFeature& f = detectedFeatures[base + k];
        f.imageIdx = i;
        f.descriptor.assign(desc.begin() + (size_t)k * MH_DESC_DIM, desc.begin() + (size_t)(k + 1) * MH_DESC_DIM);
        f.coord2D[0] = xy[2 * k];
        f.coord2D[1] = xy[2 * k + 1];
      }
    }
  }
};

}  // namespace MopedNS
