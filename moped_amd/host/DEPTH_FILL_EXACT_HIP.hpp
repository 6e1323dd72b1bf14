// DEPTH_FILL_EXACT_HIP -- moped3d only: drop-in for DEPTH_FILL_EXACT_CPU
// (moped3d/libmoped/src/depthfill/DEPTH_FILL_EXACT_CPU.hpp, config.hpp:39):
//     pipeline.addAlg( "DEPTHFILL", new DEPTH_FILL_EXACT_HIP( 8, false ) );
//     pipeline.addAlg( "DEPTHFILL", new DEPTH_FILL_EXACT_CPU( 8, false ) );   // fallback
// Same constructor arguments (scaleFactor, doBilinearInterpolation).  For every IMAGE_TYPE_DEPTH_MAP image of the
// frame: its holes (z < 0) are filled in place and a distance map named "<name>.distance" (IMAGE_TYPE_PROB_MAP, one
// Float per pixel) is appended to frameData.images (:416-428) -- the map DEPTHMAP_PROP_CPU and CLUSTER_LINKAGE look
// up by that name.  Bit-identical to the CPU step (mh_depth_fill, include/moped_hip.h).  A map the device fill cannot
// take (its downscaled size exceeds 8192 pixels: 640 x 480 needs a factor >= 8; or any device error) makes this step
// say so and declare itself NOT capable: the pipeline's getAlgs(true) then hands the slot to the next algorithm
// registered under "DEPTHFILL" -- the CPU step of the second line above -- from the next frame on (util.hpp:151-159);
// the frame at hand keeps its holes and gets no distance map, exactly what a frame without this step would have.
#pragma once
#include "hip_session.hpp"

namespace MopedNS {

class DEPTH_FILL_EXACT_HIP : public MopedAlg {
  int scaleFactor;
  bool doBilinearInterpolation;

 public:
  DEPTH_FILL_EXACT_HIP(int scaleFactor, bool doBilinearInterpolation)
      : scaleFactor(scaleFactor), doBilinearInterpolation(doBilinearInterpolation) {
    capable = HipSession::get() != 0;
  }

  void getConfig(map<string, string>& config) const {
    hipGetConfig(config, _stepName, _alg, "DEPTH_FILL_EXACT_HIP", "scaleFactor", scaleFactor);
  }
  void setConfig(map<string, string>& config) {
    if (hipSetConfig(config, _stepName, _alg, "DEPTH_FILL_EXACT_HIP", "scaleFactor", scaleFactor)) configUpdated = true;
  }

  void process(FrameData& frameData) {
    mh_ctx* ctx = HipSession::get();
    const int n_images = (int)frameData.images.size();   // (the loop appends)
    for (int i = 0; i < n_images; ++i) {
      SP_Image image = frameData.images[i];
      if (image->imageType != IMAGE_TYPE_DEPTH_MAP) continue;
      const int w = image->width, h = image->height;
      if (w <= 0 || h <= 0 || image->data.size() < (size_t)w * h * 4 * sizeof(Float)) continue;
      SP_Image distance(new Image);
      distance->imageType = IMAGE_TYPE_PROB_MAP;
      distance->width = w;
      distance->height = h;
      distance->name = image->name + ".distance";
      distance->data.resize((size_t)w * h * sizeof(Float));
      float K[4];
      for (int j = 0; j < 4; ++j) K[j] = image->intrinsicLinearCalibration[j];
      if (mh_depth_fill_host(ctx, (float*)&image->data[0], w, h, scaleFactor, doBilinearInterpolation ? 1 : 0, K,
                             (float*)&distance->data[0], 0) != MH_OK) {
        HipSession::warn("mh_depth_fill_host");
        capable = false;   // the CPU step registered behind this one takes the slot from the next frame on
        continue;
      }
      frameData.images.push_back(distance);
    }
  }
};

}  // namespace MopedNS
