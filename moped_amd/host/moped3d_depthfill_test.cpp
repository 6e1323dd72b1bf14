// moped3d_depthfill_test -- the DEPTHFILL step plugin driven the way moped3d's pipeline drives it
// (moped3d/libmoped/src/config.hpp:39, src/moped.cpp: the active algorithm of every step runs on the frame):
//   moped3d_depthfill_test in.bin out.bin [scaleFactor=8] [bilinear=0]
// in.bin : int32 w, h; float32 K[4]; float32 depth[h][w][4] (x, y, z, norm; z < 0 = hole)
// out.bin: the frame's depth map after the step, then its "<name>.distance" map [h][w]
#define MOPED_AMD_WITH_DEPTH
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "moped_types.hpp"
#include "DEPTH_FILL_EXACT_HIP.hpp"

using namespace MopedNS;

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s in.bin out.bin [scaleFactor] [bilinear]\n", argv[0]);
    return 2;
  }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 2; }
  int32_t wh[2];
  float K[4];
  if (std::fread(wh, 4, 2, f) != 2 || std::fread(K, 4, 4, f) != 4) return 2;
  FrameData frameData;
  SP_Image gray(new Image);              // a moped3d frame: the camera image first, then its depth map
  gray->imageType = IMAGE_TYPE_GRAY_IMAGE;
  gray->name = "camera";
  gray->width = wh[0];
  gray->height = wh[1];
  frameData.images.push_back(gray);
  SP_Image depth(new Image);
  depth->imageType = IMAGE_TYPE_DEPTH_MAP;
  depth->name = "camera.depth";
  depth->width = wh[0];
  depth->height = wh[1];
  for (int j = 0; j < 4; ++j) depth->intrinsicLinearCalibration[j] = K[j];
  depth->data.resize((size_t)wh[0] * wh[1] * 4 * sizeof(Float));
  if (std::fread(&depth->data[0], 1, depth->data.size(), f) != depth->data.size()) return 2;
  std::fclose(f);
  frameData.images.push_back(depth);

  MopedPipeline pipeline;
  pipeline.addAlg("DEPTHFILL", new DEPTH_FILL_EXACT_HIP(argc > 3 ? std::atoi(argv[3]) : 8, argc > 4 && std::atoi(argv[4]) != 0));
  list<MopedAlg*> algs = pipeline.getAlgs(true);
  if (algs.empty()) {
    std::fprintf(stderr, "DEPTHFILL: no gfx950 device / HIP library -- not capable\n");
    return 3;
  }
  for (list<MopedAlg*>::iterator a = algs.begin(); a != algs.end(); ++a) (*a)->process(frameData);

  if (frameData.images.size() != 3 || frameData.images[2]->imageType != IMAGE_TYPE_PROB_MAP ||
      frameData.images[2]->name != "camera.depth.distance") {
    std::fprintf(stderr, "the step did not append the distance map\n");
    return 4;
  }
  FILE* o = std::fopen(argv[2], "wb");
  if (!o) { std::perror(argv[2]); return 2; }
  std::fwrite(&frameData.images[1]->data[0], 1, frameData.images[1]->data.size(), o);
  std::fwrite(&frameData.images[2]->data[0], 1, frameData.images[2]->data.size(), o);
  std::fclose(o);
  std::printf("DEPTHFILL %dx%d -> %s\n", wh[0], wh[1], frameData.images[2]->name.c_str());
  return 0;
}
