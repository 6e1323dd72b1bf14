// moped_hip_test -- stand-alone driver in the shape of moped2/moped_test.cpp
// (:154-161: the same frame N times, per-step times printed like STATUS_DISPLAY):
// builds a MopedPipeline with the HIP steps in the config.hpp slots, feeds it a
// scene file (models + one frame of detected features) and prints the objects.
//
//   moped_hip_test scene.bin [repeats]
//   moped_hip_test --images scene.bin [repeats]   (frame with several Images, second format below)
//   moped_hip_test --sift image.pgm      (FEAT step only: binary P5 image -> keypoints)
//
// Scene file (little endian, written by scripts/dump_scene.py):
//   int32 n_models, Q ; float K[4] ; float cam[7]
//   per model: int32 n_pts ; float xyz[n_pts][3] ; float desc[n_pts][128]
//   float q_uv[Q][2] ; float q_desc[Q][128]
// With --images (dump_scene.dump_images): FrameData::images holds n_images entries, cameras and -- as a moped3d
// frame does -- maps no feature refers to; every feature names its image:
//   int32 n_models, Q, n_images ; per image: int32 is_map ; float K[4] ; float cam[7]
//   models as above ; float q_uv[Q][2] ; float q_desc[Q][128] ; int32 q_image[Q]
#include <cstdio>
#include <ctime>
#include <iostream>

#include "moped_types.hpp"

#include "FEAT_SIFT_HIP.hpp"
#include "MATCH_BRUTE_HIP.hpp"
#include "CLUSTER_MEAN_SHIFT_HIP.hpp"
#include "POSE_RANSAC_P3P_HIP.hpp"
#include "FILTER_PROJECTION_HIP.hpp"

using namespace MopedNS;

// The pipeline slots of src/config.hpp:83-120 with the HIP steps in place.
static void createPipeline(MopedPipeline& pipeline) {
  pipeline.addAlg("MATCH_SIFT", new MATCH_BRUTE_HIP(128, "SIFT", 0.8));
  pipeline.addAlg("CLUSTER", new CLUSTER_MEAN_SHIFT_HIP(200, 20, 7, 100));
  pipeline.addAlg("POSE", new POSE_RANSAC_P3P_HIP(1024, 4, 5, 6, 10));
  pipeline.addAlg("FILTER", new FILTER_PROJECTION_HIP(5, 4096., 2));
  pipeline.addAlg("POSE2", new POSE_RANSAC_P3P_HIP(1024, 4, 6, 8, 5));
  pipeline.addAlg("FILTER2", new FILTER_PROJECTION_HIP(7, 4096., 3));
}

template <typename T>
static bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }

// FEAT slot alone (src/config.hpp:69): one 8-bit gray image through FEAT_SIFT_HIP::process.
static int run_sift(const char* path) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::perror(path); return 2; }
  int w = 0, h = 0, maxv = 0;
  if (std::fscanf(f, "P5 %d %d %d", &w, &h, &maxv) != 3 || maxv != 255 || w <= 0 || h <= 0) return 2;
  std::fgetc(f);
  SP_Image image(new Image);
  image->width = w;
  image->height = h;
  image->data.resize((size_t)w * h);
  if (!rd(f, &image->data[0], image->data.size())) return 2;
  std::fclose(f);
  MopedPipeline pipeline;
  pipeline.addAlg("SIFT", new FEAT_SIFT_HIP("-1"));
  list<MopedAlg*> algs = pipeline.getAlgs(true);
  if (algs.empty() || !algs.front()->isCapable()) return 3;
  list<SP_Object> objects;
  FrameData frameData;
  frameData.objects = &objects;
  frameData.images.push_back(image);
  algs.front()->process(frameData);
  const vector<FrameData::DetectedFeature>& feats = frameData.detectedFeatures["SIFT"];
  std::printf("KEYPOINTS %zu\n", feats.size());
  for (size_t i = 0; i < feats.size(); ++i) {
    double sum = 0;
    for (int k = 0; k < 128; ++k) sum += feats[i].descriptor[k] * (k + 1);
    std::printf("KP %d %.6f %.6f %.6f\n", feats[i].imageIdx, feats[i].coord2D[0], feats[i].coord2D[1], sum);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 3 && std::string(argv[1]) == "--sift") return run_sift(argv[2]);
  const bool multi = argc >= 3 && std::string(argv[1]) == "--images";
  if (multi) { --argc; ++argv; }
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s [--images] scene.bin [repeats]\n", argv[0]);
    return 2;
  }
  const int repeats = argc > 2 ? std::atoi(argv[2]) : 1;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 2; }
  int32_t nm = 0, Q = 0, n_images = 1;
  if (!rd(f, &nm, 1) || !rd(f, &Q, 1) || (multi && !rd(f, &n_images, 1)) || n_images < 1 || n_images > 64) return 2;
  vector<SP_Image> images;
  for (int i = 0; i < n_images; ++i) {
    int32_t is_map = 0;
    float K[4], cam[7];
    if ((multi && !rd(f, &is_map, 1)) || !rd(f, K, 4) || !rd(f, cam, 7)) return 2;
    SP_Image image(new Image);
    image->width = 640;
    image->height = 480;
    image->intrinsicLinearCalibration.init(K[0], K[1], K[2], K[3]);
    image->intrinsicNonlinearCalibration.init(0.f, 0.f, 0.f, 0.f);
    image->cameraPose.rotation.init(cam[0], cam[1], cam[2], cam[3]);
    image->cameraPose.translation.init(cam[4], cam[5], cam[6]);
    image->name = is_map ? "map" : "camera";
    images.push_back(image);
  }
  vector<SP_Model> models;
  for (int m = 0; m < nm; ++m) {
    int32_t n = 0;
    if (!rd(f, &n, 1)) return 2;
    vector<float> xyz((size_t)n * 3), desc((size_t)n * 128);
    if (!rd(f, &xyz[0], xyz.size()) || !rd(f, &desc[0], desc.size())) return 2;
    SP_Model model(new Model);
    model->name = "model" + toString(m);
    vector<Model::IP>& ips = model->IPs["SIFT"];
    ips.resize(n);
    for (int i = 0; i < n; ++i) {
      ips[i].coord3D.init(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
      ips[i].descriptor.assign(desc.begin() + (size_t)i * 128, desc.begin() + (size_t)(i + 1) * 128);
    }
    models.push_back(model);
  }
  vector<float> uv((size_t)Q * 2), qd((size_t)Q * 128);
  vector<int32_t> qimg(Q, 0);
  if (!rd(f, &uv[0], uv.size()) || !rd(f, &qd[0], qd.size()) || (multi && Q > 0 && !rd(f, &qimg[0], qimg.size())))
    return 2;
  std::fclose(f);

  MopedPipeline pipeline;
  createPipeline(pipeline);
  list<MopedAlg*> all = pipeline.getAlgs();
  for (list<MopedAlg*>::iterator a = all.begin(); a != all.end(); ++a) {
    if (!(*a)->isCapable()) {
      std::fprintf(stderr, "step %s: no gfx950 device / HIP library -- not capable\n", (*a)->_stepName.c_str());
      return 3;
    }
    (*a)->modelsUpdated(models);  // as MopedPimpl::addModel does (src/moped.cpp:94-99)
  }

  list<SP_Object> objects;
  map<string, double> total;
  for (int rep = 0; rep < repeats; ++rep) {
    objects.clear();
    FrameData frameData;
    frameData.objects = &objects;
    frameData.images = images;
    vector<FrameData::DetectedFeature>& feats = frameData.detectedFeatures["SIFT"];
    feats.resize(Q);
    for (int i = 0; i < Q; ++i) {
      feats[i].imageIdx = qimg[i];
      feats[i].coord2D.init(uv[2 * i], uv[2 * i + 1]);
      feats[i].descriptor.assign(qd.begin() + (size_t)i * 128, qd.begin() + (size_t)(i + 1) * 128);
    }
    // the per-frame loop of MopedPimpl::processImages (src/moped.cpp:180-191)
    list<MopedAlg*> algs = pipeline.getAlgs(true);
    for (list<MopedAlg*>::iterator a = algs.begin(); a != algs.end(); ++a) {
      struct timespec t0, t1;
      clock_gettime(CLOCK_REALTIME, &t0);
      (*a)->process(frameData);
      clock_gettime(CLOCK_REALTIME, &t1);
      const double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
      frameData.times[(*a)->_stepName] = (Float)dt;
      if (rep > 0 || repeats == 1) total[(*a)->_stepName] += dt;
    }
    if (rep == repeats - 1) {
      size_t nmatch = 0, ncl = 0;
      for (size_t m = 0; m < frameData.matches.size(); ++m) nmatch += frameData.matches[m].size();
      for (size_t m = 0; m < frameData.oldClusters.size(); ++m) ncl += frameData.oldClusters[m].size();
      std::printf("MATCHES %zu CLUSTERS %zu OBJECTS_AFTER_POSE %zu\n", nmatch, ncl, frameData.oldObjects.size());
    }
  }
  const int timed = repeats > 1 ? repeats - 1 : 1;
  for (map<string, double>::iterator t = total.begin(); t != total.end(); ++t)
    std::printf("TIME %s %.6f\n", t->first.c_str(), t->second / timed);
  for (list<SP_Object>::iterator o = objects.begin(); o != objects.end(); ++o)
    std::printf("OBJ %s %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.4f\n", (*o)->model->name.c_str(),
                (*o)->pose.translation[0], (*o)->pose.translation[1], (*o)->pose.translation[2],
                (*o)->pose.rotation[0], (*o)->pose.rotation[1], (*o)->pose.rotation[2], (*o)->pose.rotation[3],
                (*o)->score);
  return 0;
}
