// moped_hip_test -- stand-alone driver in the shape of moped2/moped_test.cpp
// (:154-161: the same frame N times, per-step times printed like STATUS_DISPLAY):
// builds a MopedPipeline with the HIP steps in the config.hpp slots, feeds it a
// scene file (models + one frame of detected features) and prints the objects.
//
//   moped_hip_test scene.bin [repeats]
//   moped_hip_test --images scene.bin [repeats]   (frame with several Images, second format below)
//   moped_hip_test --resident [--images] scene.bin [repeats]   (FRAME_RESIDENT_HIP: MATCH .. FILTER2 as one step)
//   moped_hip_test --sift image.pgm      (FEAT step only: binary P5 image -> keypoints)
//   moped_hip_test --world W scene.bin [repeats]
//       the model DB sharded over W ranks, device-resident frames through mh_frame_enqueue_sharded*: the frame
//       loop of MopedPimpl::processImages (src/moped.cpp:166-194) for a host that owns W GPUs.  W <= the number
//       of devices: one context per device, mh_comm_create_all (RCCL), one thread.  More ranks than devices (a
//       one-GPU box): W threads share device 0 over the host transport (mh_comm_create_host).
//
// Scene file (little endian, written by scripts/dump_scene.py):
//   int32 n_models, Q ; float K[4] ; float cam[7]
//   per model: int32 n_pts ; float xyz[n_pts][3] ; float desc[n_pts][128]
//   float q_uv[Q][2] ; float q_desc[Q][128]
// With --images (dump_scene.dump_images): FrameData::images holds n_images entries, cameras and -- as a moped3d
// frame does -- maps no feature refers to; every feature names its image:
//   int32 n_models, Q, n_images ; per image: int32 is_map ; float K[4] ; float cam[7]
//   models as above ; float q_uv[Q][2] ; float q_desc[Q][128] ; int32 q_image[Q]
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <iostream>
#include <mutex>
#include <thread>

#include "moped_types.hpp"

#include "FEAT_SIFT_HIP.hpp"
#include "MATCH_BRUTE_HIP.hpp"
#include "CLUSTER_MEAN_SHIFT_HIP.hpp"
#include "POSE_RANSAC_P3P_HIP.hpp"
#include "FILTER_PROJECTION_HIP.hpp"
#include "FRAME_RESIDENT_HIP.hpp"

using namespace MopedNS;

// The same slots' work as ONE step (FRAME_RESIDENT_HIP: the frame stays on the device between MATCH and FILTER2).
static void createResidentPipeline(MopedPipeline& pipeline) {
  pipeline.addAlg("MATCH_SIFT", new FRAME_RESIDENT_HIP(128, "SIFT", 0.8, 200, 20, 7, 100, 1024, 4, 5, 6, 10, 5, 4096., 2,
                                                       1024, 4, 6, 8, 5, 7, 4096., 3));
}

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

// ---- --world W -----------------------------------------------------------------------------------------------
struct FlatScene {
  int n_models, Q;
  vector<float> desc, xyz, uv, qd;
  vector<int32_t> model_of, first_row;   // first_row[n_models + 1]
  mh_cam cam;
};

// all-gather among threads of this process (the host transport's callback): every rank copies its block in, the
// last one to arrive releases the round
struct ThreadGather {
  int world;
  std::mutex mu;
  std::condition_variable cv;
  vector<unsigned char> board;
  int arrived, round, leaving;
  explicit ThreadGather(int w) : world(w), arrived(0), round(0), leaving(0) {}
  struct Rank { ThreadGather* g; int rank; };
  static int fn(void* user, const void* send, void* recv, size_t bytes) {
    Rank* r = static_cast<Rank*>(user);
    ThreadGather& g = *r->g;
    std::unique_lock<std::mutex> lock(g.mu);
    g.cv.wait(lock, [&] { return g.leaving == 0; });   // the previous round's readers are done with the board
    if (g.arrived == 0) g.board.assign(bytes * g.world, 0);
    if (g.board.size() != bytes * g.world) return 1;
    std::memcpy(&g.board[bytes * r->rank], send, bytes);
    const int my_round = g.round;
    if (++g.arrived == g.world) {
      g.arrived = 0;
      ++g.round;
      g.leaving = g.world;
      g.cv.notify_all();
    } else {
      g.cv.wait(lock, [&] { return g.round != my_round; });
    }
    std::memcpy(recv, &g.board[0], g.board.size());
    if (--g.leaving == 0) g.cv.notify_all();
    return 0;
  }
};

#define CHECK_MH(ctx, call)                                                                   \
  do {                                                                                        \
    if ((call) != MH_OK) {                                                                    \
      std::fprintf(stderr, "%s: %s\n", #call, (ctx) ? mh_last_error(ctx) : "?");              \
      return 4;                                                                               \
    }                                                                                         \
  } while (0)

static int upload_shard(mh_ctx* ctx, const FlatScene& sc, int rank, int world) {
  const int lo = (int)((long)rank * sc.n_models / world), hi = (int)((long)(rank + 1) * sc.n_models / world);
  const int r0 = sc.first_row[lo], r1 = sc.first_row[hi];
  // Update(): this rank's models, L2-normalised on the device (MATCH_ANN_CPU.hpp:80-107); row and model ids global
  CHECK_MH(ctx, mh_db_upload_raw(ctx, &sc.desc[(size_t)r0 * 128], &sc.model_of[r0], &sc.xyz[(size_t)r0 * 3], r1 - r0,
                                 sc.n_models, r0, 1));
  CHECK_MH(ctx, mh_reserve(ctx, sc.Q, 1024, 4096));
  return 0;
}

static void print_objects(const vector<mh_object>& objs) {
  for (size_t i = 0; i < objs.size(); ++i) {
    const mh_object& o = objs[i];
    std::printf("OBJ model%d %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", o.model, o.pose[4], o.pose[5], o.pose[6],
                o.pose[0], o.pose[1], o.pose[2], o.pose[3], o.score);
  }
}

// one thread per rank, all on device 0, host transport
static int sharded_rank_thread(const FlatScene& sc, int rank, int world, int repeats, ThreadGather* tg,
                               vector<mh_object>* out, int32_t* counts) {
  mh_ctx* ctx = 0;
  if (mh_create(0, &ctx) != MH_OK) return 3;
  if (int rc = upload_shard(ctx, sc, rank, world)) return rc;
  ThreadGather::Rank me = {tg, rank};
  mh_comm* comm = 0;
  CHECK_MH(ctx, mh_comm_create_host(ctx, rank, world, &ThreadGather::fn, &me, &comm));
  float *qd = 0, *uv = 0;
  if (hipMalloc(&qd, sc.qd.size() * 4) != hipSuccess || hipMalloc(&uv, sc.uv.size() * 4) != hipSuccess) return 4;
  hipMemcpy(uv, &sc.uv[0], sc.uv.size() * 4, hipMemcpyHostToDevice);
  mh_frame_params prm;
  mh_frame_default_params(&prm);
  for (int rep = 0; rep < repeats; ++rep) {
    hipMemcpy(qd, &sc.qd[0], sc.qd.size() * 4, hipMemcpyHostToDevice);   // the frame normalises in place
    CHECK_MH(ctx, mh_frame_enqueue_sharded(ctx, comm, qd, uv, sc.Q, &sc.cam, &prm, 7));
  }
  out->resize(4096);
  int32_t n = 0;
  CHECK_MH(ctx, mh_frame_gather_objects(ctx, comm, 0, &(*out)[0], (int)out->size(), &n));
  out->resize(n);
  int32_t mine = 0;
  CHECK_MH(ctx, mh_frame_fetch(ctx, 0, 0, &mine, counts));
  hipFree(qd);
  hipFree(uv);
  mh_comm_destroy(comm);
  mh_destroy(ctx);
  return 0;
}

static int run_sharded(const FlatScene& sc, int world, int repeats) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return 3;
  vector<mh_object> objs;
  int32_t matches = 0, clusters = 0;
  struct timespec t0, t1;
  if (world <= n_dev) {
    // the host owns `world` devices: one context per device, RCCL, the W collectives of a frame as one group
    vector<mh_ctx*> ctxs(world, (mh_ctx*)0);
    vector<mh_comm*> comms(world, (mh_comm*)0);
    vector<float*> qd(world, (float*)0);
    vector<const float*> uv(world, (const float*)0);
    for (int r = 0; r < world; ++r) {
      if (mh_create(r, &ctxs[r]) != MH_OK) return 3;
      if (int rc = upload_shard(ctxs[r], sc, r, world)) return rc;
      float* u = 0;
      hipSetDevice(r);
      if (hipMalloc(&qd[r], sc.qd.size() * 4) != hipSuccess || hipMalloc(&u, sc.uv.size() * 4) != hipSuccess) return 4;
      hipMemcpy(u, &sc.uv[0], sc.uv.size() * 4, hipMemcpyHostToDevice);
      uv[r] = u;
    }
    CHECK_MH(ctxs[0], mh_comm_create_all(&ctxs[0], world, &comms[0]));
    mh_frame_params prm;
    mh_frame_default_params(&prm);
    const uint64_t seed = 7;
    double total = 0;
    for (int rep = 0; rep < repeats; ++rep) {
      for (int r = 0; r < world; ++r) {
        hipSetDevice(r);
        hipMemcpy(qd[r], &sc.qd[0], sc.qd.size() * 4, hipMemcpyHostToDevice);
      }
      clock_gettime(CLOCK_REALTIME, &t0);
      CHECK_MH(ctxs[0], mh_frame_enqueue_sharded_all(&ctxs[0], &comms[0], world, &qd[0], &uv[0], sc.Q, 1, &sc.cam, &prm,
                                                     &seed));
      objs.clear();
      matches = clusters = 0;
      for (int r = 0; r < world; ++r) {   // rank order = model order
        vector<mh_object> mine(4096);
        int32_t n = 0, counts[4];
        CHECK_MH(ctxs[r], mh_frame_fetch(ctxs[r], &mine[0], (int)mine.size(), &n, counts));
        objs.insert(objs.end(), mine.begin(), mine.begin() + n);
        matches += counts[0];
        clusters += counts[1];
      }
      clock_gettime(CLOCK_REALTIME, &t1);
      if (rep > 0 || repeats == 1) total += (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    }
    std::printf("TIME FRAME %.6f\n", total / (repeats > 1 ? repeats - 1 : 1));
    for (int r = 0; r < world; ++r) {
      mh_comm_destroy(comms[r]);
      hipSetDevice(r);
      hipFree(qd[r]);
      hipFree(const_cast<float*>(uv[r]));
      mh_destroy(ctxs[r]);
    }
    std::printf("TRANSPORT rccl\n");
  } else {
    ThreadGather tg(world);
    vector<vector<mh_object> > out(world);
    vector<int32_t> counts((size_t)4 * world, 0);
    vector<int> rcs(world, 0);
    vector<std::thread> threads;
    for (int r = 0; r < world; ++r)
      threads.push_back(std::thread([&, r] {
        rcs[r] = sharded_rank_thread(sc, r, world, repeats, &tg, &out[r], &counts[4 * r]);
      }));
    for (int r = 0; r < world; ++r) threads[r].join();
    for (int r = 0; r < world; ++r) {
      if (rcs[r]) return rcs[r];
      matches += counts[4 * r];
      clusters += counts[4 * r + 1];
      if (out[r].size() != out[0].size() ||
          (out[r].size() && std::memcmp(&out[r][0], &out[0][0], out[0].size() * sizeof(mh_object)) != 0)) {
        std::fprintf(stderr, "rank %d gathered other objects than rank 0\n", r);
        return 5;
      }
    }
    objs = out[0];
    std::printf("TRANSPORT host\n");
  }
  std::printf("MATCHES %d CLUSTERS %d WORLD %d\n", matches, clusters, world);
  print_objects(objs);
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 3 && std::string(argv[1]) == "--sift") return run_sift(argv[2]);
  int world = 0;
  if (argc >= 4 && std::string(argv[1]) == "--world") {
    world = std::atoi(argv[2]);
    if (world < 1 || world > 64) return 2;
    argc -= 2;
    argv += 2;
  }
  const bool resident = argc >= 3 && std::string(argv[1]) == "--resident";   // one step for the whole frame
  if (resident) { --argc; ++argv; }
  const bool fill_matches = resident && argc >= 3 && std::string(argv[1]) == "--fill-matches";   // + frameData.matches filled
  if (fill_matches) { --argc; ++argv; }
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
  FlatScene flat;
  flat.n_models = nm;
  flat.Q = Q;
  flat.first_row.push_back(0);
  for (int m = 0; m < nm; ++m) {
    int32_t n = 0;
    if (!rd(f, &n, 1)) return 2;
    vector<float> xyz((size_t)n * 3), desc((size_t)n * 128);
    if (!rd(f, &xyz[0], xyz.size()) || !rd(f, &desc[0], desc.size())) return 2;
    flat.xyz.insert(flat.xyz.end(), xyz.begin(), xyz.end());
    flat.desc.insert(flat.desc.end(), desc.begin(), desc.end());
    flat.model_of.insert(flat.model_of.end(), n, m);
    flat.first_row.push_back(flat.first_row.back() + n);
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
  if (world > 0) {
    if (multi) return 2;
    flat.uv = uv;
    flat.qd = qd;
    const Image& im = *images[0];
    for (int j = 0; j < 4; ++j) flat.cam.K[j] = im.intrinsicLinearCalibration[j];
    for (int j = 0; j < 4; ++j) flat.cam.cam[j] = im.cameraPose.rotation[j];
    for (int j = 0; j < 3; ++j) flat.cam.cam[4 + j] = im.cameraPose.translation[j];
    return run_sharded(flat, world, repeats);
  }

  MopedPipeline pipeline;
  if (resident) createResidentPipeline(pipeline);
  else createPipeline(pipeline);
  list<MopedAlg*> all = pipeline.getAlgs();
  for (list<MopedAlg*>::iterator a = all.begin(); a != all.end(); ++a) {
    if (!(*a)->isCapable()) {
      std::fprintf(stderr, "step %s: no gfx950 device / HIP library -- not capable\n", (*a)->_stepName.c_str());
      return 3;
    }
    (*a)->modelsUpdated(models);  // as MopedPimpl::addModel does (src/moped.cpp:94-99)
  }
  {
    // Moped::getConfig / setConfig (src/moped.cpp:196-220): every step publishes its constants, a host sets some back
    map<string, string> config;
    for (list<MopedAlg*>::iterator a = all.begin(); a != all.end(); ++a) (*a)->getConfig(config);
    std::printf("CONFIG_KEYS %zu\n", config.size());
    if (fill_matches) {
      config["MATCH_SIFT:0:FRAME_RESIDENT_HIP/FillMatches"] = "1";
      for (list<MopedAlg*>::iterator a = all.begin(); a != all.end(); ++a) (*a)->setConfig(config);
    }
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
  std::printf("HANDOVER_STEPS %lu FRAMES %d\n", HipHandover::get().taken, repeats);   // steps that ran on the device-resident frame
  for (map<string, double>::iterator t = total.begin(); t != total.end(); ++t)
    std::printf("TIME %s %.6f\n", t->first.c_str(), t->second / timed);
  for (list<SP_Object>::iterator o = objects.begin(); o != objects.end(); ++o)
    std::printf("OBJ %s %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.4f\n", (*o)->model->name.c_str(),
                (*o)->pose.translation[0], (*o)->pose.translation[1], (*o)->pose.translation[2],
                (*o)->pose.rotation[0], (*o)->pose.rotation[1], (*o)->pose.rotation[2], (*o)->pose.rotation[3],
                (*o)->score);
  return 0;
}
