// moped_hip_bench -- the frame loop of MopedPimpl::processImages (moped2/libmoped/src/moped.cpp:166-194) for a C++ host
// that keeps frames in flight: no Python, no torch, nothing but libmoped_hip.so's C ABI and the HIP runtime's memory /
// stream calls.  `slots` contexts share ONE copy of the model database (mh_db_share), each carries batches of `batch`
// frames through mh_frame_enqueue_batch on its own stream; the frames' descriptors start in PINNED HOST memory (one
// hipMemcpyAsync per batch on the slot's stream) or, for the resident figure, in device memory.  EVERY batch's objects
// come back to the host inside the timed loops (the reference's loop hands each frame's list<SP_Object> to its caller,
// moped.cpp:166-194; moped2/moped_test.cpp:205-207 prints them): mh_frame_fetch_batch_async behind each batch into the
// slot's pinned block, mh_frame_fetch_wait + a pass over the heads before the slot's next batch; the clock stops when
// the last batch's have arrived.  Also: the latency of ONE frame alone (copy + mh_frame_enqueue + mh_frame_fetch).
//
//   moped_hip_bench frames.bin [--slots 16] [--batch 8] [--steps 10] [--frames-per-step 1024] [--json]
//                              [--objects-out objs.bin]
// --objects-out: one more (untimed) pass over the file's frames, step number 1000, through the same delivered path;
// per frame int32 n + n mh_object, in file order (tests/test_gpu_cpp_host.py compares them with the Python pipeline's).
//
// frames.bin (little endian, scripts/dump_scene.py dump_frames):
//   int32 n_models, Q, n_frames ; float K[4] ; float cam[7]
//   per model: int32 n_pts ; float xyz[n_pts][3] ; float desc[n_pts][128]
//   per frame: float q_uv[Q][2] ; float q_desc[Q][128]
#include <execinfo.h>
#include <hip/hip_runtime_api.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "moped_hip.h"

using std::vector;

#define CK_HIP(x)                                                                         \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      std::fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_));                      \
      return 4;                                                                           \
    }                                                                                     \
  } while (0)
#define CK_MH(ctx, x)                                                                     \
  do {                                                                                    \
    int rc_ = (x);                                                                        \
    if (rc_ != MH_OK) {                                                                   \
      std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, (ctx) ? mh_last_error(ctx) : "?");  \
      return 4;                                                                           \
    }                                                                                     \
  } while (0)

template <typename T>
static bool rd(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void on_fault(int sig) {   // a crash says where (the binary is linked -rdynamic)
  void* pc[48];
  const int n = backtrace(pc, 48);
  static const char msg[] = "moped_hip_bench: fatal signal, call stack:\n";
  if (write(2, msg, sizeof msg - 1) < 0) _exit(128 + sig);
  backtrace_symbols_fd(pc, n, 2);
  _exit(128 + sig);
}

int main(int argc, char** argv) {
  signal(SIGSEGV, on_fault);
  signal(SIGBUS, on_fault);
  // one hardware queue per slot: the HIP runtime reads this when it creates its queues
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s frames.bin [--slots N] [--batch B] [--steps K] [--frames-per-step F] [--json]\n", argv[0]);
    return 2;
  }
  int slots = 16, B = 8, steps = 10, frames_per_step = 1024;
  bool json = false;
  std::string objects_out;
  for (int i = 2; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--slots" && i + 1 < argc) slots = std::atoi(argv[++i]);
    else if (a == "--batch" && i + 1 < argc) B = std::atoi(argv[++i]);
    else if (a == "--steps" && i + 1 < argc) steps = std::atoi(argv[++i]);
    else if (a == "--frames-per-step" && i + 1 < argc) frames_per_step = std::atoi(argv[++i]);
    else if (a == "--json") json = true;
    else if (a == "--objects-out" && i + 1 < argc) objects_out = argv[++i];
  }
  if (slots < 1 || slots > 64 || B < 1 || B > MH_MAX_BATCH || steps < 1) return 2;

  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 2; }
  int32_t nm = 0, Q = 0, n_frames = 0;
  mh_cam cam;
  if (!rd(f, &nm, 1) || !rd(f, &Q, 1) || !rd(f, &n_frames, 1) || !rd(f, cam.K, 4) || !rd(f, cam.cam, 7)) return 2;
  vector<float> desc, xyz;
  vector<int32_t> model_of;
  for (int m = 0; m < nm; ++m) {
    int32_t n = 0;
    if (!rd(f, &n, 1)) return 2;
    const size_t r0 = model_of.size();
    xyz.resize((r0 + n) * 3);
    desc.resize((r0 + n) * 128);
    if (!rd(f, &xyz[r0 * 3], (size_t)n * 3) || !rd(f, &desc[r0 * 128], (size_t)n * 128)) return 2;
    model_of.insert(model_of.end(), n, m);
  }
  const int N = (int)model_of.size();
  n_frames = n_frames / B * B;
  if (n_frames < B) { std::fprintf(stderr, "the file holds fewer frames than one batch\n"); return 2; }
  // the frames: keypoints to the device once, descriptors to pinned host memory (batch after batch)
  float *h_desc = 0, *h_uv = 0;
  const size_t fd = (size_t)Q * 128, fu = (size_t)Q * 2;
  CK_HIP(hipHostMalloc(&h_desc, fd * n_frames * 4, hipHostMallocDefault));
  h_uv = (float*)std::malloc(fu * n_frames * 4);
  for (int i = 0; i < n_frames; ++i)
    if (!rd(f, h_uv + fu * i, fu) || !rd(f, h_desc + fd * i, fd)) return 2;
  std::fclose(f);

  // contexts: one database for all (MATCH_ANN_CPU::Update once, src/match/MATCH_ANN_CPU.hpp:72-109)
  vector<mh_ctx*> ctx(slots, (mh_ctx*)0);
  vector<hipStream_t> stream(slots, (hipStream_t)0);
  if (mh_create(0, &ctx[0]) != MH_OK) { std::fprintf(stderr, "no gfx950 device\n"); return 3; }
  for (int s = 0; s < slots; ++s) {
    if (s && mh_create(0, &ctx[s]) != MH_OK) return 3;
    CK_HIP(hipStreamCreateWithFlags(&stream[s], hipStreamNonBlocking));
    CK_MH(ctx[s], mh_set_stream(ctx[s], stream[s]));
    if (s == 0) CK_MH(ctx[0], mh_db_upload_raw(ctx[0], &desc[0], &model_of[0], &xyz[0], N, nm, 0, 1));
    else CK_MH(ctx[s], mh_db_share(ctx[s], ctx[0]));
    CK_MH(ctx[s], mh_reserve_batch(ctx[s], Q, B, 1024, 4096));
  }
  const int pool_groups = n_frames / B;
  float* d_uv = 0;          // [n_frames][Q][2], resident
  float* d_pristine = 0;    // [n_frames][Q][128], resident copy for the "inputs in HBM" figure
  CK_HIP(hipMalloc(&d_uv, fu * n_frames * 4));
  CK_HIP(hipMalloc(&d_pristine, fd * n_frames * 4));
  CK_HIP(hipMemcpy(d_uv, h_uv, fu * n_frames * 4, hipMemcpyHostToDevice));
  CK_HIP(hipMemcpy(d_pristine, h_desc, fd * n_frames * 4, hipMemcpyHostToDevice));
  vector<float*> work(slots, (float*)0);
  for (int s = 0; s < slots; ++s) CK_HIP(hipMalloc(&work[s], fd * B * 4));
  mh_frame_params prm;
  mh_frame_default_params(&prm);

  const int groups = std::max(pool_groups, frames_per_step / B);
  vector<uint64_t> seeds(B);
  // delivery: one pinned block per slot; pending[s] = pool group of the batch whose objects are on their way into it
  const int CAP = 32;   // objects per frame a record carries (n_objects says if there were more)
  const size_t rec = mh_frame_block_stride(CAP);
  vector<unsigned char*> block(slots, (unsigned char*)0);
  for (int s = 0; s < slots; ++s) CK_HIP(hipHostMalloc((void**)&block[s], rec * B, hipHostMallocDefault));
  vector<int> pending(slots, -1);
  long objects = 0, frames_delivered = 0, max_per_frame = 0, min_per_frame = 1 << 30;
  FILE* dump = 0;               // --objects-out pass: the delivered records in file order
  vector<vector<unsigned char> > dump_by_group;
  // the slot's delivery reaches the host: wait for ITS event, read the heads
  auto consume = [&](int s) -> int {
    if (pending[s] < 0) return 0;
    int32_t flags = 0;
    CK_MH(ctx[s], mh_frame_fetch_wait(ctx[s], &flags));
    for (int k = 0; k < B; ++k) {
      const mh_frame_head* h = (const mh_frame_head*)(block[s] + rec * k);
      objects += h->n_objects;
      max_per_frame = std::max<long>(max_per_frame, h->n_objects);
      min_per_frame = std::min<long>(min_per_frame, h->n_objects);
      ++frames_delivered;
    }
    if (!dump_by_group.empty()) dump_by_group[pending[s]].assign(block[s], block[s] + rec * B);
    pending[s] = -1;
    return 0;
  };
  unsigned long issued = 0;   // batches handed out so far: batch i goes to slot i mod slots
  // one step = `groups` batches, round-robin over the slots; from_host: the descriptors cross PCIe inside the loop
  auto run_step = [&](int step, bool from_host, int n_groups) -> int {
    for (int g = 0; g < n_groups; ++g) {
      const int s = (int)(issued++ % (unsigned long)slots), pg = g % pool_groups;
      if (int rc = consume(s)) return rc;   // the slot's previous batch is on the host before the slot is reused
      const float* src = (from_host ? h_desc : d_pristine) + fd * B * pg;
      CK_HIP(hipMemcpyAsync(work[s], src, fd * B * 4, from_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, stream[s]));
      for (int k = 0; k < B; ++k) seeds[k] = 1000ull * (uint64_t)(step + 7) + (uint64_t)g * B + k + 1;
      CK_MH(ctx[s], mh_frame_enqueue_batch(ctx[s], work[s], d_uv + fu * B * pg, Q, B, &cam, &prm, &seeds[0]));
      CK_MH(ctx[s], mh_frame_fetch_batch_async(ctx[s], B, CAP, block[s], (uint32_t)issued));
      pending[s] = pg;
    }
    return 0;
  };
  auto drain = [&]() -> int {
    for (int s = 0; s < slots; ++s)
      if (int rc = consume(s)) return rc;
    return 0;
  };
  auto sync_all = [&]() -> int {
    for (int s = 0; s < slots; ++s) CK_HIP(hipStreamSynchronize(stream[s]));
    return 0;
  };
  double fps[2] = {0, 0};
  long frames_timed = 0;
  for (int mode = 0; mode < 2; ++mode) {   // 0: inputs resident in HBM, 1: descriptors from pinned host memory
    for (int w = 0; w < 3; ++w)
      if (int rc = run_step(-1 - w, mode == 1, groups)) return rc;
    if (int rc = drain()) return rc;
    if (int rc = sync_all()) return rc;
    objects = frames_delivered = 0;
    const double t0 = now_s();
    for (int k = 0; k < steps; ++k)
      if (int rc = run_step(k, mode == 1, groups)) return rc;
    if (int rc = drain()) return rc;       // the clock stops when the last batch's objects are on the host
    if (int rc = sync_all()) return rc;
    fps[mode] = (double)steps * groups * B / (now_s() - t0);
    frames_timed = (long)steps * groups * B;
    if (frames_delivered != frames_timed) {
      std::fprintf(stderr, "delivered %ld frames of %ld timed\n", frames_delivered, frames_timed);
      return 5;
    }
  }
  const long frames_counted = frames_delivered;
  const long objects_counted = objects;
  if (!objects_out.empty()) {
    dump = std::fopen(objects_out.c_str(), "wb");
    if (!dump) { std::perror(objects_out.c_str()); return 2; }
    dump_by_group.assign(pool_groups, vector<unsigned char>());
    if (int rc = run_step(1000, false, pool_groups)) return rc;
    if (int rc = drain()) return rc;
    for (int pg = 0; pg < pool_groups; ++pg)
      for (int k = 0; k < B; ++k) {
        const mh_frame_head* h = (const mh_frame_head*)(&dump_by_group[pg][0] + rec * k);
        const int32_t n = h->n_objects, take = std::min<int32_t>(n, CAP);
        std::fwrite(&n, 4, 1, dump);
        std::fwrite(h + 1, sizeof(mh_object), (size_t)take, dump);
      }
    std::fclose(dump);
    dump_by_group.clear();
  }
  vector<mh_object> objs(4096);
  // one frame alone: pinned host descriptors -> objects on the host
  vector<double> lat;
  for (int i = 0; i < 60; ++i) {
    const int fi = i % n_frames;
    const double t0 = now_s();
    CK_HIP(hipMemcpyAsync(work[0], h_desc + fd * fi, fd * 4, hipMemcpyHostToDevice, stream[0]));
    CK_MH(ctx[0], mh_frame_enqueue(ctx[0], work[0], d_uv + fu * fi, Q, &cam, &prm, 77 + i));
    int32_t n = 0, counts[4];
    CK_MH(ctx[0], mh_frame_fetch(ctx[0], &objs[0], (int)objs.size(), &n, counts));
    if (i >= 10) lat.push_back(now_s() - t0);
  }
  std::sort(lat.begin(), lat.end());
  const double lat_ms = 1e3 * lat[lat.size() / 2];
  // ... and with the descriptors already in HBM (the convention of the throughput figure above: the 1.5 MB of a frame's
  // descriptors do not cross PCIe inside the clock; the frame's working copy is made on the device because
  // mh_frame_enqueue normalises in place)
  lat.clear();
  for (int i = 0; i < 60; ++i) {
    const int fi = i % n_frames;
    CK_HIP(hipMemcpyAsync(work[0], d_pristine + fd * fi, fd * 4, hipMemcpyDeviceToDevice, stream[0]));
    CK_HIP(hipStreamSynchronize(stream[0]));
    const double t0 = now_s();
    CK_MH(ctx[0], mh_frame_enqueue(ctx[0], work[0], d_uv + fu * fi, Q, &cam, &prm, 77 + i));
    int32_t n = 0, counts[4];
    CK_MH(ctx[0], mh_frame_fetch(ctx[0], &objs[0], (int)objs.size(), &n, counts));
    if (i >= 10) lat.push_back(now_s() - t0);
  }
  std::sort(lat.begin(), lat.end());
  const double lat_res_ms = 1e3 * lat[lat.size() / 2];
  const double opf = frames_counted ? (double)objects_counted / frames_counted : 0.0;
  if (json)
    std::printf("{\"host\": \"moped_hip_bench (C++, C ABI only)\", \"slots\": %d, \"frames_per_batch\": %d, \"steps\": %d, "
                "\"frames_per_step\": %d, \"fps_resident\": %.2f, \"fps_pinned_host\": %.2f, \"single_frame_latency_ms\": %.4f, \"single_frame_latency_resident_ms\": %.4f, "
                "\"results_delivered\": \"every frame\", \"frames_delivered\": %ld, \"objects_per_frame\": %.3f, "
                "\"min_objects_per_frame\": %ld, \"max_objects_per_frame\": %ld, \"models\": %d, \"rows\": %d, \"queries\": %d}\n",
                slots, B, steps, groups * B, fps[0], fps[1], lat_ms, lat_res_ms, frames_counted, opf, min_per_frame, max_per_frame, nm, N, Q);
  else
    std::printf("slots %d x %d frames: %.0f frames/s (inputs in HBM), %.0f frames/s (descriptors from pinned host memory); one "
                "frame alone %.3f ms from pinned host memory, %.3f ms with its descriptors in HBM; %.2f objects per frame\n", slots, B, fps[0],
                fps[1], lat_ms, lat_res_ms, opf);
  for (int s = 0; s < slots; ++s) {
    hipFree(work[s]);
    hipHostFree(block[s]);
    mh_destroy(ctx[s]);
    hipStreamDestroy(stream[s]);
  }
  hipFree(d_uv);
  hipFree(d_pristine);
  hipHostFree(h_desc);
  std::free(h_uv);
  return 0;
}
