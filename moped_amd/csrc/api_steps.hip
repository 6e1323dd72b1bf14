// C ABI: CLUSTER / POSE / FILTER entry points and the device-resident frame.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "context.h"
#include "steps.h"

using namespace mh;

// Device buffers of one frame.  Match records are kept sorted by (model, query):
// matches[model] of the reference is the slice [model_off[m], model_off[m+1]).
constexpr int FETCH_PIN_OBJECTS = 32;   // objects that come with a frame's head in mh_frame_fetch's first copy

struct FrameState {
  int max_m = 0, max_clusters = 0, max_objects = 0, n_models_cap = 0;
  // The working arrays below (counts .. tickets) are carved out of ONE allocation of MH_MAX_BATCH equal arenas: the
  // pointers name frame 0's copy, frame f of a batch that goes through the stages in one launch (FrameBatch, steps.h)
  // has its own at + f * arena_bytes.  Results and count snapshots are slot-indexed arrays outside the arenas.
  unsigned char* arena = nullptr;
  size_t arena_bytes = 0;
  int n_arenas = 1;   // copies allocated: 1 until the first merged batch of B frames asks for B
  FrameCounts* counts = nullptr;
  int32_t* n_slots = nullptr;      // object slots in use (device scalar)
  int32_t* n_clusters = nullptr;   // rows of the current cluster table (device scalar)
  // ... and of the table FILTER rewrites for POSE2: a word of its own -- a POSE launch's workgroups read every frame's
  // cluster count as they walk the batch, and the frame's closing workgroup (fused FILTER) may have rewritten the table
  // by the time a workgroup without a task in that frame comes by
  int32_t* n_clusters2 = nullptr;
  // group
  int32_t *acc_q = nullptr, *acc_model = nullptr, *m_q = nullptr, *m_model = nullptr, *m_rep = nullptr;
  mh_corr* m_corr = nullptr;
  // several images: image of every match; the matches again in (model, image, query) order for CLUSTER / POSE
  int32_t *m_img = nullptr, *mi_img = nullptr, *off2 = nullptr;
  mh_corr* mi_corr = nullptr;
  mh_depth* m_depth = nullptr;     // per match, when the frame carries depth attributes
  int32_t* model_off = nullptr;
  // cluster
  int32_t *ms_members = nullptr, *ms_cl_start = nullptr, *ms_ncl = nullptr;
  int32_t *cl_model = nullptr, *cl_begin = nullptr, *cl_count = nullptr;
  // objects
  int32_t *obj_model = nullptr, *obj_ninl = nullptr, *obj_cluster = nullptr, *obj_valid = nullptr,
          *obj_npts = nullptr, *obj_clsize = nullptr;
  float *obj_pose = nullptr, *obj_err = nullptr, *obj_score = nullptr, *obj_score_raw = nullptr;
  // filter
  unsigned long long* best = nullptr;
  int32_t* new_members = nullptr;
  // POSE as two launches (PoseSplit, steps.h): the tasks' winning hypotheses, scratch for clusters past the refine's LDS cache
  PoseHyp* hyp = nullptr;
  float* rf_pts = nullptr;
  int32_t* rf_list = nullptr;
  // packed result {int32 n; int32 pad[3]; mh_object[max_objects]}
  unsigned char* result = nullptr;
  size_t result_bytes = 0;
  int32_t* snap = nullptr;  // [4] counts snapshot: matches, clusters, objects after POSE, after FILTER
  int task_grid = 32;   // workgroups for the POSE/FILTER launches: follows the task count of the last fetched frame
  int ms_grid = 8;      // ... and of the CLUSTER launch: its cluster count + head room
  int slot = 0;            // result / snap slot the next frame_rest writes (frames of a batch share the context)
  int list_first = 0, list_n = 1;   // result slots whose match lists are still in the arenas: [list_first, list_first + list_n)
  // What the last launches found, written by the kernels' tails into host-visible (pinned, mapped) words and read -- without
  // any synchronisation: a guess is all it is -- when the next launches are sized: [0][f] (cluster, replica) tasks of POSE
  // in frame f of the batch, [1][f] of POSE2, [2][f] models that CLUSTER had to cluster.
  int32_t* fb = nullptr;
  // mh_frame_fetch[_slot]: where the frame's head, counters and first objects land -- pinned, so that the copies are
  // enqueued together and one synchronisation ends them (into pageable memory every one of them blocks: four round
  // trips, ~40 us of one synchronous frame's 550)
  struct FetchPin {
    int32_t head[4];
    mh_object objects[FETCH_PIN_OBJECTS];
    int32_t snap[4];
    FrameCounts fc;
    int32_t n_feat;
  }* fetch_pin = nullptr;
  // one frame alone: the closing workgroup of FILTER2 writes the frame's head, counters and objects HERE itself
  // (page-locked, device-writable): mh_frame_fetch then synchronises and reads -- no copy at all
  FrameHostBlock* host_block = nullptr;
  bool host_armed = false;   // the frame enqueued last writes host_block (a batch, a frame without FILTER2: no)
  uint32_t host_seq_expect = 0;   // armed enqueues so far = what host_block->seq reads once the last of them is through
  // the fused FILTER / FILTER2 steps' arguments on the device + what the host last stored there (FilterFuse, steps.h)
  FilterFuseArgs* fuse_dev = nullptr;   // [2]
  FilterFuseArgs fuse_shadow[2];
  bool fuse_valid[2] = {false, false};
  unsigned int* tickets = nullptr;  // [8] last_workgroup() words: 0 CLUSTER, 1 POSE, 2 FILTER, 3 POSE2, 4 FILTER2
};

namespace {

template <typename T>
int dev_alloc(mh_ctx* ctx, T*& p, size_t n) {
  MH_HIP(ctx, hipMalloc(&p, (n > 0 ? n : 1) * sizeof(T)));
  return MH_OK;
}

void free_fs(FrameState* fs) {
  if (!fs) return;
  void* ptrs[] = {fs->arena, fs->result, fs->snap, fs->fuse_dev};
  for (void* p : ptrs)
    if (p) hipFree(p);
  if (fs->fb) hipHostFree(fs->fb);
  if (fs->fetch_pin) hipHostFree(fs->fetch_pin);
  if (fs->host_block) hipHostFree(fs->host_block);
  delete fs;
}

int ensure_fs(mh_ctx* ctx, int max_m, int max_clusters, int max_objects, int n_models, int n_arenas = 1) {
  FrameState* fs = ctx->fs;
  if (fs && fs->max_m >= max_m && fs->max_clusters >= max_clusters &&
      fs->max_objects >= max_objects && fs->n_models_cap >= n_models && fs->n_arenas >= n_arenas)
    return MH_OK;
  int task_grid = 0, ms_grid = 0;
  unsigned char* kept_result = nullptr;
  int32_t* kept_snap = nullptr;
  if (fs) {
    n_arenas = std::max(n_arenas, fs->n_arenas);
    task_grid = fs->task_grid;
    ms_grid = fs->ms_grid;
    max_m = std::max(max_m, fs->max_m);
    max_clusters = std::max(max_clusters, fs->max_clusters);
    max_objects = std::max(max_objects, fs->max_objects);
    n_models = std::max(n_models, fs->n_models_cap);
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (max_objects == fs->max_objects) {   // the result blocks keep their size: frames not fetched yet stay fetchable
      kept_result = fs->result;
      kept_snap = fs->snap;
      fs->result = nullptr;
      fs->snap = nullptr;
    }
    free_fs(fs);
    ctx->fs = nullptr;
  }
  fs = new FrameState;
  ctx->fs = fs;
  fs->max_m = max_m;
  fs->max_clusters = max_clusters;
  fs->max_objects = max_objects;
  fs->n_models_cap = n_models;
  fs->n_arenas = n_arenas;
  if (task_grid) {   // (what the launches had learnt about the frames' task counts)
    fs->task_grid = task_grid;
    fs->ms_grid = ms_grid;
  }
  int rc = 0;
  // two passes over the same list: sizes first (pointers are offsets into a null arena), then the real addresses
  for (int pass = 0; pass < 2 && !rc; ++pass) {
    size_t off = 0;
    unsigned char* const base = fs->arena;
    auto carve = [&](auto*& p, size_t n) {
      typedef typename std::remove_reference<decltype(*p)>::type T;
      p = reinterpret_cast<T*>(base + off);
      off += ((n > 0 ? n : 1) * sizeof(T) + 255) & ~(size_t)255;
    };
    carve(fs->counts, 1);
    carve(fs->n_slots, 1);
    carve(fs->n_clusters, 1);
    carve(fs->n_clusters2, 1);
    carve(fs->tickets, 8);
    carve(fs->acc_q, max_m);
    carve(fs->acc_model, max_m);
    carve(fs->m_q, max_m);
    carve(fs->m_model, max_m);
    carve(fs->m_rep, max_m);
    carve(fs->m_corr, max_m);
    carve(fs->m_depth, max_m);
    carve(fs->m_img, max_m);
    carve(fs->mi_img, max_m);
    carve(fs->mi_corr, max_m);
    carve(fs->off2, (size_t)n_models + 1);
    carve(fs->model_off, (size_t)n_models + 1);
    carve(fs->ms_members, max_m);
    carve(fs->ms_cl_start, (size_t)max_m + n_models + 1);
    carve(fs->ms_ncl, (size_t)n_models + 1);
    carve(fs->cl_model, max_clusters);
    carve(fs->cl_begin, max_clusters);
    carve(fs->cl_count, max_clusters);
    carve(fs->obj_model, max_objects);
    carve(fs->obj_ninl, max_objects);
    carve(fs->obj_cluster, max_objects);
    carve(fs->obj_valid, max_objects);
    carve(fs->obj_npts, max_objects);
    carve(fs->obj_clsize, (size_t)2 * max_objects);
    carve(fs->obj_pose, (size_t)7 * max_objects);
    carve(fs->obj_err, max_objects);
    carve(fs->obj_score, max_objects);
    carve(fs->obj_score_raw, max_objects);
    carve(fs->best, max_m);
    carve(fs->new_members, max_m);
    carve(fs->hyp, max_objects);
    carve(fs->rf_pts, (size_t)9 * max_m);
    carve(fs->rf_list, (size_t)4 * max_m);
    if (pass == 0) {
      fs->arena_bytes = off;
      rc |= dev_alloc(ctx, fs->arena, off * n_arenas);
    }
  }
  fs->result_bytes = 16 + sizeof(mh_object) * (size_t)max_objects;
  if (kept_result) {
    fs->result = kept_result;
    fs->snap = kept_snap;
  } else {
    rc |= dev_alloc(ctx, fs->result, fs->result_bytes * MH_MAX_BATCH);
    if (!rc) MH_HIP(ctx, hipMemsetAsync(fs->result, 0, fs->result_bytes * MH_MAX_BATCH, ctx->stream));   // "0 objects" before the first frame
    rc |= dev_alloc(ctx, fs->snap, 4 * MH_MAX_BATCH);
    if (!rc) MH_HIP(ctx, hipMemsetAsync(fs->snap, 0, sizeof(int32_t) * 4 * MH_MAX_BATCH, ctx->stream));   // (a slot fetched before it was written reads zeros)
  }
  rc |= dev_alloc(ctx, fs->fuse_dev, 2);
  if (!rc) {
    if (hipHostMalloc(&fs->fb, 3 * MH_MAX_BATCH * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) rc = MH_ERR_HIP;
    else std::memset(fs->fb, 0xFF, 3 * MH_MAX_BATCH * sizeof(int32_t));   // -1 = nothing known yet
    if (hipHostMalloc(&fs->fetch_pin, sizeof(*fs->fetch_pin), hipHostMallocDefault) != hipSuccess) rc = MH_ERR_HIP;
    if (hipHostMalloc(&fs->host_block, sizeof(*fs->host_block), hipHostMallocDefault) != hipSuccess) rc = MH_ERR_HIP;
    else std::memset(fs->host_block, 0, sizeof(*fs->host_block));
  }
  if (rc) {   // a half-built state must not look valid to the next call
    free_fs(fs);
    ctx->fs = nullptr;
    return MH_ERR_HIP;
  }
  // every frame's tickets, claim table (best), obj_valid / obj_score / obj_npts start at zero
  MH_HIP(ctx, hipMemsetAsync(fs->arena, 0, fs->arena_bytes * n_arenas, ctx->stream));
  fs->host_seq_expect = 0;   // (tickets[7], the device's count of host-block writes, is zero again)
  fs->host_armed = false;
  return MH_OK;
}

FilterBuffers make_fb(const mh_ctx* ctx, const FrameState* fs, int n_models) {
  (void)ctx;
  FilterBuffers fb;
  fb.corr = fs->m_corr;
  fb.m_rep = fs->m_rep;
  fb.m_img = nullptr;
  fb.cams = nullptr;
  fb.n_images = 1;
  fb.model_off = fs->model_off;
  fb.n_models = n_models;
  fb.max_m = fs->max_m;
  fb.obj_model = fs->obj_model;
  fb.obj_pose = fs->obj_pose;
  fb.obj_score = fs->obj_score;
  fb.obj_score_raw = fs->obj_score_raw;
  fb.obj_valid = fs->obj_valid;
  fb.obj_npts = fs->obj_npts;
  fb.max_objects = fs->max_objects;
  fb.best = fs->best;
  fb.obj_clsize = fs->obj_clsize;
  fb.new_members = fs->new_members;
  fb.cl_model = fs->cl_model;
  fb.cl_begin = fs->cl_begin;
  fb.cl_count = fs->cl_count;
  fb.max_clusters = fs->max_clusters;
  return fb;
}

__global__ void set_scalar_kernel(int32_t* p, int32_t v) { *p = v; }

// mh_frame_enqueue_image_batch: the B images' keypoints lie at a fixed stride of Q rows; rows past an image's count
// become zero rows with a norm term of -1 before MATCH -- "no such query" to the two-stage search (a plain zero query is
// its worst case: every row of a normalised DB ties, the candidate lists overflow and the query falls back to brute
// force), a finite dummy to the exact kernels ...
__global__ void image_batch_tail_kernel(float* __restrict__ desc, float* __restrict__ norm, const int32_t* __restrict__ counts,
                                        int Q, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte piece of a row
  if (i >= B * Q * (DIM / 4)) return;
  const int row = i / (DIM / 4), f = row / Q, q = row - f * Q;
  if (q < min(counts[f], Q)) return;
  reinterpret_cast<float4*>(desc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i % (DIM / 4) == 0) norm[row] = -1.f;
}
// ... and "no neighbour" after it, so that no step of the frame sees them
__global__ void image_batch_mask_kernel(int32_t* __restrict__ idx, float* __restrict__ d1, float* __restrict__ d2,
                                        const int32_t* __restrict__ counts, int Q, int B) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= B * Q) return;
  const int f = row / Q, q = row - f * Q;
  if (q < min(counts[f], Q)) return;
  idx[row] = -1;
  d1[row] = __builtin_inff();
  d2[row] = __builtin_inff();
}

// Result block of a frame that stops after POSE (run_stage2 = 0): the valid objects in
// list order.  Frames with the FILTER stages get it from the last FILTER launch.
__global__ void pack_result_kernel(unsigned char* result, const int32_t* n_slots,
                                   const int32_t* obj_valid, const int32_t* obj_model,
                                   const float* obj_pose, const float* obj_score,
                                   const int32_t* obj_npts, int max_objects, const FrameCounts* counts) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  mh_object* out = reinterpret_cast<mh_object*>(result + 16);
  int k = 0;
  const int n = *n_slots;
  for (int o = 0; o < n && k < max_objects; ++o) {
    if (!obj_valid[o]) continue;
    mh_object ob;
    ob.model = obj_model[o];
    for (int j = 0; j < 7; ++j) ob.pose[j] = obj_pose[7 * o + j];
    ob.score = obj_score[o];
    ob.n_points = obj_npts[o];
    out[k++] = ob;
  }
  reinterpret_cast<int32_t*>(result)[0] = k;
  reinterpret_cast<int32_t*>(result)[1] = counts->error;   // capacity flags, as the last FILTER launch reports them
}

int ensure_linkage_scratch(mh_ctx* ctx, size_t floats) {
  if (floats <= ctx->lk_scratch_floats) return MH_OK;
  // 3 n^2 floats per (model, frame) problem: 37 MB per frame at 3 000 matches, 0.6 GB for a batch of 16 -- and it grows
  // with the square of what a caller reserves.  Bounded per context (mh_set_linkage_scratch_limit, default 4 GiB)
  // with an error the caller can act on instead of an allocation that takes the device's memory from the other slots.
  if (floats * sizeof(float) > ctx->lk_scratch_limit) {
    ctx->err = "linkage clusterer: " + std::to_string(floats * sizeof(float) >> 20) + " MiB of similarity-matrix scratch asked for, the "
               "context's limit is " + std::to_string(ctx->lk_scratch_limit >> 20) + " MiB (fewer frames per batch, fewer queries "
               "reserved, or mh_set_linkage_scratch_limit)";
    return MH_ERR_CAPACITY;
  }
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->lk_scratch) MH_HIP(ctx, hipFree(ctx->lk_scratch));
  ctx->lk_scratch = nullptr;
  ctx->lk_scratch_floats = 0;
  MH_HIP(ctx, hipMalloc(&ctx->lk_scratch, floats * sizeof(float)));
  ctx->lk_scratch_floats = floats;
  return MH_OK;
}

// Device buffers of the depth rules: patch map, per-(model, patch) counts (kept zero), keep flags.
// (frames > 1: a merged batch -- every frame's patch map, counters and keep flags, frame after frame)
int ensure_rule_buffers(mh_ctx* ctx, int patches, int Q, int frames = 1) {
  mh_ctx::DepthRuleState& rs = ctx->rules;
  if (patches > 4096) {
    ctx->err = "depth rules: more than 4096 patches (raise PatchSize)";
    return MH_ERR_CAPACITY;
  }
  Q *= frames;
  if (patches * frames > rs.patches_cap) {
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (rs.inv_size) MH_HIP(ctx, hipFree(rs.inv_size));
    rs.inv_size = nullptr;
    MH_HIP(ctx, hipMalloc(&rs.inv_size, sizeof(double) * patches * frames));
    rs.patches_cap = patches * frames;
  }
  const size_t need = (size_t)std::max(ctx->n_models, 1) * patches * frames;
  if (need > rs.cnt_cap) {
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (rs.cnt) MH_HIP(ctx, hipFree(rs.cnt));
    rs.cnt = nullptr;
    MH_HIP(ctx, hipMalloc(&rs.cnt, sizeof(int32_t) * need));
    MH_HIP(ctx, hipMemsetAsync(rs.cnt, 0, sizeof(int32_t) * need, ctx->stream));
    rs.cnt_cap = need;
  }
  if (Q > rs.keep_cap) {
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (rs.keep1) MH_HIP(ctx, hipFree(rs.keep1));
    rs.keep1 = nullptr;
    MH_HIP(ctx, hipMalloc(&rs.keep1, (size_t)Q));
    rs.keep_cap = Q;
  }
  return MH_OK;
}

void stamp(mh_ctx* ctx, int i) {
  if (ctx->timing) hipEventRecord(ctx->ev[i], ctx->stream);
}

// CLUSTER .. FILTER2 of a device-resident frame in six launches.  gathered != nullptr:
// exchange-1 blocks ([n_shards][3][Q]) to merge first.
// batch_n > 1: the batch_n frames of a batch together, one launch per stage (FrameBatch, steps.h; the caller has checked
// merged_batch_ok) -- q_uv_dev / the top-2 arrays name frame 0's, fs->slot is 0, batch_seeds the frames' seeds.
// stage_lo .. stage_hi (mh_step_*: the six slots one call each on a frame that stays on the device): only the stages
// 0 MATCH's tail (ratio test + lists), 1 CLUSTER, 2 POSE, 3 FILTER, 4 POSE2, 5 FILTER2 in that range are launched, FILTER
// as launches of its own; everything else of the frame's state is left as the stage before wrote it.
int frame_rest(mh_ctx* ctx, const float* q_uv_dev, int Q, const int32_t* gathered, int n_shards,
               const mh_cam* cam, const mh_frame_params* prm, uint64_t seed, const uint64_t* batch_seeds = nullptr,
               int batch_n = 1, int stage_lo = 0, int stage_hi = 5) {
  FrameState* fs = ctx->fs;
  const bool stepped = stage_lo != 0 || stage_hi != 5;
  auto runs = [&](int stage) { return stage >= stage_lo && stage <= stage_hi; };
  if (!stepped) ctx->step.done = -1;   // (a whole frame overwrites whatever a stepped frame left in the arrays)
  FrameBatch fb1, fb2;
  const FrameBatch *b1 = nullptr, *b2 = nullptr;
  if (batch_n > 1) {
    fb1.arena = fs->arena_bytes;
    fb1.q = Q;
    fb1.result_bytes = (int)fs->result_bytes;
    fb1.n = batch_n;
    fb2 = fb1;
    for (int f = 0; f < batch_n; ++f) {
      fb1.seed[f] = batch_seeds[f];
      fb2.seed[f] = batch_seeds[f] ^ 0x5DEECE66Dull;
    }
    b1 = &fb1;
    b2 = &fb2;
  }
  hipStream_t s = ctx->stream;
  const DevCam dc = make_devcam(*cam);
  const int nm = ctx->n_models;
  // group_kernel keeps one LDS histogram bin per model: every entry point (fused, sharded, batched)
  // comes through here, so the bound is checked here and not in the callers
  if (nm > MH_MAX_MODELS) {
    ctx->err = "more than 8192 models per context";
    return MH_ERR_CAPACITY;
  }
  unsigned char* const result = fs->result + (size_t)fs->slot * fs->result_bytes;
  int32_t* const snap = fs->snap + 4 * fs->slot;
  // (mh_frame_fetch_matches_slot: a merged batch leaves every frame's lists in its own arena, a frame on its own in arena 0)
  fs->list_first = fs->slot;
  fs->list_n = batch_n > 1 ? batch_n : 1;
  // Every workgroup of the POSE / FILTER launches needs a free compute unit to start even if
  // it has no task, and MATCH kernels of other frames keep all of them busy: launch about as
  // many workgroups as the previous frame had tasks (experiment builds: MH_TASK_GRID pins the number, MH_MS_GRID the
  // CLUSTER launch's -- 0 = one workgroup per model, the old shape).
  static const int grid_env = exp_int("MH_TASK_GRID", 0);
  // POSE / POSE2: one row of workgroups for all frames of the launch, about as many as the launches before found tasks
  // (+ 25%); nothing known yet: the per-frame guess (the last fetched frame's task count) times the frames
  auto pose_grid = [&](const int32_t* found) {
    if (grid_env > 0) return grid_env;
    long sum = 0;
    bool known = true;
    for (int f = 0; f < batch_n; ++f) {
      known &= found[f] >= 0;
      sum += found[f] >= 0 ? found[f] : 0;
    }
    if (!known) sum = (long)fs->task_grid * batch_n / 2;
    return (int)std::min<long>(160, std::max<long>(8, sum + sum / 4 + 2));
  };
  const int grid = pose_grid(fs->fb), grid2 = pose_grid(fs->fb + MH_MAX_BATCH);
  static const int ms_grid_env = exp_int("MH_MS_GRID", -1);
  // CLUSTER: one row of workgroups for all frames of the launch, as many as the launches before found models to cluster
  // (+ 1 per 8); nothing known yet: the per-frame guess times the frames.  Every one of them takes a whole compute unit.
  int ms_grid = fs->ms_grid;
  {
    const int32_t* found = fs->fb + 2 * MH_MAX_BATCH;
    long sum = 0;
    bool known = true;
    for (int f = 0; f < batch_n; ++f) {
      known &= found[f] >= 0;
      sum += found[f] >= 0 ? found[f] : 0;
    }
    if (!known) sum = (long)std::max(2, fs->ms_grid / 2) * batch_n;
    ms_grid = (int)std::min<long>(48, std::max<long>(batch_n > 1 ? 4 : 2, sum + sum / 8 + 1));
    if (batch_n == 1) ms_grid = std::max(ms_grid, std::min(fs->ms_grid, 8));
  }
  if (ms_grid_env >= 0) ms_grid = ms_grid_env;
  const bool multi = ctx->q_img && ctx->n_images > 1 && ctx->cams_dev;
  if (multi && (ctx->q_depth || ctx->depth_img.img || ctx->linkage_on)) {
    ctx->err = "frames with several images: the moped3d depth steps are single-camera";
    return MH_ERR_ARG;
  }
  // a merged batch with a depth map per frame (mh_frame_set_depth_image_batch; merged_batch_ok has checked the count)
  DepthMaps dmaps;
  const DepthMaps* maps = nullptr;
  if (batch_n > 1 && ctx->depth_img.img) {
    for (int f = 0; f < batch_n; ++f) {
      dmaps.img[f] = ctx->batch_img[f];
      dmaps.fill[f] = ctx->batch_fill[f];
    }
    maps = &dmaps;
  }
  // moped3d depth rules: patch maps of this frame's depth image, DEPTHFILTER on the features
  DepthRules rules;
  if (ctx->rules.on && ctx->depth_img.img && runs(0)) {
    mh_ctx::DepthRuleState& rs = ctx->rules;
    const int pw = (ctx->depth_img.w + rs.patch - 1) / rs.patch, ph = (ctx->depth_img.h + rs.patch - 1) / rs.patch;
    int rc = ensure_rule_buffers(ctx, pw * ph, Q, batch_n);
    if (rc) return rc;
    const bool filters = rs.feature_filter >= 0.f || rs.match_filter >= 0.f;
    if (filters) launch_depth_patches(ctx->depth_img, rs.K, rs.patch, rs.inv_size, s, maps, batch_n);
    if (rs.feature_filter >= 0.f) {
      launch_feature_density(q_uv_dev, Q, ctx->feat_count_dev, rs.patch, pw, ph, rs.inv_size, rs.feature_filter,
                             rs.keep1, s, batch_n);
      rules.keep1 = rs.keep1;
    }
    if (rs.match_filter >= 0.f) {
      rules.inv_size = rs.inv_size;
      rules.cnt = rs.cnt;
      rules.filter2 = rs.match_filter;
    }
    rules.patch = rs.patch;
    rules.pw = pw;
    rules.ph = ph;
    if (rs.ratio_table && rs.table_models >= nm) rules.ratio_table = reinterpret_cast<const float4*>(rs.ratio_table);
    rules.max_depth = rs.max_depth;
    rules.default_depth = rs.default_depth;
    rules.cauchy_scale = rs.cauchy_scale;
  }
  // MATCH tail: (shard merge,) ratio test + per-model lists; resets the frame's counters
  const int q0 = gathered ? 0 : ctx->batch_q0;   // frame of a batch matched in one launch: its slice of the top-2 arrays
  if (runs(0))
  launch_group(gathered, n_shards, ctx->nn_idx + q0, ctx->nn_d1 + q0, ctx->nn_d2 + q0, Q, prm->ratio, q_uv_dev,
               ctx->db_model, ctx->db_xyz, ctx->N, ctx->rmap, nm, fs->max_m, fs->acc_q,
               fs->acc_model, fs->m_q, fs->m_model, fs->m_corr, fs->m_rep, fs->model_off,
               ctx->q_depth ? ctx->q_depth + q0 : nullptr,   // (a batch's depth attributes lie frame after frame like its queries)
               fs->m_depth, ctx->depth_img, fs->counts, fs->n_slots, fs->best, s, rules,
               gathered ? ctx->exchange_stride : 0, gathered ? ctx->exchange_plane : 0, b1,
               fs->slot == 0 ? ctx->exchange_tags : nullptr, maps);
  stamp(ctx, 2);
  // CLUSTER (+ flat cluster table, snap[0..1])
  const bool have_depth = ctx->q_depth || ctx->depth_img.img;
  if (!runs(1)) {
  } else if (ctx->linkage_on && have_depth && ctx->depth_img.img) {
    // moped3d: linkage over similarity matrices; 3 n^2 floats of scratch per model, n <= LK_CAP
    // (a merged batch: every frame its own region)
    const size_t need = 3 * (size_t)std::min(fs->max_m, LK_CAP) * (size_t)fs->max_m;
    int rc = ensure_linkage_scratch(ctx, need * (size_t)std::max(1, batch_n));
    if (rc) return rc;
    launch_linkage_models(fs->m_corr, reinterpret_cast<const float*>(fs->m_depth), fs->model_off, nm, ctx->depth_img,
                          ctx->linkage, ctx->lk_scratch, batch_n > 1 ? need : ctx->lk_scratch_floats, fs->ms_members,
                          fs->ms_cl_start, fs->ms_ncl, fs->max_clusters, fs->cl_model, fs->cl_begin, fs->cl_count,
                          fs->n_clusters, snap, fs->counts, fs->tickets + 0, s, ms_grid, b1, maps,
                          fs->fb + 2 * MH_MAX_BATCH);
  } else if (multi) {
    // MeanShift per (model, image) in image order (CLUSTER_MEAN_SHIFT_CPU.hpp:189-195)
    launch_image_split(fs->m_corr, fs->m_q, fs->m_model, fs->model_off, nm, ctx->q_img + (size_t)ctx->batch_f * Q, ctx->n_images, fs->counts,
                       fs->m_img, fs->m_rep, fs->mi_corr, fs->mi_img, fs->off2, s);
    launch_meanshift_models(fs->mi_corr, fs->off2, nm * ctx->n_images, prm->ms_radius, prm->ms_merge,
                            prm->ms_min_pts, prm->ms_max_iter, fs->ms_members, fs->ms_cl_start,
                            fs->ms_ncl, fs->max_clusters, fs->cl_model, fs->cl_begin, fs->cl_count,
                            fs->n_clusters, snap, fs->counts, fs->tickets + 0, s, ctx->n_images, ms_grid);
  } else
  launch_meanshift_models(fs->m_corr, fs->model_off, nm, prm->ms_radius, prm->ms_merge,
                          prm->ms_min_pts, prm->ms_max_iter, fs->ms_members, fs->ms_cl_start,
                          fs->ms_ncl, fs->max_clusters, fs->cl_model, fs->cl_begin, fs->cl_count,
                          fs->n_clusters, snap, fs->counts, fs->tickets + 0, s, 1, ms_grid, b1, fs->fb + 2 * MH_MAX_BATCH);
  stamp(ctx, 3);
  PoseImages img1, img2;   // POSE works on the (model, image, query) copy, POSE2 on FILTER's clusters over the match lists
  if (multi) {
    img1.cams = img2.cams = ctx->cams_dev;
    img1.n_images = img2.n_images = ctx->n_images;
    img1.img_of = fs->mi_img;
    img2.img_of = fs->m_img;
  }
  // POSE (+ slot count, snap[2] = objects after POSE).  With the FILTER stages on, each FILTER runs in the tail of
  // the POSE launch before it (its last workgroup: filter_dev.h) -- four launches per frame instead of six; a
  // dependent launch costs the pipeline ~5% of its throughput whatever is in it (MH_FUSE_FILTER=0: launches of
  // their own, the same objects).
  static const bool fuse_filter = exp_int("MH_FUSE_FILTER", 1) != 0;
  const float* depth4 = (ctx->q_depth || ctx->depth_img.img) ? reinterpret_cast<const float*>(fs->m_depth) : nullptr;
  FilterBuffers fb = make_fb(ctx, fs, nm);
  if (multi) {
    fb.m_img = fs->m_img;
    fb.cams = ctx->cams_dev;
    fb.n_images = ctx->n_images;
  }
  const bool fused = fuse_filter && prm->run_stage2 && !ctx->timing && !stepped;   // (stage timing wants the steps apart)
  // (one frame alone in result slot 0: FILTER2's tail also writes the host's block, mh_frame_fetch reads it without a copy)
  fs->host_armed = batch_n == 1 && fs->slot == 0 && prm->run_stage2 && fs->host_block && !stepped;
  if (fs->host_armed) ++fs->host_seq_expect;
  const FilterTail ft1{fs->tickets + 2, snap + 3, nullptr, grid, nullptr, nullptr, nullptr},
      ft2{fs->tickets + 4, nullptr, result, grid, fs->host_armed ? fs->host_block : nullptr, fs->host_armed ? snap : nullptr,
          fs->host_armed ? fs->tickets + 7 : nullptr};
  FilterFuse ff1, ff2;
  ff1.fb = ff2.fb = &fb;
  ff1.tail = &ft1;
  ff2.tail = &ft2;
  ff1.n_clusters_dev = fs->n_clusters2;   // FILTER writes POSE2's cluster count, FILTER2 the frame's final one
  ff2.n_clusters_dev = fs->n_clusters;
  ff1.dev = fs->fuse_dev;
  ff1.shadow = &fs->fuse_shadow[0];
  ff1.shadow_valid = &fs->fuse_valid[0];
  ff2.dev = fs->fuse_dev + 1;
  ff2.shadow = &fs->fuse_shadow[1];
  ff2.shadow_valid = &fs->fuse_valid[1];
  ff1.min_points = prm->f1_min_points;
  ff1.feature_distance = prm->f1_feature_distance;
  ff1.min_score = prm->f1_min_score;
  ff2.min_points = prm->f2_min_points;
  ff2.feature_distance = prm->f2_feature_distance;
  ff2.min_score = prm->f2_min_score;
  PoseSplit split;
  split.hyp = (fused && ctx->pose_split) ? fs->hyp : nullptr;   // (the refine launch closes the frames through the fused FILTER tail)
  split.pts = fs->rf_pts;
  split.list = fs->rf_list;
  split.max_m = fs->max_m;
  if (runs(2))
  launch_pose(multi ? fs->mi_corr : fs->m_corr, depth4, ctx->depth_kind, ctx->depth_alpha, fs->ms_members, fs->cl_model,
              fs->cl_begin, fs->cl_count, fs->n_clusters, fs->max_clusters, dc, prm->pose1, seed, fs->n_slots,
              fs->max_objects, fs->obj_model, fs->obj_pose, fs->obj_ninl, fs->obj_err, fs->obj_cluster,
              fs->obj_valid, fs->counts, PoseTail{fs->tickets + 1, fs->n_slots, snap + 2, grid, fs->fb}, s, img1,
              fused ? &ff1 : nullptr, b1, &split);
  stamp(ctx, 4);
  if (prm->run_stage2) {
    // FILTER (snap[3] = objects kept)
    if (!fused && runs(3))
      launch_filter(fb, dc, prm->f1_min_points, prm->f1_feature_distance, prm->f1_min_score,
                    fs->n_slots, fs->n_clusters2, fs->counts, ft1, s);
    stamp(ctx, 5);
    // POSE2 on the rewritten clusters, objects appended after the kept ones
    if (runs(4))
    launch_pose(fs->m_corr, depth4, ctx->depth_kind, ctx->depth_alpha, fs->new_members, fs->cl_model,
                fs->cl_begin, fs->cl_count, fs->n_clusters2, fs->max_clusters, dc, prm->pose2,
                seed ^ 0x5DEECE66Dull, fs->n_slots, fs->max_objects, fs->obj_model, fs->obj_pose,
                fs->obj_ninl, fs->obj_err, fs->obj_cluster, fs->obj_valid, fs->counts,
                PoseTail{fs->tickets + 3, fs->n_slots, nullptr, grid2, fs->fb + MH_MAX_BATCH}, s, img2, fused ? &ff2 : nullptr, b2,
                &split);
    stamp(ctx, 6);
    // FILTER2 (+ the frame's result block)
    if (!fused && runs(5))
      launch_filter(fb, dc, prm->f2_min_points, prm->f2_feature_distance, prm->f2_min_score,
                    fs->n_slots, fs->n_clusters, fs->counts, ft2, s);
    stamp(ctx, 7);
  } else {
    for (int i = 5; i <= 7; ++i) stamp(ctx, i);
    hipLaunchKernelGGL(pack_result_kernel, dim3(1), dim3(1), 0, s, result, fs->n_slots,
                       fs->obj_valid, fs->obj_model, fs->obj_pose, fs->obj_score, fs->obj_npts,
                       fs->max_objects, fs->counts);
  }
  stamp(ctx, 8);
  MH_HIP(ctx, hipGetLastError());
  return MH_OK;
}

// The frames of a batch can share their launches when nothing of the frame is per-context state: no depth
// attributes / map / rules (one map per context), one image, the fused FILTER tails (the stand-alone FILTER and
// result-packing kernels are per frame), no stage timing, no graph replay.  MH_MERGE_BATCH=0: frame after frame.
// Per-query depth ATTRIBUTES (mh_frame_set_depth: B Q entries, frame after frame like the queries) travel with a merged
// batch where the caller says so (`attrs_ok`: mh_frame_enqueue_batch) -- group_kernel takes frame f's slice, the per-frame
// arenas hold every frame's m_depth, pose_kernel<1 | 2> shifts its pointers like pose_kernel<0>.
// A depth MAP per frame (mh_frame_set_depth_image_batch with as many maps as the batch has frames: `maps_for`), the depth
// rules and the linkage clusterer travel with it too (round 4): depth_patch / feature_density / group / linkage_models
// take frame f's map from a DepthMaps table and its rule buffers behind those of the frames before it.
bool merged_batch_ok(const mh_ctx* ctx, const mh_frame_params* prm, bool attrs_ok = false, int maps_for = 0) {
  static const bool on = exp_int("MH_MERGE_BATCH", 1) != 0;
  static const bool fuse_filter = exp_int("MH_FUSE_FILTER", 1) != 0;
  static const bool merge_maps = exp_int("MH_MERGE_MAPS", 1) != 0;
  const bool maps_ok = merge_maps && attrs_ok && maps_for > 1 && ctx->batch_imgs == maps_for && ctx->depth_img.img;
  return on && fuse_filter && prm->run_stage2 && !ctx->timing && (attrs_ok || !ctx->q_depth) &&
         (maps_ok || (!ctx->depth_img.img && !ctx->rules.on && !ctx->linkage_on)) && !(ctx->q_img && ctx->n_images > 1);
}

int ensure_batch_arenas(mh_ctx* ctx, int B) {
  FrameState* fs = ctx->fs;
  if (fs->n_arenas >= B) return MH_OK;   // (the usual case; the first batch pays one reallocation)
  return ensure_fs(ctx, fs->max_m, fs->max_clusters, fs->max_objects, fs->n_models_cap, B);
}

// Buffers for a launch of Q queries = `frames` frames of q_frame queries each (0: one frame of Q): the MATCH side for
// all of them, the working arrays of the rest chain per frame (a frame has at most as many matches as queries) with
// `frames` copies, so that a merged batch never reallocates behind work that is already enqueued.
int prepare_frame(mh_ctx* ctx, int Q, int q_frame = 0, int frames = 1) {
  int rc = ensure_frame_buffers(ctx, Q);
  if (rc) return rc;
  if ((rc = ensure_match_scratch(ctx, Q))) return rc;
  const int want_m = ctx->fs ? ctx->fs->max_m : 0;
  const int mc = ctx->fs ? ctx->fs->max_clusters : 1024;
  const int mo = ctx->fs ? ctx->fs->max_objects : 4096;
  // (several images: CLUSTER's per-"model" tables hold one entry per (model, image) pair)
  return ensure_fs(ctx, std::max(want_m, q_frame > 0 ? q_frame : Q), mc, mo,
                   ctx->n_models * (ctx->n_images > 1 ? ctx->n_images : 1), frames);
}

// mh_frame_fetch_batch_async: frame f's head {n, flags, counts[4], tag, f} + its first min(n, max_objects) objects out of
// result slot f into record f of the caller's block (dst: device-visible host memory, or the staging buffer).
__global__ void __launch_bounds__(128) deliver_kernel(const unsigned char* __restrict__ result, size_t result_bytes,
                                                      const int32_t* __restrict__ snap, unsigned char* __restrict__ dst,
                                                      int max_objects, int result_objects, uint32_t tag) {
  const int f = blockIdx.x;
  const int32_t* src = reinterpret_cast<const int32_t*>(result + (size_t)f * result_bytes);
  int32_t* out = reinterpret_cast<int32_t*>(dst + (size_t)f * (sizeof(mh_frame_head) + sizeof(mh_object) * (size_t)max_objects));
  const int n = src[0];
  const int take = max(0, min(n, min(max_objects, result_objects)));
  const int t = threadIdx.x;
  if (t < 8) {
    int32_t w;
    if (t == 0) w = n;
    else if (t == 1) w = src[1];
    else if (t < 6) w = snap[4 * f + t - 2];
    else if (t == 6) w = (int32_t)tag;
    else w = f;
    out[t] = w;
  }
  constexpr int OW = (int)(sizeof(mh_object) / 4);
  for (int w = t; w < take * OW; w += blockDim.x) out[8 + w] = src[4 + w];
  __threadfence_system();
}

}  // namespace

namespace mh {

int delivery_begin(mh_ctx* ctx, void* host_block, size_t bytes, unsigned char** dst_dev) {
  auto& d = ctx->dlv;
  if (d.pending) {
    ctx->err = "delivery: the context's previous delivery has not been waited for (mh_frame_fetch_wait)";
    return MH_ERR_ARG;
  }
  if (!d.done) MH_HIP(ctx, hipEventCreateWithFlags(&d.done, hipEventDisableTiming));
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, host_block, 0) == hipSuccess && dev) {
    *dst_dev = static_cast<unsigned char*>(dev);
    return MH_OK;
  }
  (void)hipGetLastError();   // not pinned / not mapped: through the staging buffer
  if (d.stage_cap < bytes) {
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (d.stage) hipFree(d.stage);
    d.stage = nullptr;
    d.stage_cap = 0;
    MH_HIP(ctx, hipMalloc(&d.stage, bytes));
    d.stage_cap = bytes;
  }
  *dst_dev = d.stage;
  return MH_OK;
}

int delivery_end(mh_ctx* ctx, void* host_block, size_t bytes, unsigned char* dst_dev, int B, int max_objects, uint32_t tag) {
  auto& d = ctx->dlv;
  MH_HIP(ctx, hipGetLastError());
  if (dst_dev == d.stage)
    MH_HIP(ctx, hipMemcpyAsync(host_block, d.stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipEventRecord(d.done, ctx->stream));
  d.pending = true;
  d.host_block = static_cast<unsigned char*>(host_block);
  d.B = B;
  d.max_objects = max_objects;
  d.tag = tag;
  return MH_OK;
}

}  // namespace mh

extern "C" {

size_t mh_frame_block_stride(int max_objects) {
  return sizeof(mh_frame_head) + sizeof(mh_object) * (size_t)(max_objects > 0 ? max_objects : 0);
}

int mh_frame_fetch_batch_async(mh_ctx* ctx, int B, int max_objects, void* host_block, uint32_t tag) {
  static_assert(sizeof(mh_frame_head) == 32 && sizeof(mh_object) == 40, "block layout");
  if (!ctx || !ctx->fs || !host_block || B < 1 || B > MH_MAX_BATCH || max_objects < 0) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  FrameState* fs = ctx->fs;
  const size_t bytes = mh_frame_block_stride(max_objects) * (size_t)B;
  unsigned char* dst = nullptr;
  if (int rc = delivery_begin(ctx, host_block, bytes, &dst)) return rc;
  hipLaunchKernelGGL(deliver_kernel, dim3(B), dim3(128), 0, ctx->stream, fs->result, fs->result_bytes, fs->snap, dst,
                     max_objects, fs->max_objects, tag);
  return delivery_end(ctx, host_block, bytes, dst, B, max_objects, tag);
}

int mh_frame_fetch_query(mh_ctx* ctx) {
  if (!ctx) return MH_ERR_ARG;
  if (!ctx->dlv.pending) return MH_OK;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  const hipError_t e = hipEventQuery(ctx->dlv.done);
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    return 1;
  }
  MH_HIP(ctx, e);
  return MH_OK;
}

int mh_frame_fetch_wait(mh_ctx* ctx, int32_t* flags_or) {
  if (!ctx) return MH_ERR_ARG;
  if (flags_or) *flags_or = 0;
  auto& d = ctx->dlv;
  if (!d.pending) return MH_OK;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  MH_HIP(ctx, hipEventSynchronize(d.done));
  d.pending = false;
  int32_t flags = 0;
  const size_t stride = mh_frame_block_stride(d.max_objects);
  for (int f = 0; f < d.B; ++f) {
    const mh_frame_head* h = reinterpret_cast<const mh_frame_head*>(d.host_block + (size_t)f * stride);
    flags |= h->flags;
    if (h->tag != d.tag || h->frame != f) {   // the block was written by something else in the meantime
      ctx->err = "mh_frame_fetch_wait: the host block does not hold the delivery that was enqueued into it";
      return MH_ERR_ARG;
    }
  }
  if (flags_or) *flags_or = flags;
  if (flags) {
    ctx->err = (flags & ERR_EXCHANGE)
                   ? std::string("frame exchange: the ranks' blocks carry different sequence numbers / seeds -- the ranks issued "
                                 "their collectives in different orders (every rank must enqueue its slots in the same order)")
                   : "frame: capacity exceeded (flags " + std::to_string(flags) + ")";
    return MH_ERR_CAPACITY;
  }
  return MH_OK;
}

void mh_free_frame_state(mh_ctx* ctx) {
  free_fs(ctx->fs);
  ctx->fs = nullptr;
}

int mh_reserve(mh_ctx* ctx, int max_queries, int max_clusters, int max_objects) {
  if (!ctx || max_queries <= 0 || max_clusters <= 0 || max_objects <= 0) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  int rc = ensure_frame_buffers(ctx, max_queries);
  if (rc) return rc;
  return ensure_fs(ctx, ctx->max_q, max_clusters, max_objects, ctx->n_models);
}

int mh_reserve_batch(mh_ctx* ctx, int queries_per_frame, int frames, int max_clusters, int max_objects) {
  if (!ctx || queries_per_frame <= 0 || frames < 1 || frames > MH_MAX_BATCH || max_clusters <= 0 || max_objects <= 0)
    return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  int rc = ensure_frame_buffers(ctx, queries_per_frame * frames);
  if (rc) return rc;
  if ((rc = ensure_match_scratch(ctx, queries_per_frame * frames))) return rc;
  return ensure_fs(ctx, queries_per_frame, max_clusters, max_objects, ctx->n_models, frames);
}

void mh_frame_default_params(mh_frame_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof *p);
  p->ratio = 0.8f;            // config.hpp:83
  p->ms_radius = 200.f;       // config.hpp:101
  p->ms_merge = 20.f;
  p->ms_min_pts = 7;
  p->ms_max_iter = 100;
  // (LM caps: 2 iterations on plain residuals -- a warm start; the squared-residual phase, Newton-scaled since round 4,
  //  converges from there as fast as from a converged plain phase: single frame 0.70 -> 0.675 ms, frame_stress unchanged)
  p->pose1 = {1024, 4, 5, 6, 10.f, 2, 10};   // config.hpp:110 (…, 4, 5, 6, 10)
  p->f1_min_points = 5;       // config.hpp:115
  p->f1_feature_distance = 4096.f;
  p->f1_min_score = 2.f;
  p->pose2 = {1024, 4, 6, 8, 5.f, 2, 10};    // config.hpp:118 (…, 4, 6, 8, 5)
  p->f2_min_points = 7;       // config.hpp:120
  p->f2_feature_distance = 4096.f;
  p->f2_min_score = 3.f;
  p->run_stage2 = 1;
}

int mh_meanshift(mh_ctx* ctx, const float* pts_host, int n, int dim, float radius, float merge,
                 int min_pts, int max_iter, int32_t* label, int32_t* order, int32_t* n_clusters) {
  if (!ctx || n < 0 || (dim != 2 && dim != 3) || !n_clusters || (n > 0 && (!pts_host || !label))) {
    if (ctx) ctx->err = "mh_meanshift: bad argument";
    return MH_ERR_ARG;
  }
  *n_clusters = 0;
  if (n == 0) return MH_OK;
  if (n > MS_CAP) {
    ctx->err = "mh_meanshift: more than 2048 points";
    return MH_ERR_CAPACITY;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  const size_t b_pts = (size_t)n * dim * sizeof(float);
  const size_t b_i = (size_t)(n + 2) * sizeof(int32_t);
  int rc = ensure_scratch(ctx, b_pts + 4 * b_i + 64);
  if (rc) return rc;
  unsigned char* base = (unsigned char*)ctx->scratch;
  float* d_pts = (float*)base;
  int32_t* d_members = (int32_t*)(base + ((b_pts + 15) & ~(size_t)15));
  int32_t* d_start = d_members + (n + 2);
  int32_t* d_label = d_start + (n + 2);
  int32_t* d_misc = d_label + (n + 2);  // [0] ncl, [1] iterations
  MH_HIP(ctx, hipMemcpyAsync(d_pts, pts_host, b_pts, hipMemcpyHostToDevice, ctx->stream));
  launch_meanshift_single(d_pts, n, dim, radius, merge, min_pts, max_iter, d_members, d_start,
                          d_misc, d_label, d_misc + 1, ctx->stream);
  MH_HIP(ctx, hipGetLastError());
  int32_t misc[2];
  MH_HIP(ctx, hipMemcpyAsync(misc, d_misc, sizeof misc, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipMemcpyAsync(label, d_label, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n_clusters = misc[0];
  if (order) {
    std::vector<int32_t> st(misc[0] + 1);
    MH_HIP(ctx, hipMemcpy(st.data(), d_start, st.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    const int total = st[misc[0]];
    for (int i = 0; i < n; ++i) order[i] = -1;
    if (total > 0)
      MH_HIP(ctx, hipMemcpy(order, d_members, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  return MH_OK;
}

int mh_meanshift_batch(mh_ctx* ctx, const float* pts_host, const int32_t* off, int n_problems, int dim,
                       float radius, float merge, int min_pts, int max_iter, int32_t* label,
                       int32_t* order, int32_t* n_clusters) {
  if (!ctx || n_problems < 0 || (dim != 2 && dim != 3) || (n_problems > 0 && (!off || !n_clusters))) {
    if (ctx) ctx->err = "mh_meanshift_batch: bad argument";
    return MH_ERR_ARG;
  }
  if (n_problems == 0) return MH_OK;
  const int total = off[n_problems];
  for (int p = 0; p < n_problems; ++p) {
    n_clusters[p] = 0;
    const int n = off[p + 1] - off[p];
    if (n < 0 || off[0] != 0) {
      ctx->err = "mh_meanshift_batch: offsets must start at 0 and not decrease";
      return MH_ERR_ARG;
    }
    if (n > MS_CAP) {
      ctx->err = "mh_meanshift_batch: more than 2048 points in one problem";
      return MH_ERR_CAPACITY;
    }
  }
  if (total == 0) return MH_OK;
  if (!pts_host || !label) {
    ctx->err = "mh_meanshift_batch: bad argument";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  // device layout: pts | off | members | label | cl_start (total + n_problems + 1) | ncl
  const size_t b_pts = ((size_t)total * dim * sizeof(float) + 15) & ~(size_t)15;
  const size_t n_off = (size_t)n_problems + 1;
  const size_t n_start = (size_t)total + n_problems + 1;
  const size_t ints = n_off + 2 * (size_t)total + n_start + n_problems;
  int rc = ensure_scratch(ctx, b_pts + ints * sizeof(int32_t) + 64);
  if (rc) return rc;
  if ((rc = ensure_pinned(ctx, (2 * (size_t)total + n_start + n_problems) * sizeof(int32_t)))) return rc;
  unsigned char* base = (unsigned char*)ctx->scratch;
  float* d_pts = (float*)base;
  int32_t* d_off = (int32_t*)(base + b_pts);
  int32_t* d_members = d_off + n_off;
  int32_t* d_label = d_members + total;
  int32_t* d_start = d_label + total;
  int32_t* d_ncl = d_start + n_start;
  hipStream_t s = ctx->stream;
  MH_HIP(ctx, hipMemcpyAsync(d_pts, pts_host, (size_t)total * dim * sizeof(float), hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemcpyAsync(d_off, off, n_off * sizeof(int32_t), hipMemcpyHostToDevice, s));
  launch_meanshift_batch(d_pts, d_off, n_problems, dim, radius, merge, min_pts, max_iter, d_members, d_start,
                         d_ncl, d_label, s);
  MH_HIP(ctx, hipGetLastError());
  // members, label, cl_start, ncl are contiguous on the device: one copy back
  int32_t* h = (int32_t*)ctx->pinned;
  MH_HIP(ctx, hipMemcpyAsync(h, d_members, (2 * (size_t)total + n_start + n_problems) * sizeof(int32_t),
                             hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipStreamSynchronize(s));
  const int32_t* h_members = h;
  const int32_t* h_label = h + total;
  const int32_t* h_start = h_label + total;
  const int32_t* h_ncl = h_start + n_start;
  for (int p = 0; p < n_problems; ++p) {
    const int b = off[p], n = off[p + 1] - b;
    n_clusters[p] = n > 0 ? h_ncl[p] : 0;
    for (int i = 0; i < n; ++i) label[b + i] = h_label[b + i];
    if (order) {
      for (int i = 0; i < n; ++i) order[b + i] = -1;
      if (n > 0) {
        const int kept = h_start[b + p + h_ncl[p]];
        for (int i = 0; i < kept; ++i) order[b + i] = h_members[b + i];
      }
    }
  }
  return MH_OK;
}

// the frame's cameras into the context's device table
static int upload_cams(mh_ctx* ctx, const mh_cam* cams, int n_images) {
  if (!ctx->cams_dev) MH_HIP(ctx, hipMalloc(&ctx->cams_dev, sizeof(DevCam) * MH_MAX_IMAGES));
  DevCam h[MH_MAX_IMAGES];
  for (int i = 0; i < n_images; ++i) h[i] = make_devcam(cams[i]);
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // frames in flight still read the old table
  MH_HIP(ctx, hipMemcpy(ctx->cams_dev, h, sizeof(DevCam) * n_images, hipMemcpyHostToDevice));
  return MH_OK;
}

static int pose_ransac_impl(mh_ctx* ctx, const mh_corr* corr_host, const mh_depth* depth_host, int kind,
                            float alpha, const int32_t* cluster_off, int n_clusters, const mh_cam* cam,
                            const mh_pose_params* prm, uint64_t seed, mh_pose_out* out_host,
                            int32_t* n_out, const int32_t* image_of_host = nullptr, int n_images = 1) {
  if (!ctx || n_clusters < 0 || !cam || !prm || !n_out || (n_clusters > 0 && (!corr_host || !cluster_off || !out_host)) ||
      n_images < 1 || n_images > MH_MAX_IMAGES) {
    if (ctx) ctx->err = "mh_pose_ransac: bad argument";
    return MH_ERR_ARG;
  }
  *n_out = 0;
  if (n_clusters == 0) return MH_OK;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  const int R_ = prm->max_objects_per_cluster > 0 ? prm->max_objects_per_cluster : 1;
  const int total = cluster_off[n_clusters];
  const int n_obj = n_clusters * R_;
  ctx->step.done = -1;   // (the frame's working arrays are this call's now)
  int rc = ensure_fs(ctx, std::max(total, 1), n_clusters, n_obj, std::max(ctx->n_models, 1));
  if (rc) return rc;
  FrameState* fs = ctx->fs;
  hipStream_t s = ctx->stream;
  std::vector<int32_t> h_members(std::max(total, 1)), h_model(n_clusters, 0), h_begin(n_clusters), h_count(n_clusters);
  for (int i = 0; i < total; ++i) h_members[i] = i;
  for (int c = 0; c < n_clusters; ++c) {
    h_model[c] = c;   // the step-level call has no models: every cluster its own (keys the task's random stream)
    h_begin[c] = cluster_off[c];
    h_count[c] = cluster_off[c + 1] - cluster_off[c];
  }
  PoseImages images;
  if (image_of_host && n_images > 1) {   // every correspondence in its own image: cam[0..n_images)
    for (int i = 0; i < total; ++i)
      if (image_of_host[i] < 0 || image_of_host[i] >= n_images) {
        ctx->err = "mh_pose_ransac_images: image index outside [0, n_images)";
        return MH_ERR_ARG;
      }
    if ((rc = upload_cams(ctx, cam, n_images))) return rc;
    MH_HIP(ctx, hipMemcpyAsync(fs->m_img, image_of_host, (size_t)total * sizeof(int32_t), hipMemcpyHostToDevice, s));
    images.cams = ctx->cams_dev;
    images.img_of = fs->m_img;
    images.n_images = n_images;
  }
  MH_HIP(ctx, hipMemcpyAsync(fs->m_corr, corr_host, (size_t)total * sizeof(mh_corr), hipMemcpyHostToDevice, s));
  if (depth_host)
    MH_HIP(ctx, hipMemcpyAsync(fs->m_depth, depth_host, (size_t)total * sizeof(mh_depth), hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemcpyAsync(fs->ms_members, h_members.data(), (size_t)total * 4, hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemcpyAsync(fs->cl_model, h_model.data(), (size_t)n_clusters * 4, hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemcpyAsync(fs->cl_begin, h_begin.data(), (size_t)n_clusters * 4, hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemcpyAsync(fs->cl_count, h_count.data(), (size_t)n_clusters * 4, hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemsetAsync(fs->counts, 0, sizeof(FrameCounts), s));
  hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, s, fs->n_clusters, n_clusters);
  hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, s, fs->n_slots, 0);
  const DevCam dc = make_devcam(*cam);
  launch_pose(fs->m_corr, depth_host ? reinterpret_cast<const float*>(fs->m_depth) : nullptr, kind, alpha,
              fs->ms_members, fs->cl_model, fs->cl_begin, fs->cl_count, fs->n_clusters, n_clusters, dc, *prm, seed, fs->n_slots, fs->max_objects, fs->obj_model, fs->obj_pose,
              fs->obj_ninl, fs->obj_err, fs->obj_cluster, fs->obj_valid, fs->counts, PoseTail{nullptr, nullptr, nullptr, 0, nullptr}, s,
              images);
  MH_HIP(ctx, hipGetLastError());
  // the five result arrays into one pinned block (pageable destinations make every one of these copies a blocking one)
  if (int rc_pin = ensure_pinned(ctx, (size_t)n_obj * 11 * 4)) return rc_pin;
  int32_t* const valid = static_cast<int32_t*>(ctx->pinned);
  int32_t* const ninl = valid + n_obj;
  int32_t* const ocl = ninl + n_obj;
  float* const err = reinterpret_cast<float*>(ocl + n_obj);
  float* const pose = err + n_obj;
  MH_HIP(ctx, hipMemcpyAsync(valid, fs->obj_valid, (size_t)n_obj * 4, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipMemcpyAsync(ninl, fs->obj_ninl, (size_t)n_obj * 4, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipMemcpyAsync(ocl, fs->obj_cluster, (size_t)n_obj * 4, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipMemcpyAsync(pose, fs->obj_pose, (size_t)n_obj * 28, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipMemcpyAsync(err, fs->obj_err, (size_t)n_obj * 4, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipStreamSynchronize(s));
  int k = 0;
  for (int o = 0; o < n_obj; ++o) {
    if (!valid[o]) continue;
    mh_pose_out& po = out_host[k++];
    std::memcpy(po.pose, &pose[(size_t)7 * o], 28);
    po.cluster = ocl[o];
    po.n_inliers = ninl[o];
    po.err = err[o];
  }
  *n_out = k;
  return MH_OK;
}

int mh_pose_ransac(mh_ctx* ctx, const mh_corr* corr_host, const int32_t* cluster_off,
                   int n_clusters, const mh_cam* cam, const mh_pose_params* prm, uint64_t seed,
                   mh_pose_out* out_host, int32_t* n_out) {
  return pose_ransac_impl(ctx, corr_host, nullptr, MH_DEPTH_NONE, 0.f, cluster_off, n_clusters, cam, prm, seed,
                          out_host, n_out);
}

int mh_pose_ransac_images(mh_ctx* ctx, const mh_corr* corr_host, const int32_t* image_of_host,
                          const int32_t* cluster_off, int n_clusters, const mh_cam* cams, int n_images,
                          const mh_pose_params* prm, uint64_t seed, mh_pose_out* out_host, int32_t* n_out) {
  if (n_clusters > 0 && n_images > 1 && !image_of_host) return MH_ERR_ARG;
  return pose_ransac_impl(ctx, corr_host, nullptr, MH_DEPTH_NONE, 0.f, cluster_off, n_clusters, cams, prm, seed,
                          out_host, n_out, image_of_host, n_images);
}

int mh_pose_ransac_depth(mh_ctx* ctx, const mh_corr* corr_host, const mh_depth* depth_host,
                         const int32_t* cluster_off, int n_clusters, const mh_cam* cam,
                         const mh_pose_params* prm, int kind, float alpha, uint64_t seed,
                         mh_pose_out* out_host, int32_t* n_out) {
  if (kind != MH_DEPTH_BACKPROJECTION && kind != MH_DEPTH_REPROJECTION) return MH_ERR_ARG;
  if (n_clusters > 0 && !depth_host) return MH_ERR_ARG;
  return pose_ransac_impl(ctx, corr_host, depth_host, kind, alpha, cluster_off, n_clusters, cam, prm, seed,
                          out_host, n_out);
}

int mh_frame_set_images(mh_ctx* ctx, const int32_t* q_image_dev, const mh_cam* cams, int n_images) {
  if (!ctx) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  if (!q_image_dev || !cams || n_images <= 1) {   // back to one image: the camera of mh_frame_enqueue*
    ctx->q_img = nullptr;
    ctx->n_images = 1;
    return MH_OK;
  }
  if (n_images > MH_MAX_IMAGES) {
    ctx->err = "mh_frame_set_images: more than MH_MAX_IMAGES images";
    return MH_ERR_CAPACITY;
  }
  if (int rc = upload_cams(ctx, cams, n_images)) return rc;
  ctx->q_img = q_image_dev;
  ctx->n_images = n_images;
  return MH_OK;
}

int mh_frame_set_depth(mh_ctx* ctx, const mh_depth* q_depth_dev, int kind, float alpha) {
  if (!ctx) return MH_ERR_ARG;
  if (q_depth_dev && kind != MH_DEPTH_BACKPROJECTION && kind != MH_DEPTH_REPROJECTION) return MH_ERR_ARG;
  ctx->q_depth = q_depth_dev;
  ctx->depth_img = DepthImage{};
  ctx->depth_kind = q_depth_dev ? kind : MH_DEPTH_NONE;
  ctx->depth_alpha = alpha;
  return MH_OK;
}

int mh_frame_set_depth_image(mh_ctx* ctx, const float* depth_xyzn_dev, const float* fill_distance_dev, int width,
                             int height, int kind, float alpha, float cauchy_scale) {
  if (!ctx) return MH_ERR_ARG;
  if (depth_xyzn_dev && ((kind != MH_DEPTH_BACKPROJECTION && kind != MH_DEPTH_REPROJECTION) || width <= 0 ||
                         height <= 0 || !(cauchy_scale > 0.f)))
    return MH_ERR_ARG;
  ctx->q_depth = nullptr;
  ctx->depth_img = DepthImage{};
  ctx->batch_imgs = 0;
  if (depth_xyzn_dev) {
    ctx->depth_img.img = reinterpret_cast<const float4*>(depth_xyzn_dev);
    ctx->depth_img.fill = fill_distance_dev;
    ctx->depth_img.w = width;
    ctx->depth_img.h = height;
    ctx->depth_img.cauchy_scale = cauchy_scale;
  }
  ctx->depth_kind = depth_xyzn_dev ? kind : MH_DEPTH_NONE;
  ctx->depth_alpha = alpha;
  return MH_OK;
}

int mh_frame_set_depth_image_batch(mh_ctx* ctx, const float* const* depth_xyzn_dev, const float* const* fill_distance_dev,
                                   int n_frames, int width, int height, int kind, float alpha, float cauchy_scale) {
  if (!ctx || !depth_xyzn_dev || n_frames < 1 || n_frames > MH_MAX_BATCH) return MH_ERR_ARG;
  for (int f = 0; f < n_frames; ++f)
    if (!depth_xyzn_dev[f]) return MH_ERR_ARG;
  const int rc = mh_frame_set_depth_image(ctx, depth_xyzn_dev[0], fill_distance_dev ? fill_distance_dev[0] : nullptr, width,
                                          height, kind, alpha, cauchy_scale);
  if (rc) return rc;
  for (int f = 0; f < n_frames; ++f) {
    ctx->batch_img[f] = reinterpret_cast<const float4*>(depth_xyzn_dev[f]);
    ctx->batch_fill[f] = fill_distance_dev ? fill_distance_dev[f] : nullptr;
  }
  ctx->batch_imgs = n_frames;
  return MH_OK;
}

int mh_frame_set_depth_image_host(mh_ctx* ctx, const float* depth_xyzn_host, const float* fill_distance_host,
                                  int width, int height, int kind, float alpha, float cauchy_scale) {
  if (!ctx) return MH_ERR_ARG;
  if (!depth_xyzn_host) return mh_frame_set_depth_image(ctx, nullptr, nullptr, 0, 0, 0, alpha, cauchy_scale);
  if (width <= 0 || height <= 0) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  const size_t px = (size_t)width * height;
  if (px > ctx->own_depth_px) {
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_depth) MH_HIP(ctx, hipFree(ctx->own_depth));
    if (ctx->own_fill) MH_HIP(ctx, hipFree(ctx->own_fill));
    ctx->own_depth = ctx->own_fill = nullptr;
    ctx->own_depth_px = 0;
    MH_HIP(ctx, hipMalloc(&ctx->own_depth, px * 4 * sizeof(float)));
    MH_HIP(ctx, hipMalloc(&ctx->own_fill, px * sizeof(float)));
    ctx->own_depth_px = px;
  }
  MH_HIP(ctx, hipMemcpyAsync(ctx->own_depth, depth_xyzn_host, px * 4 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  if (fill_distance_host)
    MH_HIP(ctx, hipMemcpyAsync(ctx->own_fill, fill_distance_host, px * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the host maps may change after the call
  return mh_frame_set_depth_image(ctx, ctx->own_depth, fill_distance_host ? ctx->own_fill : nullptr, width, height, kind,
                                  alpha, cauchy_scale);
}

int mh_set_linkage_scratch_limit(mh_ctx* ctx, size_t bytes) {
  if (!ctx) return MH_ERR_ARG;
  ctx->lk_scratch_limit = bytes ? bytes : (size_t)4 << 30;
  return MH_OK;
}

int mh_frame_set_cluster_linkage(mh_ctx* ctx, const mh_linkage_params* prm) {
  if (!ctx) return MH_ERR_ARG;
  if (prm && (prm->linkage_type < 0 || prm->linkage_type > 2)) {   // (before anything of the context changes)
    ctx->err = "mh_frame_set_cluster_linkage: linkage_type must be 0 (minimum), 1 (average) or 2 (maximum)";
    return MH_ERR_ARG;
  }
  ctx->linkage_on = prm != nullptr;
  if (prm) {
    ctx->linkage.cutoff = prm->cutoff;
    ctx->linkage.min_pts = prm->min_pts;
    ctx->linkage.use3d_filter = prm->use3d_filter;
    ctx->linkage.sigma2d = prm->sigma2d;
    ctx->linkage.sigma3d = prm->sigma3d;
    ctx->linkage.linkage_type = prm->linkage_type;
  }
  return MH_OK;
}

int mh_cluster_linkage(mh_ctx* ctx, const mh_corr* corr_host, const mh_depth* depth_host, const int32_t* off,
                       int n_problems, const mh_linkage_params* prm, int32_t* label, int32_t* order,
                       int32_t* n_clusters) {
  if (!ctx || n_problems < 0 || !prm || (n_problems > 0 && (!off || !n_clusters))) {
    if (ctx) ctx->err = "mh_cluster_linkage: bad argument";
    return MH_ERR_ARG;
  }
  if (prm->linkage_type < 0 || prm->linkage_type > 2) {   // (before any upload is enqueued)
    ctx->err = "mh_cluster_linkage: linkage_type must be 0 (minimum), 1 (average) or 2 (maximum)";
    return MH_ERR_ARG;
  }
  if (n_problems == 0) return MH_OK;
  if (!ctx->depth_img.img) {
    ctx->err = "mh_cluster_linkage: no depth map (mh_frame_set_depth_image)";
    return MH_ERR_ARG;
  }
  const int total = off[n_problems];
  size_t need = 0;
  for (int p = 0; p < n_problems; ++p) {
    n_clusters[p] = 0;
    const int n = off[p + 1] - off[p];
    if (n < 0 || off[0] != 0) {
      ctx->err = "mh_cluster_linkage: offsets must start at 0 and not decrease";
      return MH_ERR_ARG;
    }
    if (n > LK_CAP) {
      ctx->err = "mh_cluster_linkage: more than 1024 points in one problem";
      return MH_ERR_CAPACITY;
    }
    need += 3 * (size_t)n * n;
  }
  if (total == 0) return MH_OK;
  if (!corr_host || !depth_host || !label) {
    ctx->err = "mh_cluster_linkage: bad argument";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  int rc = ensure_linkage_scratch(ctx, need);
  if (rc) return rc;
  // device layout: corr | depth | off | members | label | cl_start (total + n_problems + 1) | ncl
  const size_t b_corr = ((size_t)total * sizeof(mh_corr) + 15) & ~(size_t)15;
  const size_t b_depth = (size_t)total * sizeof(mh_depth);
  const size_t n_off = (size_t)n_problems + 1, n_start = (size_t)total + n_problems + 1;
  const size_t ints = n_off + 2 * (size_t)total + n_start + n_problems;
  if ((rc = ensure_scratch(ctx, b_corr + b_depth + ints * sizeof(int32_t) + 64))) return rc;
  if ((rc = ensure_pinned(ctx, (2 * (size_t)total + n_start + n_problems) * sizeof(int32_t)))) return rc;
  unsigned char* base = (unsigned char*)ctx->scratch;
  mh_corr* d_corr = (mh_corr*)base;
  float* d_depth = (float*)(base + b_corr);
  int32_t* d_off = (int32_t*)(base + b_corr + b_depth);
  int32_t* d_members = d_off + n_off;
  int32_t* d_label = d_members + total;
  int32_t* d_start = d_label + total;
  int32_t* d_ncl = d_start + n_start;
  hipStream_t s = ctx->stream;
  MH_HIP(ctx, hipMemcpyAsync(d_corr, corr_host, (size_t)total * sizeof(mh_corr), hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemcpyAsync(d_depth, depth_host, b_depth, hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemcpyAsync(d_off, off, n_off * sizeof(int32_t), hipMemcpyHostToDevice, s));
  LinkageParams lp;
  lp.cutoff = prm->cutoff;
  lp.min_pts = prm->min_pts;
  lp.use3d_filter = prm->use3d_filter;
  lp.sigma2d = prm->sigma2d;
  lp.sigma3d = prm->sigma3d;
  lp.linkage_type = prm->linkage_type;
  launch_linkage_batch(d_corr, d_depth, d_off, n_problems, ctx->depth_img, lp, ctx->lk_scratch, ctx->lk_scratch_floats,
                       d_members, d_start, d_ncl, d_label, s);
  MH_HIP(ctx, hipGetLastError());
  int32_t* hbuf = (int32_t*)ctx->pinned;
  MH_HIP(ctx, hipMemcpyAsync(hbuf, d_members, (2 * (size_t)total + n_start + n_problems) * sizeof(int32_t),
                             hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipStreamSynchronize(s));
  const int32_t* h_members = hbuf;
  const int32_t* h_label = hbuf + total;
  const int32_t* h_start = h_label + total;
  const int32_t* h_ncl = h_start + n_start;
  for (int p = 0; p < n_problems; ++p) {
    const int b = off[p], n = off[p + 1] - b;
    n_clusters[p] = n > 0 ? h_ncl[p] : 0;
    for (int i = 0; i < n; ++i) label[b + i] = h_label[b + i];
    if (order) {
      for (int i = 0; i < n; ++i) order[b + i] = -1;
      if (n > 0) {
        const int kept = h_start[b + p + h_ncl[p]];
        for (int i = 0; i < kept; ++i) order[b + i] = h_members[b + i];
      }
    }
  }
  return MH_OK;
}

int mh_frame_set_depth_rules(mh_ctx* ctx, const mh_depth_rules* r, const float K[4]) {
  if (!ctx) return MH_ERR_ARG;
  mh_ctx::DepthRuleState& rs = ctx->rules;
  if (!r) {
    rs.on = false;
    return MH_OK;
  }
  const bool filters = r->feature_density >= 0.f || r->match_density >= 0.f;
  if ((filters && (r->patch_size <= 0 || !K)) || (r->ratio_table && (r->n_models <= 0 || !(r->cauchy_scale > 0.f)))) {
    ctx->err = "mh_frame_set_depth_rules: bad argument";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  rs.patch = r->patch_size > 0 ? r->patch_size : 64;
  // `Float filter = Density*100*100` (DEPTHFILTER_CPU.hpp:132)
  rs.feature_filter = r->feature_density >= 0.f ? r->feature_density * 100 * 100 : -1.f;
  rs.match_filter = r->match_density >= 0.f ? r->match_density * 100 * 100 : -1.f;
  for (int i = 0; i < 4; ++i) rs.K[i] = K ? K[i] : 0.f;
  rs.max_depth = r->maximum_depth;
  rs.default_depth = r->default_depth;
  rs.cauchy_scale = r->cauchy_scale;
  if (r->ratio_table) {
    if (r->n_models > rs.table_models) {
      MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (rs.ratio_table) MH_HIP(ctx, hipFree(rs.ratio_table));
      rs.ratio_table = nullptr;
      rs.table_models = 0;
      MH_HIP(ctx, hipMalloc(&rs.ratio_table, sizeof(float) * 4 * r->n_models));
    }
    MH_HIP(ctx, hipMemcpyAsync(rs.ratio_table, r->ratio_table, sizeof(float) * 4 * r->n_models, hipMemcpyHostToDevice,
                               ctx->stream));
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the host table may go away after the call
    rs.table_models = r->n_models;
  } else if (rs.ratio_table) {
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MH_HIP(ctx, hipFree(rs.ratio_table));
    rs.ratio_table = nullptr;
    rs.table_models = 0;
  }
  rs.on = true;
  return MH_OK;
}

int mh_project_test(mh_ctx* ctx, const float pose[7], const mh_corr* corr_host, int n,
                    const mh_cam* cam, float thr, uint8_t* inlier_host, float* err2_host,
                    int32_t* n_inliers) {
  if (!ctx || !pose || !cam || n < 0 || (n > 0 && !corr_host)) return MH_ERR_ARG;
  if (n_inliers) *n_inliers = 0;
  if (n == 0) return MH_OK;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  const size_t b_c = (size_t)n * sizeof(mh_corr);
  int rc = ensure_scratch(ctx, b_c + (size_t)n * 5 + 256);
  if (rc) return rc;
  unsigned char* base = (unsigned char*)ctx->scratch;
  mh_corr* d_c = (mh_corr*)base;
  float* d_e = (float*)(base + ((b_c + 15) & ~(size_t)15));
  float* d_pose = d_e + n;
  int32_t* d_cnt = (int32_t*)(d_pose + 8);
  uint8_t* d_in = (uint8_t*)(d_cnt + 4);
  hipStream_t s = ctx->stream;
  MH_HIP(ctx, hipMemcpyAsync(d_c, corr_host, b_c, hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemcpyAsync(d_pose, pose, 28, hipMemcpyHostToDevice, s));
  launch_project_test(d_pose, d_c, n, make_devcam(*cam), thr, d_in, d_e, d_cnt, s);
  MH_HIP(ctx, hipGetLastError());
  int32_t cnt = 0;
  MH_HIP(ctx, hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, s));
  if (inlier_host) MH_HIP(ctx, hipMemcpyAsync(inlier_host, d_in, n, hipMemcpyDeviceToHost, s));
  if (err2_host) MH_HIP(ctx, hipMemcpyAsync(err2_host, d_e, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipStreamSynchronize(s));
  if (n_inliers) *n_inliers = cnt;
  return MH_OK;
}

int mh_filter(mh_ctx* ctx, const mh_corr* corr_host, const int32_t* model_off, int n_models,
              const int32_t* obj_model, const float* obj_pose, int n_obj, const mh_cam* cam,
              int min_points, float feature_distance, float min_score, float* score,
              uint8_t* keep, int32_t* out_order, int32_t* cl_members, int32_t* cl_off,
              int32_t* n_kept) {
  return mh_filter_images(ctx, corr_host, nullptr, model_off, n_models, obj_model, obj_pose, n_obj, cam, 1, min_points,
                          feature_distance, min_score, score, keep, out_order, cl_members, cl_off, n_kept);
}

int mh_filter_images(mh_ctx* ctx, const mh_corr* corr_host, const int32_t* image_of_host, const int32_t* model_off,
                     int n_models, const int32_t* obj_model, const float* obj_pose, int n_obj, const mh_cam* cam,
                     int n_images, int min_points, float feature_distance, float min_score, float* score,
                     uint8_t* keep, int32_t* out_order, int32_t* cl_members, int32_t* cl_off, int32_t* n_kept) {
  if (!ctx || !model_off || n_models <= 0 || n_obj < 0 || !cam || !n_kept || n_images < 1 || n_images > MH_MAX_IMAGES ||
      (n_images > 1 && !image_of_host))
    return MH_ERR_ARG;
  *n_kept = 0;
  if (cl_off) cl_off[0] = 0;
  if (n_obj == 0) return MH_OK;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  const int M = model_off[n_models];
  ctx->step.done = -1;   // (the frame's working arrays are this call's now)
  int rc = ensure_fs(ctx, std::max(M, 1), std::max(n_obj, 1), std::max(n_obj, 1), n_models);
  if (rc) return rc;
  FrameState* fs = ctx->fs;
  hipStream_t s = ctx->stream;
  std::vector<int32_t> ones(n_obj, 1);
  const bool multi = image_of_host && n_images > 1;
  if (multi) {
    for (int i = 0; i < M; ++i)
      if (image_of_host[i] < 0 || image_of_host[i] >= n_images) {
        ctx->err = "mh_filter_images: image index outside [0, n_images)";
        return MH_ERR_ARG;
      }
    if ((rc = upload_cams(ctx, cam, n_images))) return rc;
    MH_HIP(ctx, hipMemcpyAsync(fs->m_img, image_of_host, (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice, s));
  }
  MH_HIP(ctx, hipMemcpyAsync(fs->m_corr, corr_host, (size_t)M * sizeof(mh_corr), hipMemcpyHostToDevice, s));
  launch_rep(fs->m_corr, M, fs->m_rep, s, multi ? fs->m_img : nullptr);
  MH_HIP(ctx, hipMemcpyAsync(fs->model_off, model_off, (size_t)(n_models + 1) * 4, hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemcpyAsync(fs->obj_model, obj_model, (size_t)n_obj * 4, hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemcpyAsync(fs->obj_pose, obj_pose, (size_t)n_obj * 28, hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemcpyAsync(fs->obj_valid, ones.data(), (size_t)n_obj * 4, hipMemcpyHostToDevice, s));
  MH_HIP(ctx, hipMemsetAsync(fs->counts, 0, sizeof(FrameCounts), s));
  MH_HIP(ctx, hipMemsetAsync(fs->best, 0, sizeof(unsigned long long) * (size_t)std::max(M, 1), s));
  hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, s, fs->n_slots, n_obj);
  FilterBuffers fb = make_fb(ctx, fs, n_models);
  fb.max_objects = n_obj;  // grid size; arrays are at least this large
  if (multi) {
    fb.m_img = fs->m_img;
    fb.cams = ctx->cams_dev;
    fb.n_images = n_images;
  }
  launch_filter(fb, make_devcam(*cam), min_points, feature_distance, min_score, fs->n_slots,
                fs->n_clusters, fs->counts, FilterTail{fs->tickets + 5, nullptr, nullptr, 0, nullptr, nullptr, nullptr}, s);
  MH_HIP(ctx, hipGetLastError());
  // results: everything the host needs in ONE pinned block, copied behind the kernel, one synchronisation (five
  // blocking copies after it cost the step 0.1 ms: profiles/r02_host_step_timing.txt)
  const size_t words = 1 + 4 * (size_t)n_obj + (size_t)std::max(M, 1);
  if ((rc = ensure_pinned(ctx, words * 4))) return rc;
  int32_t* const hp = static_cast<int32_t*>(ctx->pinned);
  int32_t* const h_kept = hp;
  float* const sc = reinterpret_cast<float*>(hp + 1);
  int32_t* const old_of = hp + 1 + n_obj;
  int32_t* const begin = old_of + n_obj;
  int32_t* const count = begin + n_obj;
  int32_t* const mem = count + n_obj;
  MH_HIP(ctx, hipMemcpyAsync(h_kept, fs->n_slots, 4, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipMemcpyAsync(sc, fs->obj_score_raw, (size_t)n_obj * 4, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipMemcpyAsync(old_of, fs->obj_clsize + n_obj, (size_t)n_obj * 4, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipMemcpyAsync(begin, fs->cl_begin, (size_t)std::min(n_obj, fs->max_clusters) * 4, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipMemcpyAsync(count, fs->cl_count, (size_t)std::min(n_obj, fs->max_clusters) * 4, hipMemcpyDeviceToHost, s));
  if (M > 0) MH_HIP(ctx, hipMemcpyAsync(mem, fs->new_members, (size_t)M * 4, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipStreamSynchronize(s));
  const int32_t kept = std::min(*h_kept, n_obj);
  if (keep) std::memset(keep, 0, n_obj);
  if (score)
    for (int o = 0; o < n_obj; ++o) score[o] = sc[o];
  int w = 0;
  for (int k = 0; k < kept; ++k) {
    const int o = old_of[k];
    if (keep) keep[o] = 1;
    if (out_order) out_order[k] = o;
    if (cl_off) cl_off[k] = w;
    const int b = model_off[obj_model[o]];
    for (int j = 0; j < count[k]; ++j, ++w)
      if (cl_members) cl_members[w] = mem[begin[k] + j] - b;  // index inside the model
  }
  if (cl_off) cl_off[kept] = w;
  *n_kept = kept;
  return MH_OK;
}



int mh_frame_enqueue(mh_ctx* ctx, float* q_desc_dev, const float* q_uv_dev, int Q,
                     const mh_cam* cam, const mh_frame_params* prm, uint64_t seed) {
  if (!ctx || Q <= 0 || !q_desc_dev || !q_uv_dev || !cam || !prm) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  int rc = prepare_frame(ctx, Q);
  if (rc) return rc;
  ctx->feat_count_dev = nullptr;
  stamp(ctx, 0);
  launch_normalize(q_desc_dev, ctx->q_norm, Q, ctx->stream);
  if (ctx->wb_want && ctx->wb_ev) hipEventRecord(ctx->wb_ev, ctx->stream);   // (mh_frame_run_host copies them back from here)
  if (int rc_m = ctx_match(ctx, q_desc_dev, ctx->q_norm, Q, ctx->nn_idx, ctx->nn_d1, ctx->nn_d2)) return rc_m;
  stamp(ctx, 1);
  ctx->step.done = -1;   // (whatever a stepped frame left on the device is overwritten from here on)
  return frame_rest(ctx, q_uv_dev, Q, nullptr, 0, cam, prm, seed, nullptr, 1, ctx->stage_lo, ctx->stage_hi);
}

// The whole frame for a host that holds its features in HOST memory and wants the objects back before it goes on: the
// loop body of MopedPimpl::processImages (src/moped.cpp:183-191: MATCH .. FILTER2 one after the other on one FrameData)
// as ONE call -- two uploads, one stream-ordered chain of launches, one synchronisation -- instead of six steps with the
// frame's lists crossing PCIe between them (FRAME_RESIDENT_HIP, moped_amd/host).
int mh_host_alloc(mh_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out || bytes == 0) return MH_ERR_ARG;
  *out = nullptr;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  MH_HIP(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
  return MH_OK;
}

int mh_host_free(mh_ctx* ctx, void* p) {
  if (!ctx) return MH_ERR_ARG;
  if (p) MH_HIP(ctx, hipHostFree(p));
  return MH_OK;
}

int mh_frame_run_host_begin(mh_ctx* ctx, float* q_desc_host, const float* q_uv_host, const int32_t* q_image_host, int Q,
                            const mh_cam* cams, int n_images, const mh_frame_params* prm, uint64_t seed, int write_back) {
  if (!ctx || Q <= 0 || !q_desc_host || !q_uv_host || !cams || n_images < 1 || n_images > MH_MAX_IMAGES || !prm ||
      (n_images > 1 && !q_image_host)) {
    if (ctx) ctx->err = "mh_frame_run_host: bad argument";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  // A write-back of the frame before may still be copying q_desc out on wb_stream (the caller skipped
  // mh_frame_wait_descriptors, or a call failed behind its copy): this frame's upload overwrites that buffer and
  // ensure_frame_buffers may free it -- wait for the stream whatever wb_pending says (nothing in flight: no cost).
  if (ctx->wb_stream) {
    MH_HIP(ctx, hipStreamSynchronize(ctx->wb_stream));
    ctx->wb_pending = false;
  }
  int rc = ensure_frame_buffers(ctx, Q);
  if (rc) return rc;
  MH_HIP(ctx, hipMemcpyAsync(ctx->q_desc, q_desc_host, (size_t)Q * DIM * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  MH_HIP(ctx, hipMemcpyAsync(ctx->q_uv, q_uv_host, (size_t)Q * 2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  if (n_images > 1) {
    if (ctx->hf_img_cap < Q) {
      MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->hf_img) hipFree(ctx->hf_img);
      ctx->hf_img = nullptr;
      ctx->hf_img_cap = 0;
      MH_HIP(ctx, hipMalloc(&ctx->hf_img, (size_t)ctx->max_q * sizeof(int32_t)));
      ctx->hf_img_cap = ctx->max_q;
    }
    MH_HIP(ctx, hipMemcpyAsync(ctx->hf_img, q_image_host, (size_t)Q * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = mh_frame_set_images(ctx, ctx->hf_img, cams, n_images))) return rc;
  } else if (ctx->q_img) {
    if ((rc = mh_frame_set_images(ctx, nullptr, nullptr, 0))) return rc;
  }
  // the reference normalises the frame's descriptors in place (MATCH_ANN_CPU.hpp:157): the copy back starts as soon as
  // normalize_kernel is through -- on a stream of its own, beside the frame's other kernels; behind them on the frame's
  // stream it was another ~0.13 ms at the end of every frame
  if (write_back && !ctx->wb_stream) {
    MH_HIP(ctx, hipStreamCreateWithFlags(&ctx->wb_stream, hipStreamNonBlocking));
    MH_HIP(ctx, hipEventCreateWithFlags(&ctx->wb_ev, hipEventDisableTiming));
  }
  ctx->wb_want = write_back != 0;
  rc = mh_frame_enqueue(ctx, ctx->q_desc, ctx->q_uv, Q, &cams[0], prm, seed);
  ctx->wb_want = false;
  if (rc) return rc;
  if (write_back) {
    MH_HIP(ctx, hipStreamWaitEvent(ctx->wb_stream, ctx->wb_ev, 0));
    ctx->wb_pending = true;   // (set before the copy: an error below must not hide a copy that did start)
    MH_HIP(ctx, hipMemcpyAsync(q_desc_host, ctx->q_desc, (size_t)Q * DIM * sizeof(float), hipMemcpyDeviceToHost, ctx->wb_stream));
  }
  return MH_OK;
}

int mh_frame_wait_descriptors(mh_ctx* ctx) {
  if (!ctx) return MH_ERR_ARG;
  if (ctx->wb_pending) {
    MH_HIP(ctx, hipSetDevice(ctx->device));
    MH_HIP(ctx, hipStreamSynchronize(ctx->wb_stream));   // (the next frame's upload overwrites q_desc)
    ctx->wb_pending = false;
  }
  return MH_OK;
}

int mh_frame_run_host(mh_ctx* ctx, float* q_desc_host, const float* q_uv_host, const int32_t* q_image_host, int Q,
                      const mh_cam* cams, int n_images, const mh_frame_params* prm, uint64_t seed, int write_back,
                      mh_object* objects_host, int max_objects, int32_t* n_objects, int32_t* counts) {
  if (!n_objects) {
    if (ctx) ctx->err = "mh_frame_run_host: bad argument";
    return MH_ERR_ARG;
  }
  int rc = mh_frame_run_host_begin(ctx, q_desc_host, q_uv_host, q_image_host, Q, cams, n_images, prm, seed, write_back);
  if (rc) return rc;
  rc = mh_frame_fetch(ctx, objects_host, max_objects, n_objects, counts);
  const int rc_wb = mh_frame_wait_descriptors(ctx);
  return rc ? rc : rc_wb;
}

int mh_frame_enqueue_image(mh_ctx* ctx, const uint8_t* gray_dev, int width, int height, int double_size,
                           int max_keypoints, const mh_cam* cam, const mh_frame_params* prm, uint64_t seed) {
  if (!ctx || !gray_dev || width <= 0 || height <= 0 || max_keypoints <= 0 || !cam || !prm) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  const int Q = max_keypoints;
  int rc = prepare_frame(ctx, Q);
  if (rc) return rc;
  // FEAT: keypoints straight into the frame's query buffers; their number stays on the device
  int32_t* n_dev = nullptr;
  if ((rc = sift_into(ctx, gray_dev, width, height, double_size, Q, ctx->q_desc, ctx->q_uv, &n_dev))) return rc;
  ctx->feat_count_dev = n_dev;
  stamp(ctx, 0);
  launch_normalize(ctx->q_desc, ctx->q_norm, Q, ctx->stream, n_dev);
  if ((rc = ctx_match(ctx, ctx->q_desc, ctx->q_norm, Q, ctx->nn_idx, ctx->nn_d1, ctx->nn_d2, n_dev, ctx->feat_expected)))
    return rc;
  stamp(ctx, 1);
  return frame_rest(ctx, ctx->q_uv, Q, nullptr, 0, cam, prm, seed);
}

int mh_frame_enqueue_image_batch(mh_ctx* ctx, const uint8_t* const* gray_dev, int B, int width, int height, int double_size,
                                 int max_keypoints, const mh_cam* cam, const mh_frame_params* prm, const uint64_t* seeds) {
  if (!ctx || !gray_dev || B < 1 || B > MH_MAX_BATCH || width <= 0 || height <= 0 || max_keypoints <= 0 || !cam || !prm || !seeds)
    return MH_ERR_ARG;
  for (int f = 0; f < B; ++f)
    if (!gray_dev[f]) return MH_ERR_ARG;
  if (ctx->depth_img.img || ctx->rules.on || ctx->q_depth || (ctx->q_img && ctx->n_images > 1)) {
    ctx->err = "mh_frame_enqueue_image_batch: depth maps / rules / attributes and image indices belong to ONE frame";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  const int Q = max_keypoints;
  int rc = prepare_frame(ctx, B * Q, Q, B > 1 && merged_batch_ok(ctx, prm) ? B : 1);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  if (!ctx->img_counts) {
    MH_HIP(ctx, hipMalloc(&ctx->img_counts, MH_MAX_BATCH * sizeof(int32_t)));
    MH_HIP(ctx, hipMemsetAsync(ctx->img_counts, 0, MH_MAX_BATCH * sizeof(int32_t), s));
  }
  // FEAT image by image into the batch's query rows (image f: rows f Q ..), every image's count in a word of its own
  // (round 4: ONE launch per FEAT stage for all B images -- a frame's 26 dependent launches were what bounded this path)
  if ((rc = sift_into_batch(ctx, gray_dev, B, width, height, double_size, Q, ctx->q_desc, ctx->q_uv, ctx->img_counts)))
    return rc;
  launch_normalize_batch(ctx->q_desc, ctx->q_norm, Q, B, s, ctx->img_counts);   // (one launch: blockIdx.y = image)
  MH_HIP(ctx, hipGetLastError());
  ctx->feat_count_dev = nullptr;
  stamp(ctx, 0);
  hipLaunchKernelGGL(image_batch_tail_kernel, dim3((B * Q * (DIM / 4) + 255) / 256), dim3(256), 0, s, ctx->q_desc, ctx->q_norm,
                     ctx->img_counts, Q, B);
  // ONE MATCH launch sequence over the B images' keypoints
  if ((rc = ctx_match(ctx, ctx->q_desc, ctx->q_norm, B * Q, ctx->nn_idx, ctx->nn_d1, ctx->nn_d2))) return rc;
  hipLaunchKernelGGL(image_batch_mask_kernel, dim3((B * Q + 255) / 256), dim3(256), 0, s, ctx->nn_idx, ctx->nn_d1, ctx->nn_d2,
                     ctx->img_counts, Q, B);
  stamp(ctx, 1);
  if (B > 1 && merged_batch_ok(ctx, prm)) {
    if ((rc = ensure_batch_arenas(ctx, B))) return rc;
    ctx->batch_q0 = 0;
    ctx->fs->slot = 0;
    return frame_rest(ctx, ctx->q_uv, Q, nullptr, 0, cam, prm, seeds[0], seeds, B);
  }
  for (int f = 0; f < B && rc == MH_OK; ++f) {
    ctx->batch_q0 = f * Q;
    ctx->fs->slot = f;
    rc = frame_rest(ctx, ctx->q_uv + 2 * (size_t)f * Q, Q, nullptr, 0, cam, prm, seeds[f]);
  }
  ctx->batch_q0 = 0;
  ctx->fs->slot = 0;
  return rc;
}

int mh_frame_features_dev(mh_ctx* ctx, float** desc_dev, float** uv_dev, int32_t** n_dev) {
  if (!ctx || !ctx->feat_count_dev) return MH_ERR_ARG;
  if (desc_dev) *desc_dev = ctx->q_desc;
  if (uv_dev) *uv_dev = ctx->q_uv;
  if (n_dev) *n_dev = ctx->feat_count_dev;
  return MH_OK;
}

int mh_frame_keypoints(mh_ctx* ctx, int32_t* n_keypoints) {
  if (!ctx || !n_keypoints || ctx->feat_last < 0) return MH_ERR_ARG;
  *n_keypoints = ctx->feat_last;
  return MH_OK;
}

int mh_frame_enqueue_batch(mh_ctx* ctx, float* q_desc_dev, const float* q_uv_dev, int Q, int B, const mh_cam* cam,
                           const mh_frame_params* prm, const uint64_t* seeds) {
  if (!ctx || Q <= 0 || B < 1 || B > MH_MAX_BATCH || !q_desc_dev || !q_uv_dev || !cam || !prm || !seeds)
    return MH_ERR_ARG;
  if (B > 1 && (ctx->depth_img.img || ctx->rules.on) && ctx->batch_imgs != B) {
    ctx->err = "mh_frame_enqueue_batch: a depth map belongs to ONE frame (mh_frame_set_depth_image_batch hands in one per "
               "frame of the batch)";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  const bool merge = B > 1 && merged_batch_ok(ctx, prm, true, B);
  int rc = prepare_frame(ctx, B * Q, Q, merge ? B : 1);   // (the arenas before any work is enqueued)
  if (rc) return rc;
  ctx->feat_count_dev = nullptr;
  stamp(ctx, 0);
  launch_normalize(q_desc_dev, ctx->q_norm, B * Q, ctx->stream);
  if ((rc = ctx_match(ctx, q_desc_dev, ctx->q_norm, B * Q, ctx->nn_idx, ctx->nn_d1, ctx->nn_d2))) return rc;
  stamp(ctx, 1);
  if (merge) {   // the B frames through group / CLUSTER / POSE / POSE2 in one launch each
    if ((rc = ensure_batch_arenas(ctx, B))) return rc;
    ctx->batch_q0 = 0;
    ctx->fs->slot = 0;
    return frame_rest(ctx, q_uv_dev, Q, nullptr, 0, cam, prm, seeds[0], seeds, B);
  }
  for (int f = 0; f < B && rc == MH_OK; ++f) {
    ctx->batch_q0 = f * Q;
    ctx->batch_f = f;
    ctx->fs->slot = f;
    if (B > 1 && ctx->batch_imgs == B && ctx->depth_img.img) {   // the frame's own depth map
      ctx->depth_img.img = ctx->batch_img[f];
      ctx->depth_img.fill = ctx->batch_fill[f];
    }
    rc = frame_rest(ctx, q_uv_dev + 2 * (size_t)f * Q, Q, nullptr, 0, cam, prm, seeds[f]);
  }
  ctx->batch_q0 = 0;
  ctx->batch_f = 0;
  ctx->fs->slot = 0;
  if (B > 1 && ctx->batch_imgs == B && ctx->depth_img.img) {   // back to the first frame's map, as the setter left it
    ctx->depth_img.img = ctx->batch_img[0];
    ctx->depth_img.fill = ctx->batch_fill[0];
  }
  return rc;
}

int mh_frame_enqueue_match_local(mh_ctx* ctx, float* q_desc_dev, int Q, int32_t* top2_dev) {
  if (!ctx || Q <= 0 || !q_desc_dev || !top2_dev) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  // (the working arrays of the rest chain are sized by the calls that run it: a batch's MATCH covers B frames' queries)
  int rc = ensure_frame_buffers(ctx, Q);
  if (rc) return rc;
  if ((rc = ensure_match_scratch(ctx, Q))) return rc;
  if (!ctx->fs && (rc = prepare_frame(ctx, Q))) return rc;
  float* d1 = reinterpret_cast<float*>(top2_dev + Q);
  ctx->feat_count_dev = nullptr;
  stamp(ctx, 0);
  launch_normalize(q_desc_dev, ctx->q_norm, Q, ctx->stream);
  if (int rc_m = ctx_match(ctx, q_desc_dev, ctx->q_norm, Q, top2_dev, d1, d1 + Q)) return rc_m;
  MH_HIP(ctx, hipGetLastError());
  return MH_OK;
}

int mh_frame_enqueue_rest(mh_ctx* ctx, const float* q_uv_dev, int Q, const int32_t* gathered_dev,
                          int n_shards, const mh_cam* cam, const mh_frame_params* prm, uint64_t seed) {
  if (!ctx || Q <= 0 || !q_uv_dev || !gathered_dev || n_shards <= 0 || !cam || !prm) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  int rc = prepare_frame(ctx, Q);
  if (rc) return rc;
  // shard k's block is [3][Q] words at gathered_dev + k*3*Q; the first launch merges them
  // into the context's own top-2 arrays
  stamp(ctx, 1);
  return frame_rest(ctx, q_uv_dev, Q, gathered_dev, n_shards, cam, prm, seed);
}

int mh_frame_fetch(mh_ctx* ctx, mh_object* objects_host, int max_objects, int32_t* n_objects,
                   int32_t* counts) {
  if (!ctx || !n_objects || !ctx->fs) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  FrameState* fs = ctx->fs;
  FrameState::FetchPin& pin = *fs->fetch_pin;
  if (fs->host_armed && !ctx->feat_count_dev) {
    // the frame's FILTER2 wrote the host's block itself: wait for the stream, read
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const FrameHostBlock& h = *fs->host_block;
    const uint32_t seq = h.seq;
    if (seq != fs->host_seq_expect) fs->host_seq_expect = seq;   // (stale block: the stream is idle, the device's count is final -- resynchronise)
    else {
      const int n = h.head[0];
      *n_objects = n;
      if (counts) std::memcpy(counts, h.snap, 4 * sizeof(int32_t));
      const int tasks = 4 * std::max(h.snap[1], h.snap[3]);
      fs->task_grid = std::min(96, std::max(16, (tasks + tasks / 2 + 7) / 8 * 8));
      fs->ms_grid = std::min(32, std::max(4, h.snap[1] + 2));
      const int take = n < max_objects ? n : max_objects;
      if (take > 0 && objects_host) {
        const int have = std::min(take, FRAME_HOST_OBJECTS);
        std::memcpy(objects_host, h.objects, sizeof(mh_object) * (size_t)have);
        if (take > have)
          MH_HIP(ctx, hipMemcpy(objects_host + have, fs->result + 16 + sizeof(mh_object) * (size_t)have,
                                sizeof(mh_object) * (size_t)(take - have), hipMemcpyDeviceToHost));
      }
      if (h.error) {
        ctx->err = (h.error & ERR_EXCHANGE)
                       ? std::string("frame exchange: the ranks' blocks carry different sequence numbers / seeds -- the ranks issued "
                                     "their collectives in different orders (every rank must enqueue its slots in the same order)")
                       : "frame: capacity exceeded (flags " + std::to_string(h.error) + ")";
        return MH_ERR_CAPACITY;
      }
      return MH_OK;
    }
    // (the tail did not run -- a launch failed: the copies below report what there is)
  }
  const size_t first = std::min(fs->result_bytes, sizeof pin.head + sizeof pin.objects);   // head + the first objects: one copy
  MH_HIP(ctx, hipMemcpyAsync(pin.head, fs->result, first, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipMemcpyAsync(pin.snap, fs->snap, sizeof pin.snap, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipMemcpyAsync(&pin.fc, fs->counts, sizeof pin.fc, hipMemcpyDeviceToHost, ctx->stream));
  pin.n_feat = -1;
  if (ctx->feat_count_dev)
    MH_HIP(ctx, hipMemcpyAsync(&pin.n_feat, ctx->feat_count_dev, sizeof pin.n_feat, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int32_t* head = pin.head;
  const int32_t* snap = pin.snap;
  const FrameCounts fc = pin.fc;
  const int32_t n_feat = pin.n_feat;
  if (n_feat >= 0) ctx->feat_expected = ctx->feat_last = n_feat;
  const int n = head[0];
  *n_objects = n;
  if (counts) std::memcpy(counts, snap, 4 * sizeof(int32_t));
  // clusters x 4 replicas (POSE), kept objects x 4 (POSE2), + 50 % head room, in steps of 8
  const int tasks = 4 * std::max(snap[1], snap[3]);
  fs->task_grid = std::min(96, std::max(16, (tasks + tasks / 2 + 7) / 8 * 8));
  fs->ms_grid = std::min(32, std::max(4, snap[1] + 2));   // (a model with matches and no cluster still takes a turn: the workgroups loop)
  const int take = n < max_objects ? n : max_objects;
  if (take > 0 && objects_host) {
    const int have = std::min(take, FETCH_PIN_OBJECTS);
    std::memcpy(objects_host, pin.objects, sizeof(mh_object) * (size_t)have);
    if (take > have)
      MH_HIP(ctx, hipMemcpy(objects_host + have, fs->result + 16 + sizeof(mh_object) * (size_t)have,
                            sizeof(mh_object) * (size_t)(take - have), hipMemcpyDeviceToHost));
  }
  if (fc.error) {
    ctx->err = (fc.error & ERR_EXCHANGE)
                   ? std::string("frame exchange: the ranks' blocks carry different sequence numbers / seeds -- the ranks issued "
                                 "their collectives in different orders (every rank must enqueue its slots in the same order)")
                   : "frame: capacity exceeded (flags " + std::to_string(fc.error) + ")";
    return MH_ERR_CAPACITY;
  }
  return MH_OK;
}

int mh_frame_enqueue_rest_strided(mh_ctx* ctx, const float* q_uv_dev, int Q, const int32_t* gathered_dev,
                                  int n_shards, int shard_stride_words, const mh_cam* cam,
                                  const mh_frame_params* prm, uint64_t seed) {
  if (!ctx || shard_stride_words < 3 * Q) return MH_ERR_ARG;
  ctx->exchange_stride = shard_stride_words;
  const int rc = mh_frame_enqueue_rest(ctx, q_uv_dev, Q, gathered_dev, n_shards, cam, prm, seed);
  ctx->exchange_stride = 0;
  return rc;
}

int mh_frame_enqueue_rest_batch(mh_ctx* ctx, const float* q_uv_dev, int Q, const int32_t* gathered_dev, int n_shards,
                                int shard_stride_words, int plane_stride_words, int slot, const mh_cam* cam,
                                const mh_frame_params* prm, uint64_t seed) {
  if (!ctx || slot < 0 || slot >= MH_MAX_BATCH || plane_stride_words < Q || shard_stride_words < 3 * plane_stride_words)
    return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  int rc = prepare_frame(ctx, Q);
  if (rc) return rc;
  ctx->exchange_stride = shard_stride_words;
  ctx->exchange_plane = plane_stride_words;
  ctx->fs->slot = slot;
  rc = mh_frame_enqueue_rest(ctx, q_uv_dev, Q, gathered_dev, n_shards, cam, prm, seed);
  ctx->exchange_stride = 0;
  ctx->exchange_plane = 0;
  if (ctx->fs) ctx->fs->slot = 0;
  return rc;
}

int mh_frame_enqueue_rest_frames(mh_ctx* ctx, const float* q_uv_dev, int Q, const int32_t* gathered_dev, int n_shards,
                                 int shard_stride_words, int plane_stride_words, int B, const mh_cam* cam,
                                 const mh_frame_params* prm, const uint64_t* seeds) {
  if (!ctx || Q <= 0 || !q_uv_dev || !gathered_dev || n_shards <= 0 || !cam || !prm || !seeds || B < 1 || B > MH_MAX_BATCH ||
      plane_stride_words < Q || shard_stride_words < 3 * plane_stride_words)
    return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  const bool merge = B > 1 && merged_batch_ok(ctx, prm);
  int rc = prepare_frame(ctx, merge ? B * Q : Q, Q, merge ? B : 1);
  if (rc) return rc;
  // merged: frame f merges the shards' blocks into [f Q, (f + 1) Q) of the context's top-2 arrays
  if (merge) {
    if ((rc = ensure_batch_arenas(ctx, B))) return rc;
    ctx->exchange_stride = shard_stride_words;
    ctx->exchange_plane = plane_stride_words;
    ctx->batch_q0 = 0;
    ctx->fs->slot = 0;
    stamp(ctx, 1);
    rc = frame_rest(ctx, q_uv_dev, Q, gathered_dev, n_shards, cam, prm, seeds[0], seeds, B);
    ctx->exchange_stride = 0;
    ctx->exchange_plane = 0;
    return rc;
  }
  for (int f = 0; f < B && rc == MH_OK; ++f) {
    ctx->batch_f = f;   // (frames with several images: the frame's slice of the per-query image index)
    rc = mh_frame_enqueue_rest_batch(ctx, q_uv_dev + 2 * (size_t)f * Q, Q, gathered_dev + (size_t)f * Q, n_shards,
                                     shard_stride_words, plane_stride_words, f, cam, prm, seeds[f]);
  }
  ctx->batch_f = 0;
  return rc;
}

int mh_frame_fetch_slot(mh_ctx* ctx, int slot, mh_object* objects_host, int max_objects, int32_t* n_objects,
                        int32_t* counts) {
  if (!ctx || !n_objects || !ctx->fs || slot < 0 || slot >= MH_MAX_BATCH) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  FrameState* fs = ctx->fs;
  const unsigned char* result = fs->result + (size_t)slot * fs->result_bytes;
  FrameState::FetchPin& pin = *fs->fetch_pin;
  const size_t first = std::min(fs->result_bytes, sizeof pin.head + sizeof pin.objects);
  MH_HIP(ctx, hipMemcpyAsync(pin.head, result, first, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipMemcpyAsync(pin.snap, fs->snap + 4 * slot, sizeof pin.snap, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int32_t* head = pin.head;
  const int32_t* snap = pin.snap;
  *n_objects = head[0];
  if (counts) std::memcpy(counts, snap, 4 * sizeof(int32_t));
  const int tasks = 4 * std::max(snap[1], snap[3]);
  fs->task_grid = std::min(96, std::max(16, (tasks + tasks / 2 + 7) / 8 * 8));
  fs->ms_grid = std::min(32, std::max(4, snap[1] + 2));   // (a model with matches and no cluster still takes a turn: the workgroups loop)
  const int take = head[0] < max_objects ? head[0] : max_objects;
  if (take > 0 && objects_host) {
    const int have = std::min(take, FETCH_PIN_OBJECTS);
    std::memcpy(objects_host, pin.objects, sizeof(mh_object) * (size_t)have);
    if (take > have)
      MH_HIP(ctx, hipMemcpy(objects_host + have, result + 16 + sizeof(mh_object) * (size_t)have,
                            sizeof(mh_object) * (size_t)(take - have), hipMemcpyDeviceToHost));
  }
  if (head[1]) {
    ctx->err = (head[1] & ERR_EXCHANGE)
                   ? std::string("frame exchange: the ranks' blocks carry different sequence numbers / seeds -- the ranks issued "
                                 "their collectives in different orders (every rank must enqueue its slots in the same order)")
                   : "frame: capacity exceeded (flags " + std::to_string(head[1]) + ")";
    return MH_ERR_CAPACITY;
  }
  return MH_OK;
}

int mh_frame_result_copy_slots_dev(mh_ctx* ctx, void* dst_dev, int n_slots, int max_objects) {
  if (!ctx || !dst_dev || max_objects < 0 || n_slots <= 0 || n_slots > MH_MAX_BATCH) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  if (!ctx->fs)   // before the first frame: an empty result block ("0 objects")
    if (int rc = prepare_frame(ctx, ctx->max_q > 0 ? ctx->max_q : 1)) return rc;
  FrameState* fs = ctx->fs;
  const size_t bytes = std::min(fs->result_bytes, 16 + sizeof(mh_object) * (size_t)max_objects);
  MH_HIP(ctx, hipMemcpy2DAsync(dst_dev, bytes, fs->result, fs->result_bytes, bytes, (size_t)n_slots,
                               hipMemcpyDeviceToDevice, ctx->stream));
  return MH_OK;
}

int mh_frame_result_copy_dev(mh_ctx* ctx, void* dst_dev, int max_objects) {
  if (!ctx || !dst_dev || max_objects < 0) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  if (!ctx->fs)   // before the first frame: an empty result block ("0 objects")
    if (int rc = prepare_frame(ctx, ctx->max_q > 0 ? ctx->max_q : 1)) return rc;
  FrameState* fs = ctx->fs;
  const size_t bytes = std::min(fs->result_bytes, 16 + sizeof(mh_object) * (size_t)max_objects);
  MH_HIP(ctx, hipMemcpyAsync(dst_dev, fs->result, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return MH_OK;
}

int mh_frame_fetch_matches_slot(mh_ctx* ctx, int slot, int32_t* query_host, int32_t* model_host, int cap,
                                int32_t* n_matches) {
  if (!ctx || !ctx->fs || !n_matches || cap < 0 || slot < 0 || slot >= MH_MAX_BATCH) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  FrameState* fs = ctx->fs;
  if (slot < fs->list_first || slot >= fs->list_first + fs->list_n) {
    ctx->err = "mh_frame_fetch_matches_slot: the lists of that frame are gone (frames that went through the steps one "
               "after the other share one set of working arrays: only the last one's remain)";
    return MH_ERR_ARG;
  }
  const size_t a = (size_t)(slot - fs->list_first) * fs->arena_bytes;   // the frame's copy of the working arrays
  int32_t snap[4] = {0, 0, 0, 0};
  MH_HIP(ctx, hipMemcpyAsync(snap, fs->snap + 4 * slot, sizeof snap, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n_matches = snap[0];
  const int take = std::min(snap[0], cap);
  if (take > 0 && query_host)
    MH_HIP(ctx, hipMemcpy(query_host, reinterpret_cast<const unsigned char*>(fs->m_q) + a, sizeof(int32_t) * (size_t)take,
                          hipMemcpyDeviceToHost));
  if (take > 0 && model_host)
    MH_HIP(ctx, hipMemcpy(model_host, reinterpret_cast<const unsigned char*>(fs->m_model) + a, sizeof(int32_t) * (size_t)take,
                          hipMemcpyDeviceToHost));
  return MH_OK;
}

int mh_frame_fetch_match_points(mh_ctx* ctx, mh_corr* corr_host, int cap, int32_t* n_matches) {
  if (!ctx || !ctx->fs || !n_matches || cap < 0 || (cap > 0 && !corr_host)) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  FrameState* fs = ctx->fs;
  const int slot = fs->list_first + fs->list_n - 1;
  const size_t a = (size_t)(slot - fs->list_first) * fs->arena_bytes;
  int32_t snap[4] = {0, 0, 0, 0};
  MH_HIP(ctx, hipMemcpyAsync(snap, fs->snap + 4 * slot, sizeof snap, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n_matches = snap[0];
  const int take = std::min(snap[0], cap);
  if (take > 0)
    MH_HIP(ctx, hipMemcpy(corr_host, reinterpret_cast<const unsigned char*>(fs->m_corr) + a, sizeof(mh_corr) * (size_t)take,
                          hipMemcpyDeviceToHost));
  return MH_OK;
}

int mh_frame_fetch_matches(mh_ctx* ctx, int32_t* query_host, int32_t* model_host, int cap, int32_t* n_matches) {
  if (!ctx || !ctx->fs) return MH_ERR_ARG;
  return mh_frame_fetch_matches_slot(ctx, ctx->fs->list_first + ctx->fs->list_n - 1, query_host, model_host, cap, n_matches);
}

int mh_frame_result_dev(mh_ctx* ctx, void** block_dev, int64_t* bytes) {
  if (!ctx || !ctx->fs || !block_dev || !bytes) return MH_ERR_ARG;
  *block_dev = ctx->fs->result;
  *bytes = (int64_t)ctx->fs->result_bytes;
  return MH_OK;
}

int mh_frame_counters(mh_ctx* ctx, int32_t out[8]) {
  if (!ctx || !out || !ctx->fs) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  static_assert(sizeof(FrameCounts) == 8 * sizeof(int32_t), "mh_frame_counters hands the block out as 8 words");
  MH_HIP(ctx, hipMemcpyAsync(out, ctx->fs->counts, sizeof(FrameCounts), hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MH_OK;
}

int mh_pose_kernel_info(mh_ctx* ctx, int kind, int32_t out[8]) {
  if (!ctx || !out || kind < 0 || kind > 3) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  return pose_kernel_info(kind, out);
}

int mh_timing(mh_ctx* ctx, mh_times* out) {
  if (!ctx || !out || !ctx->timing || !ctx->ev_made) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  MH_HIP(ctx, hipEventSynchronize(ctx->ev[8]));
  float* dst[8] = {&out->match_ms, &out->group_ms, &out->cluster_ms, &out->pose1_ms,
                   &out->filter1_ms, &out->pose2_ms, &out->filter2_ms, nullptr};
  for (int i = 0; i < 7; ++i) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]);
    *dst[i] = ms;
  }
  float tot = 0.f;
  hipEventElapsedTime(&tot, ctx->ev[0], ctx->ev[8]);
  out->total_ms = tot;
  return MH_OK;
}

// ---- the frame's six slots ONE CALL EACH on a frame that stays on the device between them -----------------------------
// The per-step plugins' hand-over (moped_amd/host/hip_session.hpp, HipHandover): MATCH leaves the frame's lists in the
// context's working arrays, CLUSTER clusters them where they lie, POSE poses those clusters, FILTER filters those
// objects ... -- every call launches its slot's kernels behind the previous call's (frame_rest with a stage range),
// copies what the slot's contract says the host's FrameData must hold into page-locked memory, and waits once.  What
// the steps uploaded again and again before (the match list three times, the objects twice, ~25 small copies per
// FILTER) stays where it is.  ctx->step says where the resident frame stands; a call out of order, or after anything
// else has used the context's working arrays, is refused (MH_ERR_ARG) and the plugin takes its upload path.
}  // extern "C"
namespace {

int step_refuse(mh_ctx* ctx, const char* who) {
  ctx->err = std::string(who) + ": the resident frame is not at the stage before this one (the steps must run in pipeline "
             "order on one frame; any other call that uses the frame's working arrays ends the hand-over)";
  return MH_ERR_ARG;
}

int step_flags(mh_ctx* ctx, const FrameCounts& fc, const char* who) {
  if (!fc.error) return MH_OK;
  ctx->step.done = -1;
  ctx->err = std::string(who) + ": capacity exceeded (flags " + std::to_string(fc.error) + ")";
  return MH_ERR_CAPACITY;
}

// A stage's results -- a handful of short arrays -- go to the host by ONE kernel that writes them into the context's
// page-locked block, one after the other, and one stream synchronisation: six stream-ordered copies of a few hundred
// bytes each were 40-60 us of every stepped slot (each a transfer of its own behind the stage's kernel).
constexpr int GATHER_SEGS = 8;
struct GatherArgs {
  const uint32_t* src[GATHER_SEGS];
  uint32_t words[GATHER_SEGS], dst_word[GATHER_SEGS];
  int n;
};
__global__ void __launch_bounds__(256) step_gather_kernel(GatherArgs a, uint32_t* __restrict__ dst) {
  for (int k = 0; k < a.n; ++k) {
    const uint32_t* __restrict__ src = a.src[k];
    uint32_t* out = dst + a.dst_word[k];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.words[k]; i += gridDim.x * blockDim.x) out[i] = src[i];
  }
}
struct PinCursor {
  unsigned char* base;
  size_t off = 0;
  GatherArgs g = {};
  // n elements of T that the gather kernel fills from device array `src` (4-byte aligned, like every array of the frame)
  template <typename T> T* take(size_t n, const void* src) {
    T* p = reinterpret_cast<T*>(base + off);
    if (src && n > 0) {
      g.src[g.n] = static_cast<const uint32_t*>(src);
      g.words[g.n] = (uint32_t)((n * sizeof(T) + 3) / 4);
      g.dst_word[g.n] = (uint32_t)(off / 4);
      ++g.n;
    }
    off += (n * sizeof(T) + 15) & ~(size_t)15;
    return p;
  }
  int run(mh_ctx* ctx) {   // behind the stage's kernels on the context's stream; returns when the block is filled
    size_t total = 0;
    for (int k = 0; k < g.n; ++k) total += g.words[k];
    const unsigned blocks = (unsigned)std::min<size_t>(64, std::max<size_t>(1, total / 1024));
    hipLaunchKernelGGL(step_gather_kernel, dim3(blocks), dim3(256), 0, ctx->stream, g, reinterpret_cast<uint32_t*>(base));
    MH_HIP(ctx, hipGetLastError());
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MH_OK;
  }
};

}  // namespace
extern "C" {

int mh_step_match(mh_ctx* ctx, float* q_desc_host, const float* q_uv_host, int Q, const mh_cam* cam, float ratio,
                  int write_back) {
  if (!ctx || Q <= 0 || !q_desc_host || !q_uv_host || !cam) {
    if (ctx) ctx->err = "mh_step_match: bad argument";
    return MH_ERR_ARG;
  }
  mh_frame_params p;
  mh_frame_default_params(&p);
  p.ratio = ratio;
  ctx->stage_lo = ctx->stage_hi = 0;
  const int rc = mh_frame_run_host_begin(ctx, q_desc_host, q_uv_host, nullptr, Q, cam, 1, &p, 0, write_back);
  ctx->stage_lo = 0;
  ctx->stage_hi = 5;
  if (rc) return rc;
  mh_ctx::StepState& st = ctx->step;
  st.done = 0;
  st.Q = Q;
  st.M = -1;   // (known once mh_step_match_fetch has run)
  st.n_clusters = st.n_slots = 0;
  st.cam = *cam;
  st.valid.clear();
  st.valid_model.clear();
  return MH_OK;
}

int mh_step_match_fetch(mh_ctx* ctx, int32_t* model_off_host, int32_t* match_query, mh_corr* match_pts, int cap,
                        int32_t* n_matches) {
  if (!ctx || !n_matches || cap < 0 || (cap > 0 && (!match_query || !match_pts))) return MH_ERR_ARG;
  *n_matches = 0;
  if (!ctx->fs || ctx->step.done != 0) return step_refuse(ctx, "mh_step_match_fetch");
  MH_HIP(ctx, hipSetDevice(ctx->device));
  mh_ctx::StepState& st = ctx->step;
  FrameState* fs = ctx->fs;
  const int nm = ctx->n_models, take = std::min(st.Q, fs->max_m);   // (a frame accepts at most one match per query)
  if (int rc = ensure_pinned(ctx, 64 + (size_t)(nm + 1 + take) * 4 + (size_t)take * sizeof(mh_corr) + 64)) return rc;
  PinCursor pc{static_cast<unsigned char*>(ctx->pinned)};
  FrameCounts* fc = pc.take<FrameCounts>(1, fs->counts);
  int32_t* off = pc.take<int32_t>(nm + 1, fs->model_off);
  int32_t* mq = pc.take<int32_t>(take, fs->m_q);
  mh_corr* mc = pc.take<mh_corr>(take, fs->m_corr);
  if (int rc = pc.run(ctx)) return rc;
  if (int rc = step_flags(ctx, *fc, "mh_step_match_fetch")) return rc;
  const int M = off[nm];
  if (M < 0 || M > take) {
    st.done = -1;
    ctx->err = "mh_step_match_fetch: inconsistent match count";
    return MH_ERR_HIP;
  }
  st.M = M;
  st.model_off.assign(off, off + nm + 1);
  *n_matches = M;
  if (model_off_host) std::memcpy(model_off_host, off, (size_t)(nm + 1) * 4);
  const int give = std::min(M, cap);
  if (give > 0) {
    std::memcpy(match_query, mq, (size_t)give * 4);
    std::memcpy(match_pts, mc, (size_t)give * sizeof(mh_corr));
  }
  if (M > cap) {   // (the first `cap` are written, the resident frame stays valid: a larger buffer can fetch again)
    ctx->err = "mh_step_match_fetch: more matches than the caller's buffers hold (cap >= Q always suffices)";
    return MH_ERR_CAPACITY;
  }
  return MH_OK;
}

int mh_step_cluster(mh_ctx* ctx, float radius, float merge, int min_pts, int max_iter, int32_t* cl_model_host,
                    int32_t* cl_off_host, int32_t* members_host, int cap_clusters, int cap_members, int32_t* n_clusters) {
  if (!ctx || !n_clusters || cap_clusters < 0 || cap_members < 0) return MH_ERR_ARG;
  *n_clusters = 0;
  mh_ctx::StepState& st = ctx->step;
  if (!ctx->fs || st.done != 0 || st.M < 0) return step_refuse(ctx, "mh_step_cluster");
  MH_HIP(ctx, hipSetDevice(ctx->device));
  FrameState* fs = ctx->fs;
  mh_frame_params p;
  mh_frame_default_params(&p);
  p.ms_radius = radius;
  p.ms_merge = merge;
  p.ms_min_pts = min_pts;
  p.ms_max_iter = max_iter;
  if (int rc = frame_rest(ctx, ctx->q_uv, st.Q, nullptr, 0, &st.cam, &p, 0, nullptr, 1, 1, 1)) {
    st.done = -1;
    return rc;
  }
  const int M = st.M, tab = std::min(fs->max_clusters, std::max(M, 1));
  if (int rc = ensure_pinned(ctx, 256 + (size_t)(3 * tab + std::max(M, 1)) * 4)) return rc;
  PinCursor pc{static_cast<unsigned char*>(ctx->pinned)};
  FrameCounts* fc = pc.take<FrameCounts>(1, fs->counts);
  int32_t* ncl = pc.take<int32_t>(1, fs->n_clusters);
  int32_t* cm = pc.take<int32_t>(tab, fs->cl_model);
  int32_t* cb = pc.take<int32_t>(tab, fs->cl_begin);
  int32_t* cc = pc.take<int32_t>(tab, fs->cl_count);
  int32_t* mem = pc.take<int32_t>(std::max(M, 1), M > 0 ? fs->ms_members : nullptr);
  if (int rc = pc.run(ctx)) return rc;
  if (int rc = step_flags(ctx, *fc, "mh_step_cluster")) return rc;
  const int n = *ncl;
  if (n < 0 || n > tab) {
    st.done = -1;
    ctx->err = "mh_step_cluster: inconsistent cluster count";
    return MH_ERR_HIP;
  }
  *n_clusters = n;
  int w = 0;
  for (int c = 0; c < n; ++c) {
    const int model = cm[c], b = st.model_off[model];
    if (c < cap_clusters) {
      if (cl_model_host) cl_model_host[c] = model;
      if (cl_off_host) cl_off_host[c] = w;
    }
    for (int j = 0; j < cc[c]; ++j, ++w)
      if (members_host && w < cap_members) members_host[w] = mem[cb[c] + j] - b;   // index inside the model's match list
  }
  if (cl_off_host && n <= cap_clusters) cl_off_host[n] = w;
  st.n_clusters = n;
  st.done = 1;
  return (n > cap_clusters || w > cap_members) ? MH_ERR_CAPACITY : MH_OK;
}

int mh_step_pose(mh_ctx* ctx, int which, const mh_pose_params* prm, uint64_t seed, mh_step_object* out, int cap,
                 int32_t* n_out) {
  if (!ctx || !prm || !n_out || cap < 0 || (cap > 0 && !out) || (which != 1 && which != 2) ||
      prm->max_objects_per_cluster < 1)
    return MH_ERR_ARG;
  *n_out = 0;
  mh_ctx::StepState& st = ctx->step;
  if (!ctx->fs || st.done != (which == 1 ? 1 : 3)) return step_refuse(ctx, "mh_step_pose");
  MH_HIP(ctx, hipSetDevice(ctx->device));
  FrameState* fs = ctx->fs;
  const int stage = which == 1 ? 2 : 4;
  const int base = which == 1 ? 0 : st.n_slots, n_new = st.n_clusters * prm->max_objects_per_cluster;
  if (base + n_new > fs->max_objects) {
    ctx->err = "mh_step_pose: more (cluster, replica) tasks than object slots reserved";
    return MH_ERR_CAPACITY;
  }
  mh_frame_params p;
  mh_frame_default_params(&p);
  (which == 1 ? p.pose1 : p.pose2) = *prm;
  // (frame_rest keys POSE2's random streams with seed ^ 0x5DEECE66D: undone here, the caller's seed is the stage's)
  if (int rc = frame_rest(ctx, ctx->q_uv, st.Q, nullptr, 0, &st.cam, &p, which == 1 ? seed : seed ^ 0x5DEECE66Dull, nullptr, 1,
                          stage, stage)) {
    st.done = -1;
    return rc;
  }
  if (int rc = ensure_pinned(ctx, 256 + (size_t)std::max(n_new, 1) * (4 + 4 + 28))) return rc;
  PinCursor pc{static_cast<unsigned char*>(ctx->pinned)};
  FrameCounts* fc = pc.take<FrameCounts>(1, fs->counts);
  int32_t* valid = pc.take<int32_t>(n_new, fs->obj_valid + base);
  int32_t* model = pc.take<int32_t>(n_new, fs->obj_model + base);
  float* pose = pc.take<float>((size_t)7 * n_new, fs->obj_pose + (size_t)7 * base);
  if (int rc = pc.run(ctx)) return rc;
  if (int rc = step_flags(ctx, *fc, "mh_step_pose")) return rc;
  if (which == 1) {
    st.valid.clear();
    st.valid_model.clear();
  }
  int k = 0;
  for (int o = 0; o < n_new; ++o) {
    if (!valid[o]) continue;
    st.valid.push_back(base + o);
    st.valid_model.push_back(model[o]);
    if (k < cap) {
      out[k].model = model[o];
      std::memcpy(out[k].pose, pose + (size_t)7 * o, 28);
    }
    ++k;
  }
  *n_out = k;
  st.n_slots = base + n_new;
  st.done = stage;
  return k > cap ? MH_ERR_CAPACITY : MH_OK;
}

int mh_step_filter(mh_ctx* ctx, int which, int min_points, float feature_distance, float min_score, int n_objects,
                   float* score, uint8_t* keep, int32_t* out_order, int32_t* cl_members, int32_t* cl_off, int cap_members,
                   int32_t* n_kept) {
  if (!ctx || !n_kept || n_objects < 0 || cap_members < 0 || (which != 1 && which != 2)) return MH_ERR_ARG;
  *n_kept = 0;
  if (cl_off) cl_off[0] = 0;
  mh_ctx::StepState& st = ctx->step;
  if (!ctx->fs || st.done != (which == 1 ? 2 : 4)) return step_refuse(ctx, "mh_step_filter");
  if (n_objects != (int)st.valid.size()) {
    st.done = -1;
    ctx->err = "mh_step_filter: the host's object list is not the one the device holds";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  FrameState* fs = ctx->fs;
  const int stage = which == 1 ? 3 : 5;
  mh_frame_params p;
  mh_frame_default_params(&p);
  if (which == 1) {
    p.f1_min_points = min_points;
    p.f1_feature_distance = feature_distance;
    p.f1_min_score = min_score;
  } else {
    p.f2_min_points = min_points;
    p.f2_feature_distance = feature_distance;
    p.f2_min_score = min_score;
  }
  if (int rc = frame_rest(ctx, ctx->q_uv, st.Q, nullptr, 0, &st.cam, &p, 0, nullptr, 1, stage, stage)) {
    st.done = -1;
    return rc;
  }
  const int nb = std::max(st.n_slots, 1), tab = std::min(nb, fs->max_clusters), M = std::max(st.M, 1);
  if (int rc = ensure_pinned(ctx, 512 + (size_t)(2 * nb + 2 * tab + M) * 4)) return rc;
  PinCursor pc{static_cast<unsigned char*>(ctx->pinned)};
  FrameCounts* fc = pc.take<FrameCounts>(1, fs->counts);
  int32_t* kept_p = pc.take<int32_t>(1, fs->n_slots);
  float* sc = pc.take<float>(nb, fs->obj_score_raw);
  int32_t* old_of = pc.take<int32_t>(nb, fs->obj_clsize + fs->max_objects);
  int32_t* cb = pc.take<int32_t>(tab, fs->cl_begin);
  int32_t* cc = pc.take<int32_t>(tab, fs->cl_count);
  int32_t* mem = pc.take<int32_t>(M, st.M > 0 ? fs->new_members : nullptr);
  if (int rc = pc.run(ctx)) return rc;
  if (int rc = step_flags(ctx, *fc, "mh_step_filter")) return rc;
  const int kept = *kept_p;
  if (kept < 0 || kept > n_objects || kept > tab) {
    st.done = -1;
    ctx->err = "mh_step_filter: inconsistent object count";
    return MH_ERR_HIP;
  }
  for (int i = 0; i < n_objects; ++i) {
    if (score) score[i] = sc[st.valid[i]];
    if (keep) keep[i] = 0;
  }
  std::vector<int32_t> new_model(kept);
  int w = 0;
  bool over = false;
  for (int k = 0; k < kept; ++k) {
    // kept object k sat in slot old_of[k]: the slots that held an object are in ascending order = the host's list order
    const int i = (int)(std::lower_bound(st.valid.begin(), st.valid.end(), old_of[k]) - st.valid.begin());
    if (i >= n_objects || st.valid[i] != old_of[k]) {
      st.done = -1;
      ctx->err = "mh_step_filter: a kept object does not come from a slot that held one";
      return MH_ERR_HIP;
    }
    const int model = st.valid_model[i], b = st.model_off[model];
    new_model[k] = model;
    if (keep) keep[i] = 1;
    if (out_order) out_order[k] = i;
    if (cl_off) cl_off[k] = w;
    for (int j = 0; j < cc[k]; ++j, ++w) {
      if (w >= cap_members) over = true;
      else if (cl_members) cl_members[w] = mem[cb[k] + j] - b;
    }
  }
  if (cl_off) cl_off[kept] = w;
  *n_kept = kept;
  st.valid.resize(kept);
  for (int k = 0; k < kept; ++k) st.valid[k] = k;   // FILTER compacts the kept objects to slots 0 .. kept - 1, in order
  st.valid_model.swap(new_model);
  st.n_slots = kept;
  st.n_clusters = kept;
  st.done = stage;
  return over ? MH_ERR_CAPACITY : MH_OK;
}

}  // extern "C"
