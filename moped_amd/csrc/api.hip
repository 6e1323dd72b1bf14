// C ABI of libmoped_hip.so: context, model database, MATCH entry points.
// (CLUSTER / POSE / FILTER / frame entry points live in api_steps.hip.)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "context.h"
#include "steps.h"

using namespace mh;

namespace mh {

int use_stream(mh_ctx* ctx) {
  if (ctx->stream) return MH_OK;
  if (!ctx->own_stream) MH_HIP(ctx, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
  ctx->stream = ctx->own_stream;
  return MH_OK;
}

int ensure_scratch(mh_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_cap) return MH_OK;
  if (ctx->scratch) MH_HIP(ctx, hipFree(ctx->scratch));
  ctx->scratch = nullptr;
  ctx->scratch_cap = 0;
  size_t cap = bytes + bytes / 4 + 4096;
  MH_HIP(ctx, hipMalloc(&ctx->scratch, cap));
  ctx->scratch_cap = cap;
  return MH_OK;
}

int ensure_pinned(mh_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pinned_cap) return MH_OK;
  if (ctx->pinned) MH_HIP(ctx, hipHostFree(ctx->pinned));
  ctx->pinned = nullptr;
  ctx->pinned_cap = 0;
  size_t cap = bytes + bytes / 4 + 4096;
  MH_HIP(ctx, hipHostMalloc(&ctx->pinned, cap, hipHostMallocDefault));
  ctx->pinned_cap = cap;
  return MH_OK;
}

static void free_screen_bufs(mh_ctx* ctx) {
  ScreenBufs& b = ctx->sbuf;
  for (void* p : {(void*)b.qh, (void*)b.qbad, (void*)b.part, (void*)b.tau, (void*)b.ovf_cnt, (void*)b.recs, (void*)b.ovf, (void*)b.stats})
    if (p) hipFree(p);
  b = ScreenBufs();
}

// the context's view of its store
static void bind_store(mh_ctx* ctx) {
  const DbStore* st = ctx->store.get();
  ctx->N = st ? st->N : 0;
  ctx->n_models = st ? st->n_models : 0;
  ctx->index_base = st ? st->index_base : 0;
  ctx->rmap = RowMap();
  if (st) {
    ctx->rmap.base = st->index_base;
    ctx->rmap.nb = st->n_blocks;
    ctx->rmap.glo = st->blk_glo;
    ctx->rmap.llo = st->blk_llo;
  }
  ctx->db_desc = st ? st->desc : nullptr;
  ctx->db_norm = st ? st->norm : nullptr;
  ctx->db_xyz = st ? st->xyz : nullptr;
  ctx->db_model = st ? st->model : nullptr;
  ctx->sdb = st ? st->screen : ScreenDb();
}

int ensure_match_scratch(mh_ctx* ctx, int Q) {
  const size_t need_pack = match_pack_floats(Q);
  if (need_pack > ctx->match_pack_cap) {
    if (ctx->match_pack) MH_HIP(ctx, hipFree(ctx->match_pack));
    ctx->match_pack = nullptr;
    ctx->match_pack_cap = 0;
    MH_HIP(ctx, hipMalloc(&ctx->match_pack, need_pack * sizeof(float)));
    ctx->match_pack_cap = need_pack;
  }
  size_t need = match_scratch_elems(Q, ctx->N > 0 ? ctx->N : 1);
  if (need > ctx->match_scratch_cap) {
    if (ctx->match_scratch) MH_HIP(ctx, hipFree(ctx->match_scratch));
    ctx->match_scratch = nullptr;
    ctx->match_scratch_cap = 0;
    MH_HIP(ctx, hipMalloc(&ctx->match_scratch, need * sizeof(Top2)));
    ctx->match_scratch_cap = need;
  }
  // the screen's scratch (only for a DB the screen can serve)
  const int q_pad = screen_q_pad(Q);
  if (ctx->sdb.usable && q_pad > ctx->sbuf.q_pad) {
    ScreenBufs& b = ctx->sbuf;
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_screen_bufs(ctx);
    const size_t slots = (size_t)q_pad * screen_rec_slots();
    MH_HIP(ctx, hipMalloc(&b.qh, (size_t)q_pad * DIM * sizeof(_Float16)));
    MH_HIP(ctx, hipMalloc(&b.qbad, (size_t)q_pad));
    MH_HIP(ctx, hipMalloc(&b.part, (size_t)screen_max_splits_a() * q_pad * sizeof(float2)));
    MH_HIP(ctx, hipMalloc(&b.tau, (size_t)q_pad * sizeof(float)));
    MH_HIP(ctx, hipMalloc(&b.recs, slots * sizeof(uint2)));
    MH_HIP(ctx, hipMalloc(&b.ovf_cnt, (size_t)q_pad * sizeof(int32_t)));
    MH_HIP(ctx, hipMalloc(&b.ovf, (size_t)q_pad * SCREEN_OVF_CAP * sizeof(uint2)));
    MH_HIP(ctx, hipMalloc(&b.stats, (size_t)q_pad * 3 * sizeof(unsigned int)));
    MH_HIP(ctx, hipMemsetAsync(b.recs, 0, slots * sizeof(uint2), ctx->stream));
    MH_HIP(ctx, hipMemsetAsync(b.ovf_cnt, 0, (size_t)q_pad * sizeof(int32_t), ctx->stream));
    MH_HIP(ctx, hipMemsetAsync(b.stats, 0, (size_t)q_pad * 3 * sizeof(unsigned int), ctx->stream));
    b.ovf_cap = SCREEN_OVF_CAP;
    b.q_pad = q_pad;   // only now: a failed allocation above leaves q_pad = 0 and the next call starts over
  }
  return MH_OK;
}

int ctx_match(mh_ctx* ctx, const float* qn, const float* qnorm, int Q, int32_t* idx1, float* d1, float* d2,
              const int32_t* q_count, int q_expected) {
  if (Q <= 0) return MH_OK;
  int rc = ensure_match_scratch(ctx, Q);
  if (rc) return rc;
  const int qe = (q_expected > 0 && q_expected < Q) ? q_expected : Q;
  const int screen_mode = ctx->match_mode >= 2 ? 0 : ctx->match_mode;        // modes 2 / 3 pin one of the exact kernels
  const int kernel_pin = ctx->match_mode >= 2 ? ctx->match_mode - 2 : -1;
  if (ctx->sdb.usable && ctx->sbuf.q_pad >= screen_q_pad(Q) && screen_wanted(qe, ctx->N, screen_mode)) {
    ctx->sbuf.ev = nullptr;
    if (ctx->timing && ctx->ev_made) {   // one event set per launch sequence, a ring of them (mh_match_timing)
      ctx->sbuf.ev = ctx->mev[ctx->mev_next];
      ctx->mev_next = (ctx->mev_next + 1) % mh_ctx::MEV_SETS;
      ++ctx->mev_used;
    }
    ctx->sbuf.big = ctx->lane_stream;
    ctx->sbuf.ev_in = ctx->lane_in;
    ctx->sbuf.ev_out = ctx->lane_out;
    launch_match_screen(qn, qnorm, Q, ctx->db_desc, ctx->db_norm, ctx->N, ctx->rmap, ctx->sdb, ctx->sbuf, idx1, d1,
                        d2, ctx->stream, q_count, q_expected);
    ++ctx->match_launches[2];
  } else {
    const int k = launch_match(qn, qnorm, Q, ctx->db_desc, ctx->db_norm, ctx->N, ctx->rmap, ctx->match_scratch,
                               ctx->match_pack, idx1, d1, d2, ctx->stream, q_count, q_expected, kernel_pin);
    if (k >= 0) ++ctx->match_launches[k];
  }
  return MH_OK;
}

template <typename T>
static int realloc_dev(mh_ctx* ctx, T*& p, size_t n) {
  if (p) MH_HIP(ctx, hipFree(p));
  p = nullptr;
  MH_HIP(ctx, hipMalloc(&p, (n > 0 ? n : 1) * sizeof(T)));
  return MH_OK;
}

int ensure_frame_buffers(mh_ctx* ctx, int Q) {
  if (Q <= ctx->max_q) return MH_OK;
  int cap = ctx->max_q > 0 ? ctx->max_q : 4096;
  while (cap < Q) cap *= 2;
  int rc;
  if ((rc = realloc_dev(ctx, ctx->q_desc, (size_t)cap * DIM))) return rc;
  if ((rc = realloc_dev(ctx, ctx->q_norm, cap))) return rc;
  if ((rc = realloc_dev(ctx, ctx->q_uv, (size_t)cap * 2))) return rc;
  if ((rc = realloc_dev(ctx, ctx->nn_idx, cap))) return rc;
  if ((rc = realloc_dev(ctx, ctx->nn_d1, cap))) return rc;
  if ((rc = realloc_dev(ctx, ctx->nn_d2, cap))) return rc;
  ctx->max_q = cap;
  return MH_OK;
}

}  // namespace mh

extern "C" {

int mh_create(int device, mh_ctx** out) {
  if (!out) return MH_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
    return MH_ERR_NODEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MH_ERR_NODEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MH_ERR_NODEVICE;  // gfx950 code objects only
  if (hipSetDevice(device) != hipSuccess) return MH_ERR_HIP;
  mh_ctx* ctx = new (std::nothrow) mh_ctx;
  if (!ctx) return MH_ERR_HIP;
  ctx->device = device;
  // The context's own stream is created on first use (mh_use_stream): a host that
  // installs its own stream with mh_set_stream never takes a hardware queue for it.
  ctx->stream = nullptr;
  *out = ctx;
  return MH_OK;
}

void mh_free_frame_state(mh_ctx* ctx);  // api_steps.hip
void mh_free_sift_state(mh_ctx* ctx);   // api_sift.hip

void mh_destroy(mh_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  mh_free_frame_state(ctx);
  mh_free_sift_state(ctx);
  free_screen_bufs(ctx);
  mh::free_exchange(ctx);
  ctx->store.reset();   // the DB goes with its last user
  void* ptrs[] = {ctx->q_desc, ctx->q_norm,
                  ctx->q_uv,    ctx->nn_idx,  ctx->nn_d1,  ctx->nn_d2,    ctx->match_scratch,
                  ctx->scratch, ctx->match_pack, ctx->rules.ratio_table, ctx->rules.inv_size, ctx->rules.cnt,
                  ctx->rules.keep1, ctx->lk_scratch, ctx->own_depth, ctx->own_fill, ctx->cams_dev, ctx->df_buf, ctx->img_counts, ctx->hf_img};
  for (void* p : ptrs)
    if (p) hipFree(p);
  if (ctx->pinned) hipHostFree(ctx->pinned);
  if (ctx->wb_ev) hipEventDestroy(ctx->wb_ev);
  if (ctx->wb_stream) hipStreamDestroy(ctx->wb_stream);
  if (ctx->ev_made)
    for (auto& e : ctx->ev) hipEventDestroy(e);
  for (auto& set : ctx->mev)
    for (auto& e : set)
      if (e) hipEventDestroy(e);
  if (ctx->dlv.done) hipEventDestroy(ctx->dlv.done);
  if (ctx->dlv.stage) hipFree(ctx->dlv.stage);
  if (ctx->lane_in) hipEventDestroy(ctx->lane_in);
  if (ctx->lane_out) hipEventDestroy(ctx->lane_out);
  if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

#ifdef MH_EXPERIMENTS
// ---- lanes (experiment builds only; tried in round 3 and NOT shipped) ------------------------------------------------
// A host that keeps several frames in flight mixes two kinds of kernels on the device: passes A and B of MATCH, whose
// workgroups each take the whole register file of a compute unit, and a dozen small, latency-bound launches per batch.
// scripts/cu_trace.py: a fifth of the chip's unit-time holds only small workgroups, a sixth nothing.  A lane = a few
// extra streams for the big passes, confined to a CU mask that leaves `reserve` units of each XCD to everything else
// (mask bit i = unit i / 8 of XCD i % 8: scripts/experiments/cu_mask_probe.hip); a context with a lane hands over to
// it before pass A and takes its stream back after pass B (two event waits per MATCH launch sequence, ~12 us each).
// Measured (config 1, one box, bench.py --lane streams,reserve): off 12 000 frames/s; 4,1 11 940; 4,2 11 690; 4,4 11 710;
// 2,2 10 710; 8,2 11 020 -- what the reserved units give the small kernels, the hand-offs and the lane's in-order
// streams take back.  Lowest-priority lane streams instead of a mask HUNG the run (events between streams of different
// priority on this runtime): that variant is refused.
extern "C" int mh_lane_create(int device, int n_streams, int reserve_cus_per_xcd, int low_priority, mh_lane** out) {
  if (low_priority) return MH_ERR_ARG;
  if (!out || n_streams < 1 || n_streams > 16 || reserve_cus_per_xcd < 0 || reserve_cus_per_xcd > 16) return MH_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return MH_ERR_NODEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MH_ERR_HIP;
  const int n_cu = prop.multiProcessorCount;
  mh_lane* lane = new mh_lane;
  lane->device = device;
  lane->reserve = reserve_cus_per_xcd;
  lane->low_priority = low_priority;
  // CU mask bit i = unit i / 8 of XCD i % 8 on this part (scripts/experiments/cu_mask_probe.hip): the last `reserve`
  // units of every XCD stay out of the lane
  std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
  const int per_xcd = n_cu / 8;
  for (int i = 0; i < n_cu; ++i)
    if (i / 8 < per_xcd - reserve_cus_per_xcd) mask[i / 32] |= 1u << (i % 32);
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = numerically largest = least urgent
  for (int k = 0; k < n_streams; ++k) {
    hipStream_t s = nullptr;
    hipError_t e;
    if (reserve_cus_per_xcd > 0) e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    else if (low_priority) e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo);
    else e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
      for (hipStream_t t : lane->streams) hipStreamDestroy(t);
      delete lane;
      return MH_ERR_HIP;
    }
    lane->streams.push_back(s);
  }
  *out = lane;
  return MH_OK;
}

extern "C" int mh_lane_destroy(mh_lane* lane) {
  if (!lane) return MH_OK;
  hipSetDevice(lane->device);
  for (hipStream_t s : lane->streams) {
    hipStreamSynchronize(s);
    hipStreamDestroy(s);
  }
  delete lane;
  return MH_OK;
}

extern "C" int mh_set_lane(mh_ctx* ctx, mh_lane* lane) {
  if (!ctx) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->stream) MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->lane_stream) MH_HIP(ctx, hipStreamSynchronize(ctx->lane_stream));
  ctx->lane = nullptr;
  ctx->lane_stream = nullptr;
  if (!lane) return MH_OK;
  if (lane->device != ctx->device || lane->streams.empty()) {
    ctx->err = "mh_set_lane: the lane belongs to another device";
    return MH_ERR_ARG;
  }
  if (!ctx->lane_in) MH_HIP(ctx, hipEventCreateWithFlags(&ctx->lane_in, hipEventDisableTiming));
  if (!ctx->lane_out) MH_HIP(ctx, hipEventCreateWithFlags(&ctx->lane_out, hipEventDisableTiming));
  ctx->lane = lane;
  ctx->lane_stream = lane->streams[lane->next++ % lane->streams.size()];
  return MH_OK;
}
#endif   // MH_EXPERIMENTS

const char* mh_last_error(const mh_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int mh_set_stream(mh_ctx* ctx, void* hip_stream) {
  if (!ctx) return MH_ERR_ARG;
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : nullptr;
  return hip_stream ? MH_OK : mh::use_stream(ctx);
}

int mh_synchronize(mh_ctx* ctx) {
  if (!ctx) return MH_ERR_ARG;
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MH_OK;
}

int mh_db_upload(mh_ctx* ctx, const float* desc_host, const int32_t* model_of_host,
                 const float* xyz_host, int N, int n_models, int32_t index_base) {
  return mh_db_upload_raw(ctx, desc_host, model_of_host, xyz_host, N, n_models, index_base, 0);
}

int mh_db_upload_raw(mh_ctx* ctx, const float* desc_host, const int32_t* model_of_host,
                     const float* xyz_host, int N, int n_models, int32_t index_base, int normalize) {
  if (!ctx || N < 0 || n_models < 0 || (N > 0 && (!desc_host || !model_of_host || !xyz_host))) {
    if (ctx) ctx->err = "mh_db_upload: bad argument";
    return MH_ERR_ARG;
  }
  // every later kernel indexes per-model tables with these values
  for (int i = 0; i < N; ++i)
    if (model_of_host[i] < 0 || model_of_host[i] >= n_models) {
      ctx->err = "mh_db_upload: model_of value outside [0, n_models)";
      return MH_ERR_ARG;
    }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // rows are padded to whole 128-row tiles for the match kernels: zero descriptors,
  // +inf norm terms (a padding row can never enter a top-2)
  const size_t Npad = ((size_t)N + 127) / 128 * 128;
  if (!ctx->store || ctx->store.use_count() > 1) {   // other contexts read the old store: leave it alone
    ctx->store = std::make_shared<DbStore>();
    ctx->store->device = ctx->device;
  }
  DbStore* st = ctx->store.get();
  st->N = 0;
  st->screen = ScreenDb();
  bind_store(ctx);
  if (Npad > st->cap) {
    st->cap = 0;   // until every array has its new size
    int rc;
    if ((rc = realloc_dev(ctx, st->desc, Npad * DIM))) return rc;
    if ((rc = realloc_dev(ctx, st->norm, Npad))) return rc;
    if ((rc = realloc_dev(ctx, st->xyz, Npad * 3))) return rc;
    if ((rc = realloc_dev(ctx, st->model, Npad))) return rc;
    st->cap = Npad;
  }
  st->n_models = n_models;
  st->index_base = index_base;
  st->n_blocks = 0;   // one run of global rows starting at index_base (mh_db_upload_blocks sets the tables afterwards)
  if (N > 0) {
    MH_HIP(ctx, hipMemcpyAsync(st->desc, desc_host, (size_t)N * DIM * sizeof(float),
                               hipMemcpyHostToDevice, ctx->stream));
    MH_HIP(ctx, hipMemcpyAsync(st->xyz, xyz_host, (size_t)N * 3 * sizeof(float),
                               hipMemcpyHostToDevice, ctx->stream));
    MH_HIP(ctx, hipMemcpyAsync(st->model, model_of_host, (size_t)N * sizeof(int32_t),
                               hipMemcpyHostToDevice, ctx->stream));
    if (Npad > (size_t)N) {
      MH_HIP(ctx, hipMemsetAsync(st->desc + (size_t)N * DIM, 0, (Npad - N) * DIM * sizeof(float), ctx->stream));
      MH_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)(st->norm + N), 0x7F800000, Npad - N, ctx->stream));
    }
    if (normalize)
      launch_normalize(st->desc, st->norm, N, ctx->stream);   // A1 on the device, like Update() (:94)
    else
      launch_row_norms(st->desc, st->norm, N, ctx->stream);
    MH_HIP(ctx, hipGetLastError());
    // the screen's f16 image + what it needs to know about the rows (match_screen.hip)
    if (screen_wanted(1 << 30, N)) {
      const size_t need_h = screen_db_half_elems(N);
      if (need_h > st->cap_h) {
        st->cap_h = 0;
        int rc;
        if ((rc = realloc_dev(ctx, st->desc_h, need_h))) return rc;
        if ((rc = realloc_dev(ctx, st->neg_h, screen_dneg_elems(N)))) return rc;
        st->cap_h = need_h;
      }
      if (!st->stats) MH_HIP(ctx, hipMalloc(&st->stats, 8 * sizeof(unsigned int)));
      MH_HIP(ctx, hipMemsetAsync(st->stats, 0, 8 * sizeof(unsigned int), ctx->stream));
      launch_db_to_half(st->desc, st->norm, N, st->desc_h, st->neg_h, st->stats, ctx->stream);
      MH_HIP(ctx, hipGetLastError());
      unsigned int h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      MH_HIP(ctx, hipMemcpyAsync(h, st->stats, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
      MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
      float dd_max, x_max;
      std::memcpy(&dd_max, &h[0], 4);
      std::memcpy(&x_max, &h[1], 4);
      st->screen.dbh = st->desc_h;
      st->screen.dneg = st->neg_h;
      st->screen.dmax = std::sqrt(dd_max);
      std::memcpy(&st->screen.spread, &h[3], 4);   // (0 for a DB of fewer than 32 rows: no whole block, never used)
      st->screen.zero_idx = (int32_t)h[4];
      std::memcpy(&st->screen.zero_d1, &h[5], 4);
      std::memcpy(&st->screen.zero_d2, &h[6], 4);
      st->screen.usable = h[2] == 0 && x_max < 60000.f;   // (a NaN coordinate reads as a huge bit pattern: not < 60000)
    }
  }
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  st->N = N;
  bind_store(ctx);
  return MH_OK;
}

int mh_db_upload_blocks(mh_ctx* ctx, const float* desc_host, const int32_t* model_of_host, const float* xyz_host, int N,
                        int n_models, const int32_t* block_global_row, const int32_t* block_rows, int n_blocks,
                        int normalize) {
  if (!ctx || n_blocks < 0 || (n_blocks > 0 && (!block_global_row || !block_rows))) {
    if (ctx) ctx->err = "mh_db_upload_blocks: bad argument";
    return MH_ERR_ARG;
  }
  long long rows = 0;
  for (int b = 0; b < n_blocks; ++b) {
    if (block_rows[b] <= 0 || block_global_row[b] < 0 ||
        (b > 0 && (long long)block_global_row[b] < (long long)block_global_row[b - 1] + block_rows[b - 1])) {
      ctx->err = "mh_db_upload_blocks: blocks must be non-empty runs of global rows in ascending order";
      return MH_ERR_ARG;
    }
    rows += block_rows[b];
  }
  if (rows != N) {
    ctx->err = "mh_db_upload_blocks: the blocks' rows do not add up to N";
    return MH_ERR_ARG;
  }
  int rc = mh_db_upload_raw(ctx, desc_host, model_of_host, xyz_host, N, n_models, n_blocks > 0 ? block_global_row[0] : 0,
                            normalize);
  if (rc || n_blocks <= 1) return rc;
  DbStore* st = ctx->store.get();
  std::vector<int32_t> llo(n_blocks + 1);
  llo[0] = 0;
  for (int b = 0; b < n_blocks; ++b) llo[b + 1] = llo[b] + block_rows[b];
  if (st->blk_glo) MH_HIP(ctx, hipFree(st->blk_glo));
  if (st->blk_llo) MH_HIP(ctx, hipFree(st->blk_llo));
  st->blk_glo = st->blk_llo = nullptr;
  MH_HIP(ctx, hipMalloc(&st->blk_glo, sizeof(int32_t) * n_blocks));
  MH_HIP(ctx, hipMalloc(&st->blk_llo, sizeof(int32_t) * (n_blocks + 1)));
  MH_HIP(ctx, hipMemcpy(st->blk_glo, block_global_row, sizeof(int32_t) * n_blocks, hipMemcpyHostToDevice));
  MH_HIP(ctx, hipMemcpy(st->blk_llo, llo.data(), sizeof(int32_t) * (n_blocks + 1), hipMemcpyHostToDevice));
  st->n_blocks = n_blocks;
  bind_store(ctx);
  return MH_OK;
}

int mh_db_share(mh_ctx* dst, mh_ctx* src) {
  if (!dst || !src || !src->store) {
    if (dst) dst->err = "mh_db_share: the source context holds no database";
    return MH_ERR_ARG;
  }
  if (dst == src) return MH_OK;
  if (dst->device != src->device) {
    dst->err = "mh_db_share: contexts on different devices";
    return MH_ERR_ARG;
  }
  MH_HIP(dst, hipSetDevice(dst->device));
  if (dst->stream) MH_HIP(dst, hipStreamSynchronize(dst->stream));   // frames of dst still reading its old DB
  if (src->stream) MH_HIP(src, hipStreamSynchronize(src->stream));   // the upload into src has completed (it is synchronous) -- cheap
  dst->store = src->store;
  bind_store(dst);
  return MH_OK;
}

int mh_db_size(const mh_ctx* ctx, int* N, int* n_models) {
  if (!ctx) return MH_ERR_ARG;
  if (N) *N = ctx->N;
  if (n_models) *n_models = ctx->n_models;
  return MH_OK;
}

int mh_normalize(mh_ctx* ctx, float* desc_host, int n) {
  if (!ctx || n < 0 || (n > 0 && !desc_host)) return MH_ERR_ARG;
  if (n == 0) return MH_OK;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  int rc = ensure_frame_buffers(ctx, n);
  if (rc) return rc;
  const size_t bytes = (size_t)n * DIM * sizeof(float);
  MH_HIP(ctx, hipMemcpyAsync(ctx->q_desc, desc_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  launch_normalize(ctx->q_desc, ctx->q_norm, n, ctx->stream);
  MH_HIP(ctx, hipGetLastError());
  MH_HIP(ctx, hipMemcpyAsync(desc_host, ctx->q_desc, bytes, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MH_OK;
}

int mh_normalize_dev(mh_ctx* ctx, float* q_dev, float* qnorm_dev, int Q) {
  if (!ctx || Q < 0 || (Q > 0 && (!q_dev || !qnorm_dev))) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  launch_normalize(q_dev, qnorm_dev, Q, ctx->stream);
  MH_HIP(ctx, hipGetLastError());
  return MH_OK;
}

int mh_match_local_dev(mh_ctx* ctx, const float* qn_dev, const float* qnorm_dev, int Q,
                       int32_t* idx1_dev, float* d1_dev, float* d2_dev) {
  if (!ctx || Q < 0 || (Q > 0 && (!qn_dev || !qnorm_dev || !idx1_dev || !d1_dev || !d2_dev)))
    return MH_ERR_ARG;
  if (Q == 0) return MH_OK;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  int rc = ctx_match(ctx, qn_dev, qnorm_dev, Q, idx1_dev, d1_dev, d2_dev);
  if (rc) return rc;
  MH_HIP(ctx, hipGetLastError());
  return MH_OK;
}

int mh_match_merge_dev(mh_ctx* ctx, const int32_t* idx1_s_dev, const float* d1_s_dev,
                       const float* d2_s_dev, int n_shards, int Q, int32_t* idx1_dev,
                       float* d1_dev, float* d2_dev) {
  if (!ctx || Q < 0 || n_shards < 0) return MH_ERR_ARG;
  if (Q == 0) return MH_OK;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  launch_match_merge(idx1_s_dev, d1_s_dev, d2_s_dev, n_shards, Q, (size_t)Q, idx1_dev, d1_dev, d2_dev,
                     ctx->stream);
  MH_HIP(ctx, hipGetLastError());
  return MH_OK;
}

// MATCH of host descriptors; `normalized_out` != nullptr: the rows are raw, are normalised on the device first and
// come back normalised (what MATCH_ANN_CPU::process does to frameData's descriptors in place, :157) -- one upload
// and one synchronisation for both.
static int match_host(mh_ctx* ctx, const float* q_host, float* normalized_out, int Q, float ratio, int32_t* nn_idx,
                      int32_t* nn_raw, float* d1, float* d2) {
  if (!ctx || Q < 0 || (Q > 0 && (!q_host || !nn_idx))) return MH_ERR_ARG;
  if (Q == 0) return MH_OK;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  int rc = ensure_frame_buffers(ctx, Q);
  if (rc) return rc;
  if ((rc = ensure_match_scratch(ctx, Q))) return rc;
  if ((rc = ensure_pinned(ctx, (size_t)Q * 16))) return rc;
  if ((rc = ensure_scratch(ctx, (size_t)Q * 4))) return rc;
  const size_t bytes = (size_t)Q * DIM * sizeof(float);
  MH_HIP(ctx, hipMemcpyAsync(ctx->q_desc, q_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  if (normalized_out)
    launch_normalize(ctx->q_desc, ctx->q_norm, Q, ctx->stream);
  else
    launch_row_norms(ctx->q_desc, ctx->q_norm, Q, ctx->stream);
  if ((rc = ctx_match(ctx, ctx->q_desc, ctx->q_norm, Q, ctx->nn_idx, ctx->nn_d1, ctx->nn_d2))) return rc;
  // the reference's acceptance test on squared distances (MATCH_ANN_CPU.hpp:165)
  int32_t* d_acc = (int32_t*)ctx->scratch;
  launch_accept(ctx->nn_idx, ctx->nn_d1, ctx->nn_d2, Q, ratio, d_acc, ctx->stream);
  MH_HIP(ctx, hipGetLastError());
  int32_t* h_acc = (int32_t*)ctx->pinned;
  int32_t* h_raw = h_acc + Q;
  float* h_d1 = (float*)(h_raw + Q);
  float* h_d2 = h_d1 + Q;
  MH_HIP(ctx, hipMemcpyAsync(h_acc, d_acc, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipMemcpyAsync(h_raw, ctx->nn_idx, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipMemcpyAsync(h_d1, ctx->nn_d1, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipMemcpyAsync(h_d2, ctx->nn_d2, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (normalized_out) MH_HIP(ctx, hipMemcpyAsync(normalized_out, ctx->q_desc, bytes, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(nn_idx, h_acc, (size_t)Q * 4);
  if (nn_raw) std::memcpy(nn_raw, h_raw, (size_t)Q * 4);
  if (d1) std::memcpy(d1, h_d1, (size_t)Q * 4);
  if (d2) std::memcpy(d2, h_d2, (size_t)Q * 4);
  return MH_OK;
}

int mh_match(mh_ctx* ctx, const float* q_host, int Q, float ratio, int32_t* nn_idx,
             int32_t* nn_raw, float* d1, float* d2) {
  return match_host(ctx, q_host, nullptr, Q, ratio, nn_idx, nn_raw, d1, d2);
}

int mh_normalize_match(mh_ctx* ctx, float* q_host, int Q, float ratio, int32_t* nn_idx, int32_t* nn_raw, float* d1,
                       float* d2) {
  return match_host(ctx, q_host, q_host, Q, ratio, nn_idx, nn_raw, d1, d2);
}

#ifdef MH_TRACE
// experiment builds: the workgroup residency trace (common.h).  on = 1 allocates the ring and points every translation
// unit's kernels at it; mh_trace_fetch copies up to cap records {kernel << 32 | HW_ID, block << 32 | XCC_ID, t0, t1}
// (100 MHz ticks) and rewinds the ring.
static unsigned long long* g_trace_dev = nullptr;
int mh_trace_enable(int on) {
  if (on && !g_trace_dev) {
    if (hipMalloc(&g_trace_dev, (8 + 4 * TRACE_CAP) * sizeof(unsigned long long)) != hipSuccess) return MH_ERR_HIP;
    hipMemset(g_trace_dev, 0, 64);
  }
  unsigned long long* p = on ? g_trace_dev : nullptr;
  const int n = std::min(trace_n_binds().load(), 32);
  for (int i = 0; i < n; ++i) trace_binds()[i](p);
  hipDeviceSynchronize();
  return MH_OK;
}
long long mh_trace_fetch(unsigned long long* out, long long cap) {
  if (!g_trace_dev) return -1;
  hipDeviceSynchronize();
  unsigned long long n = 0;
  hipMemcpy(&n, g_trace_dev, sizeof n, hipMemcpyDeviceToHost);
  const unsigned long long have = std::min<unsigned long long>(n, TRACE_CAP);
  const unsigned long long take = std::min<unsigned long long>(have, (unsigned long long)std::max(cap, 0ll));
  if (take && out) hipMemcpy(out, g_trace_dev + 8, take * 32, hipMemcpyDeviceToHost);
  hipMemset(g_trace_dev, 0, 8);
  return (long long)n;
}
#endif

float mh_screen_margin(float qq, float dmax) { return screen_margin_host(qq, dmax); }

uint16_t mh_screen_record_value(float top, float thr) { return screen_record_value(top, thr); }

void mh_screen_record_bounds(uint16_t value_bits, uint32_t row0, float tau, float spread, int N, float dmax, float* lo,
                             float* hi) {
  float l, h;
  screen_record_bounds(value_bits, row0, tau, spread, N, dmax, l, h);
  if (lo) *lo = l;
  if (hi) *hi = h;
}

int mh_screen_values(mh_ctx* ctx, const float* q_host, int Q, int n_rows, float* out_host, float* dmax, float* spread,
                     int shape) {
  if (!ctx || !q_host || !out_host || Q <= 0 || n_rows <= 0 || (Q & 31) || (n_rows & 31)) {
    if (ctx) ctx->err = "mh_screen_values: Q and n_rows must be positive multiples of 32";
    return MH_ERR_ARG;
  }
  if (!ctx->sdb.dbh || n_rows > (ctx->N + 127) / 128 * 128) {
    ctx->err = "mh_screen_values: the database has no f16 image (fewer than 4096 rows?) or fewer rows than asked for";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  const int q_pad = screen_q_pad(Q);
  float *qd = nullptr, *qn = nullptr, *out = nullptr;
  _Float16* qh = nullptr;
  uint8_t* qbad = nullptr;
  int rc = MH_OK;
  auto done = [&]() {
    for (void* p : {(void*)qd, (void*)qn, (void*)out, (void*)qh, (void*)qbad})
      if (p) hipFree(p);
  };
  if (hipMalloc(&qd, (size_t)Q * DIM * 4) != hipSuccess || hipMalloc(&qn, (size_t)Q * 4) != hipSuccess ||
      hipMalloc(&out, (size_t)Q * n_rows * 4) != hipSuccess || hipMalloc(&qh, (size_t)q_pad * DIM * 2) != hipSuccess ||
      hipMalloc(&qbad, (size_t)q_pad) != hipSuccess) {
    done();
    ctx->err = "mh_screen_values: out of device memory";
    return MH_ERR_HIP;
  }
  hipMemcpyAsync(qd, q_host, (size_t)Q * DIM * 4, hipMemcpyHostToDevice, ctx->stream);
  launch_row_norms(qd, qn, Q, ctx->stream);
  launch_screen_prepare(qd, qn, Q, q_pad, qh, qbad, ctx->stream);
  launch_screen_values(qh, Q, ctx->sdb, n_rows, out, ctx->stream, shape);
  if (hipMemcpyAsync(out_host, out, (size_t)Q * n_rows * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
    ctx->err = "mh_screen_values: device error";
    rc = MH_ERR_HIP;
  }
  done();
  if (dmax) *dmax = ctx->sdb.dmax;
  if (spread) *spread = ctx->sdb.spread;
  return rc;
}

int mh_match_set_mode(mh_ctx* ctx, int mode) {
  if (!ctx || mode < -1 || mode > 3) return MH_ERR_ARG;
  ctx->match_mode = mode;
  return MH_OK;
}

int mh_pose_set_split(mh_ctx* ctx, int on) {
  if (!ctx) return MH_ERR_ARG;
  ctx->pose_split = on ? 1 : 0;
  return MH_OK;
}

int mh_match_launches(mh_ctx* ctx, uint32_t out[3]) {
  if (!ctx || !out) return MH_ERR_ARG;
  for (int k = 0; k < 3; ++k) out[k] = ctx->match_launches[k];
  return MH_OK;
}

int mh_match_stats(mh_ctx* ctx, int Q, uint32_t stats[4], int reset) {
  if (!ctx || !stats) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  stats[0] = stats[1] = stats[2] = stats[3] = 0;
  if (Q > 0) {
    int rc = ensure_match_scratch(ctx, Q);
    if (rc) return rc;
    stats[3] = (ctx->sdb.usable && ctx->sbuf.q_pad >= screen_q_pad(Q) && screen_wanted(Q, ctx->N, ctx->match_mode >= 2 ? 0 : ctx->match_mode)) ? 1u : 0u;
  }
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->sbuf.stats) {
    std::vector<unsigned int> h((size_t)ctx->sbuf.q_pad * 3);
    MH_HIP(ctx, hipMemcpy(h.data(), ctx->sbuf.stats, h.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
    unsigned long long sum[3] = {0, 0, 0};
    for (size_t i = 0; i < h.size(); ++i) sum[i % 3] += h[i];
    for (int k = 0; k < 3; ++k) stats[k] = (uint32_t)std::min<unsigned long long>(sum[k], 0xFFFFFFFFull);
    if (reset) MH_HIP(ctx, hipMemset(ctx->sbuf.stats, 0, h.size() * sizeof(unsigned int)));
  }
  return MH_OK;
}

int mh_enable_timing(mh_ctx* ctx, int on) {
  if (!ctx) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  if (on && !ctx->ev_made) {
    for (auto& e : ctx->ev) MH_HIP(ctx, hipEventCreate(&e));
    for (auto& set : ctx->mev)
      for (auto& e : set) MH_HIP(ctx, hipEventCreate(&e));
    ctx->ev_made = true;
  }
  ctx->timing = on != 0;
  ctx->mev_next = ctx->mev_used = 0;
  return MH_OK;
}

int mh_match_timing(mh_ctx* ctx, float ms[5]) {
  if (!ctx || !ms || !ctx->timing || !ctx->ev_made || ctx->mev_used == 0) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int n = std::min(ctx->mev_used, (int)mh_ctx::MEV_SETS);
  double sum[5] = {0, 0, 0, 0, 0};
  for (int k = 0; k < n; ++k) {
    const int set = ((ctx->mev_next - 1 - k) % mh_ctx::MEV_SETS + mh_ctx::MEV_SETS) % mh_ctx::MEV_SETS;
    for (int i = 0; i < 5; ++i) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, ctx->mev[set][i], ctx->mev[set][i + 1]) != hipSuccess) return MH_ERR_HIP;
      sum[i] += t;
    }
  }
  for (int i = 0; i < 5; ++i) ms[i] = (float)(sum[i] / n);
  ctx->mev_used = 0;
  return MH_OK;
}

}  // extern "C"
