// Shared declarations of libmoped_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>
#include <string>

#include "../../include/moped_hip.h"

namespace mh {

constexpr int DIM = MH_DESC_DIM;

// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize,
// and a function's attributes belong to the device that is current when they are set: one
// `DynLds` per kernel remembers the devices done (bit per device id), so a second device in the
// process -- or a second host thread -- gets the attribute before its first launch too.  Setting
// it twice from two racing threads is harmless.
struct DynLds {
  std::atomic<unsigned long long> done[4];   // device ids 0..255
  template <typename K>
  void ensure(K kernel, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::atomic<unsigned long long>& word = done[(dev >> 6) & 3];
    const unsigned long long bit = 1ull << (dev & 63);
    if (word.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bytes);
    word.fetch_or(bit, std::memory_order_release);
  }
};

// ---- A/B switches ------------------------------------------------------------------------------------------------
// The product library reads NO environment variable that changes which kernels run or how (a C++ host must get the same
// kernels as the bench).  The switches the measurement scripts flip -- launch shapes, fused / unfused steps, kernel
// variants -- exist only in experiment builds: make EXTRA=-DMH_EXPERIMENTS BUILD=build_exp OUT=../libmoped_hip_exp.so
// (scripts/README.md).  exp_int(name, def) is `def` in the product.
#ifdef MH_EXPERIMENTS
inline int exp_int(const char* name, int def) {
  const char* e = getenv(name);
  return e ? atoi(e) : def;
}
#else
constexpr int exp_int(const char*, int def) { return def; }
#endif

// ---- workgroup residency trace (experiment builds only: make EXTRA=-DMH_TRACE) --------------------------------
// Which compute unit every workgroup of a frame's kernels ran on, and when: thread 0 of a workgroup stamps
// s_memrealtime (100 MHz) at entry and exit and files {kernel id, HW_ID, XCC_ID, t0, t1} in a global ring.  From these
// records scripts/cu_trace.py rebuilds every compute unit's timeline -- how long it held a MATCH workgroup, how long
// only small ones (that a MATCH workgroup cannot join), how long nothing.  Nothing of this exists in the product build.
#ifdef MH_TRACE
enum TraceKernel : unsigned { TK_NORMALIZE = 1, TK_PREPARE, TK_PASS_A, TK_TAU, TK_PASS_B, TK_PASS_C, TK_GROUP, TK_CLUSTER, TK_POSE, TK_OTHER, TK_PASS_B_LOOP, TK_PASS_A_LOOP };   // (the last two: phases INSIDE a pass workgroup)
typedef void (*TraceBindFn)(unsigned long long*);
inline std::atomic<int>& trace_n_binds() { static std::atomic<int> n{0}; return n; }
inline TraceBindFn* trace_binds() { static TraceBindFn fns[32]; return fns; }
inline void trace_register(TraceBindFn f) { const int i = trace_n_binds().fetch_add(1); if (i < 32) trace_binds()[i] = f; }
constexpr unsigned long long TRACE_CAP = 1ull << 21;   // records (32 bytes each)
#ifdef __HIPCC__
// one copy of the buffer pointer per translation unit (no relocatable device code): every unit registers a binder
#define MH_TRACE_TU()                                                                                              \
  namespace {                                                                                                      \
  __device__ unsigned long long* g_trace_buf;                                                                      \
  struct TraceReg_ {                                                                                               \
    TraceReg_() {                                                                                                  \
      mh::trace_register([](unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &p, sizeof p); }); \
    }                                                                                                              \
  } trace_reg_;                                                                                                    \
  struct TraceScope {                                                                                              \
    unsigned long long t0;                                                                                         \
    unsigned kid;                                                                                                  \
    __device__ __forceinline__ explicit TraceScope(unsigned k) : t0(0), kid(k) {                                   \
      if (threadIdx.x == 0 && g_trace_buf) t0 = __builtin_amdgcn_s_memrealtime();                                  \
    }                                                                                                              \
    __device__ __forceinline__ ~TraceScope() {                                                                     \
      if (threadIdx.x == 0 && g_trace_buf) {                                                                       \
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();                                            \
        const unsigned long long i = atomicAdd(g_trace_buf, 1ull);                                                 \
        if (i < mh::TRACE_CAP) {                                                                                   \
          unsigned long long* r = g_trace_buf + 8 + 4 * i;                                                         \
          r[0] = ((unsigned long long)kid << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);   /* HW_REG_HW_ID */   \
          r[1] = ((unsigned long long)blockIdx.y << 48) | ((unsigned long long)(blockIdx.x & 0xFFFF) << 32) |      \
                 __builtin_amdgcn_s_getreg((31 << 11) | 20);                                   /* HW_REG_XCC_ID */  \
          r[2] = t0;                                                                                               \
          r[3] = t1;                                                                                               \
        }                                                                                                          \
      }                                                                                                            \
    }                                                                                                              \
  };                                                                                                               \
  }
#define MH_TRACE_SCOPE(kid) TraceScope trace_scope_(kid)
#define MH_TRACE_PHASE(name, kid) TraceScope name(kid)   // a part of a workgroup's life (ends with the enclosing block)
#endif
#else
#define MH_TRACE_TU()
#define MH_TRACE_SCOPE(kid) do { } while (0)
#define MH_TRACE_PHASE(name, kid) do { } while (0)
#endif

// ---- match ------------------------------------------------------------------
// Per-query local result of one DB split / shard.
struct Top2 {
  float d1;     // best squared distance
  float d2;     // second best
  int32_t i1;   // row of the best (index_base applied), -1 = none
  int32_t pad;
};

// Local row <-> global row of a context's database.  A shard's rows are a concatenation of blocks, each a run of
// consecutive global rows (a model's rows are one run: MATCH_ANN_CPU::Update flattens model after model,
// src/match/MATCH_ANN_CPU.hpp:76-99), blocks in ascending global order -- so local order IS global order and ties
// between rows break the same way in either numbering.  nb <= 1: one block starting at `base` (the usual
// contiguous shard, mh_db_upload's index_base); else glo[nb] = first global row of each block, llo[nb + 1] =
// first local row of each block (llo[nb] = N), both in device memory (mh_db_upload_blocks: a round-robin model
// assignment).
struct RowMap {
  const int32_t* glo = nullptr;
  const int32_t* llo = nullptr;
  int nb = 0;
  int32_t base = 0;
};
#ifdef __HIPCC__
__device__ __forceinline__ int32_t row_to_global(const RowMap& m, int32_t li) {
  if (m.nb <= 1) return li + m.base;
  int lo = 0, hi = m.nb - 1;   // largest b with llo[b] <= li
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (m.llo[mid] <= li) lo = mid; else hi = mid - 1;
  }
  return m.glo[lo] + (li - m.llo[lo]);
}
// -1: the global row belongs to another shard
__device__ __forceinline__ int32_t row_to_local(const RowMap& m, int32_t gi, int N) {
  if (m.nb <= 1) {
    const int32_t li = gi - m.base;
    return (gi >= 0 && li >= 0 && li < N) ? li : -1;
  }
  if (gi < m.glo[0]) return -1;
  int lo = 0, hi = m.nb - 1;   // largest b with glo[b] <= gi
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (m.glo[mid] <= gi) lo = mid; else hi = mid - 1;
  }
  const int32_t off = gi - m.glo[lo];
  return off < m.llo[lo + 1] - m.llo[lo] ? m.llo[lo] + off : -1;
}
#endif

// A1: normalise rows in place + norm term of the normalised rows.
// n_dev (optional): device-side row count, rows [min(n, *n_dev), n) are left alone.
void launch_normalize(float* desc, float* norm_out, int n, hipStream_t s, const int32_t* n_dev = nullptr);
void launch_normalize_batch(float* desc, float* norm_out, int n, int B, hipStream_t s, const int32_t* n_dev);
// dot(d,d) chain for every DB row.
void launch_row_norms(const float* desc, float* norm_out, int n, hipStream_t s);
// Scratch (in Top2 units) the match kernel needs for Q queries.
size_t match_scratch_elems(int Q, int N);
// Local top-2 of Q queries vs N rows.
// Floats of packed-query scratch launch_match needs for Q queries.
size_t match_pack_floats(int Q);
// Returns the kernel that searched: 0 = match_kernel (VALU), 1 = match_mfma_kernel, -1 = none (no queries / no rows).
int launch_match(const float* qn, const float* qnorm, int Q, const float* db, const float* dnorm,
                 int N, const RowMap& rmap, Top2* scratch, float* pack, int32_t* idx1, float* d1,
                 float* d2, hipStream_t s, const int32_t* q_count = nullptr, int q_expected = 0,
                 int kernel_pin = -1);
// kernel_pin: -1 = by query count, 0 = match_kernel (VALU), 1 = match_mfma_kernel (mh_match_set_mode 2 / 3).
// q_count (optional): device-side query count; queries [min(Q, *q_count), Q) get "no neighbour"
// (idx -1) without being searched.  q_expected: host estimate of it (sizes the DB splits).
// shard k's arrays start k * shard_stride elements after the given pointers
void launch_match_merge(const int32_t* idx1_s, const float* d1_s, const float* d2_s, int S, int Q,
                        size_t shard_stride, int32_t* idx1, float* d1, float* d2, hipStream_t s);

// ---- launch fusion --------------------------------------------------------------
// The GPU front end retires ~0.2 M dependent dispatches per second over all queues
// (scripts/dispatch_rate.py), so a frame is kept to a handful of launches: the small
// serial step that used to follow a grid-wide kernel in its own launch runs instead in
// the LAST workgroup of that kernel to finish.
// Returns true in exactly one workgroup of the launch -- the last one to call it; every
// global write the other workgroups made before their call is visible to it afterwards.
// All threads of every workgroup must call it exactly once.  The ticket (zero before
// the first launch) re-arms itself.
#ifdef __HIPCC__
__device__ __forceinline__ bool last_workgroup(unsigned int* ticket) {
  __shared__ int is_last_s;
  __threadfence();   // release: this workgroup's results
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int n = gridDim.x;   // (gridDim.y = frames of a batch, each with a ticket of its own)
    const unsigned int t = atomicAdd(ticket, 1u);
    is_last_s = (t == n - 1u);
    if (t == n - 1u) atomicExch(ticket, 0u);
  }
  __syncthreads();
  const bool last = is_last_s != 0;
  if (last) __threadfence();   // acquire: everybody else's results
  return last;
}
// The same for launches whose workgroups share the work of SEVERAL frames (a batch through one launch per stage): a
// frame's ticket counts finished work items instead of workgroups.  `add` items of the frame's `total` are done: true
// in exactly one workgroup, the one whose call completes the count (every global write the others made before theirs is
// visible to it); workgroups without an item in the frame never call.  All threads of the workgroup call it.
__device__ __forceinline__ bool frame_work_done(unsigned int* ticket, unsigned int add, unsigned int total) {
  __shared__ int is_last_w;
  __threadfence();   // release: this workgroup's results
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(ticket, add);
    is_last_w = (t + add == total);
    if (t + add == total) atomicExch(ticket, 0u);
  }
  __syncthreads();
  const bool last = is_last_w != 0;
  if (last) __threadfence();   // acquire: everybody else's results
  return last;
}
// The frame's cluster table in (model, emission) order -- the order POSE walks `clusters[model]`
// (POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:276-280) -- written by the WHOLE workgroup that closes the frame's CLUSTER
// step: thread mm reads model mm's cluster count (zero, and stored as zero, for a model `work(points)` says was never
// clustered), a prefix sum over the models gives its clusters their places; clusters past `max_clusters` are dropped.
// Returns the number of clusters the models hold (the caller reports more than max_clusters as ERR_CLUSTER_CAP).  All
// threads call it; lay_s: blockDim.x / 64 words of LDS.  (One thread walking the models -- two or three dependent global
// loads per model, a store between them -- kept the closing workgroup ~15 us longer on a 20-model frame.)
template <typename WorkFn>
__device__ __forceinline__ int layout_cluster_table(int n_models, const int32_t* __restrict__ model_off, int32_t* ncl,
                                                    const int32_t* cl_start, int models_div, int max_clusters,
                                                    int32_t* __restrict__ cl_model, int32_t* __restrict__ cl_begin,
                                                    int32_t* __restrict__ cl_count, int* lay_s, WorkFn work) {
  const int n_waves = (int)(blockDim.x >> 6), wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  int k0 = 0;
  for (int c0 = 0; c0 < n_models; c0 += (int)blockDim.x) {
    const int mm = c0 + (int)threadIdx.x;
    int nc = 0, bb = 0;
    if (mm < n_models) {
      bb = model_off[mm];
      const bool w = work(model_off[mm + 1] - bb);
      nc = w ? ncl[mm] : 0;
      if (!w) ncl[mm] = 0;
    }
    int incl = nc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    __syncthreads();   // (lay_s of the chunk before)
    if (lane == 63) lay_s[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < n_waves; ++w) {
      const int c = lay_s[w];
      if (w < wave) before += c;
      total += c;
    }
    const int first = k0 + before + incl - nc;
    const int32_t* st = cl_start + bb + mm;
    for (int c = 0; c < nc; ++c) {
      const int k = first + c;
      if (k >= max_clusters) break;
      cl_model[k] = mm / models_div;
      cl_begin[k] = bb + st[c];
      cl_count[k] = st[c + 1] - st[c];
    }
    k0 += total;
  }
  return k0;
}
#endif

// Depth map of the frame as moped3d holds it (moped3d/moped3d.cpp:279-333): 4 floats per pixel
// (x, y, z in the camera frame, norm or negative = invalid) and, optionally, the per-pixel
// fill distance (DEPTH_FILL's ".distance" map).  img == nullptr: none.
struct DepthImage {
  const float4* img = nullptr;
  const float* fill = nullptr;
  int w = 0, h = 0;
  float cauchy_scale = 0.1f;
};

// ---- group ------------------------------------------------------------------
struct FrameCounts {
  int32_t n_matches;
  int32_t n_clusters;
  int32_t reserved[3];   // (object counts live in the frame's snap[] / result block)
  int32_t n_pose_tasks;  // (cluster, replica) tasks of POSE + POSE2 that evaluated hypotheses
  int32_t error;         // sticky capacity/overflow flags
  int32_t n_hyp;         // P3P hypotheses evaluated by POSE + POSE2 (all tasks of the frame)
};

}  // namespace mh
