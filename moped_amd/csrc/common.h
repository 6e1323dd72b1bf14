// Shared declarations of libmoped_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
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

// ---- match ------------------------------------------------------------------
// Per-query local result of one DB split / shard.
struct Top2 {
  float d1;     // best squared distance
  float d2;     // second best
  int32_t i1;   // row of the best (index_base applied), -1 = none
  int32_t pad;
};

// A1: normalise rows in place + norm term of the normalised rows.
// n_dev (optional): device-side row count, rows [min(n, *n_dev), n) are left alone.
void launch_normalize(float* desc, float* norm_out, int n, hipStream_t s, const int32_t* n_dev = nullptr);
// dot(d,d) chain for every DB row.
void launch_row_norms(const float* desc, float* norm_out, int n, hipStream_t s);
// Scratch (in Top2 units) the match kernel needs for Q queries.
size_t match_scratch_elems(int Q, int N);
// Local top-2 of Q queries vs N rows.
// Floats of packed-query scratch launch_match needs for Q queries.
size_t match_pack_floats(int Q);
void launch_match(const float* qn, const float* qnorm, int Q, const float* db, const float* dnorm,
                  int N, int32_t index_base, Top2* scratch, float* pack, int32_t* idx1, float* d1,
                  float* d2, hipStream_t s, const int32_t* q_count = nullptr, int q_expected = 0);
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
