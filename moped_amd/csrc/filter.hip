// FILTER: FILTER_PROJECTION_CPU::process on gfx950
// (moped2/libmoped/src/filter/FILTER_PROJECTION_CPU.hpp:80-162).
//
//  F1  per object: project every match of its model, in-cluster flag where the
//      squared error < FeatureDistance, score = sum 1/(err+1) accumulated in the
//      reference's order and precision (Float += double, :102-112); then claim
//      each in-cluster keypoint with a 64-bit atomicMax on (score, first object)
//      -- the reference's "point.first < score" sweep in (model, list) order
//      (:117-129) keeps the FIRST object with the maximal score.
//  F2  per object: count the matches of its model whose keypoint it owns (:136-145)
//  F3  one thread: erase objects with too few points or too low a score, keep list
//      order, build the rewritten cluster table (:150-160)
//  F4  per kept object: ordered member list
#include "geom.h"

namespace mh {

namespace {

constexpr int FT = 256;
constexpr int FILTER_GRID = 128;  // grid-stride over the object slots (their count lives on the device)

__device__ __forceinline__ unsigned long long pack_best(float score, int obj) {
  return ((unsigned long long)__float_as_uint(score) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)obj);
}
__device__ __forceinline__ int best_obj(unsigned long long k) {
  return k == 0ull ? -1 : (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
}

__global__ __launch_bounds__(FT) void filter_score_kernel(FilterBuffers fb, DevCam cam,
                                                          float feature_distance,
                                                          const int32_t* __restrict__ n_slots_dev) {
  __shared__ float err_s[FT];
  __shared__ float score_s;
  const int tid = threadIdx.x;
  for (int o = blockIdx.x; o < *n_slots_dev; o += gridDim.x) {
  __syncthreads();
  if (!fb.obj_valid[o]) continue;
  const int m = fb.obj_model[o];
  const int b = fb.model_off[m];
  const int n = fb.model_off[m + 1] - b;
  TM T;
  tm_from_pose(T, fb.obj_pose + 7 * (size_t)o, fb.obj_pose + 7 * (size_t)o + 4);
  float score = 0.f;  // tid 0 only
  for (int base = 0; base < n; base += FT) {
    const int i = base + tid;
    float e = __builtin_inff();
    if (i < n) {
      const mh_corr c = fb.corr[b + i];
      e = reproj_err2(T.r, T.t, cam, c.x, c.y, c.z, c.u, c.v);
    }
    err_s[tid] = e;
    __syncthreads();
    if (tid == 0) {
      const int cnt = min(FT, n - base);
      for (int j = 0; j < cnt; ++j)
        if (err_s[j] < feature_distance)
          score = (float)((double)score + 1. / ((double)err_s[j] + 1.));  // score += 1./(err+1.)
    }
    __syncthreads();
  }
  if (tid == 0) {
    score_s = score;
    fb.obj_score[o] = score;
  }
  __syncthreads();
  score = score_s;
  if (!(score > 0.f)) continue;
  const unsigned long long key = pack_best(score, o);
  for (int i = tid; i < n; i += FT) {
    const mh_corr c = fb.corr[b + i];
    if (reproj_err2(T.r, T.t, cam, c.x, c.y, c.z, c.u, c.v) < feature_distance)
      atomicMax(&fb.best[fb.m_rep[b + i]], key);
  }
  }  // object loop
}

__global__ __launch_bounds__(FT) void filter_count_kernel(FilterBuffers fb,
                                                          const int32_t* __restrict__ n_slots_dev) {
  __shared__ int cnt_s;
  const int tid = threadIdx.x;
  for (int o = blockIdx.x; o < *n_slots_dev; o += gridDim.x) {
  __syncthreads();
  if (tid == 0) cnt_s = 0;
  __syncthreads();
  if (fb.obj_valid[o]) {
    const int m = fb.obj_model[o];
    const int b = fb.model_off[m];
    const int n = fb.model_off[m + 1] - b;
    int mine = 0;
    for (int i = tid; i < n; i += FT) mine += (best_obj(fb.best[fb.m_rep[b + i]]) == o);
    if (mine) atomicAdd(&cnt_s, mine);
  }
  __syncthreads();
  if (tid == 0) fb.obj_clsize[o] = cnt_s;
  }  // object loop
}

// Single thread: erase / compact the object list in place (ascending, so a move
// never overwrites an unread slot) and lay out the new cluster table.
__global__ void filter_compact_kernel(FilterBuffers fb, int min_points, float min_score,
                                      int32_t* n_slots_dev, int32_t* n_clusters_dev,
                                      int32_t* old_of, FrameCounts* counts) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int n = *n_slots_dev;
  int k = 0, w = 0;
  for (int o = 0; o < n; ++o) {
    fb.obj_score_raw[o] = fb.obj_valid[o] ? fb.obj_score[o] : 0.f;
    if (!fb.obj_valid[o]) continue;
    const int sz = fb.obj_clsize[o];
    if (sz < min_points || fb.obj_score[o] < min_score) continue;
    if (k >= fb.max_clusters) {
      atomicOr(&counts->error, ERR_CLUSTER_CAP);
      break;
    }
    old_of[k] = o;
    fb.obj_model[k] = fb.obj_model[o];
    for (int j = 0; j < 7; ++j) fb.obj_pose[7 * k + j] = fb.obj_pose[7 * o + j];
    fb.obj_score[k] = fb.obj_score[o];
    fb.obj_npts[k] = sz;
    fb.obj_valid[k] = 1;
    fb.cl_model[k] = fb.obj_model[k];
    fb.cl_begin[k] = w;
    fb.cl_count[k] = sz;
    w += sz;
    ++k;
  }
  for (int o = k; o < n; ++o) fb.obj_valid[o] = 0;
  *n_slots_dev = k;
  *n_clusters_dev = k;
}

__global__ __launch_bounds__(64) void filter_fill_kernel(FilterBuffers fb,
                                                         const int32_t* __restrict__ n_slots_dev,
                                                         const int32_t* __restrict__ old_of) {
  const int lane = threadIdx.x;
  for (int r = blockIdx.x; r < *n_slots_dev; r += gridDim.x) {
  const int o = old_of[r];
  const int m = fb.obj_model[r];
  const int b = fb.model_off[m];
  const int n = fb.model_off[m + 1] - b;
  int w = fb.cl_begin[r];
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const bool mine = i < n && best_obj(fb.best[fb.m_rep[b + i]]) == o;
    const unsigned long long bal = __ballot(mine);
    if (mine) fb.new_members[w + __popcll(bal & ((1ull << lane) - 1ull))] = b + i;
    w += __popcll(bal);
  }
  }  // kept-object loop
}

}  // namespace

void launch_filter(const FilterBuffers& fb, const DevCam& cam, int min_points,
                   float feature_distance, float min_score, int32_t* n_slots_dev,
                   int32_t* n_clusters_dev, FrameCounts* counts, hipStream_t s) {
  hipMemsetAsync(fb.best, 0, (size_t)fb.max_m * sizeof(unsigned long long), s);
  const int grid = fb.max_objects < FILTER_GRID ? fb.max_objects : FILTER_GRID;
  hipLaunchKernelGGL(filter_score_kernel, dim3(grid), dim3(FT), 0, s, fb, cam,
                     feature_distance, n_slots_dev);
  hipLaunchKernelGGL(filter_count_kernel, dim3(grid), dim3(FT), 0, s, fb, n_slots_dev);
  int32_t* old_of = fb.obj_clsize + fb.max_objects;  // second half of the scratch array
  hipLaunchKernelGGL(filter_compact_kernel, dim3(1), dim3(64), 0, s, fb, min_points, min_score,
                     n_slots_dev, n_clusters_dev, old_of, counts);
  hipLaunchKernelGGL(filter_fill_kernel, dim3(grid), dim3(64), 0, s, fb, n_slots_dev,
                     old_of);
}

}  // namespace mh
