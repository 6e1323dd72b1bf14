// FILTER as device code shared by filter_kernel (filter.hip) and the fused tail of pose_kernel (pose.hip): see
// filter.hip for what F1..F4 are.  256 threads per workgroup in both.
#pragma once
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

// LDS of one FILTER workgroup
struct FilterLds {
  double term_s[FT];
  float score_s;
  int cnt_s, kept_s;
  DevCam cams_s[MH_MAX_IMAGES];   // several images: every match is projected through its own image's camera
};

// F1 for the object slots first, first + stride, ... < n_slots (every thread of the workgroup calls it)
__device__ __forceinline__ void filter_score(FilterLds& S, const FilterBuffers& fb, const DevCam& cam, float feature_distance,
                                             int n_slots, int first, int stride) {
  double (&term_s)[FT] = S.term_s;
  float& score_s = S.score_s;
  DevCam (&cams_s)[MH_MAX_IMAGES] = S.cams_s;

  const int tid = threadIdx.x;
  const bool multi = fb.m_img != nullptr;
  if (multi) {
    for (int i = tid; i < fb.n_images * (int)(sizeof(DevCam) / 4); i += FT)
      reinterpret_cast<float*>(cams_s)[i] = reinterpret_cast<const float*>(fb.cams)[i];
    __syncthreads();
  }
  auto cam_of = [&](int match) -> const DevCam& { return multi ? cams_s[fb.m_img[match]] : cam; };
  // ---- F1 ----
  for (int o = first; o < n_slots; o += stride) {
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
        e = reproj_err2(T.r, T.t, cam_of(b + i), c.x, c.y, c.z, c.u, c.v);
      }
      // score += 1./(err+1.) over the in-cluster matches, in list order: the quotients in parallel, the
      // Float += double chain by one thread (adding 0. leaves a float unchanged)
      term_s[tid] = e < feature_distance ? 1. / ((double)e + 1.) : 0.;
      __syncthreads();
      if (tid == 0) {
        const int cnt = min(FT, n - base);
        for (int j = 0; j < cnt; ++j) score = (float)((double)score + term_s[j]);
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
      if (reproj_err2(T.r, T.t, cam_of(b + i), c.x, c.y, c.z, c.u, c.v) < feature_distance)
        atomicMax(&fb.best[fb.m_rep[b + i]], key);
    }
  }
}

// F2..F4 (+ the result block) by ONE workgroup, after every object has been scored
__device__ __forceinline__ void filter_finish(FilterLds& S, const FilterBuffers& fb, int min_points, float min_score, int n_slots,
                                              int32_t* n_slots_dev, int32_t* n_clusters_dev, FrameCounts* counts,
                                              const FilterTail& tail) {
  int& cnt_s = S.cnt_s;
  int& kept_s = S.kept_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- F2: keypoints each object owns ----
  for (int o = 0; o < n_slots; ++o) {
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
  }
  __syncthreads();

  // ---- F3: one thread erases / compacts the object list in place (ascending, so a move
  // never overwrites an unread slot) and lays out the new cluster table ----
  int32_t* old_of = fb.obj_clsize + fb.max_objects;  // second half of the scratch array
  if (tid == 0) {
    int k = 0, w = 0;
    for (int o = 0; o < n_slots; ++o) {
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
    for (int o = k; o < n_slots; ++o) fb.obj_valid[o] = 0;
    *n_slots_dev = k;
    *n_clusters_dev = k;
    if (tail.snap_kept) *tail.snap_kept = k;
    kept_s = k;
  }
  __syncthreads();
  const int kept = kept_s;

  // ---- F4: ordered member list of each kept object, one wavefront per object ----
  for (int r = wave; r < kept; r += FT / 64) {
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
  }
  __syncthreads();
  // the claim table goes back to "unclaimed" for the next FILTER of this frame
  const int M = fb.model_off[fb.n_models];
  for (int i = tid; i < M && i < fb.max_m; i += FT) fb.best[i] = 0ull;

  // ---- result block {int32 n; int32 pad[3]; mh_object[n]} (list order) ----
  if (tail.result && tid == 0) {
    mh_object* out = reinterpret_cast<mh_object*>(tail.result + 16);
    for (int r = 0; r < kept; ++r) {
      mh_object ob;
      ob.model = fb.obj_model[r];
      for (int j = 0; j < 7; ++j) ob.pose[j] = fb.obj_pose[7 * r + j];
      ob.score = fb.obj_score[r];
      ob.n_points = fb.obj_npts[r];
      out[r] = ob;
    }
    reinterpret_cast<int32_t*>(tail.result)[0] = kept;
    reinterpret_cast<int32_t*>(tail.result)[1] = counts->error;   // sticky capacity flags of this frame
  }
}

}  // namespace

}  // namespace mh
