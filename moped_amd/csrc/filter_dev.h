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
constexpr int FL_SLOTS = 256;   // object slots whose model / list / score / cluster size wait in LDS (more: read from global)
struct FilterLds {
  double term_s[FT];            // 64 per wavefront
  int cnt_s, kept_s;
  DevCam cams_s[MH_MAX_IMAGES];   // several images: every match is projected through its own image's camera
  int model_s[FL_SLOTS], b_s[FL_SLOTS], n_s[FL_SLOTS], cl_s[FL_SLOTS];   // n_s < 0: the slot holds no object
  float score_s[FL_SLOTS];
  int vlist_s[FL_SLOTS];        // the slots (below FL_SLOTS) that hold an object, in no particular order: cnt_s of them
};

__device__ __forceinline__ void filter_wave_sync() {   // LDS written by lanes of this wavefront, read by others of it
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The slots' model, list and (with_scores) score into LDS: one round trip for all of them instead of a chain of four
// dependent loads per slot inside the loops below (all threads; ends with a barrier).
// An object holds one of four replicas' slots of a cluster at most times: the loops below walk the LIST of slots with an
// object (vlist_s), not the slots.
__device__ __forceinline__ void filter_load_slots(FilterLds& S, const FilterBuffers& fb, int n_slots, bool with_scores) {
  if (threadIdx.x == 0) S.cnt_s = 0;
  __syncthreads();
  for (int o = threadIdx.x; o < n_slots && o < FL_SLOTS; o += FT) {
    const bool v = fb.obj_valid[o] != 0;
    const int m = v ? fb.obj_model[o] : 0;
    const int b = fb.model_off[m];
    S.model_s[o] = m;
    S.b_s[o] = b;
    S.n_s[o] = v ? fb.model_off[m + 1] - b : -1;
    if (v) S.vlist_s[atomicAdd(&S.cnt_s, 1)] = o;
    if (with_scores) {
      S.score_s[o] = v ? fb.obj_score[o] : 0.f;
      if (!v) {   // F2's answer for a slot without an object
        S.cl_s[o] = 0;
        fb.obj_clsize[o] = 0;
      }
    }
  }
  __syncthreads();
}
// (slot o's list: from LDS, or -- slots past FL_SLOTS -- from the arrays)
__device__ __forceinline__ bool filter_slot(const FilterLds& S, const FilterBuffers& fb, int o, int& m, int& b, int& n) {
  if (o < FL_SLOTS) {
    m = S.model_s[o];
    b = S.b_s[o];
    n = S.n_s[o];
    return n >= 0;
  }
  if (!fb.obj_valid[o]) return false;
  m = fb.obj_model[o];
  b = fb.model_off[m];
  n = fb.model_off[m + 1] - b;
  return true;
}

// F1 for the object slots first, first + stride, ... < n_slots (every thread of the workgroup calls it).  One WAVEFRONT
// per object, four objects at a time: 64 matches per step, their quotients into the wavefront's 64 LDS words, lane 0
// adds them in list order (the reference's Float += double chain; adding 0. leaves a float unchanged); no workgroup
// barrier inside.  (Round 4: the whole workgroup per object, slot after slot with four barriers and four dependent
// loads each, was 85-100 us of one synchronous frame's 670 -- the closing workgroup of a POSE launch runs this.)
__device__ __forceinline__ void filter_score(FilterLds& S, const FilterBuffers& fb, const DevCam& cam, float feature_distance,
                                             int n_slots, int first, int stride) {
  DevCam (&cams_s)[MH_MAX_IMAGES] = S.cams_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool multi = fb.m_img != nullptr;
  if (multi)
    for (int i = tid; i < fb.n_images * (int)(sizeof(DevCam) / 4); i += FT)
      reinterpret_cast<float*>(cams_s)[i] = reinterpret_cast<const float*>(fb.cams)[i];
  filter_load_slots(S, fb, n_slots, false);
  auto cam_of = [&](int match) -> const DevCam& { return multi ? cams_s[fb.m_img[match]] : cam; };
  const int tw = 64 * wave;   // the wavefront's 64 words of term_s
  // ---- F1 ----
  const int n_listed = S.cnt_s, n_low = min(n_slots, FL_SLOTS);
  // the listed slots, then (more than FL_SLOTS slots) the rest slot by slot
  for (int idx = first + stride * wave; idx < n_listed + (n_slots - n_low); idx += stride * (FT / 64)) {
    const int o = idx < n_listed ? S.vlist_s[idx] : n_low + (idx - n_listed);
    int m, b, n;
    if (!filter_slot(S, fb, o, m, b, n)) continue;
    TM T;
    tm_from_pose(T, fb.obj_pose + 7 * (size_t)o, fb.obj_pose + 7 * (size_t)o + 4);
    float score = 0.f;          // lane 0 only
    unsigned inl_bits = 0u;     // this lane's in-cluster flags of the first 32 steps
    for (int base = 0, step = 0; base < n; base += 64, ++step) {
      const int i = base + lane;
      float e = __builtin_inff();
      if (i < n) {
        const mh_corr c = fb.corr[b + i];
        e = reproj_err2(T.r, T.t, cam_of(b + i), c.x, c.y, c.z, c.u, c.v);
      }
      const bool in = e < feature_distance;
      if (in && step < 32) inl_bits |= 1u << step;
      S.term_s[tw + lane] = in ? 1. / ((double)e + 1.) : 0.;
      filter_wave_sync();
      if (lane == 0) {
        const int cnt = min(64, n - base);
        for (int j = 0; j < cnt; ++j) score = (float)((double)score + S.term_s[tw + j]);
      }
      filter_wave_sync();
    }
    score = __shfl(score, 0);
    if (lane == 0) fb.obj_score[o] = score;
    if (!(score > 0.f)) continue;
    const unsigned long long key = pack_best(score, o);
    for (int base = 0, step = 0; base < n; base += 64, ++step) {
      const int i = base + lane;
      if (i >= n) continue;
      bool in;
      if (step < 32) {
        in = (inl_bits >> step) & 1u;
      } else {
        const mh_corr c = fb.corr[b + i];
        in = reproj_err2(T.r, T.t, cam_of(b + i), c.x, c.y, c.z, c.u, c.v) < feature_distance;
      }
      if (in) atomicMax(&fb.best[fb.m_rep[b + i]], key);
    }
  }
}

// F1 of ONE object by the wavefront that has just refined it (fused FILTER, round 4's end): the closing workgroup of the
// launch then starts at F2 -- with ten objects in a frame their F1s were 40 us of its tail, one after the other; here they
// run where the objects are made, on as many compute units.  Same arithmetic as filter_score: 64 matches per step, the
// quotients added in list order (every lane runs the chain on the lanes' values, read lane by lane), the claims by
// atomic max.  `quat` / `trans`: the pose as obj_pose holds it (all lanes); fb: the frame's buffers (arena applied).
__device__ __forceinline__ void filter_score_wave(const FilterBuffers& fb, const DevCam& cam, float feature_distance, int o,
                                                  int m, const float* quat, const float* trans, int lane) {
  const bool multi = fb.m_img != nullptr;
  const int b = fb.model_off[m];
  const int n = fb.model_off[m + 1] - b;
  TM T;
  tm_from_pose(T, quat, trans);
  float score = 0.f;
  unsigned inl_bits = 0u;
  for (int base = 0, step = 0; base < n; base += 64, ++step) {
    const int i = base + lane;
    float e = __builtin_inff();
    if (i < n) {
      const mh_corr c = fb.corr[b + i];
      e = reproj_err2(T.r, T.t, multi ? fb.cams[fb.m_img[b + i]] : cam, c.x, c.y, c.z, c.u, c.v);
    }
    const bool in = e < feature_distance;
    if (in && step < 32) inl_bits |= 1u << step;
    const double term = in ? 1. / ((double)e + 1.) : 0.;
    const unsigned lo = (unsigned)__double_as_longlong(term), hi = (unsigned)((unsigned long long)__double_as_longlong(term) >> 32);
    const int cnt = min(64, n - base);
    for (int j = 0; j < cnt; ++j) {
      const unsigned long long bits = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, j) << 32) |
                                      (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)lo, j);
      score = (float)((double)score + __longlong_as_double((long long)bits));
    }
  }
  if (lane == 0) fb.obj_score[o] = score;
  if (!(score > 0.f)) return;
  const unsigned long long key = pack_best(score, o);
  for (int base = 0, step = 0; base < n; base += 64, ++step) {
    const int i = base + lane;
    if (i >= n) continue;
    bool in;
    if (step < 32) {
      in = (inl_bits >> step) & 1u;
    } else {
      const mh_corr c = fb.corr[b + i];
      in = reproj_err2(T.r, T.t, multi ? fb.cams[fb.m_img[b + i]] : cam, c.x, c.y, c.z, c.u, c.v) < feature_distance;
    }
    if (in) atomicMax(&fb.best[fb.m_rep[b + i]], key);
  }
}

// F2..F4 (+ the result block) by ONE workgroup, after every object has been scored
__device__ __forceinline__ void filter_finish(FilterLds& S, const FilterBuffers& fb, int min_points, float min_score, int n_slots,
                                              int32_t* n_slots_dev, int32_t* n_clusters_dev, FrameCounts* counts,
                                              const FilterTail& tail) {
  int& kept_s = S.kept_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  filter_load_slots(S, fb, n_slots, true);   // (the scores: other workgroups may have written them)

  // ---- F2: keypoints each object owns (one wavefront per object) ----
  const int n_listed = S.cnt_s, n_low = min(n_slots, FL_SLOTS);
  for (int idx = wave; idx < n_listed + (n_slots - n_low); idx += FT / 64) {
    const int o = idx < n_listed ? S.vlist_s[idx] : n_low + (idx - n_listed);
    int m, b, n, mine = 0;
    if (filter_slot(S, fb, o, m, b, n))
      for (int i = lane; i < n; i += 64) mine += (best_obj(fb.best[fb.m_rep[b + i]]) == o);
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0) {
      fb.obj_clsize[o] = mine;
      if (o < FL_SLOTS) S.cl_s[o] = mine;
    }
  }
  __syncthreads();

  // ---- F3: one thread erases / compacts the object list in place (ascending, so a move
  // never overwrites an unread slot) and lays out the new cluster table ----
  int32_t* old_of = fb.obj_clsize + fb.max_objects;  // second half of the scratch array
  if (tid == 0) {
    int k = 0, w = 0;
    for (int o = 0; o < n_slots; ++o) {
      // (explicit branches: `cond ? lds : global` makes one generic pointer of the two, and this compiler's cast of an
      // LDS address to a generic one does not assemble)
      bool valid;
      float score;
      int sz, model;
      if (o < FL_SLOTS) {
        valid = S.n_s[o] >= 0;
        score = S.score_s[o];
        sz = S.cl_s[o];
        model = S.model_s[o];
      } else {
        valid = fb.obj_valid[o] != 0;
        score = valid ? fb.obj_score[o] : 0.f;
        sz = fb.obj_clsize[o];
        model = fb.obj_model[o];
      }
      fb.obj_score_raw[o] = score;
      if (!valid) continue;
      if (sz < min_points || score < min_score) continue;
      if (k >= fb.max_clusters) {
        atomicOr(&counts->error, ERR_CLUSTER_CAP);
        break;
      }
      old_of[k] = o;
      fb.obj_model[k] = model;
      {   // (the seven words requested together: copied one by one, every store waited for its load -- seven trips to L2
          // per kept object in a single thread, ~3.5 us each time an object moved)
        float pz[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) pz[j] = fb.obj_pose[7 * o + j];
#pragma unroll
        for (int j = 0; j < 7; ++j) fb.obj_pose[7 * k + j] = pz[j];
      }
      fb.obj_score[k] = score;
      fb.obj_npts[k] = sz;
      fb.obj_valid[k] = 1;
      fb.cl_model[k] = model;
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
    int m, b, n;
    if (o < FL_SLOTS) {
      b = S.b_s[o];
      n = S.n_s[o];
    } else {
      m = fb.obj_model[r];
      b = fb.model_off[m];
      n = fb.model_off[m + 1] - b;
    }
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
    FrameHostBlock* const h = (tail.host && tail.host_seq) ? tail.host : nullptr;   // one frame alone: the same straight into the
    for (int r = 0; r < kept; ++r) {                                                 // host's page-locked block (no copies at fetch)
      mh_object ob;
      ob.model = fb.obj_model[r];
#pragma unroll
      for (int j = 0; j < 7; ++j) ob.pose[j] = fb.obj_pose[7 * r + j];
      ob.score = fb.obj_score[r];
      ob.n_points = fb.obj_npts[r];
      out[r] = ob;
      if (h && r < FRAME_HOST_OBJECTS) h->objects[r] = ob;   // (from the registers: not read back from the result block)
    }
    reinterpret_cast<int32_t*>(tail.result)[0] = kept;
    reinterpret_cast<int32_t*>(tail.result)[1] = counts->error;   // sticky capacity flags of this frame
    if (h) {
      h->head[0] = kept;
      h->head[1] = counts->error;
      h->head[2] = h->head[3] = 0;
      for (int i = 0; i < 4; ++i) h->snap[i] = tail.snap_all ? tail.snap_all[i] : 0;
      h->error = counts->error;
      __threadfence_system();
      h->seq = atomicAdd(tail.host_seq, 1u) + 1u;
    }
  }
}

}  // namespace

}  // namespace mh
