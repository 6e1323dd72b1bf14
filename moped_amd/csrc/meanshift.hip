// CLUSTER: CLUSTER_MEAN_SHIFT_CPU::MeanShift<T,N> on gfx950
// (moped2/libmoped/src/cluster/CLUSTER_MEAN_SHIFT_CPU.hpp:80-158).
//
// The reference is not textbook mean shift: every iteration (1) computes a
// flat-kernel weighted mean per canopy, (2) marks merges between canopies whose
// means are within Merge with an ORDER-DEPENDENT pointer scheme and (3) folds
// marked canopies into their targets in list order.  The partition and the
// within-cluster order depend on input order, so this kernel reproduces the
// sequential semantics exactly, with the same fp32 operation order:
//
//  (1) one thread per canopy, inner loop over the others in list order
//      (:102-120), no fused multiply-add;
//  (2) one wavefront walks the list; for canopy i all earlier canopies are
//      tested in parallel (64 per pass).  Every write of step i stores the
//      value i, so the only order dependence inside a step is whether an
//      earlier canopy j' (whose pointer is j) has already redirected j before
//      j reads its own pointer.  That predicate ("ov") is resolved exactly by a
//      short relaxation over the pointer chains inside the step (:122-132);
//  (3) one lane folds the marked canopies in list order (:134-148) -- the
//      weighted-mean updates of a target are sequential by definition.
//
// One workgroup per point set (per model); all state in LDS; <= MS_CAP points.
#include "steps.h"

namespace mh {

namespace {

constexpr int MS_THREADS = 256;

template <int ND>
struct MsLds {
  float c[ND][MS_CAP];    // canopy centre
  float a[ND][MS_CAP];    // touchPtsAggregate
  int size[MS_CAP];       // boundPointsSize
  int merges[MS_CAP];     // canopy this one merges into (self = none)
  int order[MS_CAP];      // canopiesRemaining, list order
  int order2[MS_CAP];     // compaction target
  int head[MS_CAP];       // boundPoints as a linked list over point ids
  int tail[MS_CAP];
  int next[MS_CAP];
  int flag[MS_CAP];       // per-step stamps for the ov relaxation
  int nrem;
  int merged_any;
  int stamp;
};

template <int ND>
__device__ void meanshift_body(MsLds<ND>& L, const float* __restrict__ pts, int pts_stride, int n,
                               float radius, float merge, int min_pts, int max_iter,
                               int32_t* __restrict__ members_out, int32_t member_base,
                               int32_t* __restrict__ cl_start_out, int32_t* __restrict__ ncl_out,
                               int32_t* __restrict__ label_out, int32_t* __restrict__ iters_out) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const float sq_radius = __fmul_rn(radius, radius);
  const float sq_merge = __fmul_rn(merge, merge);

  for (int i = tid; i < n; i += MS_THREADS) {
#pragma unroll
    for (int x = 0; x < ND; ++x) L.c[x][i] = pts[(size_t)i * pts_stride + x];
    L.size[i] = 1;
    L.merges[i] = i;
    L.order[i] = i;
    L.head[i] = i;
    L.tail[i] = i;
    L.next[i] = -1;
    L.flag[i] = 0;
  }
  if (tid == 0) {
    L.nrem = n;
    L.stamp = 0;
  }
  __syncthreads();

  int it = 0;
  for (; it < max_iter; ++it) {
    const int nrem = L.nrem;
    // ---- (1) weighted mean of the canopies within Radius --------------------
    for (int p = tid; p < nrem; p += MS_THREADS) {
      const int c = L.order[p];
      float cc[ND], agg[ND];
      const float fsz = (float)L.size[c];
#pragma unroll
      for (int x = 0; x < ND; ++x) {
        cc[x] = L.c[x][c];
        agg[x] = __fmul_rn(cc[x], fsz);
      }
      int touch = L.size[c];
      for (int p2 = 0; p2 < nrem; ++p2) {
        const int o = L.order[p2];
        if (o == c) continue;
        float dist = 0.f;
#pragma unroll
        for (int x = 0; x < ND; ++x) {
          const float d = __fsub_rn(L.c[x][o], cc[x]);
          dist = __fadd_rn(dist, __fmul_rn(d, d));
        }
        if (dist < sq_radius) {
          const int os = L.size[o];
          touch += os;
#pragma unroll
          for (int x = 0; x < ND; ++x) agg[x] = __fadd_rn(agg[x], __fmul_rn(L.c[x][o], (float)os));
        }
      }
#pragma unroll
      for (int x = 0; x < ND; ++x) L.a[x][c] = __fdiv_rn(agg[x], (float)touch);
    }
    __syncthreads();

    // ---- (2) merge marking, one wavefront, list order --------------------------
    if (wave == 0) {
      int stamp = L.stamp;
      for (int pi = 1; pi < nrem; ++pi) {
        const int ci = L.order[pi];
        float ai[ND];
#pragma unroll
        for (int x = 0; x < ND; ++x) ai[x] = L.a[x][ci];
        // S_i membership of the earlier canopies, 64 per pass; remember which
        // passes had any member so the later sweeps can skip the rest
        const int npass = (pi + 63) >> 6;
        bool any_close = false;
        for (int ps = 0; ps < npass; ++ps) {
          const int pj = ps * 64 + lane;
          bool close = false;
          if (pj < pi) {
            const int cj = L.order[pj];
            float dist = 0.f;
#pragma unroll
            for (int x = 0; x < ND; ++x) {
              const float d = __fsub_rn(L.a[x][cj], ai[x]);  // sqEuclDist(other) : other - this
              dist = __fadd_rn(dist, __fmul_rn(d, d));
            }
            close = dist < sq_merge;
            // S_i marker: flag = 2*stamp_base+... kept in a register-free way: order2 as scratch
            L.order2[pj] = close ? 1 : 0;
          }
          any_close |= (__ballot(close) != 0ull);
        }
        if (!any_close) continue;
        // ov relaxation: ov(j) = exists j' in S_i, !ov(j'), merges[j'] == j.
        // ov state lives in order2 (1 = member & not ov, 2 = member & ov).
        for (int round = 0; round < MS_CAP; ++round) {
          ++stamp;
          for (int ps = 0; ps < npass; ++ps) {
            const int pj = ps * 64 + lane;
            if (pj < pi && L.order2[pj] == 1) {
              const int cj = L.order[pj];
              const int tj = L.merges[cj];
              if (tj != cj) L.flag[tj] = stamp;  // a canopy is not its own predecessor
            }
          }
          bool changed = false;
          for (int ps = 0; ps < npass; ++ps) {
            const int pj = ps * 64 + lane;
            bool ch = false;
            if (pj < pi) {
              const int st = L.order2[pj];
              if (st != 0) {
                const int nst = (L.flag[L.order[pj]] == stamp) ? 2 : 1;
                ch = nst != st;
                L.order2[pj] = nst;
              }
            }
            changed |= (__ballot(ch) != 0ull);
          }
          if (!changed) break;
        }
        // writes of step i: every store is the value ci
        for (int ps = 0; ps < npass; ++ps) {
          const int pj = ps * 64 + lane;
          if (pj < pi && L.order2[pj] == 1) L.merges[L.merges[L.order[pj]]] = ci;
        }
        for (int ps = 0; ps < npass; ++ps) {
          const int pj = ps * 64 + lane;
          if (pj < pi && L.order2[pj] != 0) L.merges[L.order[pj]] = ci;
        }
      }
      if (lane == 0) L.stamp = stamp;
    }
    __syncthreads();

    // ---- (3) fold marked canopies into their targets, list order ----------------
    if (tid == 0) {
      int merged = 0;
      for (int p = 0; p < nrem; ++p) {
        const int c = L.order[p];
        const int t = L.merges[c];
        if (t == c) continue;
        const int csz = L.size[c];
        const int tsz = L.size[t];  // == boundPoints.size() of the target
        const int nsz = tsz + csz;
#pragma unroll
        for (int x = 0; x < ND; ++x) {
          const float v = __fadd_rn(__fmul_rn(L.c[x][t], (float)tsz), __fmul_rn(L.c[x][c], (float)csz));
          L.c[x][t] = __fdiv_rn(v, (float)nsz);
        }
        L.next[L.tail[t]] = L.head[c];  // splice c's points after t's
        L.tail[t] = L.tail[c];
        L.size[t] = nsz;
        merged = 1;
      }
      L.merged_any = merged;
    }
    __syncthreads();
    const bool merged_any = L.merged_any != 0;
    if (!merged_any) {
      ++it;
      break;
    }
    // compact canopiesRemaining (erase merged), keep order; reset pointers
    if (wave == 0) {
      int base = 0;
      for (int ps = 0; ps * 64 < nrem; ++ps) {
        const int p = ps * 64 + lane;
        int c = -1;
        bool keep = false;
        if (p < nrem) {
          c = L.order[p];
          keep = L.merges[c] == c;
        }
        const unsigned long long m = __ballot(keep);
        if (keep) L.order2[base + __popcll(m & ((1ull << lane) - 1ull))] = c;
        base += __popcll(m);
      }
      if (lane == 0) L.nrem = base;
    }
    __syncthreads();
    const int nn = L.nrem;
    for (int p = tid; p < nn; p += MS_THREADS) L.order[p] = L.order2[p];
    __syncthreads();
  }
  if (tid == 0 && iters_out) *iters_out = it;

  // ---- emit clusters of size >= MinPts in list order (:151-157) -----------------
  __syncthreads();
  if (label_out)
    for (int i = tid; i < n; i += MS_THREADS) label_out[i] = -1;
  __syncthreads();
  if (tid == 0) {
    const int nrem = L.nrem;
    int ncl = 0, w = 0;
    for (int p = 0; p < nrem; ++p) {
      const int c = L.order[p];
      if (L.size[c] < min_pts) continue;
      if (cl_start_out) cl_start_out[ncl] = w;
      for (int pt = L.head[c]; pt >= 0; pt = L.next[pt]) {
        if (members_out) members_out[w] = member_base + pt;
        if (label_out) label_out[pt] = ncl;
        ++w;
      }
      ++ncl;
    }
    if (cl_start_out) cl_start_out[ncl] = w;
    *ncl_out = ncl;
  }
}

// Frame form: one workgroup per model, points = uv of the model's matches.  The last
// workgroup to finish lays the per-model cluster lists out as the frame's flat cluster
// table in (model, emission) order -- the order POSE walks `clusters[model]`
// (POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:276-280) -- and publishes the counts.
__global__ __launch_bounds__(MS_THREADS) void meanshift_models_kernel(
    const mh_corr* __restrict__ corr, const int32_t* __restrict__ model_off, int n_models, float radius,
    float merge, int min_pts, int max_iter, int32_t* members, int32_t* cl_start, int32_t* ncl,
    int max_clusters, int32_t* __restrict__ cl_model, int32_t* __restrict__ cl_begin,
    int32_t* __restrict__ cl_count, int32_t* __restrict__ n_clusters_out, int32_t* __restrict__ snap,
    FrameCounts* counts, unsigned int* ticket) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  MsLds<2>& L = *reinterpret_cast<MsLds<2>*>(smem);
  const int m = blockIdx.x;
  const int b = m < n_models ? model_off[m] : 0;
  int n = m < n_models ? model_off[m + 1] - b : 0;
  if (n <= 0) {
    if (threadIdx.x == 0 && m < n_models) ncl[m] = 0;
  } else {
    if (n > MS_CAP) {
      if (threadIdx.x == 0) atomicOr(&counts->error, ERR_MS_CAP);
      n = MS_CAP;
    }
    // cl_start needs n+1 slots inside a region of n: the final offset of the last
    // cluster is implied by the region, so write starts only (see cluster table).
    meanshift_body<2>(L, reinterpret_cast<const float*>(corr + b), sizeof(mh_corr) / sizeof(float), n,
                      radius, merge, min_pts, max_iter, members + b, b, cl_start + b + m, ncl + m,
                      nullptr, nullptr);
  }
  if (!last_workgroup(ticket) || threadIdx.x != 0) return;
  int k = 0;
  for (int mm = 0; mm < n_models; ++mm) {
    const int bb = model_off[mm];
    const int32_t* st = cl_start + bb + mm;
    const int nc = ncl[mm];
    for (int c = 0; c < nc; ++c) {
      if (k >= max_clusters) {
        atomicOr(&counts->error, ERR_CLUSTER_CAP);
        break;
      }
      cl_model[k] = mm;
      cl_begin[k] = bb + st[c];
      cl_count[k] = st[c + 1] - st[c];
      ++k;
    }
  }
  counts->n_clusters = k;
  *n_clusters_out = k;
  if (snap) {
    snap[0] = counts->n_matches;
    snap[1] = k;
  }
}

template <int ND>
__global__ __launch_bounds__(MS_THREADS) void meanshift_single_kernel(
    const float* __restrict__ pts, int n, float radius, float merge, int min_pts, int max_iter,
    int32_t* __restrict__ members, int32_t* __restrict__ cl_start, int32_t* __restrict__ ncl,
    int32_t* __restrict__ label, int32_t* __restrict__ iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  MsLds<ND>& L = *reinterpret_cast<MsLds<ND>*>(smem);
  meanshift_body<ND>(L, pts, ND, n, radius, merge, min_pts, max_iter, members, 0, cl_start, ncl,
                     label, iters);
}

// Batch form: workgroup p clusters points [off[p], off[p+1]) of a concatenated array;
// members / labels are problem-local indices, cl_start of problem p starts at off[p] + p.
template <int ND>
__global__ __launch_bounds__(MS_THREADS) void meanshift_batch_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ off, float radius, float merge, int min_pts,
    int max_iter, int32_t* __restrict__ members, int32_t* __restrict__ cl_start, int32_t* __restrict__ ncl,
    int32_t* __restrict__ label) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  MsLds<ND>& L = *reinterpret_cast<MsLds<ND>*>(smem);
  const int p = blockIdx.x;
  const int b = off[p];
  const int n = off[p + 1] - b;
  if (n <= 0 || n > MS_CAP) {   // the host checks the capacity before launching
    if (threadIdx.x == 0) ncl[p] = 0;
    return;
  }
  meanshift_body<ND>(L, pts + (size_t)b * ND, ND, n, radius, merge, min_pts, max_iter, members + b, 0,
                     cl_start + b + p, ncl + p, label + b, nullptr);
}

template <typename K>
void set_lds_attr(K kernel, size_t bytes) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

void launch_meanshift_models(const mh_corr* corr, const int32_t* model_off, int n_models,
                             float radius, float merge, int min_pts, int max_iter, int32_t* members,
                             int32_t* cl_start, int32_t* ncl, int max_clusters, int32_t* cl_model,
                             int32_t* cl_begin, int32_t* cl_count, int32_t* n_clusters_out, int32_t* snap,
                             FrameCounts* counts, unsigned int* ticket, hipStream_t s) {
  static bool once = false;
  if (!once) {
    set_lds_attr(meanshift_models_kernel, sizeof(MsLds<2>));
    once = true;
  }
  // an empty database still gets one workgroup: it publishes "0 clusters"
  hipLaunchKernelGGL(meanshift_models_kernel, dim3(n_models > 0 ? n_models : 1), dim3(MS_THREADS), sizeof(MsLds<2>), s,
                     corr, model_off, n_models, radius, merge, min_pts, max_iter, members, cl_start, ncl,
                     max_clusters, cl_model, cl_begin, cl_count, n_clusters_out, snap, counts, ticket);
}

void launch_meanshift_single(const float* pts, int n, int dim, float radius, float merge,
                             int min_pts, int max_iter, int32_t* members, int32_t* cl_start,
                             int32_t* ncl, int32_t* label, int32_t* iters, hipStream_t s) {
  static bool once = false;
  if (!once) {
    set_lds_attr(meanshift_single_kernel<2>, sizeof(MsLds<2>));
    set_lds_attr(meanshift_single_kernel<3>, sizeof(MsLds<3>));
    once = true;
  }
  if (dim == 3)
    hipLaunchKernelGGL(meanshift_single_kernel<3>, dim3(1), dim3(MS_THREADS), sizeof(MsLds<3>), s,
                       pts, n, radius, merge, min_pts, max_iter, members, cl_start, ncl, label, iters);
  else
    hipLaunchKernelGGL(meanshift_single_kernel<2>, dim3(1), dim3(MS_THREADS), sizeof(MsLds<2>), s,
                       pts, n, radius, merge, min_pts, max_iter, members, cl_start, ncl, label, iters);
}

void launch_meanshift_batch(const float* pts, const int32_t* off, int n_problems, int dim, float radius,
                            float merge, int min_pts, int max_iter, int32_t* members, int32_t* cl_start,
                            int32_t* ncl, int32_t* label, hipStream_t s) {
  if (n_problems <= 0) return;
  static bool once = false;
  if (!once) {
    set_lds_attr(meanshift_batch_kernel<2>, sizeof(MsLds<2>));
    set_lds_attr(meanshift_batch_kernel<3>, sizeof(MsLds<3>));
    once = true;
  }
  if (dim == 3)
    hipLaunchKernelGGL(meanshift_batch_kernel<3>, dim3(n_problems), dim3(MS_THREADS), sizeof(MsLds<3>), s,
                       pts, off, radius, merge, min_pts, max_iter, members, cl_start, ncl, label);
  else
    hipLaunchKernelGGL(meanshift_batch_kernel<2>, dim3(n_problems), dim3(MS_THREADS), sizeof(MsLds<2>), s,
                       pts, off, radius, merge, min_pts, max_iter, members, cl_start, ncl, label);
}

}  // namespace mh
