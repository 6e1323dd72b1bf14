// CLUSTER (moped3d): CLUSTER_LINKAGE_CPU on gfx950
// (moped3d/libmoped/src/cluster/CLUSTER_LINKAGE_CPU.hpp; line numbers below are that file's),
// as moped3d's config.hpp:45 constructs it: Use3DFilter = 2, average linkage, sigmas from the
// average nearest-neighbour distances.
//
// One 1024-thread workgroup per model.  The model's matches (image point, model point,
// camera-frame point from the depth map) sit in LDS; the n x n matrices live in a per-model
// region of a global scratch buffer (L2 resident: n = 150 -> 90 KB each):
//   A  : K3D + discontinuity kernel, then (x K3F), normalised in two passes over the maxima
//   D  : K2D, then the final similarity K = the clustering's `distances`
//   mem: cluster member lists, row c = cluster c (capacity n)
// Every matrix element is computed by one thread with the reference's expression order
// (Float = float, unsuffixed literals = double); what differs from the CPU is the device's
// expf / atan2f (a few ulp).  The agglomeration is the reference's, quirks included (see
// oracle/linkage_oracle.cpp): per merge, a parallel arg-max over the candidate pairs in the
// reference's scan order (first maximum wins), the average-linkage row update, the reversed
// append of the absorbed cluster.
#include <algorithm>

#include "steps.h"

namespace mh {

namespace {

constexpr int LK_THREADS = 1024;
constexpr int LK_DLDS = 160;   // up to this many matches the clustering's similarity matrix stays in LDS

struct LkLds {
  float uv[LK_CAP][2];
  float mx[LK_CAP][3];   // Match::coord3D (model point)
  float wx[LK_CAP][3];   // Match::depthData.coord3D (camera frame)
  float nn2[LK_CAP], nn3[LK_CAP];   // nearest-neighbour distances; later the fill weights (nn2)
  int clsize[LK_CAP];
  unsigned char inlist[LK_CAP];
  float red_v[LK_THREADS / 64];
  int red_a[LK_THREADS / 64], red_b[LK_THREADS / 64];
  float red2_v[2][LK_THREADS / 64];   // the agglomeration's per-wavefront candidates, by iteration parity
  int red2_a[2][LK_THREADS / 64], red2_b[2][LK_THREADS / 64];
  float sigma2, sigma3, maxv;
  float best_v;
  int best_a, best_b;
  float dmat[LK_DLDS * LK_DLDS];
};

__device__ __forceinline__ float sq_dist(const float* a, const float* b, int n) {   // Pt::sqEuclDist: d = pt - this
  float r = 0.f;
  for (int x = 0; x < n; ++x) {
    const float d = __fsub_rn(b[x], a[x]);
    r = __fadd_rn(r, __fmul_rn(d, d));
  }
  return r;
}

__device__ __forceinline__ void saturate(int& x, int& y, int w, int h) {
  x = x < 0 ? 0 : (x >= w ? w - 1 : x);
  y = y < 0 ? 0 : (y >= h ? h - 1 : y);
}

// getDiscontinuityMatrix's element (:244-283): the largest difference between the slope of the
// straight depth line from p to q and the slopes between consecutive Bresenham samples (:176-220).
__device__ float discontinuity(const DepthImage& D, int px, int py, int qx, int qy) {
  const float depthStart = D.img[(size_t)py * D.w + px].z, depthEnd = D.img[(size_t)qy * D.w + qx].z;
  const int xDiff = px - qx, yDiff = py - qy;
  const float imagePlaneDist = sqrtf((float)(xDiff * xDiff + yDiff * yDiff));
  const float directAngle = atan2f(__fsub_rn(depthEnd, depthStart), imagePlaneDist);
  int x0 = px, y0 = py, x1 = qx, y1 = qy, t;
  const bool steep = abs(y1 - y0) > abs(x1 - x0);
  if (steep) {
    t = x0; x0 = y0; y0 = t;
    t = x1; x1 = y1; y1 = t;
  }
  if (x0 > x1) {
    t = x0; x0 = x1; x1 = t;
    t = y0; y0 = y1; y1 = t;
  }
  const float deltaX = __fsub_rn((float)x1, (float)x0), deltaY = fabsf(__fsub_rn((float)y1, (float)y0));
  const int yStep = (y0 < y1) ? 1 : -1;
  int perStep = (x1 - x0) / 20;
  if (perStep < 1) perStep = 1;
  float error = 0.f;
  const float deltaError = __fdiv_rn(deltaY, deltaX);
  int y = y0;
  float maxAngleDiff = -1.f;
  int pax = 0, pay = 0;
  bool have = false;
  for (int x = x0; x <= x1;) {
    const int cx = steep ? y : x, cy = steep ? x : y;
    if (have) {
      int ax = pax, ay = pay, bx = cx, by = cy;
      saturate(ax, ay, D.w, D.h);
      saturate(bx, by, D.w, D.h);
      const float depth1 = D.img[(size_t)ay * D.w + ax].z, depth2 = D.img[(size_t)by * D.w + bx].z;
      const float dx = (float)(pax - cx), dy = (float)(pay - cy);
      const float pixDistance = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
      const float pixAngle = atan2f(__fsub_rn(depth2, depth1), pixDistance);
      const float angleDiff = fabsf(__fsub_rn(directAngle, pixAngle));
      if (angleDiff > maxAngleDiff) maxAngleDiff = angleDiff;
    }
    pax = cx;
    pay = cy;
    have = true;
    x += perStep;
    if (x > x1) break;   // the values below are not used any more (and are NaN when p == q)
    error = __fadd_rn(error, __fmul_rn(__fmul_rn(deltaError, (float)perStep), (float)yStep));
    float intPart;
    error = modff(error, &intPart);
    y = (int)__fadd_rn((float)y, intPart);
  }
  const float discontinuityDiv = (float)(-2 * (M_PI / 128) * (M_PI / 128));
  return expf(__fdiv_rn(__fmul_rn(maxAngleDiff, maxAngleDiff), discontinuityDiv));
}

// workgroup-wide maximum of `v` under the reference's `if (value > maxValue)` (NaN never wins), from -1
__device__ float wg_max(LkLds& L, float v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  __syncthreads();
  if (lane == 0) L.red_v[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = -1.f;
    for (int k = 0; k < LK_THREADS / 64; ++k)
      if (L.red_v[k] > m) m = L.red_v[k];
    L.maxv = m;
  }
  __syncthreads();
  return L.maxv;
}

__device__ void linkage_body(LkLds& L, const mh_corr* __restrict__ corr, const float4* __restrict__ depth4, int n,
                             const DepthImage& D, const LinkageParams& P, float* __restrict__ A,
                             float* Dm, int32_t* __restrict__ mem, int32_t* __restrict__ members_out,
                             int32_t member_base, int32_t* __restrict__ cl_start_out, int32_t* __restrict__ ncl_out,
                             int32_t* __restrict__ label_out) {
  const int tid = threadIdx.x;
  const int N = n;
  for (int i = tid; i < N; i += LK_THREADS) {
    const mh_corr c = corr[i];
    const float4 d = depth4[i];
    L.uv[i][0] = c.u;
    L.uv[i][1] = c.v;
    L.mx[i][0] = c.x;
    L.mx[i][1] = c.y;
    L.mx[i][2] = c.z;
    L.wx[i][0] = d.x;
    L.wx[i][1] = d.y;
    L.wx[i][2] = d.z;
    L.clsize[i] = 1;
    L.inlist[i] = 1;
    mem[(size_t)i * N] = i;
  }
  __syncthreads();
  // ---- sigmas: getAverageNNDistances (:97-123) ----
  float k2DSigma = P.sigma2d, k3DSigma = P.sigma3d;
  if (P.sigma2d == -1.f || P.sigma3d == -1.f) {
    for (int i = tid; i < N; i += LK_THREADS) {
      float nn2 = __builtin_inff(), nn3 = __builtin_inff();   // Float = DBL_MAX
      for (int j = 0; j < N; ++j) {
        if (i == j) continue;
        const float d2 = sqrtf(sq_dist(L.uv[i], L.uv[j], 2)), d3 = sqrtf(sq_dist(L.mx[i], L.mx[j], 3));
        if (nn2 > d2) nn2 = d2;
        if (nn3 > d3) nn3 = d3;
      }
      L.nn2[i] = nn2;
      L.nn3[i] = nn3;
    }
    __syncthreads();
    if (tid == 0) {
      float s2 = 0.f, s3 = 0.f;
      for (int i = 0; i < N; ++i) {
        s2 = __fadd_rn(s2, L.nn2[i]);
        s3 = __fadd_rn(s3, L.nn3[i]);
      }
      L.sigma2 = __fdiv_rn(s2, (float)N);
      L.sigma3 = __fdiv_rn(s3, (float)N);
    }
    __syncthreads();
    if (P.sigma2d == -1.f) k2DSigma = L.sigma2;
    if (P.sigma3d == -1.f) k3DSigma = L.sigma3;
    __syncthreads();
  }
  // ---- fill weights of adaptiveWeightSum (:331-337): 1.0 / (1 + (d*d/gammaSq)), gamma = 25 ----
  for (int i = tid; i < N; i += LK_THREADS) {
    int x = (int)L.uv[i][0], y = (int)L.uv[i][1];
    saturate(x, y, D.w, D.h);
    const float d = D.fill ? D.fill[(size_t)y * D.w + x] : 0.f;
    L.nn2[i] = (float)(1.0 / (double)__fadd_rn(1.f, __fdiv_rn(__fmul_rn(d, d), 625.f)));
  }
  // ---- pass 1: K2D -> Dm, K3D + BK -> A, maximum (:133-149, 231-285, 681-682) ----
  const long long pairs = (long long)N * (N + 1) / 2;
  const float two2 = __fmul_rn(__fmul_rn(2.f, k2DSigma), k2DSigma), two3 = __fmul_rn(__fmul_rn(2.f, k3DSigma), k3DSigma);
  float mx = -1.f;
  // pair index p -> (i, j), i <= j, row-major over the upper triangle: a thread walks its pairs incrementally
  auto row_start = [&](long long i) { return i * N - i * (i - 1) / 2; };
  auto first_pair = [&](long long p, int& i, int& j) {
    long long r = (long long)(((double)(2 * N + 1) - sqrt((double)(2 * N + 1) * (2 * N + 1) - 8.0 * (double)p)) * 0.5);
    r = r < 0 ? 0 : (r > N - 1 ? N - 1 : r);
    while (r + 1 < N && row_start(r + 1) <= p) ++r;
    while (r > 0 && row_start(r) > p) --r;
    i = (int)r;
    j = i + (int)(p - row_start(r));
  };
  for (long long p = tid; p < pairs; p += LK_THREADS) {
    int i, j;
    first_pair(p, i, j);
    const float v2 = expf(__fdiv_rn(__fmul_rn(-1.f, sq_dist(L.uv[i], L.uv[j], 2)), two2));
    const float v3 = expf(__fdiv_rn(__fmul_rn(-1.f, sq_dist(L.wx[i], L.wx[j], 3)), two3));
    int ax = (int)L.uv[i][0], ay = (int)L.uv[i][1], bx = (int)L.uv[j][0], by = (int)L.uv[j][1];
    saturate(ax, ay, D.w, D.h);
    saturate(bx, by, D.w, D.h);
    const float s = __fadd_rn(v3, discontinuity(D, ax, ay, bx, by));
    Dm[(size_t)i * N + j] = v2;
    A[(size_t)i * N + j] = s;
    if (s > mx) mx = s;
  }
  const float max1 = wg_max(L, mx);
  // ---- pass 2: normalise, x / + K3F, maximum (:151-173, 683-692) ----
  mx = -1.f;
  if (P.use3d_filter) {
    const float sigma = 0.1f;
    const float twoSigmaSq = __fmul_rn(__fmul_rn(2.f, sigma), sigma);
    for (long long p = tid; p < pairs; p += LK_THREADS) {
      int i, j;
      first_pair(p, i, j);
      float val = 1.f;
      if (i != j) {
        const float dm = sqrtf(sq_dist(L.mx[i], L.mx[j], 3)), dr = sqrtf(sq_dist(L.wx[i], L.wx[j], 3));
        const float de = __fdiv_rn(fabsf(__fsub_rn(dm, dr)), dm);
        val = expf(__fdiv_rn(__fmul_rn(__fmul_rn(-1.f, de), de), twoSigmaSq));
      }
      const float e = __fdiv_rn(A[(size_t)i * N + j], max1);
      const float s = P.use3d_filter == 1 ? __fadd_rn(e, val) : __fmul_rn(e, val);
      A[(size_t)i * N + j] = s;
      if (s > mx) mx = s;
    }
  }
  const float max2 = P.use3d_filter ? wg_max(L, mx) : max1;
  // ---- pass 3: the final similarity (:339-365) into both halves of Dm ----
  for (long long p = tid; p < pairs; p += LK_THREADS) {
    int i, j;
    first_pair(p, i, j);
    const float K2De = Dm[(size_t)i * N + j];
    const float K3De = __fdiv_rn(A[(size_t)i * N + j], max2);
    const float jointWeight = __fmul_rn(L.nn2[i], L.nn2[j]);
    const float w2D = (float)(0.5 + 0.5 * (1.0 - (double)jointWeight));   // alpha + alphaBar*(1.0 - jointWeight)
    const float w3D = __fmul_rn(0.5f, jointWeight);
    const float val = __fadd_rn(__fmul_rn(w2D, K2De), __fmul_rn(w3D, K3De));
    Dm[(size_t)i * N + j] = val;
    Dm[(size_t)j * N + i] = val;
  }
  __threadfence_block();
  __syncthreads();
  if (N <= LK_DLDS) {   // the agglomeration re-reads the matrix once per merge: keep it in LDS when it fits
    for (int e = tid; e < N * N; e += LK_THREADS) L.dmat[e] = Dm[e];
    Dm = L.dmat;
    __syncthreads();
  }
  // ---- hierarchicalCluster (:416-540) ----
  int removeValue = -1;
  for (int iter = 0; iter < 2 * N + 2; ++iter) {
    // the element right after removeValue in the index list is skipped as first index (:446-450)
    int skip = -1;
    if (removeValue >= 0 && L.inlist[removeValue]) {
      for (int k = removeValue + 1; k < N; ++k)
        if (L.inlist[k]) {
          skip = k;
          break;
        }
    }
    const bool rv_listed = removeValue >= 0 && L.inlist[removeValue];
    float bv = -1.f;
    int ba = 0x7fffffff, bb = 0x7fffffff;
    // one wavefront per row, lanes along it: every lane sees its candidates in scan order, so its
    // running first-maximum is exact; the cross-lane reduction below breaks ties the same way
    for (int a = (tid >> 6); a < N; a += LK_THREADS / 64) {
      if (!L.inlist[a] || a == removeValue || a == skip) continue;
      const float* row = Dm + (size_t)a * N;
      for (int b = a + 1 + (tid & 63); b < N; b += 64) {
        if (!L.inlist[b]) continue;   // (the removed cluster is still listed in the scan right after its merge)
        const float v = row[b];
        if (v > bv) {
          bv = v;
          ba = a;
          bb = b;
        }
      }
    }
    // first maximum in scan order: larger value, then smaller (a, b)
    {
      const int lane = tid & 63, wave = tid >> 6;
      for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oa = __shfl_xor(ba, off), ob = __shfl_xor(bb, off);
        const bool take = ov > bv || (ov == bv && (oa < ba || (oa == ba && ob < bb)));
        if (take) {
          bv = ov;
          ba = oa;
          bb = ob;
        }
      }
      // every wavefront's candidate, by iteration parity (the slots of the iteration before are still being read), one
      // barrier, and every thread folds the 16 of them itself (wave-uniform reads): no single-thread section, three
      // barriers per merge instead of five
      const int par = iter & 1;
      if (lane == 0) {
        L.red2_v[par][wave] = bv;
        L.red2_a[par][wave] = ba;
        L.red2_b[par][wave] = bb;
      }
      __syncthreads();
      bv = -1.f;
      ba = 0x7fffffff;
      bb = 0x7fffffff;
      for (int k = 0; k < LK_THREADS / 64; ++k) {
        const float ov = L.red2_v[par][k];
        const int oa = L.red2_a[par][k], ob = L.red2_b[par][k];
        if (ov > bv || (ov == bv && (oa < ba || (oa == ba && ob < bb)))) {
          bv = ov;
          ba = oa;
          bb = ob;
        }
      }
      if (tid == 0 && rv_listed) L.inlist[removeValue] = 0;   // erased during this scan (read again only behind the barriers below)
    }
    const float maxSimilarity = bv;
    if (maxSimilarity < P.cutoff) break;
    const int first = ba, second = bb;
    const int S1 = L.clsize[first], S2 = L.clsize[second];
    // merge: the absorbed cluster's members are appended back to front (:493-496)
    for (int k = tid; k < S2; k += LK_THREADS) mem[(size_t)first * N + S1 + k] = mem[(size_t)second * N + (S2 - 1 - k)];
    // average-linkage row update (:515-523), all reads before any write
    float nv[(LK_CAP + LK_THREADS - 1) / LK_THREADS];
#pragma unroll
    for (int u = 0; u < (LK_CAP + LK_THREADS - 1) / LK_THREADS; ++u) {
      const int i = tid + u * LK_THREADS;
      nv[u] = 0.f;
      if (i < N) {
        const float da = Dm[(size_t)first * N + i], db = Dm[(size_t)second * N + i];
        if (P.linkage_type == 1) {
          const float a = __fmul_rn((float)S1, da), b = __fmul_rn((float)S2, db);
          nv[u] = (float)((1.0 / (double)(S1 + S2)) * (double)__fadd_rn(a, b));
        } else {
          // minimum / maximum linkage (:506-507, :525): the reference recomputes min / max of K over the merged
          // cluster's pairs (minimumLinkage :404-413, maximumLinkage :390-399); min over a union is the min of the two
          // clusters' minima -- the row entries hold exactly those (singletons start as K itself) -- so the update is
          // exact, no arithmetic.  A cluster i that is EMPTY (absorbed earlier) gets the reference's empty-loop value.
          // (`second` is empty by the time the reference's loop runs; a merge may also have absorbed a cluster that was
          // empty already -- the stale list entry of the scan, S2 = 0 -- and then changes nothing)
          const bool empty = L.clsize[i] == 0 || i == second;
          if (empty) nv[u] = P.linkage_type == 0 ? 1e20f : -1.f;
          else if (S2 == 0) nv[u] = da;
          else nv[u] = P.linkage_type == 0 ? fminf(da, db) : fmaxf(da, db);
        }
      }
    }
    __threadfence_block();
    __syncthreads();
#pragma unroll
    for (int u = 0; u < (LK_CAP + LK_THREADS - 1) / LK_THREADS; ++u) {
      const int i = tid + u * LK_THREADS;
      if (i < N) {
        Dm[(size_t)first * N + i] = nv[u];
        Dm[(size_t)i * N + first] = nv[u];
      }
    }
    if (tid == 0) {
      L.clsize[first] = S1 + S2;
      L.clsize[second] = 0;
    }
    removeValue = second;
    __threadfence_block();
    __syncthreads();
  }
  // ---- clusters with MORE than MinPts members, in index order (:530-538) ----
  if (label_out)
    for (int i = tid; i < N; i += LK_THREADS) label_out[i] = -1;
  __syncthreads();
  if (tid == 0) {
    int ncl = 0, w = 0;
    for (int c = 0; c < N; ++c) {
      const int sz = L.clsize[c];
      if (sz <= P.min_pts) continue;
      if (cl_start_out) cl_start_out[ncl] = w;
      for (int k = 0; k < sz; ++k) {
        const int m = mem[(size_t)c * N + k];
        if (members_out) members_out[w] = member_base + m;
        if (label_out) label_out[m] = ncl;
        ++w;
      }
      ++ncl;
    }
    if (cl_start_out) cl_start_out[ncl] = w;
    *ncl_out = ncl;
  }
}

// Frame form: a few workgroups over the frame's (a batch's frames') match lists; a model's scratch region starts after
// the regions of the models before it, a frame's after the frames' before it (`scratch_floats` each).  As in
// meanshift_models_kernel: ONE row of workgroups for all frames of the launch, every workgroup walks the frames, lists
// the models with more than MinPts matches and takes those whose number -- counted through the frames -- is its own
// modulo the grid; the workgroup that finishes a frame's last model lays the frame's cluster table out.  Frame f of a
// batch reads ITS depth map (maps.img[f]) and its copy of the working arrays (FrameBatch).
__global__ __launch_bounds__(LK_THREADS) void linkage_models_kernel(
    const mh_corr* __restrict__ corr0, const float4* __restrict__ depth40, const int32_t* __restrict__ model_off0,
    int n_models, DepthImage dimg, LinkageParams P, float* __restrict__ scratch0, size_t scratch_floats,
    int32_t* members0, int32_t* cl_start0, int32_t* ncl0, int max_clusters, int32_t* __restrict__ cl_model0,
    int32_t* __restrict__ cl_begin0, int32_t* __restrict__ cl_count0, int32_t* __restrict__ n_clusters_out0,
    int32_t* __restrict__ snap0, FrameCounts* counts0, unsigned int* ticket0, FrameBatch fbx, DepthMaps maps,
    int32_t* __restrict__ feedback) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LkLds& L = *reinterpret_cast<LkLds*>(smem);
  __shared__ unsigned long long busy[LK_THREADS / 64];
  __shared__ int lay_s[LK_THREADS / 64];
  const int G = (int)gridDim.x;
  const int n_frames = fbx.n > 1 ? fbx.n : 1;
  int rank = 0;   // models with work, counted through the frames of the launch
  for (int f = 0; f < n_frames; ++f) {
    const unsigned long long a = (unsigned long long)f * fbx.arena;
    const mh_corr* corr = frame_ptr(corr0, a);
    const float4* depth4 = frame_ptr(depth40, a);
    const int32_t* model_off = frame_ptr(model_off0, a);
    int32_t* members = frame_ptr(members0, a);
    int32_t* cl_start = frame_ptr(cl_start0, a);
    int32_t* ncl = frame_ptr(ncl0, a);
    FrameCounts* counts = frame_ptr(counts0, a);
    unsigned int* ticket = frame_ptr(ticket0, a);
    float* scratch = scratch0 + (size_t)f * scratch_floats;
    if (f) {
      dimg.img = maps.img[f];
      dimg.fill = maps.fill[f];
    }
    int n_busy = 0, mine = 0;
    for (int c0 = 0; c0 < n_models; c0 += LK_THREADS) {
      {
        const int mm = c0 + threadIdx.x;
        int nn = mm < n_models ? model_off[mm + 1] - model_off[mm] : 0;
        if (nn <= P.min_pts) nn = 0;   // a cluster is only emitted with MORE than MinPts members (CLUSTER_LINKAGE_CPU.hpp:537)
        const unsigned long long bl = __ballot(nn > 0);
        if ((threadIdx.x & 63) == 0) busy[threadIdx.x >> 6] = bl;
      }
      __syncthreads();
      for (int wd = 0; wd < LK_THREADS / 64; ++wd) {
        for (unsigned long long bits = busy[wd]; bits; bits &= bits - 1ull, ++rank, ++n_busy) {
          if (rank % G != (int)blockIdx.x) continue;
          ++mine;
          const int m = c0 + wd * 64 + __builtin_ctzll(bits);
          const int b = model_off[m];
          int n = model_off[m + 1] - b;
          size_t base = 0;   // floats before this model's region: 3 n'^2 per earlier model
          for (int mm = 0; mm < m; ++mm) {
            size_t k = (size_t)(model_off[mm + 1] - model_off[mm]);
            if (k > LK_CAP) k = LK_CAP;
            base += 3 * k * k;
          }
          if (n > LK_CAP) {
            if (threadIdx.x == 0) atomicOr(&counts->error, ERR_MS_CAP);
            n = LK_CAP;
          }
          __syncthreads();   // the previous model's LDS is done with
          if (base + 3 * (size_t)n * n > scratch_floats) {
            if (threadIdx.x == 0) {
              atomicOr(&counts->error, ERR_MS_CAP);
              ncl[m] = 0;
            }
          } else {
            float* A = scratch + base;
            float* Dm = A + (size_t)n * n;
            int32_t* mem = reinterpret_cast<int32_t*>(Dm + (size_t)n * n);
            linkage_body(L, corr + b, depth4 + b, n, dimg, P, A, Dm, mem, members + b, b, cl_start + b + m, ncl + m, nullptr);
          }
        }
      }
      __syncthreads();   // (busy[] is rewritten for the next thousand)
    }
    bool last = false;
    if (mine > 0) last = frame_work_done(ticket, (unsigned)mine, (unsigned)n_busy);
    else if (n_busy == 0) last = (int)blockIdx.x == f % G;
    if (!last) continue;   // (`last` is the same in every thread of the workgroup)
    if (threadIdx.x == 0 && feedback) feedback[f] = n_busy;   // what the next launches size their grids by
    // the frame's cluster table, by the whole workgroup (layout_cluster_table, common.h)
    const int min_pts = P.min_pts;
    const int k0 = layout_cluster_table(n_models, model_off, ncl, cl_start, 1, max_clusters, frame_ptr(cl_model0, a),
                                        frame_ptr(cl_begin0, a), frame_ptr(cl_count0, a), lay_s,
                                        [min_pts](int nn) { return nn > min_pts; });   // (a model without work was never clustered)
    if (threadIdx.x == 0) {
      if (k0 > max_clusters) atomicOr(&counts->error, ERR_CLUSTER_CAP);
      const int k = k0 < max_clusters ? k0 : max_clusters;
      counts->n_clusters = k;
      *frame_ptr(n_clusters_out0, a) = k;
      if (snap0) {
        snap0[4 * f] = counts->n_matches;
        snap0[4 * f + 1] = k;
      }
    }
  }
}

// Per-step form: problems [off[p], off[p+1]) of concatenated match arrays; labels per point.
__global__ __launch_bounds__(LK_THREADS) void linkage_batch_kernel(
    const mh_corr* __restrict__ corr, const float4* __restrict__ depth4, const int32_t* __restrict__ off,
    DepthImage dimg, LinkageParams P, float* __restrict__ scratch, size_t scratch_floats, int32_t* __restrict__ members,
    int32_t* __restrict__ cl_start, int32_t* __restrict__ ncl, int32_t* __restrict__ label) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LkLds& L = *reinterpret_cast<LkLds*>(smem);
  const int p = blockIdx.x;
  const int b = off[p];
  const int n = off[p + 1] - b;
  size_t base = 0;
  for (int q = 0; q < p; ++q) {
    const size_t k = (size_t)(off[q + 1] - off[q]);
    base += 3 * k * k;
  }
  if (n <= 0 || n > LK_CAP || base + 3 * (size_t)n * n > scratch_floats) {   // the host checks both before launching
    if (threadIdx.x == 0) ncl[p] = 0;
    return;
  }
  float* A = scratch + base;
  float* Dm = A + (size_t)n * n;
  int32_t* mem = reinterpret_cast<int32_t*>(Dm + (size_t)n * n);
  linkage_body(L, corr + b, depth4 + b, n, dimg, P, A, Dm, mem, members + b, 0, cl_start + b + p, ncl + p, label + b);
}

template <typename K>
void set_lds(K kernel) {
  static DynLds attr;   // one per kernel (template instantiation)
  attr.ensure(kernel, sizeof(LkLds));
}

}  // namespace

void launch_linkage_models(const mh_corr* corr, const float* depth4, const int32_t* model_off, int n_models,
                           const DepthImage& dimg, const LinkageParams& prm, float* scratch, size_t scratch_floats,
                           int32_t* members, int32_t* cl_start, int32_t* ncl, int max_clusters, int32_t* cl_model,
                           int32_t* cl_begin, int32_t* cl_count, int32_t* n_clusters_out, int32_t* snap,
                           FrameCounts* counts, unsigned int* ticket, hipStream_t s, int grid, const FrameBatch* batch,
                           const DepthMaps* maps, int32_t* feedback) {
  set_lds(linkage_models_kernel);
  const long n_frames = batch && batch->n > 1 ? batch->n : 1;
  const long all = std::max(1L, (long)n_models * n_frames);
  const int wgs = (int)std::max(1L, grid > 0 ? std::min((long)grid, all) : std::min(all, 256L));
  hipLaunchKernelGGL(linkage_models_kernel, dim3(wgs), dim3(LK_THREADS), sizeof(LkLds), s, corr,
                     reinterpret_cast<const float4*>(depth4), model_off, n_models, dimg, prm, scratch, scratch_floats,
                     members, cl_start, ncl, max_clusters, cl_model, cl_begin, cl_count, n_clusters_out, snap, counts,
                     ticket, batch ? *batch : FrameBatch(), (maps && n_frames > 1) ? *maps : DepthMaps(), feedback);
}

void launch_linkage_batch(const mh_corr* corr, const float* depth4, const int32_t* off, int n_problems,
                          const DepthImage& dimg, const LinkageParams& prm, float* scratch, size_t scratch_floats,
                          int32_t* members, int32_t* cl_start, int32_t* ncl, int32_t* label, hipStream_t s) {
  if (n_problems <= 0) return;
  set_lds(linkage_batch_kernel);
  hipLaunchKernelGGL(linkage_batch_kernel, dim3(n_problems), dim3(LK_THREADS), sizeof(LkLds), s, corr,
                     reinterpret_cast<const float4*>(depth4), off, dimg, prm, scratch, scratch_floats, members, cl_start,
                     ncl, label);
}

}  // namespace mh
