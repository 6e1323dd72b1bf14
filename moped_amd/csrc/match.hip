// MATCH: brute-force 128-D 2-NN by squared L2 on gfx950.
//
// Replaces the kd-tree search of MATCH_ANN_CPU::process
// (moped2/libmoped/src/match/MATCH_ANN_CPU.hpp:155-165) and
// MATCH_FLANN_CPU::process (…/MATCH_FLANN_CPU.hpp:133-191) with an exact search.
//
// Canonical arithmetic (shared bit-for-bit with oracle/oracle.cpp):
//   dot(a,b)  = fmaf chain over k = 0..127 starting from 0
//   dist(q,d) = max(0, fmaf(-2, dot(q,d), dot(q,q) + dot(d,d)))
//   ties on distance -> lower row index
// so the result does not depend on tiling, split count or shard layout.
//
// Kernel shape ("one query group per wavefront"): a wavefront owns TQ queries
// whose coordinates are wave-uniform and therefore live in SGPRs (scalar loads);
// its 64 lanes each own TD database rows of the current LDS tile, so every
// v_fma_f32 is (SGPR query coordinate) x (VGPR row coordinate) -> per-lane
// accumulator.  A 1024-thread workgroup (16 wavefronts = 128 queries) shares one
// double-buffered 128-row LDS tile, which is what makes the database stream
// through L2/HBM only Q/128 times.  Per-lane running top-2 per query, one
// wavefront min-reduce at the end of the block's database split.
#include "common.h"

namespace mh {

namespace {

constexpr int TQ = 8;                  // queries per wavefront
constexpr int NWAVES = 16;             // wavefronts per workgroup
constexpr int QB = TQ * NWAVES;        // queries per workgroup (128)
constexpr int TILE_ROWS = 128;         // database rows per LDS tile (2 per lane)
constexpr int LDS_STRIDE = DIM + 4;    // floats per staged row: +16 B keeps ds_read_b128 conflict-free
constexpr int TILE_FLOATS = TILE_ROWS * LDS_STRIDE;
constexpr int MATCH_THREADS = NWAVES * 64;
constexpr int TARGET_BLOCKS = 768;     // 3 workgroups per CU over the launch

__device__ __forceinline__ float dot_chain_lds(const float* a, const float* b) {
  float s = 0.f;
  for (int k = 0; k < DIM; ++k) s = fmaf(a[k], b[k], s);
  return s;
}

// ---- A1 ----------------------------------------------------------------------
// One thread per descriptor, rows staged through LDS so global traffic is
// coalesced.  Arithmetic is MATCH_ANN_CPU::norm's: sequential fp32 sum of
// squares (no contraction), scale = (float)(1.0 / sqrtf(sum)).
constexpr int NORM_ROWS = 64;
constexpr int NORM_STRIDE = DIM + 1;

template <bool NORMALIZE>
__global__ __launch_bounds__(NORM_ROWS) void normalize_kernel(float* __restrict__ desc,
                                                             float* __restrict__ norm_out, int n) {
  __shared__ float tile[NORM_ROWS * NORM_STRIDE];
  const int t = threadIdx.x;
  const size_t row0 = (size_t)blockIdx.x * NORM_ROWS;
  const int rows = min(NORM_ROWS, n - (int)row0);
  const float* src = desc + row0 * DIM;
  for (int i = 0; i < DIM; ++i) {
    int e = i * NORM_ROWS + t;  // element of the block's [rows x 128] slab
    int r = e >> 7, c = e & 127;
    tile[r * NORM_STRIDE + c] = (r < rows) ? src[e] : 0.f;
  }
  __syncthreads();
  float* mine = tile + t * NORM_STRIDE;
  if (NORMALIZE) {
    float s = 0.f;
    for (int x = 0; x < DIM; ++x) s = __fadd_rn(s, __fmul_rn(mine[x], mine[x]));
    const float inv = (float)(1.0 / (double)sqrtf(s));
    for (int x = 0; x < DIM; ++x) mine[x] = __fmul_rn(mine[x], inv);
  }
  const float nn = dot_chain_lds(mine, mine);
  if (t < rows && norm_out) norm_out[row0 + t] = nn;
  if (NORMALIZE) {
    __syncthreads();
    float* dst = desc + row0 * DIM;
    for (int i = 0; i < DIM; ++i) {
      int e = i * NORM_ROWS + t;
      int r = e >> 7, c = e & 127;
      if (r < rows) dst[e] = tile[r * NORM_STRIDE + c];
    }
  }
}

// ---- top-2 helpers -------------------------------------------------------------
struct Best {
  float b1, b2;
  int i1;
};

// Fold candidate (v, idx) into a lane-private running top-2.  Rows reach a lane
// in increasing index order, so strict '<' keeps the lower index on ties.
__device__ __forceinline__ void fold(Best& s, float v, int idx) {
  const bool lt = v < s.b1;
  s.b2 = __builtin_amdgcn_fmed3f(s.b1, s.b2, v);  // median(b1 <= b2, v) = new second best
  s.i1 = lt ? idx : s.i1;
  s.b1 = fminf(s.b1, v);
}

// Merge two disjoint top-2 sets; lower index wins a tie on the best distance.
__device__ __forceinline__ void merge(Best& a, float ob1, float ob2, int oi1) {
  const bool take = (ob1 < a.b1) || (ob1 == a.b1 && (unsigned)oi1 < (unsigned)a.i1);
  const float lose1 = take ? a.b1 : ob1;          // the best that did not win
  const float s2 = fminf(a.b2, ob2);
  a.b2 = fminf(lose1, s2);
  a.b1 = take ? ob1 : a.b1;
  a.i1 = take ? oi1 : a.i1;
}

// ---- the match kernel -----------------------------------------------------------
// grid.x = query groups of QB, grid.y = database splits.  Each block scans rows
// [split*rows_per_split, +rows_per_split) and writes one Top2 per query.
__global__ __launch_bounds__(MATCH_THREADS) void match_kernel(
    const float* __restrict__ qn, const float* __restrict__ qnorm, int Q,
    const float* __restrict__ db, const float* __restrict__ dnorm, int N,
    int tiles_per_split, int32_t index_base, Top2* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 tiles
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q0 = (blockIdx.x * NWAVES + wave) * TQ;  // wave-uniform
  const int n_tiles = (N + TILE_ROWS - 1) / TILE_ROWS;
  const int tile_begin = blockIdx.y * tiles_per_split;
  const int tile_end = min(tile_begin + tiles_per_split, n_tiles);

  // wave-uniform query row pointers (clamped so out-of-range groups stay in bounds)
  const float* qrow[TQ];
  float qn2[TQ];
#pragma unroll
  for (int t = 0; t < TQ; ++t) {
    const int qi = min(q0 + t, Q - 1);
    qrow[t] = qn + (size_t)qi * DIM;
    qn2[t] = qnorm[qi];
  }

  Best st[TQ];
#pragma unroll
  for (int t = 0; t < TQ; ++t) {
    st[t].b1 = __builtin_inff();
    st[t].b2 = __builtin_inff();
    st[t].i1 = -1;
  }

  // staging: 4096 float4 per tile over 1024 threads = 4 each, coalesced
  float4 stage[4];
  auto stage_load = [&](int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * MATCH_THREADS;
      const int r = c >> 5, c4 = c & 31;
      const int row = tile * TILE_ROWS + r;
      stage[i] = (row < N) ? *reinterpret_cast<const float4*>(db + (size_t)row * DIM + c4 * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * MATCH_THREADS;
      const int r = c >> 5, c4 = c & 31;
      *reinterpret_cast<float4*>(lds + buf * TILE_FLOATS + r * LDS_STRIDE + c4 * 4) = stage[i];
    }
  };

  if (tile_begin < tile_end) {
    stage_load(tile_begin);
    stage_store(0);
  }
  __syncthreads();

  int buf = 0;
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const bool more = tile + 1 < tile_end;
    if (more) stage_load(tile + 1);

    const int row_a = tile * TILE_ROWS + lane;
    const int row_b = row_a + 64;
    const float dn_a = (row_a < N) ? dnorm[row_a] : 0.f;
    const float dn_b = (row_b < N) ? dnorm[row_b] : 0.f;

    const float* la = lds + buf * TILE_FLOATS + lane * LDS_STRIDE;
    const float* lb = la + 64 * LDS_STRIDE;
    float acc_a[TQ], acc_b[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
      acc_a[t] = 0.f;
      acc_b[t] = 0.f;
    }
#pragma unroll 2
    for (int kc = 0; kc < DIM / 4; ++kc) {
      const float4 a = *reinterpret_cast<const float4*>(la + kc * 4);
      const float4 b = *reinterpret_cast<const float4*>(lb + kc * 4);
#pragma unroll
      for (int t = 0; t < TQ; ++t) {
        const float4 q = *reinterpret_cast<const float4*>(qrow[t] + kc * 4);  // scalar load
        acc_a[t] = fmaf(q.x, a.x, acc_a[t]);
        acc_b[t] = fmaf(q.x, b.x, acc_b[t]);
        acc_a[t] = fmaf(q.y, a.y, acc_a[t]);
        acc_b[t] = fmaf(q.y, b.y, acc_b[t]);
        acc_a[t] = fmaf(q.z, a.z, acc_a[t]);
        acc_b[t] = fmaf(q.z, b.z, acc_b[t]);
        acc_a[t] = fmaf(q.w, a.w, acc_a[t]);
        acc_b[t] = fmaf(q.w, b.w, acc_b[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
      const float da = fmaxf(0.f, fmaf(-2.f, acc_a[t], qn2[t] + dn_a));
      const float db2 = fmaxf(0.f, fmaf(-2.f, acc_b[t], qn2[t] + dn_b));
      if (row_a < N) fold(st[t], da, row_a);
      if (row_b < N) fold(st[t], db2, row_b);
    }

    if (more) stage_store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // wavefront min-reduce of the per-lane top-2, one query at a time
#pragma unroll
  for (int t = 0; t < TQ; ++t) {
    Best s = st[t];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ob1 = __shfl_xor(s.b1, off);
      const float ob2 = __shfl_xor(s.b2, off);
      const int oi1 = __shfl_xor(s.i1, off);
      merge(s, ob1, ob2, oi1);
    }
    const int qi = q0 + t;
    if (lane == 0 && qi < Q) {
      Top2 o;
      o.d1 = s.b1;
      o.d2 = s.b2;
      o.i1 = (s.i1 >= 0) ? s.i1 + index_base : -1;
      o.pad = 0;
      partial[(size_t)blockIdx.y * Q + qi] = o;
    }
  }
}

// Combine the splits of one shard: one thread per query.
__global__ void combine_splits_kernel(const Top2* __restrict__ partial, int S, int Q,
                                      int32_t* __restrict__ idx1, float* __restrict__ d1,
                                      float* __restrict__ d2) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q) return;
  Best s = {__builtin_inff(), __builtin_inff(), -1};
  for (int k = 0; k < S; ++k) {
    const Top2 p = partial[(size_t)k * Q + q];
    if (p.i1 < 0) continue;
    merge(s, p.d1, p.d2, p.i1);
  }
  idx1[q] = s.i1;
  d1[q] = s.b1;
  d2[q] = s.b2;
}

// Exchange-1 merge of S shards' local top-2, laid out [S][Q].
__global__ void merge_shards_kernel(const int32_t* __restrict__ idx1_s,
                                    const float* __restrict__ d1_s,
                                    const float* __restrict__ d2_s, int S, int Q,
                                    int32_t* __restrict__ idx1, float* __restrict__ d1,
                                    float* __restrict__ d2) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q) return;
  Best s = {__builtin_inff(), __builtin_inff(), -1};
  for (int k = 0; k < S; ++k) {
    const int32_t i = idx1_s[(size_t)k * Q + q];
    if (i < 0) continue;
    merge(s, d1_s[(size_t)k * Q + q], d2_s[(size_t)k * Q + q], i);
  }
  idx1[q] = s.i1;
  d1[q] = s.b1;
  d2[q] = s.b2;
}

int splits_for(int Q, int N) {
  const int qgroups = (Q + QB - 1) / QB;
  const int n_tiles = (N + TILE_ROWS - 1) / TILE_ROWS;
  int S = (TARGET_BLOCKS + qgroups - 1) / qgroups;
  if (S > n_tiles) S = n_tiles;
  if (S < 1) S = 1;
  return S;
}

}  // namespace

void launch_normalize(float* desc, float* norm_out, int n, hipStream_t s) {
  if (n <= 0) return;
  const int blocks = (n + NORM_ROWS - 1) / NORM_ROWS;
  hipLaunchKernelGGL(normalize_kernel<true>, dim3(blocks), dim3(NORM_ROWS), 0, s, desc, norm_out, n);
}

void launch_row_norms(const float* desc, float* norm_out, int n, hipStream_t s) {
  if (n <= 0) return;
  const int blocks = (n + NORM_ROWS - 1) / NORM_ROWS;
  hipLaunchKernelGGL(normalize_kernel<false>, dim3(blocks), dim3(NORM_ROWS), 0, s,
                     const_cast<float*>(desc), norm_out, n);
}

size_t match_scratch_elems(int Q, int N) { return (size_t)splits_for(Q, N) * (size_t)(Q > 0 ? Q : 1); }

void launch_match(const float* qn, const float* qnorm, int Q, const float* db, const float* dnorm,
                  int N, int32_t index_base, Top2* scratch, int32_t* idx1, float* d1, float* d2,
                  hipStream_t s) {
  if (Q <= 0) return;
  const int S = (N > 0) ? splits_for(Q, N) : 0;
  if (S > 0) {
    const int qgroups = (Q + QB - 1) / QB;
    const int n_tiles = (N + TILE_ROWS - 1) / TILE_ROWS;
    const int tiles_per_split = (n_tiles + S - 1) / S;
    const size_t lds_bytes = 2 * TILE_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(match_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      attr_set = true;
    }
    hipLaunchKernelGGL(match_kernel, dim3(qgroups, S), dim3(MATCH_THREADS), lds_bytes, s, qn, qnorm,
                       Q, db, dnorm, N, tiles_per_split, index_base, scratch);
  }
  hipLaunchKernelGGL(combine_splits_kernel, dim3((Q + 255) / 256), dim3(256), 0, s, scratch, S, Q,
                     idx1, d1, d2);
}

void launch_match_merge(const int32_t* idx1_s, const float* d1_s, const float* d2_s, int S, int Q,
                        int32_t* idx1, float* d1, float* d2, hipStream_t s) {
  if (Q <= 0) return;
  hipLaunchKernelGGL(merge_shards_kernel, dim3((Q + 255) / 256), dim3(256), 0, s, idx1_s, d1_s, d2_s,
                     S, Q, idx1, d1, d2);
}

}  // namespace mh
