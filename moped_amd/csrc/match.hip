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
#include <cstdlib>
#include <algorithm>

#include "common.h"

MH_TRACE_TU()

namespace mh {

namespace {

constexpr int TQ = 8;                  // queries per wavefront (pack_queries_kernel assumes 8)
constexpr int NWAVES = 16;             // wavefronts per workgroup
constexpr int QB = TQ * NWAVES;        // queries per workgroup (128)
constexpr int TILE_ROWS = 128;         // database rows per LDS tile (2 per lane)
constexpr int MATCH_THREADS = NWAVES * 64;
constexpr int TARGET_BLOCKS_DEFAULT = 768;  // 3 workgroups per CU over the launch (MH_MATCH_BLOCKS overrides)

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

constexpr int NORM_THREADS = 256;   // four wavefronts move the slab (16-byte accesses, all in flight); the first does the rows

template <bool NORMALIZE>
__global__ __launch_bounds__(NORM_THREADS) void normalize_kernel(float* __restrict__ desc,
                                                                float* __restrict__ norm_out, int n,
                                                                const int32_t* __restrict__ n_dev) {
  MH_TRACE_SCOPE(mh::TK_NORMALIZE);
  __shared__ float tile[NORM_ROWS * NORM_STRIDE];
  const int t = threadIdx.x;
  const size_t row0 = (size_t)blockIdx.x * NORM_ROWS;
  if (blockIdx.y) {   // image of a batch: its n rows, its count word
    desc += (size_t)blockIdx.y * n * DIM;
    norm_out += (size_t)blockIdx.y * n;
    if (n_dev) n_dev += blockIdx.y;
  }
  if (n_dev) n = min(n, *n_dev);   // row count known only on the device (features extracted there)
  const int rows = min(NORM_ROWS, n - (int)row0);
  if (rows <= 0) return;
  constexpr int PER_THREAD = NORM_ROWS * DIM / 4 / NORM_THREADS;   // float4s of the [rows x 128] slab per thread
  const float4* src = reinterpret_cast<const float4*>(desc + row0 * DIM);
  float4 v[PER_THREAD];
#pragma unroll
  for (int i = 0; i < PER_THREAD; ++i) {
    const int e = i * NORM_THREADS + t;   // float4 index: row e / 32, columns 4 (e % 32) ..
    v[i] = (e >> 5) < rows ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < PER_THREAD; ++i) {
    const int e = i * NORM_THREADS + t;
    float* d = tile + (e >> 5) * NORM_STRIDE + (e & 31) * 4;
    d[0] = v[i].x;
    d[1] = v[i].y;
    d[2] = v[i].z;
    d[3] = v[i].w;
  }
  __syncthreads();
  if (t < NORM_ROWS) {
    float* mine = tile + t * NORM_STRIDE;
    if (NORMALIZE) {
      float s = 0.f;
      for (int x = 0; x < DIM; ++x) s = __fadd_rn(s, __fmul_rn(mine[x], mine[x]));
      const float inv = (float)(1.0 / (double)sqrtf(s));
      for (int x = 0; x < DIM; ++x) mine[x] = __fmul_rn(mine[x], inv);
    }
    const float nn = dot_chain_lds(mine, mine);
    if (t < rows && norm_out) norm_out[row0 + t] = nn;
  }
  if (NORMALIZE) {
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(desc + row0 * DIM);
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int e = i * NORM_THREADS + t;
      const float* d = tile + (e >> 5) * NORM_STRIDE + (e & 31) * 4;
      if ((e >> 5) < rows) dst[e] = make_float4(d[0], d[1], d[2], d[3]);
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
__device__ __forceinline__ void fold(float& b1, float& b2, int& i1, float v, int idx) {
  const bool lt = v < b1;
  b2 = __builtin_amdgcn_fmed3f(b1, b2, v);  // median(b1 <= b2, v) = new second best
  i1 = lt ? idx : i1;
  b1 = lt ? v : b1;                         // select on the same compare (a min would re-canonicalise b1)
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

// ---- query packing ----------------------------------------------------------------
// P[g][k][8]: the 8 queries of group g interleaved per coordinate, zero padded, so a
// wavefront fetches "coordinate k of its 8 queries" as 8 consecutive dwords and four
// coordinates with two s_load_dwordx16.  Even-aligned SGPR pairs = query pairs.
__global__ void pack_queries_kernel(const float* __restrict__ qn, const float* __restrict__ qnorm,
                                    int Q, const int32_t* __restrict__ q_count, float* __restrict__ P,
                                    float* __restrict__ Pnorm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one element of P
  const int G = (Q + TQ - 1) / TQ;
  if (i >= G * DIM * TQ) return;
  if (q_count) Q = min(Q, *q_count);
  const int t = i & (TQ - 1), k = (i >> 3) & (DIM - 1), g = i >> 10;
  const int q = g * TQ + t;
  P[i] = (q < Q) ? qn[(size_t)q * DIM + k] : 0.f;
  if (k == 0) Pnorm[g * TQ + t] = (q < Q) ? qnorm[q] : 0.f;
}

// ---- the match kernel -----------------------------------------------------------
// 1-D grid of (query blocks of QB) x (database splits).  Each block scans the rows of
// its split and writes one Top2 per query.
//
// Inner loop, per wavefront and 2 coordinates ("chunk"): one s_load_dwordx16 brings
// 2 coordinates x 8 queries into SGPRs, two ds_read_b64 bring 2 coordinates of the
// lane's two rows into VGPRs, then 16 v_pk_fma_f32: each packs a QUERY PAIR (an
// even-aligned SGPR pair) against one row coordinate broadcast to both halves by
// op_sel -- no register moves on either pipe.  The loads of chunk c+1 are issued
// (inline asm, so the compiler cannot sink them) before the FMAs of chunk c and
// waited for with one lgkmcnt(0) after them: SMEM returns out of order, so a
// counted wait is not available, but a full chunk of FMAs (x4 wavefronts per SIMD)
// covers the latency.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int LDS_STRIDE = DIM + 4;    // floats per staged row: +16 B keeps ds_read_b128 conflict-free
constexpr int TILE_FLOATS = TILE_ROWS * LDS_STRIDE;
constexpr int NCHUNK = DIM / 4;        // four coordinates per pipeline stage

// The accumulators ride through the issue/wait statements as in-out operands: that
// pins the FMAs of chunk c between the issue of chunk c+1 and its wait (otherwise
// the scheduler is free to hoist every load of the tile to the front and spill).
#define ACC_TIE(a) "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
template <int KC>
__device__ __forceinline__ void chunk_issue(const float* qp, unsigned la, v16f& qa, v16f& qb, v4f& ra, v4f& rb,
                                            v2f (&acc)[TQ]) {
  asm volatile("s_load_dwordx16 %0, %12, %14\n\ts_load_dwordx16 %1, %12, %15\n\t"
               "ds_read_b128 %2, %13 offset:%16\n\tds_read_b128 %3, %13 offset:%17"
               : "=s"(qa), "=s"(qb), "=v"(ra), "=v"(rb), ACC_TIE(acc)
               : "s"(qp), "v"(la), "i"(KC * 128), "i"(KC * 128 + 64), "i"(KC * 16),
                 "i"(KC * 16 + 64 * LDS_STRIDE * 4)
               : "memory");
}
// One wait for everything in flight; the operands tie the loaded registers to the
// wait so no consumer can be scheduled above it.
__device__ __forceinline__ void chunk_wait(v16f& qa, v16f& qb, v4f& ra, v4f& rb, v2f (&acc)[TQ]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(qa), "+s"(qb), "+v"(ra), "+v"(rb), ACC_TIE(acc) : : "memory");
}
// qa = coordinate k0: queries 0..7, k1: 0..7 ; qb = k2, k3.  ra / rb = row a / b at k0..k3.
// acc[2*tp + row]: query pair tp x row.  Each chain sees k ascending.
// v_pk_fma_f32 D = S0 * S1 + D with S0 = an aligned SGPR pair (two queries) and S1 =
// one half of a VGPR pair broadcast to both lanes of the packed op:
//   LO: op_sel_hi:[1,0,1]  -> S1.lo for both halves ; HI: op_sel:[0,1,0] -> S1.hi for both.
#define PKFMA_LO(acc, q, r) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(q), "v"(r))
#define PKFMA_HI(acc, q, r) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "s"(q), "v"(r))
__device__ __forceinline__ void chunk_fma(const v16f& qa, const v16f& qb, const v4f& ra, const v4f& rb,
                                          v2f (&acc)[TQ]) {
  const v2f a01 = (v2f){ra.x, ra.y}, a23 = (v2f){ra.z, ra.w};
  const v2f b01 = (v2f){rb.x, rb.y}, b23 = (v2f){rb.z, rb.w};
#pragma unroll
  for (int tp = 0; tp < TQ / 2; ++tp) {
    const v2f q0 = (v2f){qa[2 * tp], qa[2 * tp + 1]}, q1 = (v2f){qa[8 + 2 * tp], qa[8 + 2 * tp + 1]};
    const v2f q2 = (v2f){qb[2 * tp], qb[2 * tp + 1]}, q3 = (v2f){qb[8 + 2 * tp], qb[8 + 2 * tp + 1]};
    PKFMA_LO(acc[2 * tp], q0, a01);
    PKFMA_LO(acc[2 * tp + 1], q0, b01);
    PKFMA_HI(acc[2 * tp], q1, a01);
    PKFMA_HI(acc[2 * tp + 1], q1, b01);
    PKFMA_LO(acc[2 * tp], q2, a23);
    PKFMA_LO(acc[2 * tp + 1], q2, b23);
    PKFMA_HI(acc[2 * tp], q3, a23);
    PKFMA_HI(acc[2 * tp + 1], q3, b23);
  }
}
template <int KC>
__device__ __forceinline__ void chunk_pipe(const float* qp, unsigned la, v16f& qa0, v16f& qb0, v4f& ra0, v4f& rb0,
                                           v16f& qa1, v16f& qb1, v4f& ra1, v4f& rb1, v2f (&acc)[TQ]) {
  if constexpr (KC < NCHUNK) {
    if constexpr (KC + 1 < NCHUNK) chunk_issue<KC + 1>(qp, la, qa1, qb1, ra1, rb1, acc);
    chunk_fma(qa0, qb0, ra0, rb0, acc);
    if constexpr (KC + 1 < NCHUNK) chunk_wait(qa1, qb1, ra1, rb1, acc);
    chunk_pipe<KC + 1>(qp, la, qa1, qb1, ra1, rb1, qa0, qb0, ra0, rb0, acc);
  }
}

__global__ __launch_bounds__(MATCH_THREADS) void match_kernel(
    const float* __restrict__ P, const float* __restrict__ Pnorm, int Q,
    const float* __restrict__ db, const float* __restrict__ dnorm, int N,
    int tiles_per_split, int n_splits, int32_t index_base, Top2* __restrict__ partial,
    const int32_t* __restrict__ q_count) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 tiles
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware block -> (query group block, split) map.  Workgroups are dealt to the
  // 8 XCDs round-robin by linear id, and every XCD has its own L2: giving each XCD
  // whole splits (all query blocks of a split share that split's DB rows) makes the
  // DB leave HBM/MALL once instead of once per XCD.  Pure locality: any placement is
  // correct.
  const int nqb = gridDim.x / n_splits;                  // query blocks
  int qblock, split;
  {
    const int L = blockIdx.x;
    if ((n_splits & 7) == 0) {
      const int x = L & 7, j = L >> 3;
      split = x + 8 * (j / nqb);
      qblock = j % nqb;
    } else {
      split = L / nqb;
      qblock = L % nqb;
    }
  }
  // Q = capacity (stride of `partial`); with q_count the live queries are [0, min(Q, *q_count))
  // and the query blocks beyond them leave at once
  const int Qe = q_count ? min(Q, *q_count) : Q;
  const int n_groups = (Qe + TQ - 1) / TQ;
  if (qblock * NWAVES >= n_groups) return;                           // uniform over the workgroup
  const int g_raw = qblock * NWAVES + wave;                          // wave-uniform
  const bool live = g_raw < n_groups;
  const int g = live ? g_raw : n_groups - 1;                         // surplus waves redo the last group
  const int q0 = g * TQ;
  // the packed-query base must sit in SGPRs for s_load: make the uniformity explicit
  const unsigned long long qp_bits = (unsigned long long)(P + (size_t)g * (DIM * TQ));
  const unsigned qp_lo = __builtin_amdgcn_readfirstlane((unsigned)qp_bits);
  const unsigned qp_hi = __builtin_amdgcn_readfirstlane((unsigned)(qp_bits >> 32));
  const float* qp = (const float*)(((unsigned long long)qp_hi << 32) | qp_lo);
  const int n_tiles = (N + TILE_ROWS - 1) / TILE_ROWS;
  const int tile_begin = split * tiles_per_split;
  const int tile_end = min(tile_begin + tiles_per_split, n_tiles);

  v2f nq[TQ / 2];  // dot(q,q) of the query pairs; parked in VGPRs (the SGPR file is for the query stream)
#pragma unroll
  for (int tp = 0; tp < TQ / 2; ++tp) {
    float x = Pnorm[q0 + 2 * tp], y = Pnorm[q0 + 2 * tp + 1];
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(nq[tp].x), "=v"(nq[tp].y) : "s"(x), "s"(y));
  }

  // per-lane running top-2 per query
  float b1[TQ], b2[TQ];
  int i1[TQ];
#pragma unroll
  for (int t = 0; t < TQ; ++t) {
    b1[t] = __builtin_inff();
    b2[t] = __builtin_inff();
    i1[t] = -1;
  }

  // Staging.  The DB is padded to whole tiles (zero rows, +inf norms), so there are
  // no bounds checks; a tile is 64 KB contiguous, each thread copies 4 x 16 B at
  // (wave-uniform tile base) + (constant per-thread offset).
  float4 st0, st1, st2, st3;  // named registers (an array captured by a lambda ends up in scratch)
  const unsigned st_off = (unsigned)tid * 16u;
#define STAGE_LOAD(tile_)                                                                          \
  do {                                                                                             \
    const char* tb_ = reinterpret_cast<const char*>(db) + (size_t)(tile_) * (TILE_ROWS * DIM * 4); \
    st0 = *reinterpret_cast<const float4*>(tb_ + st_off);                                          \
    st1 = *reinterpret_cast<const float4*>(tb_ + st_off + 1 * (MATCH_THREADS * 16));               \
    st2 = *reinterpret_cast<const float4*>(tb_ + st_off + 2 * (MATCH_THREADS * 16));               \
    st3 = *reinterpret_cast<const float4*>(tb_ + st_off + 3 * (MATCH_THREADS * 16));               \
  } while (0)
  // thread c = tid + i*1024 holds chunk c4 = c & 31 of tile row r = c >> 5
  float* const st_dst = lds + (tid >> 5) * LDS_STRIDE + (tid & 31) * 4;
#define STAGE_STORE(buf_)                                                                  \
  do {                                                                                     \
    float* d_ = st_dst + (buf_) * TILE_FLOATS;                                             \
    *reinterpret_cast<float4*>(d_) = st0;                                                  \
    *reinterpret_cast<float4*>(d_ + 32 * LDS_STRIDE) = st1;                                \
    *reinterpret_cast<float4*>(d_ + 64 * LDS_STRIDE) = st2;                                \
    *reinterpret_cast<float4*>(d_ + 96 * LDS_STRIDE) = st3;                                \
  } while (0)

  if (tile_begin < tile_end) {
    STAGE_LOAD(tile_begin);
    STAGE_STORE(0);
  }
  __syncthreads();

  const unsigned lds_base = (unsigned)(size_t)lds;  // LDS byte address of the dynamic region
  int buf = 0;
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const bool more = tile + 1 < tile_end;
    if (more) STAGE_LOAD(tile + 1);

    const int row_a = tile * TILE_ROWS + lane;
    const int row_b = row_a + 64;
    const float dn_a = dnorm[row_a];  // +inf on padding rows: they can never enter a top-2
    const float dn_b = dnorm[row_b];

    const unsigned la = lds_base + (unsigned)((buf * TILE_FLOATS + lane * LDS_STRIDE) * 4);
    v2f acc[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) acc[t] = (v2f){0.f, 0.f};
    if (live) {  // wavefronts beyond the last query group only help with staging (small-Q launches)
      v16f qa0, qb0, qa1, qb1;
      v4f ra0, rb0, ra1, rb1;
      chunk_issue<0>(qp, la, qa0, qb0, ra0, rb0, acc);
      chunk_wait(qa0, qb0, ra0, rb0, acc);
      chunk_pipe<0>(qp, la, qa0, qb0, ra0, rb0, qa1, qb1, ra1, rb1, acc);
    }
    // distances + fold: row a first, then row b (ascending row index within the lane)
#pragma unroll
    for (int tp = 0; tp < TQ / 2; ++tp) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float dnr = r ? dn_b : dn_a;
        const int row = r ? row_b : row_a;
        const v2f d = __builtin_elementwise_fma((v2f){-2.f, -2.f}, acc[2 * tp + r], nq[tp] + (v2f){dnr, dnr});
        fold(b1[2 * tp], b2[2 * tp], i1[2 * tp], fmaxf(d.x, 0.f), row);
        fold(b1[2 * tp + 1], b2[2 * tp + 1], i1[2 * tp + 1], fmaxf(d.y, 0.f), row);
      }
    }

    if (more) STAGE_STORE(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // wavefront min-reduce of the per-lane top-2, one query at a time
#pragma unroll
  for (int t = 0; t < TQ; ++t) {
    Best s = {b1[t], b2[t], i1[t]};
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ob1 = __shfl_xor(s.b1, off);
      const float ob2 = __shfl_xor(s.b2, off);
      const int oi1 = __shfl_xor(s.i1, off);
      if (oi1 >= 0) merge(s, ob1, ob2, oi1);
    }
    const int qi = q0 + t;
    if (live && lane == 0 && qi < Qe) {
      Top2 o;
      o.d1 = s.b1;
      o.d2 = s.b2;
      o.i1 = (s.i1 >= 0) ? s.i1 + index_base : -1;
      o.pad = 0;
      partial[(size_t)split * Q + qi] = o;
    }
  }
}

// Combine the splits of one shard: one thread per query.
__global__ void combine_splits_kernel(const Top2* __restrict__ partial, int S, int Q,
                                      const int32_t* __restrict__ q_count, int32_t* __restrict__ idx1,
                                      float* __restrict__ d1, float* __restrict__ d2, RowMap rmap) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q) return;
  Best s = {__builtin_inff(), __builtin_inff(), -1};
  if (q_count && q >= *q_count) S = 0;   // no such query in this frame: "no neighbour"
  // eight partial results in flight per step (the merge itself is a short dependent chain)
  for (int k0 = 0; k0 < S; k0 += 8) {
    Top2 p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      p[j] = (k0 + j < S) ? partial[(size_t)(k0 + j) * Q + q] : Top2{__builtin_inff(), __builtin_inff(), -1};
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (p[j].i1 >= 0) merge(s, p[j].d1, p[j].d2, p[j].i1);
  }
  idx1[q] = s.i1 >= 0 ? row_to_global(rmap, s.i1) : -1;   // (the partials hold local rows: local order = global order)
  d1[q] = s.b1;
  d2[q] = s.b2;
}

// Exchange-1 merge of S shards' local top-2, laid out [S][Q].
__global__ void merge_shards_kernel(const int32_t* __restrict__ idx1_s,
                                    const float* __restrict__ d1_s,
                                    const float* __restrict__ d2_s, int S, int Q,
                                    size_t shard_stride, int32_t* __restrict__ idx1,
                                    float* __restrict__ d1, float* __restrict__ d2) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q) return;
  Best s = {__builtin_inff(), __builtin_inff(), -1};
  for (int k = 0; k < S; ++k) {
    const int32_t i = idx1_s[k * shard_stride + q];
    if (i < 0) continue;
    merge(s, d1_s[k * shard_stride + q], d2_s[k * shard_stride + q], i);
  }
  idx1[q] = s.i1;
  d1[q] = s.b1;
  d2[q] = s.b2;
}

int splits_for(int Q, int N) {
  const int qgroups = (Q + QB - 1) / QB;
  const int n_tiles = (N + TILE_ROWS - 1) / TILE_ROWS;
  static const int target = std::max(1, exp_int("MH_MATCH_BLOCKS", TARGET_BLOCKS_DEFAULT));
  int S = (target + qgroups - 1) / qgroups;
  if (S >= 8) S = (S + 3) / 8 * 8;  // whole splits per XCD (see match_kernel); to the nearest multiple: fewer, longer workgroups win
  if (S > n_tiles) S = n_tiles;
  if (S < 1) S = 1;
  return S;
}

}  // namespace

void launch_normalize(float* desc, float* norm_out, int n, hipStream_t s, const int32_t* n_dev) {
  if (n <= 0) return;
  const int blocks = (n + NORM_ROWS - 1) / NORM_ROWS;
  hipLaunchKernelGGL(normalize_kernel<true>, dim3(blocks), dim3(NORM_THREADS), 0, s, desc, norm_out, n, n_dev);
}

// B images' rows in one launch: image f's descriptors n rows behind image f - 1's, its count in n_dev[f]
void launch_normalize_batch(float* desc, float* norm_out, int n, int B, hipStream_t s, const int32_t* n_dev) {
  if (n <= 0 || B <= 0) return;
  const int blocks = (n + NORM_ROWS - 1) / NORM_ROWS;
  hipLaunchKernelGGL(normalize_kernel<true>, dim3(blocks, B), dim3(NORM_THREADS), 0, s, desc, norm_out, n, n_dev);
}

void launch_row_norms(const float* desc, float* norm_out, int n, hipStream_t s) {
  if (n <= 0) return;
  const int blocks = (n + NORM_ROWS - 1) / NORM_ROWS;
  hipLaunchKernelGGL(normalize_kernel<false>, dim3(blocks), dim3(NORM_THREADS), 0, s,
                     const_cast<float*>(desc), norm_out, n, (const int32_t*)nullptr);
}

// (the split count can follow a smaller expected query count: size for the largest)
static int expected_queries(int Q, int q_expected) {
  const int lo = std::max(QB, Q / 8);   // bounds the split count (and the scratch) at 8x the full-Q one
  if (q_expected <= 0 || q_expected > Q) return Q;
  return std::min(Q, std::max(q_expected, lo));
}
// the matrix-pipe variant (match_mfma.hip)
int mfma_splits_for(int Q, int N);
int mfma_max_splits(int N);
void launch_match_mfma(const float* qn, const float* qnorm, int Q, const float* db, const float* dnorm, int N,
                       Top2* scratch, int S, const int32_t* q_count, hipStream_t s);
// Which kernel searches: the matrix-pipe one when there are enough queries to fill its 256-query
// blocks (measured cross-over between 600 and 3000 queries), the VALU one below that.  Both give the
// same bits.  A context pins the choice with mh_match_set_mode 2 / 3 (kernel_pin 0 / 1); MH_MATCH_MFMA = 0 / 1 pins
// it for a process of an experiment build (A/B runs).
bool match_uses_mfma(int q_expected, int kernel_pin) {
  static const int pinned = exp_int("MH_MATCH_MFMA", -1);
  if (kernel_pin >= 0) return kernel_pin != 0;
  return pinned >= 0 ? pinned != 0 : q_expected >= 1536;
}

size_t match_scratch_elems(int Q, int N) {
  const int e = expected_queries(Q, 1);
  const int S = std::max(splits_for(e, N), mfma_max_splits(N));
  return (size_t)S * (size_t)(Q > 0 ? Q : 1);
}

size_t match_pack_floats(int Q) { return (size_t)((Q + TQ - 1) / TQ) * TQ * (DIM + 1); }

int launch_match(const float* qn, const float* qnorm, int Q, const float* db, const float* dnorm,
                 int N, const RowMap& rmap, Top2* scratch, float* pack, int32_t* idx1, float* d1,
                 float* d2, hipStream_t s, const int32_t* q_count, int q_expected, int kernel_pin) {
  if (Q <= 0) return -1;
  // the split count follows the number of queries expected (device-side counts: the caller's
  // estimate), the grid covers the capacity
  if (N > 0 && match_uses_mfma(expected_queries(Q, q_expected), kernel_pin)) {
    const int Sm = mfma_splits_for(expected_queries(Q, q_expected), N);
    launch_match_mfma(qn, qnorm, Q, db, dnorm, N, scratch, Sm, q_count, s);
    hipLaunchKernelGGL(combine_splits_kernel, dim3((Q + 63) / 64), dim3(64), 0, s, scratch, Sm, Q, q_count, idx1, d1,
                       d2, rmap);
    return 1;
  }
  const int S = (N > 0) ? splits_for(expected_queries(Q, q_expected), N) : 0;
  if (S > 0) {
    const int n_groups = (Q + TQ - 1) / TQ;
    float* P = pack;
    float* Pnorm = pack + (size_t)n_groups * TQ * DIM;
    const int pack_elems = n_groups * TQ * DIM;
    hipLaunchKernelGGL(pack_queries_kernel, dim3((pack_elems + 255) / 256), dim3(256), 0, s, qn, qnorm,
                       Q, q_count, P, Pnorm);
    const int qgroups = (Q + QB - 1) / QB;
    const int n_tiles = (N + TILE_ROWS - 1) / TILE_ROWS;
    const int tiles_per_split = (n_tiles + S - 1) / S;
    const size_t lds_bytes = 2 * TILE_FLOATS * sizeof(float);
    static DynLds attr;
    attr.ensure(match_kernel, lds_bytes);
    hipLaunchKernelGGL(match_kernel, dim3(qgroups * S), dim3(MATCH_THREADS), lds_bytes, s, P, Pnorm, Q,
                       db, dnorm, N, tiles_per_split, S, 0, scratch, q_count);
  }
  hipLaunchKernelGGL(combine_splits_kernel, dim3((Q + 63) / 64), dim3(64), 0, s, scratch, S, Q,
                     q_count, idx1, d1, d2, rmap);
  return S > 0 ? 0 : -1;
}

void launch_match_merge(const int32_t* idx1_s, const float* d1_s, const float* d2_s, int S, int Q,
                        size_t shard_stride, int32_t* idx1, float* d1, float* d2, hipStream_t s) {
  if (Q <= 0) return;
  hipLaunchKernelGGL(merge_shards_kernel, dim3((Q + 255) / 256), dim3(256), 0, s, idx1_s, d1_s, d2_s,
                     S, Q, shard_stride, idx1, d1, d2);
}

}  // namespace mh
