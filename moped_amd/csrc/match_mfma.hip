// MATCH on the matrix pipe: the same exact 2-NN search as match.hip, same canonical arithmetic,
// same results bit for bit -- v_mfma_f32_32x32x2_f32 accumulates C + a0*b0 + a1*b1 as two fused
// multiply-adds in k order (scripts/experiments/mfma_f32_chain.hip: 1024 / 1024 results identical
// to the fmaf chain), so dot(q, d) over K = 128 is the chain the oracle computes.
//
// Shape: a wavefront owns 32 queries and keeps them in registers as the A operands of all 64
// k-pairs (64 VGPRs: lane l holds q[l % 32][2 t + l / 32]).  The 128-row DB tile sits in LDS
// (double buffered, rows 129 floats apart: the B operand read -- lane l takes
// d[row block * 32 + l % 32][2 t + l / 32] -- touches 32 distinct banks per half wave).  Per tile a
// wavefront issues 4 row blocks x 64 k-pairs = 256 MFMAs into four 32x32 accumulators; the
// distances and the running top-2 fold (16 queries x 4 row blocks per lane, ascending rows) are
// VALU work that runs beside the MFMAs of the SIMD's other wavefront.  Eight wavefronts (256
// queries) share a tile; grid, split handling, XCD-aware block map and the per-split Top2 output
// are match.hip's.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace mh {

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int MW = 8;                    // wavefronts per workgroup
constexpr int MQ = 32 * MW;              // queries per workgroup (256)
constexpr int M_THREADS = 64 * MW;
constexpr int M_TILE = 128;              // DB rows per LDS tile
constexpr int M_STRIDE = DIM + 1;        // floats between rows in LDS (odd: conflict-free B reads)
constexpr int M_TILE_FLOATS = M_TILE * M_STRIDE;

__device__ __forceinline__ void fold2(float& b1, float& b2, int& i1, float v, int idx) {
  const bool lt = v < b1;
  b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
  i1 = lt ? idx : i1;
  b1 = lt ? v : b1;
}
struct Best2 {
  float b1, b2;
  int i1;
};
__device__ __forceinline__ void merge2(Best2& a, float ob1, float ob2, int oi1) {
  const bool take = (ob1 < a.b1) || (ob1 == a.b1 && (unsigned)oi1 < (unsigned)a.i1);
  const float lose1 = take ? a.b1 : ob1;
  const float s2 = fminf(a.b2, ob2);
  a.b2 = fminf(lose1, s2);
  a.b1 = take ? ob1 : a.b1;
  a.i1 = take ? oi1 : a.i1;
}

#ifdef MM_PROF   // phase timing build (make EXTRA=-DMM_PROF): cycles of thread 0 per phase, summed over workgroups
__device__ unsigned long long g_mm_prof[8];
#define MM_T(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); atomicAdd(&g_mm_prof[k], now_ - t_prof); t_prof = now_; } } while (0)
#else
#define MM_T(k) do { } while (0)
#endif

__global__ __launch_bounds__(M_THREADS) void match_mfma_kernel(
    const float* __restrict__ qn, const float* __restrict__ qnorm, int Q, const float* __restrict__ db,
    const float* __restrict__ dnorm, int N, int tiles_base, int tiles_rem, int n_splits, int32_t index_base,
    Top2* __restrict__ partial, const int32_t* __restrict__ q_count) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 tiles
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  // XCD-aware (query block, split) map, as in match_kernel
  const int nqb = gridDim.x / n_splits;
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
  const int Qe = q_count ? min(Q, *q_count) : Q;
  if (qblock * MQ >= Qe) return;   // uniform over the workgroup
#ifdef MM_PROF
  unsigned long long t_prof = clock64();
  if (threadIdx.x == 0) atomicAdd(&g_mm_prof[7], 1ull);
#endif
  const int q0 = qblock * MQ + wave * 32;
  const int n_tiles = (N + M_TILE - 1) / M_TILE;
  // split s takes tiles_base tiles, the first tiles_rem splits one more
  const int tile_begin = split * tiles_base + min(split, tiles_rem);
  const int tile_end = min(tile_begin + tiles_base + (split < tiles_rem ? 1 : 0), n_tiles);

  // the first tile's eight 16-byte loads go out before the A operands' 32: one trip to memory for both
  float4 f[8];
  if (tile_begin < tile_end) {
    const char* tb = reinterpret_cast<const char*>(db) + (size_t)tile_begin * (M_TILE * DIM * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const float4*>(tb + (size_t)threadIdx.x * 16 + (size_t)i * (M_THREADS * 16));
  }
  // ---- A operands: this lane's query row, every second coordinate starting at `half` ----
  float A[64];
  {
    const int q = q0 + l32;
    const bool live = q < Qe;
    const float4* row = reinterpret_cast<const float4*>(qn + (size_t)(live ? q : 0) * DIM);
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      float4 v = row[c];
      if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
      A[2 * c] = half ? v.y : v.x;
      A[2 * c + 1] = half ? v.w : v.z;
    }
  }
  MM_T(0);
  // accumulator register r of a 32x32 block belongs to query (r / 4) * 8 + half * 4 + r % 4;
  // the queries' norm terms wait in LDS (two broadcast reads per use instead of 16 registers)
  float* const nq_s = lds + 2 * M_TILE_FLOATS + wave * 32;
  if (lane < 32) nq_s[lane] = (q0 + lane < Qe) ? qnorm[q0 + lane] : 0.f;
  float b1[16], b2[16];
  int i1[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    b1[r] = __builtin_inff();
    b2[r] = __builtin_inff();
    i1[r] = -1;
  }

  // ---- staging: a tile is 64 KB contiguous (the DB is padded to whole tiles), copied in four quarters
  // of 32 rows so that only two 16-byte registers per thread are in flight: per quarter thread t takes
  // 2 x 16 B at (quarter base) + t * 16 + i * 8 KB = row (t >> 5) + 16 i, coordinates 4 (t & 31) .. + 3
  float4 st[2];
  const int st_row = tid >> 5, st_k = (tid & 31) * 4;
  auto stage_load = [&](int tile, int qt) {
    const char* tb = reinterpret_cast<const char*>(db) + (size_t)tile * (M_TILE * DIM * 4) + (size_t)qt * (32 * DIM * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) st[i] = *reinterpret_cast<const float4*>(tb + (size_t)tid * 16 + (size_t)i * (M_THREADS * 16));
  };
  auto stage_store = [&](int buf, int qt) {
    float* d = lds + buf * M_TILE_FLOATS + (qt * 32 + st_row) * M_STRIDE + st_k;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* p = d + i * 16 * M_STRIDE;
      p[0] = st[i].x;
      p[1] = st[i].y;
      p[2] = st[i].z;
      p[3] = st[i].w;
    }
  };
  if (tile_begin < tile_end) {   // the first tile (its loads left before the A operands')
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float* p = lds + (st_row + 16 * i) * M_STRIDE + st_k;
      p[0] = f[i].x;
      p[1] = f[i].y;
      p[2] = f[i].z;
      p[3] = f[i].w;
    }
  }
  __syncthreads();
  MM_T(1);

  // distances + fold of one pass (two row blocks), rows ascending within the lane
  auto fold_pass = [&](const v16f (&a)[2], const float (&dnv)[2], int row0) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int row = row0 + rb * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#ifdef MM_NOFOLD   // experiment: no distance / fold work (results are wrong)
        b1[r] += a[rb][r];
#else
        const float d = fmaxf(fmaf(-2.f, a[rb][r], nq_s[(r >> 2) * 8 + half * 4 + (r & 3)] + dnv[rb]), 0.f);
        fold2(b1[r], b2[r], i1[r], d, row);
#endif
      }
    }
  };
  // The two wavefronts of a SIMD would reach their folds together and leave the matrix pipe idle
  // meanwhile.  The later four wavefronts (4..7: second wavefront of each SIMD) therefore postpone the fold
  // of a tile's second pass to the start of the next tile: they fold while the others multiply and
  // vice versa.  Per lane the rows still arrive in ascending order.
  const bool late = wave >= MW / 2;
  v16f acc[2];
  float dn_kept[2] = {0.f, 0.f};
  int row_kept = 0;
  bool pending = false;

  int buf = 0;
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const bool more = tile + 1 < tile_end;
    const float* B = lds + buf * M_TILE_FLOATS + l32 * M_STRIDE + half;
    if (pending) {
      fold_pass(acc, dn_kept, row_kept);
      pending = false;
    }
    // two passes of 64 rows: two 32x32 accumulators live at a time; each pass in two segments of 32
    // k-pairs, a quarter of the next tile in flight during each
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float dn[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) dn[rb] = dnorm[tile * M_TILE + (2 * h + rb) * 32 + l32];   // +inf on padding rows
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
      // B operands one k-pair ahead of the MFMAs that use them; the scheduling fence keeps the
      // compiler from hoisting all the LDS reads of the pass to its top (and spilling)
      const float* Bh = B + (2 * h) * 32 * M_STRIDE;
      // B operands two k-pairs ahead of the MFMAs that use them
      float bc0 = Bh[0], bc1 = Bh[32 * M_STRIDE];
      float bd0 = Bh[2], bd1 = Bh[32 * M_STRIDE + 2];
#pragma unroll
      for (int seg = 0; seg < 2; ++seg) {
#ifndef MM_NOSTAGE   // experiment: MM_NOSTAGE keeps re-using the first tile (results are wrong)
        if (more) stage_load(tile + 1, 2 * h + seg);
#endif
#pragma unroll
        for (int tt = 0; tt < 32; ++tt) {
          const int t = seg * 32 + tt;
          float bn0 = 0.f, bn1 = 0.f;
          if (t + 2 < 64) {
            bn0 = Bh[2 * (t + 2)];
            bn1 = Bh[32 * M_STRIDE + 2 * (t + 2)];
          }
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[t], bc0, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[t], bc1, acc[1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          bc0 = bd0;
          bc1 = bd1;
          bd0 = bn0;
          bd1 = bn1;
        }
#ifndef MM_NOSTAGE
        if (more) stage_store(buf ^ 1, 2 * h + seg);
#endif
      }
      const int row0 = tile * M_TILE + (2 * h) * 32 + l32;
      if (h == 1 && late) {
        dn_kept[0] = dn[0];
        dn_kept[1] = dn[1];
        row_kept = row0;
        pending = true;
      } else {
        fold_pass(acc, dn, row0);
      }
    }
#ifndef MM_NOSTAGE
    __syncthreads();
    buf ^= 1;
#endif
  }
  if (pending) fold_pass(acc, dn_kept, row_kept);

  MM_T(2);
  // ---- the 32 lanes of a half hold different rows for the same 16 queries: min-reduce over the lanes (the
  // merge is commutative and associative: any tree gives the same top-2).  A halving butterfly: in every
  // step a lane keeps one half of its queries and hands the other half to a partner that keeps those, so
  // the merges number 8 + 4 + 2 + 1 (+ 1 across the two rows of 16) per lane instead of 16 x 5.  Partners:
  // lane ^ 1, lane ^ 2 (quad permutes), lane -/+ 4 and lane -/+ 8 inside the row of 16 (row rotations: the
  // sender differs from the receiver in exactly the bit that decides what each keeps), lane ^ 16 at the end.
  // A lane with bits b0..b3 ends with accumulator register r = 8 b0 + 4 b1 + 2 b2 + b3. ----
  Best2 v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = Best2{b1[r], b2[r], i1[r]};
#define MM_HALVE(CNT, BIT, CTRL)                                                                              \
  {                                                                                                          \
    const bool up = (l32 & (BIT)) != 0;                                                                      \
    _Pragma("unroll") for (int j = 0; j < (CNT) / 2; ++j) {                                                  \
      const Best2 send = up ? v[j] : v[j + (CNT) / 2];                                                       \
      Best2 keep = up ? v[j + (CNT) / 2] : v[j];                                                             \
      const float ob1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send.b1), CTRL, 0xF, 0xF, false)); \
      const float ob2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send.b2), CTRL, 0xF, 0xF, false)); \
      const int oi1 = __builtin_amdgcn_update_dpp(0, send.i1, CTRL, 0xF, 0xF, false);                        \
      if (oi1 >= 0) merge2(keep, ob1, ob2, oi1);                                                             \
      v[j] = keep;                                                                                           \
    }                                                                                                        \
  }
  MM_HALVE(16, 1, 0xB1)    // quad_perm [1,0,3,2]
  MM_HALVE(8, 2, 0x4E)     // quad_perm [2,3,0,1]
  MM_HALVE(4, 4, 0x124)    // row_ror:4
  MM_HALVE(2, 8, 0x128)    // row_ror:8
#undef MM_HALVE
  {
    Best2 s = v[0];
    const float ob1 = __shfl_xor(s.b1, 16);
    const float ob2 = __shfl_xor(s.b2, 16);
    const int oi1 = __shfl_xor(s.i1, 16);
    if (oi1 >= 0) merge2(s, ob1, ob2, oi1);
    const int r = ((l32 & 1) << 3) | ((l32 & 2) << 1) | ((l32 & 4) >> 1) | ((l32 & 8) >> 3);
    const int qi = q0 + (r >> 2) * 8 + half * 4 + (r & 3);
    if (l32 < 16 && qi < Qe) {
      Top2 o;
      o.d1 = s.b1;
      o.d2 = s.b2;
      o.i1 = (s.i1 >= 0) ? s.i1 + index_base : -1;
      o.pad = 0;
      partial[(size_t)split * Q + qi] = o;
    }
  }
  MM_T(3);
}

}  // namespace

#ifdef MM_PROF
extern "C" int mh_debug_mm_prof(unsigned long long out[8], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mm_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_mm_prof), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// How many splits of the DB: enough workgroups for three rounds over the chip's CUs (a workgroup owns a CU:
// two tiles of LDS, 8 x 223 VGPRs).  Fewer, longer workgroups would save pipeline fills (one round of 21 splits
// at config 1: measured 4% SLOWER, and 20% slower on an 8-way shard) - with several frames in flight the tail
// of one launch overlaps the next frame's kernels, and short workgroups let the latency-bound steps of the other
// streams in.  MH_MATCH_SPLITS pins the count, MH_MATCH_BLOCKS the workgroup target (sweeps).
constexpr int M_MAX_SPLITS = 128;
constexpr int M_TARGET_BLOCKS = 768;
int mfma_max_splits(int N) { return std::max(1, std::min((N + M_TILE - 1) / M_TILE, M_MAX_SPLITS)); }
int mfma_splits_for(int Q, int N) {
  const int qblocks = (Q + MQ - 1) / MQ;
  static const int pinned = exp_int("MH_MATCH_SPLITS", 0);
  static const int target = exp_int("MH_MATCH_BLOCKS", 0);
  const int s_max = mfma_max_splits(N);
  if (pinned > 0) return std::min(pinned, s_max);
  int S = ((target > 0 ? target : M_TARGET_BLOCKS) + qblocks - 1) / qblocks;
  if (S >= 8) S = (S + 3) / 8 * 8;   // whole splits per XCD
  return std::max(1, std::min(S, s_max));
}

// Same contract as the VALU path of launch_match (match.hip) from the normalised queries on.
void launch_match_mfma(const float* qn, const float* qnorm, int Q, const float* db, const float* dnorm, int N,
                       Top2* scratch, int S, const int32_t* q_count, hipStream_t s) {
  const int qblocks = (Q + MQ - 1) / MQ;
  const int n_tiles = (N + M_TILE - 1) / M_TILE;
  const size_t lds_bytes = (2 * M_TILE_FLOATS + MQ) * sizeof(float);   // two tiles + the queries' norm terms
  static DynLds attr;
  attr.ensure(match_mfma_kernel, lds_bytes);
  hipLaunchKernelGGL(match_mfma_kernel, dim3(qblocks * S), dim3(M_THREADS), lds_bytes, s, qn, qnorm, Q, db, dnorm, N,
                     n_tiles / S, n_tiles % S, S, 0 /* local rows: combine_splits_kernel maps them */, scratch, q_count);
}

}  // namespace mh
