// MATCH in two stages: an f16 matrix-pipe SCREEN of all (query, row) pairs, then the canonical f32
// arithmetic only on the few rows per query that can still be one of its two nearest.  Results --
// (idx1, d1, d2) per query -- are bit for bit those of match.hip / match_mfma.hip (and of the oracle):
// the screen only decides WHICH rows get the exact treatment, and it errs on the side of too many.
//
// Replaces the kd-tree search of MATCH_ANN_CPU::process (moped2/libmoped/src/match/MATCH_ANN_CPU.hpp:
// 155-165) like the exact kernels do; what changes is the cost: 2*128*Q*N flops on
// v_mfma_f32_32x32x16_f16 (16x the f32 rate) instead of on v_mfma_f32_32x32x2_f32.
//
// Why it is exact.  Canonical distance (match.hip): a(q,r) = max(0, fmaf(-2, p, qq + dd_r)) with p the
// f32 fmaf chain of dot(q, d_r).  Ranking by a is ranking by w(q,r) = p - dd_r / 2 (a = qq - 2w up to
// f32 rounding).  The screen computes w~ = sum_k f16(q_k) f16(d_rk) - dd_r / 2 with f32 accumulation
// (the -dd_r/2 term is the accumulator's initial value, exact).  With u = 2^-11 (f16 unit roundoff)
//     |w~ - w| <= E(q) = (2u + u^2) |q| Dmax            rounding of both operands (Cauchy-Schwarz)
//                      + 2^-25 sqrt(128) (|q| + Dmax)    f16 subnormal range (absolute 2^-25 per element)
//                      + 3e-5 (|q| Dmax + Dmax^2 / 2)    136 f32 accumulations on either side + the chain's own rounding
// where Dmax = max_r |d_r| (taken at upload).  Let T be any value known to be <= the second largest w~
// over distinct rows (pass A: the second largest over a SAMPLE of the rows).  A row r that is one of the
// two nearest by a satisfies w_r >= (second largest w) - slack, hence w~_r >= T - 2E - slack.  Pass B emits
// every row above tau = T - (2E + slack); pass C evaluates the canonical distance of the emitted rows and
// folds the exact top-2 (ties -> lower row, second best = second smallest value), which is the global
// answer because every row it did not see is strictly farther than the two it reports.  A query whose
// candidate list overflows, or whose descriptor is not finite / too large for f16, is searched by brute
// force inside pass C: the screen can be slow, never wrong.  The bound is exercised by
// tests/test_gpu_screen.py (near-ties inside the bound, duplicates, zero rows, unnormalised rows) and
// its arithmetic by tests/test_screen_bound_cpu.py.
//
// Shape (gfx950).  A workgroup = 4 wavefronts, 2 workgroups per CU.  A wavefront keeps 64 queries in
// registers as the B operands of all 8 k-steps (2 blocks x 8 x 4 VGPRs); the f16 DB streams through
// LDS in 128-row tiles (32 KB, double buffered, filled by global_load_lds_dwordx4 with the 16-byte
// chunks of row r stored at chunk position c ^ (r & 15), so the A-operand ds_read_b128 of 16 lanes
// touch all 64 banks once).  Queries sit on the accumulator's lane axis, rows on its register axis:
// a lane's 16 values of a 32x32 block are 16 rows of ONE query, so the running maxima are lane-local
// v_max3_f32 (8 per block) and only a lane with a hit looks at individual values.
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "screen.h"

MH_TRACE_TU()

namespace mh {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define MH_AS1 __attribute__((address_space(1)))
#define MH_AS3 __attribute__((address_space(3)))

// Wavefronts per workgroup (template parameter NW of the passes): 8 = one workgroup per CU, two of ITS wavefronts per
// SIMD; 4 = one wavefront per SIMD, so that TWO workgroups share a CU -- the two wavefronts of a SIMD then belong to
// different workgroups (no common barrier, phases drift apart: one's LDS waits and hit paths sit under the other's MFMAs),
// and a CU on which a small kernel of another frame holds a few wavefronts still takes one such workgroup (the 8-wavefront
// shape needs the whole register file of the CU: scripts/cu_trace.py found 21% of all CU time spent with only small
// workgroups resident and 16-20% with none).  Price: every tile is staged from L2 by twice as many workgroups.
constexpr int SC_QPAD = 3 * 1024;                  // the query image is padded to whole workgroups of every kernel width
constexpr int SC_TILE = 128;                       // DB rows per LDS tile (= the DB's row padding)
constexpr int SC_TILE_BYTES = SC_TILE * DIM * 2;   // 32 KB of f16
// hits a wavefront parks in LDS (80 bytes each: the block's 16 values go along) before they go to memory: 112 with eight
// wavefronts (the whole workgroup's hits, written out at its end), 40 with four (two workgroups' LDS must fit a CU:
// 2 x (64 KB tiles + 1.5 KB + 4 x 40 x 80 B) = 156 KB) -- written out whenever the buffer is three quarters full
constexpr bool SC_SHAPE16_LARGE = true;   // the four-query-block launches run on v_mfma_f32_16x16x32_f16 (screen16_kernel)
constexpr int SC_NW_LARGE = 8;   // wavefronts per workgroup of the four-query-block launches (4 or 8; MH_SCREEN_NW in experiment builds)
__host__ __device__ constexpr int sc_recbuf(int nw) { return nw >= 8 ? 112 : 40; }
constexpr int SC_GROUP = 1;                        // tiles per barrier: the wavefronts of a workgroup drift apart within a group
constexpr int SC_NBUF = 2 * SC_GROUP;              // (a hit costs its wavefront ~300 cycles), every barrier makes seven wait for the slowest
constexpr int SC_DD = 192;                         // floats per tile of the -dd/2 array: 128 rows, the 4 row blocks' maxima, their minima, padding
constexpr int SC_LDS_TILES = SC_NBUF * SC_TILE_BYTES + SC_NBUF * SC_DD * 4;   // the tiles + their -dd/2 terms
constexpr int SC_REC_BYTES = 80;                   // {slot, row0, thr, top, 16 dots}
__host__ __device__ constexpr int sc_lds_bytes(int nw) { return SC_LDS_TILES + nw * sc_recbuf(nw) * SC_REC_BYTES; }   // + the wavefronts' hit buffers
static_assert(SC_TILE == 128 && DIM == 128, "tile image and chunk swizzle assume 128 x 128");

// error model of the screen (see the header comment)
__host__ __device__ inline float screen_margin(float qq, float dmax) {
  const float nq = sqrtf(fmaxf(qq, 0.f));
  const float E = 0.000978f * nq * dmax + 4e-7f * (nq + dmax) + 3e-5f * (nq * dmax + 0.5f * dmax * dmax);
  return 2.f * E + 2e-6f * (qq + dmax * dmax);
}

// ---- f16 images ------------------------------------------------------------------------------
// DB: [n_pad][128] f16 (round to nearest), dneg[n_pad] = -dd/2, + statistics over the real rows: max dd, max |x|, any non-finite dd.
__global__ __launch_bounds__(256) void db_to_half_kernel(const float* __restrict__ db, const float* __restrict__ dnorm, int N,
                                                         size_t n_chunks, _Float16* __restrict__ dbh, float* __restrict__ dneg,
                                                         unsigned int* __restrict__ stats) {
  __shared__ unsigned int red[3];
  if (threadIdx.x < 3) red[threadIdx.x] = 0;
  __syncthreads();
  float x_max = 0.f, dd_max = 0.f;
  unsigned int bad = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_chunks; i += (size_t)gridDim.x * blockDim.x) {   // one 8-element chunk
    const float4 a = reinterpret_cast<const float4*>(db)[2 * i], b = reinterpret_cast<const float4*>(db)[2 * i + 1];
    half8 h;
    h[0] = (_Float16)a.x; h[1] = (_Float16)a.y; h[2] = (_Float16)a.z; h[3] = (_Float16)a.w;
    h[4] = (_Float16)b.x; h[5] = (_Float16)b.y; h[6] = (_Float16)b.z; h[7] = (_Float16)b.w;
    reinterpret_cast<half8*>(dbh)[i] = h;
    const int row = (int)(i >> 4);
    if ((i & 15) == 0) dneg[(size_t)(row >> 7) * SC_DD + (row & 127)] = -0.5f * dnorm[row];   // the rows' -dd/2; -inf on padding rows
    if (row < N) {
      // (fmaxf drops NaNs: a NaN coordinate shows in the row's norm term)
      x_max = fmaxf(x_max, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                                 fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
      if ((i & 15) == 0) {
        const float dd = dnorm[row];
        if (!(dd >= 0.f) || dd == __builtin_inff()) bad = 1;
        else dd_max = fmaxf(dd_max, dd);
      }
    }
  }
  // non-negative floats order like their bit patterns
  atomicMax(&red[0], __float_as_uint(dd_max));
  atomicMax(&red[1], __float_as_uint(x_max));
  if (bad) atomicOr(&red[2], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(&stats[0], red[0]);
    atomicMax(&stats[1], red[1]);
    if (red[2]) atomicOr(&stats[2], 1u);
  }
}

// Per 32-row block the largest and the smallest of its rows' -dd/2 (entries 128..131 and 132..135 of the tile's part of
// the array): dot + largest >= every row's screen value >= dot + smallest.  Pass B works on the dots alone and
// decides with the block's largest term (a superset of the rows above the threshold: for the L2-normalised rows of a
// MOPED database the two differ in the last bit); pass C reads both to turn a record's value into bounds.
__global__ void db_block_bounds_kernel(const float* __restrict__ dnorm, int N, int n_tiles, float* __restrict__ dneg,
                                       unsigned int* __restrict__ stats) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;   // 32-row block
  if (b >= n_tiles * 4) return;
  float hi = -__builtin_inff(), lo = __builtin_inff();
  for (int r = 0; r < 32; ++r) {
    const int row = b * 32 + r;
    const float v = row < N ? -0.5f * dnorm[row] : -__builtin_inff();
    if (row < N) hi = fmaxf(hi, v);
    lo = fminf(lo, v);
  }
  float* t = dneg + (size_t)(b >> 2) * SC_DD;
  t[128 + (b & 3)] = hi;
  t[132 + (b & 3)] = lo;
  // the largest spread of -dd/2 inside a block of real rows (stats[3]; non-negative floats order like their bits):
  // what a record's value can overstate its block's largest screen value by
  if (b * 32 + 32 <= N && hi - lo >= 0.f) atomicMax(&stats[3], __float_as_uint(hi - lo));
  if ((b & 3) == 0)
    for (int i = 136; i < SC_DD; ++i) t[i] = 0.f;
}

// The exact answer for an all-zero query: canonical distance max(0, fmaf(-2, 0, 0 + dd_r)) = dd_r, so the two smallest
// norm terms over the real rows, the lower row on a tie (what the exact kernels' top-2 gives).  One workgroup.
__global__ __launch_bounds__(1024) void db_zero_query_kernel(const float* __restrict__ dnorm, int N, unsigned int* __restrict__ stats) {
  __shared__ float s1[1024], s2[1024];
  __shared__ int si[1024];
  float b1 = __builtin_inff(), b2 = __builtin_inff();
  int i1 = -1;
  for (int r = threadIdx.x; r < N; r += 1024) {   // ascending rows: a strict < keeps the lower row on ties
    const float d = fmaxf(dnorm[r], 0.f);
    if (d < b1 || i1 < 0) {
      if (i1 >= 0) b2 = fminf(b2, b1);
      b1 = d;
      i1 = r;
    } else {
      b2 = fminf(b2, d);
    }
  }
  s1[threadIdx.x] = b1;
  s2[threadIdx.x] = b2;
  si[threadIdx.x] = i1;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      const float o1 = s1[threadIdx.x + st], o2 = s2[threadIdx.x + st];
      const int oi = si[threadIdx.x + st];
      float a1 = s1[threadIdx.x], a2 = s2[threadIdx.x];
      int ai = si[threadIdx.x];
      if (oi >= 0) {
        const bool take = ai < 0 || o1 < a1 || (o1 == a1 && oi < ai);
        const float lose = take ? a1 : o1;
        a2 = fminf(fminf(a2, o2), ai >= 0 ? lose : __builtin_inff());
        if (take) {
          a1 = o1;
          ai = oi;
        }
      }
      s1[threadIdx.x] = a1;
      s2[threadIdx.x] = a2;
      si[threadIdx.x] = ai;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    stats[4] = (unsigned int)si[0];
    stats[5] = __float_as_uint(s1[0]);
    stats[6] = __float_as_uint(s2[0]);
  }
}

// Queries of the frame: f16 rows (zero rows up to the padded count) + a flag for queries the screen
// cannot vouch for (non-finite norm term, or a coordinate outside f16's range).
__global__ void screen_prepare_kernel(const float* __restrict__ qn, const float* __restrict__ qnorm, int Q,
                                      const int32_t* __restrict__ q_count, int q_pad, _Float16* __restrict__ qh,
                                      uint8_t* __restrict__ qbad) {
  MH_TRACE_SCOPE(mh::TK_PREPARE);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // one 8-element chunk; 16 consecutive threads = one row
  if (i >= q_pad * 16) return;
  const int row = i >> 4;
  const int Qe = q_count ? min(Q, *q_count) : Q;
  half8 h = {0, 0, 0, 0, 0, 0, 0, 0};
  float m = 0.f;
  if (row < Qe) {
    const float4 a = reinterpret_cast<const float4*>(qn)[2 * (size_t)i], b = reinterpret_cast<const float4*>(qn)[2 * (size_t)i + 1];
    h[0] = (_Float16)a.x; h[1] = (_Float16)a.y; h[2] = (_Float16)a.z; h[3] = (_Float16)a.w;
    h[4] = (_Float16)b.x; h[5] = (_Float16)b.y; h[6] = (_Float16)b.z; h[7] = (_Float16)b.w;
    m = fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
              fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
  }
  reinterpret_cast<half8*>(qh)[i] = h;
  // row maximum over its 16 threads (fmaxf drops NaNs: a NaN coordinate shows in the norm term instead)
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((i & 15) == 0) {
    int bad = 0;
    if (row < Qe) {
      const float qq = qnorm[row];
      bad = (!(qq >= 0.f) || qq == __builtin_inff() || !(m < 60000.f)) ? 1 : 0;
      // norm term exactly -1 = "no such query" (the rows past an image's keypoint count in a batch of images,
      // mh_frame_enqueue_image_batch): no threshold, no records, no neighbour -- not a query the screen cannot vouch for
      if (qq == -1.f) bad = 2;
      else if (!bad && m == 0.f && qq == 0.f) bad = 3;   // an all-zero query: its answer is known per DB (ScreenDb::zero_*)
    }
    qbad[row] = (uint8_t)bad;
  }
}

// ---- passes A and B ------------------------------------------------------------------------------
// MODE 0 (pass A): top-2 VALUES of w~ per (split, query) over the sampled tiles -> part[split][q].
// MODE 1 (pass B): every row with w~ > tau(q) -> the query's candidate records {row0, bits}: rows
//                  row0 + (r & 3) + 8 (r >> 2) for the set bits r.  A lane owns the sub-list (query, split, half)
//                  -- `sub_cap` slots nobody else writes, no atomics; a sub-list that is full spills into the
//                  query's overflow list (atomic append, rare).  bits = 0 marks an empty slot.
constexpr int SC_SLOTS_MAX = 256;   // record slots per query: 2 halves x splits x sub_cap <= this

#ifdef SC_PROF   // experiment build (make EXTRA=-DSC_PROF ...): switch parts of passes A/B off (results are wrong then)
// and stamp a workgroup's time; scripts/screen_prof.py
__device__ unsigned long long g_sc_prof[8];
__device__ unsigned long long g_sc_trace[2][1024];   // (event << 56 | cycles since the workgroup's start) of wavefronts 0 and 4 of workgroup 3
static int g_sc_ablate = 0;
#define SC_ABL(bit) ((A.ablate >> (bit)) & 1)
#define SC_EV(e) do { if (tracing && n_ev < 1024) { g_sc_trace[wave >> 2][n_ev++] = ((unsigned long long)(e) << 56) | (__builtin_amdgcn_s_memtime() - t_start); } } while (0)
#else
#define SC_ABL(bit) 0
#define SC_EV(e) do { } while (0)
#endif

struct ScreenArgs {
  const _Float16* qh;
  const _Float16* dbh;
  const float* dneg;     // [padded rows] -dd/2, -inf on padding rows
  const float* qnorm;
  const uint8_t* qbad;
  const int32_t* q_count;
  float2* part;          // pass A's output, [n_splits_a][q_pad]
  const float* tau;      // pass B's input, [q_pad] (screen_tau_kernel)
  uint2* recs;           // [q_pad][SC_SLOTS_MAX], empty (bits 0) on entry of pass B (pass C re-empties)
  int32_t* ovf_cnt;      // [q_pad], zero on entry
  uint2* ovf;            // [q_pad][ovf_cap]
  int Q, q_pad, ovf_cap, sub_cap;
  int n_sel, tile_first, tile_stride;   // the pass covers tiles tile_first + j * tile_stride, j < n_sel
  int tiles_base, tiles_rem, n_splits;  // split s takes tiles_base selected tiles, the first tiles_rem one more
  int n_splits_a;
  float dmax;
  int ablate;            // SC_PROF builds only: 1 = no finish(), 2 = stage only the first tile, 4 = no MFMAs, 8 = no end-of-tile barrier
};

// Between the passes: a query's threshold from pass A's per-split top-2 values.  tau = +inf for queries that do not
// exist in this frame or that the screen cannot vouch for (pass B then leaves them no records; pass C searches the
// latter by brute force).  One thread per query; the splits' values of neighbouring queries are neighbours in memory.
__global__ void screen_tau_kernel(const float2* __restrict__ part, int n_splits_a, int q_pad, int Q,
                                  const int32_t* __restrict__ q_count, const float* __restrict__ qnorm,
                                  const uint8_t* __restrict__ qbad, float dmax, float* __restrict__ tau) {
  MH_TRACE_SCOPE(mh::TK_TAU);
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= q_pad) return;
  const int Qe = q_count ? min(Q, *q_count) : Q;
  float t = __builtin_inff();
  if (q < Qe && !qbad[q]) {
    float B = -__builtin_inff(), S = -__builtin_inff();
    for (int s0 = 0; s0 < n_splits_a; s0 += 8) {
      float2 p[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        p[j] = (s0 + j < n_splits_a) ? part[(size_t)(s0 + j) * q_pad + q] : make_float2(-__builtin_inff(), -__builtin_inff());
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        S = fmaxf(fminf(B, p[j].x), fmaxf(S, p[j].y));
        B = fmaxf(B, p[j].x);
      }
    }
    t = S - screen_margin(qnorm[q], dmax);
  }
  tau[q] = t;
}

// Four LDS-DMA pieces (1 KB each: 16 bytes per lane to lds_i + lane * 16) in one statement: the source of piece i
// is base_i + voff (the same per-lane offset for all four), M0 is saved once.  Issued as asm so that hipcc neither
// counts the transfers nor waits for them before the next ds_read (it would: vmcnt(0) at the first LDS read after a
// __builtin_amdgcn_global_load_lds).  Completion: the s_waitcnt vmcnt(0) + barrier that end every tile.
__device__ __forceinline__ void dma16x4(unsigned voff, const void* b0, const void* b1, const void* b2, const void* b3,
                                        unsigned l0, unsigned l1, unsigned l2, unsigned l3) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
      "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
      "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(l0), "s"(l1), "s"(l2), "s"(l3)
      : "memory");
}
__device__ __forceinline__ void dma4(unsigned voff, const void* base, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

template <int MODE, int NQB, int NW>
__global__ __launch_bounds__(64 * NW, 2) void screen_kernel(const ScreenArgs A) {
  MH_TRACE_SCOPE(MODE == 0 ? mh::TK_PASS_A : mh::TK_PASS_B);
  constexpr int SC_WAVES = NW;
  constexpr int SC_RECBUF = sc_recbuf(NW);
  constexpr int QW = 32 * NQB;            // queries per wavefront
  constexpr int QB = QW * SC_WAVES;       // queries per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const unsigned lds_base = (unsigned)(uintptr_t)(MH_AS3 unsigned char*)lds;
  // XCD-aware (query block, split) map as in match.hip: workgroups are dealt round-robin to the 8 XCDs, so
  // giving XCD x the splits x, x + 8, ... keeps a split's rows in one L2
  const int nqb = gridDim.x / A.n_splits;
  int qblock, split;
  {
    // workgroup L runs on XCD L % 8: XCD x takes the x-th eighth of the (split-major) unit list, so the rows of a
    // split -- read by all nqb query blocks -- pass through one or two of the eight L2s and not through all of them,
    // whatever the number of splits
    const int L = blockIdx.x, U = gridDim.x;
    const int x = L & 7, j = L >> 3;
    const int unit = x * (U >> 3) + min(x, U & 7) + j;
    split = unit / nqb;
    qblock = unit - split * nqb;
  }
  const int Qe = A.q_count ? min(A.Q, *A.q_count) : A.Q;
  if (qblock * QB >= Qe) return;   // uniform over the workgroup
  const int q0 = qblock * QB + wave * QW;
  const int sel_begin = split * A.tiles_base + min(split, A.tiles_rem);
  const int sel_end = min(sel_begin + A.tiles_base + (split < A.tiles_rem ? 1 : 0), A.n_sel);

  // ---- staging: one tile = 32 pieces of 1 KB (64 lanes x 16 B, lane-linear in LDS); wavefront w brings pieces
  // w, w + NW, w + 2 NW, ... (four of them with eight wavefronts, eight with four).  Chunk g = 64 piece + lane of the LDS
  // image is row g >> 4, position g & 15, and holds source chunk (g & 15) ^ (row & 15); row & 15 = (4 w + (lane >> 4)) & 15
  // for all pieces of a wavefront (they lie 4 NW rows apart: a multiple of 16), so one per-lane offset serves them and the
  // pieces differ by NW KB in a scalar base.  The rows' -dd/2 terms (512 B per tile) and the row blocks' extrema (768 B
  // per tile) come the same way: wavefronts 0..2, 4 bytes per lane.
  const unsigned voff = (unsigned)(wave * 1024 + (lane >> 4) * 256 + (((lane & 15) ^ ((wave * 4 + (lane >> 4)) & 15)) << 4));
  const unsigned voff_dd = (unsigned)((wave * 64 + lane) * 4);
  auto stage = [&](int sel, int buf) {
    const int tile = A.tile_first + sel * A.tile_stride;
    const unsigned char* tb = reinterpret_cast<const unsigned char*>(A.dbh) + (size_t)tile * SC_TILE_BYTES;
    const unsigned l = lds_base + buf * SC_TILE_BYTES + wave * 1024;
    constexpr int STEP = NW * 1024;   // bytes between a wavefront's pieces (the same in the source tile and in LDS)
#pragma unroll
    for (int g = 0; g < 32 / NW; g += 4)
      dma16x4(voff, tb + g * STEP, tb + (g + 1) * STEP, tb + (g + 2) * STEP, tb + (g + 3) * STEP, l + g * STEP, l + (g + 1) * STEP,
              l + (g + 2) * STEP, l + (g + 3) * STEP);
    if (wave < 3)
      dma4(voff_dd, A.dneg + (size_t)tile * SC_DD, lds_base + SC_NBUF * SC_TILE_BYTES + buf * (SC_DD * 4) + wave * 256);
  };

#ifdef SC_PROF
  const unsigned long long t_start = __builtin_amdgcn_s_memtime(), r_start = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_wait = 0;
#endif
#pragma unroll
  for (int j = 0; j < SC_GROUP; ++j)
    if (sel_begin + j < sel_end) stage(sel_begin + j, j);
  // ---- B operands: this lane's query of each block, k = 16 s + 8 half .. + 7 of every k-step s ----
  half8 bq[NQB][8];
#pragma unroll
  for (int nb = 0; nb < NQB; ++nb) {
    const _Float16* row = A.qh + (size_t)(q0 + nb * 32 + l32) * DIM + 8 * half;
#pragma unroll
    for (int s = 0; s < 8; ++s) bq[nb][s] = *reinterpret_cast<const half8*>(row + 16 * s);
  }
  // ---- per-query state ----
  float b1[NQB], b2[NQB], tau[NQB];
  int n_rec[NQB];
#pragma unroll
  for (int nb = 0; nb < NQB; ++nb) {
    b1[nb] = -__builtin_inff();
    b2[nb] = -__builtin_inff();
    tau[nb] = __builtin_inff();
    n_rec[nb] = 0;
    if (MODE == 1) tau[nb] = A.tau[q0 + nb * 32 + l32];
  }
  // vmcnt(0) as a builtin, not asm: hipcc must KNOW that the B operands (and everything else) have arrived, or it
  // keeps counted vmcnt waits for them inside the tile loop -- where they would wait for the LDS-DMA of the next tile
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  // ---- what a lane does with one finished 32 x 32 block: 16 rows of one of its queries ----
  // pass A: running top-2 values
  // (the two largest BLOCK MAXIMA, not the two largest values: 10 instructions per block instead of 32.  Two blocks
  // are different rows, so the second largest block maximum is still some second row's value -- a lower bound of the
  // query's second best, weaker than the exact one only when a lane's two best rows sit in the same 16-row group)
  auto fold_a = [&](const v16f& acc, int nb) {
    float m = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, acc[r]), acc[r + 1]);
    m = fmaxf(m, acc[15]);
    b2[nb] = __builtin_amdgcn_fmed3f(b1[nb], b2[nb], m);
    b1[nb] = fmaxf(b1[nb], m);
  };
  // pass B, rare part: this lane has rows above its query's threshold -> one record.  The record does not go to
  // memory now: the s_waitcnt vmcnt(0) that ends every tile (it is there for the LDS-DMA) would wait for the store's
  // round trip too, and the tile's barrier would hand that wait to all eight wavefronts.  It is parked in the
  // wavefront's LDS buffer {destination slot, row0, bits} and written out when the workgroup is done.
  uint4* const recbuf = reinterpret_cast<uint4*>(lds + SC_LDS_TILES + wave * (SC_RECBUF * SC_REC_BYTES));
  int n_parked = 0;   // records parked so far: wave-uniform (advanced by the hit lanes' count outside the divergent part)
  // `acc` = the block's 16 DOT PRODUCTS (pass B's accumulators start at zero), `top` their maximum, `thr` = tau - the
  // row block's largest -dd/2: dot > thr holds for every row whose screen value dot - dd/2 exceeds tau, and for hardly
  // any other (a superset is all pass C needs).
  // A record from a block's 16 dots: which of them exceed thr (bit r = sign(thr - acc[r]), shifted in from r = 15
  // down), and the largest dot + the block's largest -dd/2 -- an upper bound of the block's largest screen value, at
  // most the block's spread of -dd/2 above it -- as an f16 of its distance above tau (a small positive number, so the
  // f16 costs ~1e-5 and not 5e-4): pass C ranks the records by it and runs the exact arithmetic only on those that can
  // still hold one of the two nearest rows.
  auto record_bits = [](const v16f& acc, float top, float thr) {
    unsigned bits = 0;
#pragma unroll
    for (int r = 15; r >= 0; --r) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(thr - acc[r]), 31);
    return (bits & 0xFFFFu) | ((unsigned)screen_record_value(top, thr) << 16);
  };
  // The hit path proper is as short as it can be -- a wavefront meets four or five hit blocks per tile and every
  // cycle it spends here its MFMAs do not issue (a quarter of pass B before this): the lane takes a slot of its
  // sub-list and parks the block's raw values in the wavefront's LDS buffer; the record is made from them when the
  // workgroup is done, all lanes in parallel.  Only when the buffer or the sub-list is full is it made on the spot.
  auto emit = [&](const v16f& acc, int nb, int row0, float top, float thr, int pos) {
    const int q = q0 + nb * 32 + l32;
    if (n_rec[nb] < A.sub_cap) {
      const unsigned dst = (unsigned)q * SC_SLOTS_MAX + (2 * split + half) * A.sub_cap + n_rec[nb];
      ++n_rec[nb];
      if (pos < SC_RECBUF) {
        uint4* e = recbuf + 5 * pos;
        e[0] = make_uint4(dst, (unsigned)row0, __float_as_uint(thr), __float_as_uint(top));
#pragma unroll
        for (int g = 0; g < 4; ++g)
          e[1 + g] = make_uint4(__float_as_uint(acc[4 * g]), __float_as_uint(acc[4 * g + 1]), __float_as_uint(acc[4 * g + 2]),
                                __float_as_uint(acc[4 * g + 3]));
      } else {
        A.recs[dst] = make_uint2((unsigned)row0, record_bits(acc, top, thr));   // buffer full
      }
    } else {
      // the lane's sub-list is full: the query's overflow list (rare)
      const int opos = atomicAdd(&A.ovf_cnt[q], 1);
      if (opos < A.ovf_cap) A.ovf[(size_t)q * A.ovf_cap + opos] = make_uint2((unsigned)row0, record_bits(acc, top, thr));
      if (pos < SC_RECBUF) recbuf[5 * pos] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);   // its place in the buffer stays empty
    }
  };
  // the hit lanes of a block get consecutive places in the wavefront's buffer: ballot + prefix count, no LDS atomic
  // (its round trip was a quarter of the hit path)
  auto emit_hits = [&](bool hit, const v16f& acc, int nb, int row0, float top, float thr) {
    const unsigned long long hm = __ballot(hit);
    if (hm == 0ull) return;
    if (hit)
      emit(acc, nb, row0, top, thr,
           n_parked + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u)));
    n_parked += __popcll(hm);
  };

  // the parked records to their slots (all lanes in parallel); the buffer is empty afterwards
  auto flush_parked = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n = min(n_parked, SC_RECBUF);
    for (int j = lane; j < n; j += 64) {
      const uint4 e = recbuf[5 * j];
      if (e.x == 0xFFFFFFFFu) continue;
      v16f v;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint4 x = recbuf[5 * j + 1 + g];
        v[4 * g] = __uint_as_float(x.x);
        v[4 * g + 1] = __uint_as_float(x.y);
        v[4 * g + 2] = __uint_as_float(x.z);
        v[4 * g + 3] = __uint_as_float(x.w);
      }
      A.recs[e.x] = make_uint2(e.y, record_bits(v, __uint_as_float(e.w), __uint_as_float(e.z)));
    }
    __builtin_amdgcn_wave_barrier();   // (everybody has read its entries before the next hit overwrites them)
    n_parked = 0;
  };

  const int swz = l32 & 15;
  // A operands of a row block: row rb * 32 + l32, k-step s -> chunk 2 s + half, stored at position chunk ^ (row & 15);
  // accumulator register r belongs to row (r & 3) + 8 (r >> 2) + 4 half of the block: its -dd/2 goes in as C
  auto load_rb = [&](const unsigned char* T, const float* ddp, int rb, half8 (&a)[8], v16f& init) {
    const unsigned char* rowp = T + (rb * 32 + l32) * 256;
#pragma unroll
    for (int s = 0; s < 8; ++s) a[s] = *reinterpret_cast<const half8*>(rowp + (((2 * s + half) ^ swz) << 4));
    if (MODE == 1) return;   // pass B's accumulators start at zero
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(ddp + rb * 32 + 8 * g + 4 * half);
      init[4 * g] = v.x;
      init[4 * g + 1] = v.y;
      init[4 * g + 2] = v.z;
      init[4 * g + 3] = v.w;
    }
  };

#ifdef SC_PROF
  const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
  int side = 0;         // which half of the LDS buffers holds the current group of tiles
  v16f pend;            // pass B: the block whose MFMAs were issued last; looked at behind the next block's MFMAs
  float pend_hi = 0.f;  // ... and the largest -dd/2 of its row block
  int pend_row0 = 0;    // (its query block is (nb + NQB - 1) % NQB when block nb is issued: a constant after unrolling --
  bool have_pend = false;   // a run-time index would put tau[] and n_rec[] into scratch memory)
  for (int grp = sel_begin; grp < sel_end; grp += SC_GROUP) {
    const bool more = grp + SC_GROUP < sel_end;
    if (!SC_ABL(1)) {
#pragma unroll
      for (int j = 0; j < SC_GROUP; ++j)
        if (grp + SC_GROUP + j < sel_end) stage(grp + SC_GROUP + j, (side ^ 1) * SC_GROUP + j);
    }
#pragma unroll 1
    for (int j = 0; j < SC_GROUP; ++j) {
    const int sel = grp + j;
    if (sel >= sel_end) break;
    const int buf = SC_ABL(1) ? 0 : side * SC_GROUP + j;
    const unsigned char* T = lds + buf * SC_TILE_BYTES;
    const float* ddp = reinterpret_cast<const float*>(lds + SC_NBUF * SC_TILE_BYTES + buf * (SC_DD * 4));
    const int row_tile = (A.tile_first + sel * A.tile_stride) * SC_TILE;
    if (MODE == 0) {
#pragma unroll 1   // (pass A unrolled: 123 spilled registers at three query blocks)
      for (int rb = 0; rb < SC_TILE / 32; ++rb) {
        half8 a[8];
        v16f init;
        load_rb(T, ddp, rb, a, init);
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) {
          v16f acc = init;
#pragma unroll
          for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], bq[nb][s], acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);   // (left alone, the scheduler overlaps several blocks' accumulators and spills)
          fold_a(acc, nb);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      // Software pipeline: while the MFMAs of block b are issued, (1) the A operands of the NEXT row block arrive from
      // LDS and (2) the maximum over the 16 values of block b - 1 is taken, two values per MFMA -- vector instructions
      // in the shadow of the matrix pipe.  Only a lane whose maximum beats its threshold looks at single values.
      // (three and four query blocks per wavefront: no room for the prefetched copy of the next row block's operands;
      // the other wavefront of the SIMD covers the LDS latency at the head of a row block)
      constexpr bool PREF = NQB <= 2;
      half8 a_cur[8], a_nxt[PREF ? 8 : 1];
      v16f unused;
      const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (PREF) load_rb(T, ddp, 0, a_cur, unused);
      const float4 hi4 = *reinterpret_cast<const float4*>(ddp + 128);   // the four row blocks' largest -dd/2 (the same for all lanes)
      const float hi[4] = {hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
      for (int rb = 0; rb < SC_TILE / 32; ++rb) {
        if constexpr (PREF) {
          if (rb + 1 < SC_TILE / 32) load_rb(T, ddp, rb + 1, a_nxt, unused);
        } else {
          load_rb(T, ddp, rb, a_cur, unused);
        }
        const int row0 = row_tile + rb * 32 + 4 * half;
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) {
          const int pnb = (nb + NQB - 1) % NQB;
          v16f acc = zero;
          float m = -__builtin_inff();
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            if (!SC_ABL(2)) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[s], bq[nb][s], acc, 0, 0, 0);
            // (nothing behind the first four MFMAs: the pending block's last values are still on their way out of the
            // matrix pipe, and a vector instruction that reads them stalls the wavefront -- and the MFMAs behind it)
            if (have_pend && s >= 4) {
              if (s == 4) m = fmaxf(fmaxf(fmaxf(fmaxf(pend[0], pend[1]), pend[2]), pend[3]), pend[4]);
              else if (s < 7) m = fmaxf(fmaxf(fmaxf(fmaxf(m, pend[4 * s - 15]), pend[4 * s - 14]), pend[4 * s - 13]), pend[4 * s - 12]);
              else m = fmaxf(fmaxf(fmaxf(m, pend[13]), pend[14]), pend[15]);
            }
          }
          // pin the interleave: four MFMAs, then one MFMA and the two vector instructions of its shadow, four times
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
          for (int s = 4; s < 8; ++s) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (have_pend && !SC_ABL(0)) emit_hits(m > tau[pnb] - pend_hi, pend, pnb, pend_row0, m, tau[pnb] - pend_hi);
          __builtin_amdgcn_sched_barrier(0);
          pend = acc;
          pend_row0 = row0;
          pend_hi = hi[rb];
          have_pend = true;
        }
        if constexpr (PREF) {
          if (rb + 1 < SC_TILE / 32) {
#pragma unroll
            for (int s = 0; s < 8; ++s) a_cur[s] = a_nxt[s];
          }
        }
      }
    }
    }   // tiles of the group
#ifdef SC_PROF
    const unsigned long long t_w0 = __builtin_amdgcn_s_memtime();
#endif
    // a small buffer (four-wavefront workgroups) is written out when three quarters full -- behind the tile's wait for the
    // next tile's pieces, so that the stores' round trip is not in it
    const bool flush_now = MODE == 1 && NW < 8 && n_parked > SC_RECBUF * 3 / 4;
    if (!SC_ABL(3)) {   // (experiment builds: 8 = no end-of-tile wait and barrier -- what the synchronisation costs)
    if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's pieces of the next group have landed
    __syncthreads();   // everybody's pieces have; nobody reads this group's buffers any more
    }
    if (flush_now) flush_parked();
#ifdef SC_PROF
    t_wait += __builtin_amdgcn_s_memtime() - t_w0;
#endif
    side ^= 1;
  }
  if (MODE == 1 && have_pend) {
    float m = pend[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, pend[r]);
    emit_hits(m > tau[NQB - 1] - pend_hi, pend, NQB - 1, pend_row0, m, tau[NQB - 1] - pend_hi);
  }
  if (MODE == 1) flush_parked();
#ifdef SC_PROF
  if (MODE == 1 && tid == 0) {
    atomicAdd(&g_sc_prof[0], __builtin_amdgcn_s_memtime() - t_start);       // shader cycles of the workgroup
    atomicAdd(&g_sc_prof[1], __builtin_amdgcn_s_memrealtime() - r_start);   // 100 MHz ticks
    atomicAdd(&g_sc_prof[2], t_wait);                                       // cycles in the end-of-tile wait + barrier
    atomicAdd(&g_sc_prof[3], (unsigned long long)(sel_end - sel_begin));
    atomicAdd(&g_sc_prof[4], 1ull);
    atomicAdd(&g_sc_prof[5], t_loop - t_start);                             // prologue
    if (SC_ABL(0)) A.part[0].x = b1[0] + pend[3];
  }
#endif

  if (MODE == 0) {
    // the two halves of the wavefront hold different rows of the same queries
#pragma unroll
    for (int nb = 0; nb < NQB; ++nb) {
      const float ob = __shfl_xor(b1[nb], 32), os = __shfl_xor(b2[nb], 32);
      const float S = fmaxf(fminf(b1[nb], ob), fmaxf(b2[nb], os));
      const float B = fmaxf(b1[nb], ob);
      const int q = q0 + nb * 32 + l32;
      if (half == 0 && q < A.q_pad) A.part[(size_t)split * A.q_pad + q] = make_float2(B, S);
    }
  }
}

// ---- passes A and B on v_mfma_f32_16x16x32_f16 ------------------------------------------------------------------
// The same passes with the other f16 MFMA shape.  On random operands out of registers, two wavefronts per SIMD, every
// CU busy, the 16x16x32 instruction sustains 1 842 TFLOP/s on this part and the 32x32x16 one 1 625 (the power budget's
// clock, not the issue rate: scripts/experiments/mfma_shapes_rate.hip, profiles/r03_mfma_shapes_rate.txt) -- pass B on
// 32x32x16 ran at 83% of what that instruction can deliver at all.
// A 32-query x 32-row block is 2 x 2 tiles of 16 x 16 and four k-steps of 32: 16 MFMAs of 16 cycles (the other shape: 8
// of 32).  Operands (cdna_hip_programming.md 3): lane l holds A[row l & 15][k = 8 (l >> 4) + j] and B[k][col l & 15];
// result register r of a tile: col = l & 15, row = 4 (l >> 4) + r.  Rows (A) = DB rows, columns (B) = queries, so a
// lane holds, of a block, TWO queries (l & 15 of either 16-query tile) x EIGHT rows (16 ti + 4 (l >> 4) + r): the four
// "quarters" l >> 4 of a wavefront see different rows of the same queries.  What follows from that:
//   - per-query state (thresholds, running maxima, record counts) is [query block][2] per lane;
//   - a record covers the 8 rows a lane holds of one query in a block: row0 = block + 4 quarter, bit 4 ti + r = row
//     row0 + r + 16 ti; bit 31 of the record's row word marks the layout for pass C (the 32x32x16 records: bit r = row
//     row0 + (r & 3) + 8 (r >> 2), 16 rows, all 16 mask bits in use);
//   - a query's record slots are shared out over 4 x splits lane-private sub-lists (quarters) instead of 2 x splits;
//   - pass A folds the lane's 8 rows of a block into ONE maximum per query (two blocks are still different rows).
// Staging, swizzle (chunk ^ (row & 15): a 16-lane group reads 16 rows at one chunk index -- all 64 banks once), the
// XCD-aware unit map, the parked records and the exactness argument are those of screen_kernel.
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr unsigned SC_REC_LAYOUT16 = 0x80000000u;   // in a record's row word: the rows of the 16x16x32 passes
constexpr int SC_REC16_BYTES = 48;                  // {slot, row0, thr, top, 8 dots}
constexpr int SC_RECBUF16 = 8 * 1024 / SC_REC16_BYTES;   // 170 parked hits per wavefront in 8 KB
constexpr int SC_LDS16_TAU = SC_LDS_TILES + 8 * SC_RECBUF16 * SC_REC16_BYTES;   // the wavefronts' thresholds: 8 x 128 floats
constexpr int SC_LDS16_BYTES = SC_LDS16_TAU + 8 * 128 * 4;

template <int MODE, int NQB>
__global__ __launch_bounds__(512, 2) void screen16_kernel(const ScreenArgs A) {
  MH_TRACE_SCOPE(MODE == 0 ? mh::TK_PASS_A : mh::TK_PASS_B);
  constexpr int NW = 8;
  constexpr int QW = 32 * NQB;            // queries per wavefront
  constexpr int QB = QW * NW;             // queries per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int quarter = lane >> 4, l16 = lane & 15;
  const unsigned lds_base = (unsigned)(uintptr_t)(MH_AS3 unsigned char*)lds;
  const int nqb = gridDim.x / A.n_splits;
  int qblock, split;
  {
    const int L = blockIdx.x, U = gridDim.x;
    const int x = L & 7, j = L >> 3;
    const int unit = x * (U >> 3) + min(x, U & 7) + j;
    split = unit / nqb;
    qblock = unit - split * nqb;
  }
  const int Qe = A.q_count ? min(A.Q, *A.q_count) : A.Q;
  if (qblock * QB >= Qe) return;   // uniform over the workgroup
  const int q0 = qblock * QB + wave * QW;
  const bool live = q0 < Qe;   // uniform over the wavefront
  const int sel_begin = split * A.tiles_base + min(split, A.tiles_rem);
  const int sel_end = min(sel_begin + A.tiles_base + (split < A.tiles_rem ? 1 : 0), A.n_sel);

  const unsigned voff = (unsigned)(wave * 1024 + (lane >> 4) * 256 + (((lane & 15) ^ ((wave * 4 + (lane >> 4)) & 15)) << 4));
  const unsigned voff_dd = (unsigned)((wave * 64 + lane) * 4);
  auto stage = [&](int sel, int buf) {
    const int tile = A.tile_first + sel * A.tile_stride;
    const unsigned char* tb = reinterpret_cast<const unsigned char*>(A.dbh) + (size_t)tile * SC_TILE_BYTES;
    const unsigned l = lds_base + buf * SC_TILE_BYTES + wave * 1024;
    dma16x4(voff, tb, tb + 8192, tb + 16384, tb + 24576, l, l + 8192, l + 16384, l + 24576);
    if (wave < 3)
      dma4(voff_dd, A.dneg + (size_t)tile * SC_DD, lds_base + SC_NBUF * SC_TILE_BYTES + buf * (SC_DD * 4) + wave * 256);
  };
  if (sel_begin < sel_end) stage(sel_begin, 0);
  // ---- B operands: the lane's query of either 16-query tile of each block, k = 32 s + 8 quarter .. + 7 ----
  half8 bq[NQB][2][4];
#pragma unroll
  for (int nb = 0; nb < NQB; ++nb)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const _Float16* row = A.qh + (size_t)(q0 + nb * 32 + 16 * tj + l16) * DIM + 8 * quarter;
#pragma unroll
      for (int s = 0; s < 4; ++s) bq[nb][tj][s] = *reinterpret_cast<const half8*>(row + 32 * s);
    }
  // per-query state of the lane's 2 NQB queries.  Pass B is at the register limit (128 for the queries' operands, 32 + 16
  // + 16 for a row block's fragments, the accumulators and the pending block): the thresholds wait in LDS (two ds_read_b32
  // per block, in the MFMAs' shadow) and the sub-lists' record counts are bytes of NQB / 2 registers.
  float b1[NQB][2], b2[NQB][2];
  unsigned n_rec[(NQB + 1) / 2] = {};   // [nb >> 1]: byte 2 (nb & 1) + tj
  float* const tau_lds = reinterpret_cast<float*>(lds + SC_LDS16_TAU) + wave * 128;
#pragma unroll
  for (int nb = 0; nb < NQB; ++nb)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      b1[nb][tj] = -__builtin_inff();
      b2[nb][tj] = -__builtin_inff();
    }
  if (MODE == 1)
    for (int i = lane; i < QW; i += 64) tau_lds[i] = A.tau[q0 + i];
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  uint4* const recbuf = reinterpret_cast<uint4*>(lds + SC_LDS_TILES + wave * (SC_RECBUF16 * SC_REC16_BYTES));
  int n_parked = 0;
  // a record from the lane's 8 dots of one query: bit 4 ti + r = sign(thr - dot)
  auto record_bits8 = [](const v4f& d0, const v4f& d1, float top, float thr) {
    unsigned bits = 0;
#pragma unroll
    for (int r = 3; r >= 0; --r) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(thr - d1[r]), 31);
#pragma unroll
    for (int r = 3; r >= 0; --r) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(thr - d0[r]), 31);
    return (bits & 0xFFu) | ((unsigned)screen_record_value(top, thr) << 16);
  };
  auto emit = [&](const v4f& d0, const v4f& d1, unsigned& nrec_word, int shift, int q, int row0, float top, float thr, int pos) {
    const int nrec = (int)((nrec_word >> shift) & 0xFFu);
    if (nrec < A.sub_cap) {   // (sub_cap <= SC_SLOTS_MAX / 4 = 64 < 256, static_assert at SB16_MAX: the byte cannot wrap)
      const unsigned dst = (unsigned)q * SC_SLOTS_MAX + (4 * split + quarter) * A.sub_cap + nrec;
      nrec_word += 1u << shift;
      if (pos < SC_RECBUF16) {
        uint4* e = recbuf + 3 * pos;
        e[0] = make_uint4(dst, (unsigned)row0, __float_as_uint(thr), __float_as_uint(top));
        e[1] = make_uint4(__float_as_uint(d0[0]), __float_as_uint(d0[1]), __float_as_uint(d0[2]), __float_as_uint(d0[3]));
        e[2] = make_uint4(__float_as_uint(d1[0]), __float_as_uint(d1[1]), __float_as_uint(d1[2]), __float_as_uint(d1[3]));
      } else {
        A.recs[dst] = make_uint2((unsigned)row0 | SC_REC_LAYOUT16, record_bits8(d0, d1, top, thr));   // buffer full
      }
    } else {
      const int opos = atomicAdd(&A.ovf_cnt[q], 1);
      if (opos < A.ovf_cap) A.ovf[(size_t)q * A.ovf_cap + opos] = make_uint2((unsigned)row0 | SC_REC_LAYOUT16, record_bits8(d0, d1, top, thr));
      if (pos < SC_RECBUF16) recbuf[3 * pos] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
    }
  };
  // the pending block: acc[ti][tj]; query tj of the lane = tiles (0, tj) and (1, tj)
  auto emit_block = [&](bool hit0, bool hit1, const v4f (&p)[2][2], int nb, int row0, float m0, float m1, float thr0, float thr1) {
    const unsigned long long h0 = __ballot(hit0), h1 = __ballot(hit1);
    if ((h0 | h1) == 0ull) return;
    if (hit0)
      emit(p[0][0], p[1][0], n_rec[nb >> 1], 16 * (nb & 1), q0 + nb * 32 + l16, row0, m0, thr0,
           n_parked + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(h0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)h0, 0u)));
    n_parked += __popcll(h0);
    if (hit1)
      emit(p[0][1], p[1][1], n_rec[nb >> 1], 16 * (nb & 1) + 8, q0 + nb * 32 + 16 + l16, row0, m1, thr1,
           n_parked + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(h1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)h1, 0u)));
    n_parked += __popcll(h1);
  };
  auto flush_parked = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n = min(n_parked, SC_RECBUF16);
    for (int j = lane; j < n; j += 64) {
      const uint4 e = recbuf[3 * j];
      if (e.x == 0xFFFFFFFFu) continue;
      const uint4 x0 = recbuf[3 * j + 1], x1 = recbuf[3 * j + 2];
      const v4f d0 = {__uint_as_float(x0.x), __uint_as_float(x0.y), __uint_as_float(x0.z), __uint_as_float(x0.w)};
      const v4f d1 = {__uint_as_float(x1.x), __uint_as_float(x1.y), __uint_as_float(x1.z), __uint_as_float(x1.w)};
      A.recs[e.x] = make_uint2(e.y | SC_REC_LAYOUT16, record_bits8(d0, d1, __uint_as_float(e.w), __uint_as_float(e.z)));
    }
    __builtin_amdgcn_wave_barrier();
    n_parked = 0;
  };

  // A operands of a row block: tile ti = rows rb 32 + 16 ti + l16, k-step s -> chunk 4 s + quarter at position
  // chunk ^ (row & 15) = chunk ^ l16
  auto load_rb = [&](const unsigned char* T, int rb, half8 (&a)[2][4]) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const unsigned char* rowp = T + (rb * 32 + 16 * ti + l16) * 256;
#pragma unroll
      for (int s = 0; s < 4; ++s) a[ti][s] = *reinterpret_cast<const half8*>(rowp + (((4 * s + quarter) ^ l16) << 4));
    }
  };
  // (as instructions: fmaxf() quiets every operand first -- a v_max_f32 x, x per value, 28 vector instructions per block
  // where 8 do; the dots of finite operands are never NaN)
  auto max3 = [](float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
  };
  auto max8 = [&](const v4f& x, const v4f& y) {
    return max3(max3(max3(x[0], x[1], x[2]), x[3], y[0]), max3(y[1], y[2], y[3]), -__builtin_inff());
  };

  int side = 0;
  // the pending block starts as one no threshold lets through (its dots -inf): the loop needs no "is there one yet"
  const v4f ninf4 = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  v4f pend[2][2] = {{ninf4, ninf4}, {ninf4, ninf4}};
  float pend_hi = 0.f;
  int pend_row0 = 0;
  const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
  {
  MH_TRACE_PHASE(trace_loop_, MODE == 1 ? mh::TK_PASS_B_LOOP : mh::TK_PASS_A_LOOP);
  for (int sel = sel_begin; sel < sel_end; ++sel) {
    const bool more = sel + 1 < sel_end;
    if (more) stage(sel + 1, side ^ 1);
    const unsigned char* T = lds + side * SC_TILE_BYTES;
    const float* ddp = reinterpret_cast<const float*>(lds + SC_NBUF * SC_TILE_BYTES + side * (SC_DD * 4));
    const int row_tile = (A.tile_first + sel * A.tile_stride) * SC_TILE;
    // (a wavefront none of whose 128 queries exists -- the tail of the last query block -- only helps with the staging
    // and the barriers: the SIMD's matrix pipe is then its neighbour's alone)
    if (!live) {
    } else if (MODE == 0) {
#pragma unroll 1
      for (int rb = 0; rb < SC_TILE / 32; ++rb) {
        half8 a[2][4];
        load_rb(T, rb, a);
        v4f init[2];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          const float4 v = *reinterpret_cast<const float4*>(ddp + rb * 32 + 16 * ti + 4 * quarter);
          init[ti] = v4f{v.x, v.y, v.z, v.w};
        }
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) {
          v4f acc[2][2] = {{init[0], init[0]}, {init[1], init[1]}};
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
              for (int tj = 0; tj < 2; ++tj)
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ti][s], bq[nb][tj][s], acc[ti][tj], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int tj = 0; tj < 2; ++tj) {
            const float m = max8(acc[0][tj], acc[1][tj]);
            b2[nb][tj] = __builtin_amdgcn_fmed3f(b1[nb][tj], b2[nb][tj], m);
            b1[nb][tj] = fmaxf(b1[nb][tj], m);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      const float4 hi4 = *reinterpret_cast<const float4*>(ddp + 128);
      const float hi[4] = {hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
      for (int rb = 0; rb < SC_TILE / 32; ++rb) {
        half8 a[2][4];
        load_rb(T, rb, a);
        const int row0 = row_tile + rb * 32 + 4 * quarter;
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) {
          const int pnb = (nb + NQB - 1) % NQB;
          v4f acc[2][2] = {{zero4, zero4}, {zero4, zero4}};
          float m0 = -__builtin_inff(), m1 = -__builtin_inff();
          const float t0 = tau_lds[pnb * 32 + l16], t1 = tau_lds[pnb * 32 + 16 + l16];   // the pending block's thresholds
#pragma unroll
          for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
              for (int tj = 0; tj < 2; ++tj)
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ti][s], bq[nb][tj][s], acc[ti][tj], 0, 0, 0);
            // the pending block's maxima in the shadow of the second half of the MFMAs (its values have left the pipe by then)
            if (s == 2) m0 = max8(pend[0][0], pend[1][0]);
            if (s == 3) m1 = max8(pend[0][1], pend[1][1]);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          {
            const float thr0 = t0 - pend_hi, thr1 = t1 - pend_hi;
            emit_block(m0 > thr0, m1 > thr1, pend, pnb, pend_row0, m0, m1, thr0, thr1);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) pend[ti][tj] = acc[ti][tj];
          pend_row0 = row0;
          pend_hi = hi[rb];
        }
      }
    }
    if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    side ^= 1;
  }
  }
  if (MODE == 1) {
    {
      const float m0 = max8(pend[0][0], pend[1][0]), m1 = max8(pend[0][1], pend[1][1]);
      const float thr0 = tau_lds[(NQB - 1) * 32 + l16] - pend_hi, thr1 = tau_lds[(NQB - 1) * 32 + 16 + l16] - pend_hi;
      emit_block(m0 > thr0, m1 > thr1, pend, NQB - 1, pend_row0, m0, m1, thr0, thr1);
    }
    flush_parked();
  }
  if (MODE == 0) {
    // the four quarters of the wavefront hold different rows of the same queries
#pragma unroll
    for (int nb = 0; nb < NQB; ++nb)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        float B = b1[nb][tj], S = b2[nb][tj];
#pragma unroll
        for (int d = 16; d <= 32; d <<= 1) {
          const float ob = __shfl_xor(B, d), os = __shfl_xor(S, d);
          S = fmaxf(fminf(B, ob), fmaxf(S, os));
          B = fmaxf(B, ob);
        }
        const int q = q0 + nb * 32 + 16 * tj + l16;
        if (quarter == 0 && q < A.q_pad) A.part[(size_t)split * A.q_pad + q] = make_float2(B, S);
      }
  }
}

// ---- pass C: canonical arithmetic on the candidates ------------------------------------------------
struct Best {
  float b1, b2;
  int i1;
};
// one more row into a running top-2; rows arrive in any order: lower row wins a tie on the best distance,
// b2 = second smallest value.  A first row at distance +inf leaves "none" (i1 = -1) as the exact kernels do.
__device__ __forceinline__ void take(Best& a, float d, int row) {
  const bool better = d < a.b1 || (d == a.b1 && a.i1 >= 0 && row < a.i1);
  const float lose = better ? a.b1 : d;
  a.b2 = fminf(a.b2, lose);
  a.b1 = better ? d : a.b1;
  a.i1 = better ? row : a.i1;
}
__device__ __forceinline__ void merge(Best& a, float ob1, float ob2, int oi1) {   // match.hip's merge
  const bool take_o = (ob1 < a.b1) || (ob1 == a.b1 && (unsigned)oi1 < (unsigned)a.i1);
  const float lose1 = take_o ? a.b1 : ob1;
  const float s2 = fminf(a.b2, ob2);
  a.b2 = fminf(lose1, s2);
  a.b1 = take_o ? ob1 : a.b1;
  a.i1 = take_o ? oi1 : a.i1;
}

// One wavefront per query; every lane runs the canonical chain of one candidate row at a time:
// dot = fmaf chain over k = 0..127 from 0, dist = max(0, fmaf(-2, dot, qq + dd)).  The query's coordinates are
// wave-uniform (LDS broadcast reads), the row comes straight from HBM / the Infinity Cache, 16 bytes per load.
constexpr int RS_WAVES = 4;          // queries per workgroup
constexpr int RS_MAXC = 1024;        // candidate rows a query may have before it is searched by brute force
constexpr int RS_STAGE = 8;          // candidate rows staged through LDS (8 x 512 bytes = the candidate list's 4 KB)
static_assert(RS_STAGE * DIM * 4 <= RS_MAXC * 4 && RS_STAGE % 2 == 0 && RS_STAGE <= 32, "staged rows live in the candidate list's LDS");

__device__ __forceinline__ float exact_dist(const float* __restrict__ q_lds, const float* __restrict__ row, float nq, float dn) {
  // 64 bytes x 2 in flight per step: 48 VGPRs, ten wavefronts per SIMD.  (Round 3: the whole 512-byte row in flight before
  // the chain -- one round trip per row instead of four -- takes 160 VGPRs, three wavefronts per SIMD, and pass C went
  // from 0.047 to 0.071 ms per 24 000 queries: it lives on the number of queries in flight, not on one query's latency.)
  float s = 0.f;
  const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll 2
  for (int c = 0; c < DIM / 16; ++c) {
    const float4 x0 = r4[4 * c], x1 = r4[4 * c + 1], x2 = r4[4 * c + 2], x3 = r4[4 * c + 3];
    const float4 y0 = *reinterpret_cast<const float4*>(q_lds + 16 * c), y1 = *reinterpret_cast<const float4*>(q_lds + 16 * c + 4);
    const float4 y2 = *reinterpret_cast<const float4*>(q_lds + 16 * c + 8), y3 = *reinterpret_cast<const float4*>(q_lds + 16 * c + 12);
    s = fmaf(y0.x, x0.x, s); s = fmaf(y0.y, x0.y, s); s = fmaf(y0.z, x0.z, s); s = fmaf(y0.w, x0.w, s);
    s = fmaf(y1.x, x1.x, s); s = fmaf(y1.y, x1.y, s); s = fmaf(y1.z, x1.z, s); s = fmaf(y1.w, x1.w, s);
    s = fmaf(y2.x, x2.x, s); s = fmaf(y2.y, x2.y, s); s = fmaf(y2.z, x2.z, s); s = fmaf(y2.w, x2.w, s);
    s = fmaf(y3.x, x3.x, s); s = fmaf(y3.y, x3.y, s); s = fmaf(y3.z, x3.z, s); s = fmaf(y3.w, x3.w, s);
  }
  return fmaxf(fmaf(-2.f, s, nq + dn), 0.f);
}

__device__ __forceinline__ void wave_lds_sync() {   // LDS written by some lanes of this wavefront, read by others
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64 * RS_WAVES) void rescore_kernel(
    const float* __restrict__ qn, const float* __restrict__ qnorm, const uint8_t* __restrict__ qbad, int Q,
    const int32_t* __restrict__ q_count, const float* __restrict__ db, const float* __restrict__ dnorm, int N,
    RowMap rmap, uint2* __restrict__ recs, int n_slots, int32_t* __restrict__ ovf_cnt, uint2* __restrict__ ovf,
    int ovf_cap, float dmax, const float* __restrict__ tau, float spread, int32_t* __restrict__ idx1, float* __restrict__ d1, float* __restrict__ d2,
    unsigned int* __restrict__ stats, int32_t zero_idx, float zero_d1, float zero_d2) {
  MH_TRACE_SCOPE(mh::TK_PASS_C);
  __shared__ __attribute__((aligned(16))) float q_s[RS_WAVES][DIM];
  __shared__ __attribute__((aligned(16))) int cand_s[RS_WAVES][RS_MAXC];   // the candidate list, then RS_STAGE rows of them
  __shared__ int ncand_s[RS_WAVES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = blockIdx.x * RS_WAVES + wave;
  if (q >= Q) return;
  // Everything the query's wavefront reads before it knows its candidates is asked for HERE, in one go: the kernel lives
  // on round trips (a query is ~a dozen loads and ~400 fmaf), and loads issued behind the branches below each cost one.
  constexpr int RS_ITERS = (SC_SLOTS_MAX + SCREEN_OVF_CAP + 63) / 64;
  uint2* mine = recs + (size_t)q * SC_SLOTS_MAX;
  uint2 recv[RS_ITERS];
#pragma unroll
  for (int it = 0; it < RS_ITERS; ++it) {
    const int j = it * 64 + lane;
    recv[it] = j < n_slots ? mine[j] : make_uint2(0u, 0u);
  }
  const int Qe = q_count ? min(Q, *q_count) : Q;
  const int bad = qbad[q];
  const float nq = qnorm[q];
  const float tau_q = tau[q];
  const int n_ovf = ovf_cnt[q];
  const float2 qv = reinterpret_cast<const float2*>(qn + (size_t)q * DIM)[lane];
  // (the values are wanted HERE: without this the compiler moves the loads behind the tests below, one round trip each)
  asm volatile("" ::"v"(bad), "v"(nq), "v"(tau_q), "v"(n_ovf), "v"(qv.x), "v"(qv.y), "s"(Qe));
  if (q >= Qe) {   // no such query in this frame: "no neighbour", like combine_splits_kernel (pass B left it no records)
    if (lane == 0) {
      idx1[q] = -1;
      d1[q] = __builtin_inff();
      d2[q] = __builtin_inff();
    }
    return;
  }
  if (bad == 2) {   // an absent row inside the launch (see screen_prepare_kernel)
    if (lane == 0) {
      idx1[q] = -1;
      d1[q] = __builtin_inff();
      d2[q] = __builtin_inff();
    }
    return;
  }
  if (bad == 3) {   // an all-zero query: every row's distance is its norm term
    if (lane == 0) {
      idx1[q] = zero_idx >= 0 ? row_to_global(rmap, zero_idx) : -1;
      d1[q] = zero_d1;
      d2[q] = zero_d2;
      if (stats) stats[3 * q + 2] += 1u;   // a search, without candidates
    }
    return;
  }
  if (lane == 0) ncand_s[wave] = 0;
  q_s[wave][2 * lane] = qv.x;
  q_s[wave][2 * lane + 1] = qv.y;
  // ---- records -> candidate row list in LDS; the slots read are emptied for the next frame ----
  bool brute = n_ovf > ovf_cap || bad == 1;
  int n_cand = 0;
  if (!brute) {
    // Every record carries the largest screen value of its rows (f16, nearest).  Two different records hold different
    // rows, so the second largest of these values (minus the f16 rounding) is a lower bound of the DB's second largest
    // screen value s2, and a row that belongs to the exact top-2 has a screen value >= s2 - 2 E (pass B's argument
    // with the exact s2 in the place of pass A's sampled one): records whose largest value (plus the rounding) lies
    // more than screen_margin below that bound cannot hold one.  Pass A's threshold comes from an eighth of the rows
    // and lets ~25 rows per query through; ~3 survive this one and get their 512-byte row fetched.
    float l1 = -__builtin_inff(), l2 = -__builtin_inff();   // the lane's two largest lower bounds
    auto bounds = [tau_q, spread, N, dmax](uint2 rec, float& lo, float& hi) {
      // the record's value: (largest dot of the block + the block's largest -dd/2) - tau, rounded to f16.  The block's
      // largest screen value is at most that, and at least that less the spread of -dd/2 inside a block (`spread`: the
      // largest over the DB's blocks of real rows, ~1e-7 for L2-normalised rows); a block with padding rows (their
      // dot is 0, their -dd/2 is -inf) gives no lower bound.  (screen.h; checked in host arithmetic by the CPU tests)
      screen_record_bounds((unsigned short)(rec.y >> 16), rec.x & 0x7FFFFFFFu, tau_q, spread, N, dmax, lo, hi);
    };
#pragma unroll
    for (int it = 0; it < RS_ITERS; ++it) {
      const int j = it * 64 + lane;
      uint2 rec = recv[it];
      if (j < n_slots) {
        if (rec.y) mine[j] = make_uint2(0u, 0u);
      } else if (j < n_slots + n_ovf) {
        rec = ovf[(size_t)q * ovf_cap + (j - n_slots)];
      }
      recv[it] = rec;
      if (rec.y & 0xFFFFu) {
        float lo, hi;
        bounds(rec, lo, hi);
        l2 = fmaxf(l2, fminf(l1, lo));
        l1 = fmaxf(l1, lo);
      }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float o1 = __shfl_xor(l1, d), o2 = __shfl_xor(l2, d);
      l2 = fmaxf(fmaxf(l2, o2), fminf(l1, o1));
      l1 = fmaxf(l1, o1);
    }
    const float keep_from = l2 - screen_margin(nq, dmax);   // -inf with fewer than two records: everything stays
#pragma unroll
    for (int it = 0; it < RS_ITERS; ++it) {
      const uint2 rec = recv[it];
      unsigned bits = rec.y & 0xFFFFu;
      if (bits) {
        float lo, hi;
        bounds(rec, lo, hi);
        if (hi < keep_from) bits = 0;
      }
      // which rows a bit stands for: records of the 32x32x16 passes hold 16 rows, row0 + (r & 3) + 8 (r >> 2); those of
      // the 16x16x32 passes (bit 31 of the row word) 8 rows, row0 + (r & 3) + 16 (r >> 2)
      const int hi_stride = (rec.x & 0x80000000u) ? 16 : 8;
      const int row0 = (int)(rec.x & 0x7FFFFFFFu);
      // candidates in any order (the exact top-2 below breaks ties by row number): an LDS counter hands out places
      while (bits) {
        const int r = __builtin_ctz(bits);
        bits &= bits - 1;
        const int w = atomicAdd(&ncand_s[wave], 1);
        if (w < RS_MAXC) cand_s[wave][w] = row0 + (r & 3) + hi_stride * (r >> 2);
      }
    }
    wave_lds_sync();
    n_cand = ncand_s[wave];
    if (n_cand > RS_MAXC) brute = true;
  } else if (n_ovf > ovf_cap) {
    for (int j = lane; j < n_slots; j += 64) mine[j] = make_uint2(0u, 0u);   // the lists overflowed: empty every slot
  }
  if (n_ovf && lane == 0) ovf_cnt[q] = 0;
  wave_lds_sync();

  Best best = {__builtin_inff(), __builtin_inff(), -1};
  const int n_rows = brute ? N : n_cand;
  if (!brute && n_cand <= 64) {
    // The usual query: a handful of candidates.  The first RS_STAGE rows come in as ONE round trip -- 32 lanes x 16 bytes
    // per row, all of them in flight at once -- into the LDS the candidate list lay in (chunk c of staged row r at
    // c ^ r: the lanes that run the chains read different banks), and lane r runs row r's chain from there: the same
    // fmaf chain in the same order as exact_dist, its 512 bytes read from LDS instead of in four dependent round trips.
    const int my_row = lane < n_cand ? cand_s[wave][lane] : -1;
    const bool my_ok = my_row >= 0 && my_row < N;
    const float my_dn = my_ok ? dnorm[my_row] : 0.f;
    wave_lds_sync();   // every lane holds its candidate: the list's memory is free
    float4* stage = reinterpret_cast<float4*>(cand_s[wave]);
    const int n_st = min(n_cand, RS_STAGE);
    const int half = lane >> 5, c = lane & 31;
    float4 v[RS_STAGE / 2];
#pragma unroll
    for (int i = 0; i < RS_STAGE / 2; ++i) {
      const int r = 2 * i + half;
      const int row_r = __shfl(my_row, r);
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < n_st && row_r >= 0 && row_r < N) v[i] = reinterpret_cast<const float4*>(db + (size_t)row_r * DIM)[c];
    }
#pragma unroll
    for (int i = 0; i < RS_STAGE / 2; ++i) {
      const int r = 2 * i + half;
      if (r < n_st) stage[r * 32 + (c ^ r)] = v[i];
    }
    wave_lds_sync();
    if (lane < n_st) {
      if (my_ok) {
        float sdot = 0.f;
        const float4* qs4 = reinterpret_cast<const float4*>(q_s[wave]);
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
          const float4 x = stage[lane * 32 + (k ^ lane)], y = qs4[k];
          sdot = fmaf(y.x, x.x, sdot); sdot = fmaf(y.y, x.y, sdot); sdot = fmaf(y.z, x.z, sdot); sdot = fmaf(y.w, x.w, sdot);
        }
        take(best, fmaxf(fmaf(-2.f, sdot, nq + my_dn), 0.f), my_row);
      }
    } else if (my_ok) {
      take(best, exact_dist(q_s[wave], db + (size_t)my_row * DIM, nq, my_dn), my_row);
    }
  } else {
    for (int k = lane; k < n_rows; k += 64) {
      const int row = brute ? k : cand_s[wave][k];
      if (row >= 0 && row < N) take(best, exact_dist(q_s[wave], db + (size_t)row * DIM, nq, dnorm[row]), row);
    }
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float ob1 = __shfl_xor(best.b1, d), ob2 = __shfl_xor(best.b2, d);
    const int oi1 = __shfl_xor(best.i1, d);
    if (oi1 >= 0) merge(best, ob1, ob2, oi1);
  }
  if (lane == 0) {
    idx1[q] = best.i1 >= 0 ? row_to_global(rmap, best.i1) : -1;
    d1[q] = best.b1;
    d2[q] = best.b2;
    if (stats) {   // per-query tallies (this wavefront is the query's only writer; launches are stream ordered): no atomics
      stats[3 * q] += (unsigned)n_cand;
      stats[3 * q + 1] += brute ? 1u : 0u;
      stats[3 * q + 2] += 1u;
    }
  }
}

// The screen value of every (query, row) pair of a small problem, by the arithmetic of pass A: one wavefront per
// (32 queries, 32 rows), operands straight from the f16 images, the accumulator seeded with the rows' -dd/2, the eight
// MFMAs of a block in ascending k.  Register r of lane (l32, half) = query l32, row (r & 3) + 8 (r >> 2) + 4 half.
__global__ __launch_bounds__(64) void screen_values_kernel(const _Float16* __restrict__ qh, const _Float16* __restrict__ dbh,
                                                           const float* __restrict__ dneg, int n_rows, float* __restrict__ out,
                                                           int shape16) {
  const int lane = threadIdx.x, half = lane >> 5, l32 = lane & 31;
  const int q0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  if (shape16) {   // screen16_kernel's arithmetic: 2 x 2 tiles of 16 x 16, four k-steps of 32, seeded with -dd/2
    const int quarter = lane >> 4, l16 = lane & 15;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        v4f acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = r0 + 16 * ti + 4 * quarter + r;
          acc[r] = dneg[(size_t)(row >> 7) * SC_DD + (row & 127)];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const half8 a = *reinterpret_cast<const half8*>(dbh + (size_t)(r0 + 16 * ti + l16) * DIM + 32 * s + 8 * quarter);
          const half8 b = *reinterpret_cast<const half8*>(qh + (size_t)(q0 + 16 * tj + l16) * DIM + 32 * s + 8 * quarter);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(q0 + 16 * tj + l16) * n_rows + r0 + 16 * ti + 4 * quarter + r] = acc[r];
      }
    return;
  }
  half8 a[8], b[8];
  const _Float16* arow = dbh + (size_t)(r0 + l32) * DIM + 8 * half;
  const _Float16* brow = qh + (size_t)(q0 + l32) * DIM + 8 * half;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    a[s] = *reinterpret_cast<const half8*>(arow + 16 * s);
    b[s] = *reinterpret_cast<const half8*>(brow + 16 * s);
  }
  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    acc[r] = dneg[(size_t)(row >> 7) * SC_DD + (row & 127)];
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[s], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) out[(size_t)(q0 + l32) * n_rows + r0 + (r & 3) + 8 * (r >> 2) + 4 * half] = acc[r];
}


}  // namespace

#ifdef SC_PROF
extern "C" int mh_debug_screen_trace(unsigned long long* out /* [2][1024] */) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sc_trace), 2 * 1024 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int mh_debug_screen_prof(unsigned long long out[8], int reset, int ablate) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sc_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_sc_prof), z, sizeof z) != hipSuccess) return -1;
  }
  g_sc_ablate = ablate;
  return 0;
}
#endif

// ---- host side -----------------------------------------------------------------------------------
float screen_margin_host(float qq, float dmax) { return screen_margin(qq, dmax); }

size_t screen_db_half_elems(int N) { return ((size_t)N + SC_TILE - 1) / SC_TILE * SC_TILE * DIM; }
size_t screen_dneg_elems(int N) { return ((size_t)N + SC_TILE - 1) / SC_TILE * SC_DD; }

void launch_db_to_half(const float* db, const float* dnorm, int N, _Float16* dbh, float* dneg, unsigned int* stats,
                       hipStream_t s) {
  const size_t n_chunks = screen_db_half_elems(N) / 8;
  if (n_chunks == 0) return;
  const unsigned blocks = (unsigned)std::min<size_t>((n_chunks + 255) / 256, 2048);
  hipLaunchKernelGGL(db_to_half_kernel, dim3(blocks), dim3(256), 0, s, db, dnorm, N, n_chunks, dbh, dneg, stats);
  const int n_tiles = (int)(n_chunks * 8 / ((size_t)SC_TILE * DIM));
  hipLaunchKernelGGL(db_block_bounds_kernel, dim3((n_tiles * 4 + 255) / 256), dim3(256), 0, s, dnorm, N, n_tiles, dneg, stats);
  hipLaunchKernelGGL(db_zero_query_kernel, dim3(1), dim3(1024), 0, s, dnorm, N, stats);
}

void launch_screen_values(const _Float16* qh, int Q, const ScreenDb& sdb, int n_rows, float* out, hipStream_t s, int shape) {
  if (Q <= 0 || n_rows <= 0) return;
  const int shape16 = shape == 2 || (shape == 0 && SC_SHAPE16_LARGE);
  hipLaunchKernelGGL(screen_values_kernel, dim3(Q / 32, n_rows / 32), dim3(64), 0, s, qh, sdb.dbh, sdb.dneg, n_rows, out, shape16);
}
void launch_screen_prepare(const float* qn, const float* qnorm, int Q, int q_pad, _Float16* qh, uint8_t* qbad, hipStream_t s) {
  hipLaunchKernelGGL(screen_prepare_kernel, dim3((q_pad * 16 + 255) / 256), dim3(256), 0, s, qn, qnorm, Q, (const int32_t*)nullptr,
                     q_pad, qh, qbad);
}

size_t screen_rec_slots() { return SC_SLOTS_MAX; }

int screen_q_pad(int Q) { return (Q + SC_QPAD - 1) / SC_QPAD * SC_QPAD; }
int screen_max_splits_a() { return 64; }

// When the screen pays: enough (query, row) pairs to be bound by arithmetic rather than by its launches, and
// enough rows for pass A's sample to mean something.  MH_MATCH_SCREEN = 0 / 1 pins the choice for the process
// (A/B runs; bench.py records it), mh_match_set_mode for a context.
bool screen_wanted(int q_expected, int N, int mode) {
  static const int pinned = exp_int("MH_MATCH_SCREEN", -1);
  if (mode < 0) mode = pinned;
  if (mode == 0 || N < 4096) return false;
  if (mode == 1) return true;
  return (double)q_expected * (double)N >= 8e6;
}

namespace {

template <int NQB, int NW>
void launch_passes(ScreenArgs a, int Q, int qe, int n_tiles, int sample, int blocks_a, int blocks_b, int* n_slots_out,
                   hipEvent_t* ev, hipStream_t s, int cu_avail) {
  constexpr int QB = 32 * NQB * NW;
  constexpr int SC_THREADS = 64 * NW, SC_LDS_BYTES = sc_lds_bytes(NW);
  static DynLds attr0, attr1;
  attr0.ensure(screen_kernel<0, NQB, NW>, SC_LDS_BYTES);
  attr1.ensure(screen_kernel<1, NQB, NW>, SC_LDS_BYTES);
  // (the workgroup targets below are in units of eight wavefronts: a four-wavefront grid has twice the workgroups)
  blocks_a *= 8 / NW;
  (void)cu_avail;
  const int nqb = (Q + QB - 1) / QB;        // the grid covers the capacity ...
  const int nqb_e = (qe + QB - 1) / QB;     // ... the splits are sized for the queries expected
  auto splits_for = [&](int n_sel, int target, int s_max) {
    int S = std::max(1, target / nqb_e);
    S = std::min(std::min(S, n_sel), s_max);
    // whole splits per XCD (a split's rows stay in one L2) unless the rounding would leave CUs without a workgroup:
    // 24 query blocks x 8 splits fill 192 of 256 CUs, x 10 fill 240 (measured: 10% on the stage at Q = 12000)
    if (S >= 8 && (S / 8 * 8) * 33 >= S * 32) S = S / 8 * 8;
    return std::max(S, 1);
  };
  // pass A: every `stride`-th tile, starting in the middle of the first stride
  // (a shard of a few models: the records a query leaves do not shrink with the shard, the MFMA work beside them does --
  // every 4th tile there: half the records for 1/8 more of pass A's rows.  15 000 rows x 96 000 queries: pass B 0.38 ->
  // 0.34 ms, pass A 0.055 -> 0.09, the rank's frames/s +2.5%)
  const int every = sample == 8 && n_tiles < 256 ? 4 : sample;
  const int stride = n_tiles >= 4 * every ? every : 1;
  const int n_sel_a = (n_tiles + stride - 1) / stride;
  const int Sa = splits_for(n_sel_a, blocks_a, screen_max_splits_a());
  a.n_sel = n_sel_a;
  a.tile_first = std::min(stride / 2, n_tiles - 1 - (n_sel_a - 1) * stride);
  if (a.tile_first < 0) a.tile_first = 0;
  a.tile_stride = stride;
  a.n_splits = Sa;
  a.tiles_base = n_sel_a / Sa;
  a.tiles_rem = n_sel_a % Sa;
  a.n_splits_a = Sa;
  hipLaunchKernelGGL((screen_kernel<0, NQB, NW>), dim3(nqb * Sa), dim3(SC_THREADS), SC_LDS_BYTES, s, a);
  if (ev) hipEventRecord(ev[2], s);
  hipLaunchKernelGGL(screen_tau_kernel, dim3((a.q_pad + 255) / 256), dim3(256), 0, s, a.part, Sa, a.q_pad, a.Q, a.q_count,
                     a.qnorm, a.qbad, a.dmax, const_cast<float*>(a.tau));
  if (ev) hipEventRecord(ev[3], s);
  // pass B: all tiles; a query's record slots are shared out over 2 x Sb lane-private sub-lists
  static const int sb_pin = exp_int("MH_SCREEN_SPLITS_B", 0);   // experiments
  // Twice as many workgroups as CUs when every one of them still sweeps >= 24 tiles: two rounds of half-length
  // workgroups finish more evenly than one round of ~240 (pass B alone 0.272 -> 0.261 ms at Q = 12000, N = 100k), and
  // under load a one-round grid stalls on every CU another frame's kernel holds (+5% frames/s at config 1, +3% at
  // config 2).  With fewer tiles per workgroup the query-operand prologue costs more than that (Q = 3000: 0.445 ->
  // 0.394 of peak), so small launches keep one workgroup per CU.
  if (blocks_b <= 0) blocks_b = (long)n_tiles * nqb_e * NW / 8 >= (NQB >= 4 ? 16L : 24L) * 512 ? 512 : 256;
  blocks_b *= 8 / NW;
  const int Sb = sb_pin > 0 ? std::min(std::min(sb_pin, n_tiles), SC_SLOTS_MAX / 2)
                            : splits_for(n_tiles, blocks_b, SC_SLOTS_MAX / 2);
  a.n_sel = n_tiles;
  a.tile_first = 0;
  a.tile_stride = 1;
  a.n_splits = Sb;
  a.tiles_base = n_tiles / Sb;
  a.tiles_rem = n_tiles % Sb;
  // (few splits = long sub-streams: a lane sees many of its query's candidates, its sub-list must hold them)
  a.sub_cap = std::max(1, std::min(48, SC_SLOTS_MAX / (2 * Sb)));
  hipLaunchKernelGGL((screen_kernel<1, NQB, NW>), dim3(nqb * Sb), dim3(SC_THREADS), SC_LDS_BYTES, s, a);
  if (ev) hipEventRecord(ev[4], s);
  *n_slots_out = 2 * Sb * a.sub_cap;
}

// the same launch policy for the 16x16x32 kernels (eight wavefronts; a query's slots shared out over 4 x splits sub-lists)
template <int NQB>
void launch_passes16(ScreenArgs a, int Q, int qe, int n_tiles, int sample, int blocks_a, int blocks_b, int* n_slots_out,
                     hipEvent_t* ev, hipStream_t s) {
  constexpr int QB = 32 * NQB * 8;
  static DynLds attr0, attr1;
  attr0.ensure(screen16_kernel<0, NQB>, SC_LDS16_BYTES);
  attr1.ensure(screen16_kernel<1, NQB>, SC_LDS16_BYTES);
  const int nqb = (Q + QB - 1) / QB;
  const int nqb_e = (qe + QB - 1) / QB;
  auto splits_for = [&](int n_sel, int target, int s_max) {
    int S = std::max(1, target / nqb_e);
    S = std::min(std::min(S, n_sel), s_max);
    if (S >= 8 && (S / 8 * 8) * 33 >= S * 32) S = S / 8 * 8;
    return std::max(S, 1);
  };
  // (a shard of a few models: the records a query leaves do not shrink with the shard, the MFMA work beside them does --
  // every 4th tile there: half the records for 1/8 more of pass A's rows.  15 000 rows x 96 000 queries: pass B 0.38 ->
  // 0.34 ms, pass A 0.055 -> 0.09, the rank's frames/s +2.5%)
  const int every = sample == 8 && n_tiles < 256 ? 4 : sample;
  const int stride = n_tiles >= 4 * every ? every : 1;
  const int n_sel_a = (n_tiles + stride - 1) / stride;
  const int Sa = splits_for(n_sel_a, blocks_a, screen_max_splits_a());
  a.n_sel = n_sel_a;
  a.tile_first = std::min(stride / 2, n_tiles - 1 - (n_sel_a - 1) * stride);
  if (a.tile_first < 0) a.tile_first = 0;
  a.tile_stride = stride;
  a.n_splits = Sa;
  a.tiles_base = n_sel_a / Sa;
  a.tiles_rem = n_sel_a % Sa;
  a.n_splits_a = Sa;
  hipLaunchKernelGGL((screen16_kernel<0, NQB>), dim3(nqb * Sa), dim3(512), SC_LDS16_BYTES, s, a);
  if (ev) hipEventRecord(ev[2], s);
  hipLaunchKernelGGL(screen_tau_kernel, dim3((a.q_pad + 255) / 256), dim3(256), 0, s, a.part, Sa, a.q_pad, a.Q, a.q_count,
                     a.qnorm, a.qbad, a.dmax, const_cast<float*>(a.tau));
#ifdef MH_EXPERIMENTS
  // what the records cost pass B: thresholds no value reaches (WRONG results: timing only)
  if (exp_int("MH_SCREEN_NO_HITS", 0)) hipMemsetAsync(const_cast<float*>(a.tau), 0x7f, (size_t)a.q_pad * sizeof(float), s);
#endif
  if (ev) hipEventRecord(ev[3], s);
  // Pass B's splits.  A query's 256 record slots are shared out over 4 x Sb lane-private sub-lists here (the 32x32x16
  // passes: 2 x Sb), and a sub-list needs room for three records or queries spill into the overflow list and from there
  // into pass C's brute-force search (12 000 queries x 250 000 rows ran 42 splits with ONE slot per sub-list for a while:
  // 42 brute-force queries per launch, pass C 10 ms instead of 0.03): Sb <= 21.  One workgroup per compute unit, so the
  // launch takes ceil(workgroups / 256) rounds of n_tiles / Sb tiles each: the Sb with the least rounds x tiles, the
  // larger of equals.  (blocks_b > 0: an experiment's request, capped the same way.)
  constexpr int SB16_MAX = SC_SLOTS_MAX / 12;
  // screen16_kernel counts a query's records per sub-list in one byte of n_rec[]; sub_cap <= SC_SLOTS_MAX / (4 * Sb)
  static_assert(SC_SLOTS_MAX / 4 < 256, "a sub-list's record count must fit the byte screen16_kernel keeps it in");
  int Sb = 1;
  if (blocks_b > 0) {
    Sb = std::min(splits_for(n_tiles, blocks_b, SC_SLOTS_MAX / 4), SB16_MAX);
  } else {
    double best = 1e30;
    for (int c = 1; c <= std::min(SB16_MAX, std::max(1, n_tiles / 8)); ++c) {
      const double cost = (double)((nqb_e * c + 255) / 256) * (double)((n_tiles + c - 1) / c);
      if (cost <= best) {
        best = cost;
        Sb = c;
      }
    }
  }
  a.n_sel = n_tiles;
  a.tile_first = 0;
  a.tile_stride = 1;
  a.n_splits = Sb;
  a.tiles_base = n_tiles / Sb;
  a.tiles_rem = n_tiles % Sb;
  // slots per sub-list: room for three records, and on a small DB ~96 slots per query in all (a query gets ~20 records)
  // rather than everything the 256 slots allow -- pass C reads (and re-empties) every slot of every query, and on a
  // shard of a few models, where few splits fill the chip, 8 slots per sub-list made that scan 2 KB per query where 768
  // bytes do.  Large DBs keep all 256: their queries' record counts have a long tail (1 M rows, 32 000 queries, 8
  // splits: 96 slots sent 117 queries per launch into pass C's brute-force search, 69 ms instead of 3).
  const int slots_target = n_tiles >= 2048 ? SC_SLOTS_MAX : (n_tiles >= 512 ? 160 : 96);
  a.sub_cap = std::max(1, std::min(std::max(3, (slots_target + 4 * Sb - 1) / (4 * Sb)), SC_SLOTS_MAX / (4 * Sb)));
  hipLaunchKernelGGL((screen16_kernel<1, NQB>), dim3(nqb * Sb), dim3(512), SC_LDS16_BYTES, s, a);
  if (ev) hipEventRecord(ev[4], s);
  *n_slots_out = 4 * Sb * a.sub_cap;
}

}  // namespace

void launch_match_screen(const float* qn, const float* qnorm, int Q, const float* db, const float* dnorm, int N,
                         const RowMap& rmap, const ScreenDb& sdb, const ScreenBufs& sb, int32_t* idx1, float* d1,
                         float* d2, hipStream_t s, const int32_t* q_count, int q_expected) {
  static const int sample = std::max(1, exp_int("MH_SCREEN_SAMPLE", 8));      // pass A looks at every `sample`-th tile
  static const int blocks_b = std::max(0, exp_int("MH_SCREEN_BLOCKS", 0));    // workgroups of pass B; 0 = by size (launch_passes)
  static const int blocks_a = std::max(1, exp_int("MH_SCREEN_BLOCKS_A", 256));
  static const int nqb_pin = exp_int("MH_SCREEN_NQB", 0);
  const int q_pad = screen_q_pad(Q);
  const int qe = (q_expected > 0 && q_expected < Q) ? std::max(q_expected, std::min(Q, 256)) : Q;
  const int n_tiles = (N + SC_TILE - 1) / SC_TILE;

  if (sb.ev) hipEventRecord(sb.ev[0], s);
  hipLaunchKernelGGL(screen_prepare_kernel, dim3((q_pad * 16 + 255) / 256), dim3(256), 0, s, qn, qnorm, Q, q_count, q_pad,
                     sb.qh, sb.qbad);
  // the lane: everything from pass A to pass B on its stream, handed over by events
  hipStream_t sbig = s;
  if (sb.big && sb.ev_in && sb.ev_out) {
    hipEventRecord(sb.ev_in, s);
    hipStreamWaitEvent(sb.big, sb.ev_in, 0);
    sbig = sb.big;
  }
  if (sb.ev) hipEventRecord(sb.ev[1], sbig);
  ScreenArgs a;
  a.qh = sb.qh;
  a.dbh = sdb.dbh;
  a.dneg = sdb.dneg;
  a.qnorm = qnorm;
  a.qbad = sb.qbad;
  a.q_count = q_count;
  a.part = sb.part;
  a.tau = sb.tau;
  a.recs = sb.recs;
  a.ovf_cnt = sb.ovf_cnt;
  a.ovf = sb.ovf;
  a.ovf_cap = sb.ovf_cap;
  a.sub_cap = 1;
  a.Q = Q;
  a.q_pad = q_pad;
  a.dmax = sdb.dmax;
#ifdef SC_PROF
  a.ablate = g_sc_ablate;
#else
  a.ablate = 0;
#endif
  // queries per workgroup: 1024 (four 32-query blocks per wavefront: every row fragment read from LDS feeds four
  // MFMAs; 256 VGPRs, no prefetched copy of the next row block) when the launch is large enough for workgroups of
  // >= 16 tiles -- pass B alone 8-10% faster, config 2 +9.7% frames/s --, else 512 (two blocks, row fragments
  // prefetched), or 256 for small frames.  768 is selectable for experiments.
  int nqb_sel = qe > 640 ? 2 : 1;
  if (qe >= 2048 && (long)n_tiles * ((qe + 1023) / 1024) >= 16L * 256) nqb_sel = 4;
  if (nqb_pin >= 1 && nqb_pin <= 4) nqb_sel = nqb_pin;
  int n_slots = 0;
  // wavefronts per workgroup: four (two workgroups share a CU) for the large launches, see the top of the file
  static const int nw_pin = exp_int("MH_SCREEN_NW", 0);
  const int nw_sel = nw_pin == 4 || nw_pin == 8 ? nw_pin : (nqb_sel == 4 ? SC_NW_LARGE : 8);
  // the MFMA shape of the large launches: 16x16x32 (SC_SHAPE16_LARGE; MH_SCREEN_SHAPE = 1 / 2 pins 32x32x16 / 16x16x32 in experiment builds)
  static const int shape_pin = exp_int("MH_SCREEN_SHAPE", 0);
  const bool shape16 = shape_pin == 2 || (shape_pin == 0 && SC_SHAPE16_LARGE);
  // (16x16x32 from 12 query blocks of 1024: with fewer, 21 splits leave compute units without a workgroup)
  if (nqb_sel == 4 && shape16 && (shape_pin == 2 || (qe + 1023) / 1024 >= 12)) launch_passes16<4>(a, Q, qe, n_tiles, sample, blocks_a, blocks_b, &n_slots, sb.ev, sbig);
  else if (nqb_sel == 2 && shape16 && shape_pin == 2) launch_passes16<2>(a, Q, qe, n_tiles, sample, blocks_a, blocks_b, &n_slots, sb.ev, sbig);
  else if (nqb_sel == 4 && nw_sel == 4) launch_passes<4, 4>(a, Q, qe, n_tiles, sample, blocks_a, blocks_b, &n_slots, sb.ev, sbig, 256);
  else if (nqb_sel == 4) launch_passes<4, 8>(a, Q, qe, n_tiles, sample, blocks_a, blocks_b, &n_slots, sb.ev, sbig, 256);
  else if (nqb_sel == 3) launch_passes<3, 8>(a, Q, qe, n_tiles, sample, blocks_a, blocks_b, &n_slots, sb.ev, sbig, 256);
  else if (nqb_sel == 2 && nw_sel == 4) launch_passes<2, 4>(a, Q, qe, n_tiles, sample, blocks_a, blocks_b, &n_slots, sb.ev, sbig, 256);
  else if (nqb_sel == 2) launch_passes<2, 8>(a, Q, qe, n_tiles, sample, blocks_a, blocks_b, &n_slots, sb.ev, sbig, 256);
  else launch_passes<1, 8>(a, Q, qe, n_tiles, sample, blocks_a, blocks_b, &n_slots, sb.ev, sbig, 256);
  if (sbig != s) {
    hipEventRecord(sb.ev_out, sbig);
    hipStreamWaitEvent(s, sb.ev_out, 0);
  }
  // pass C
  hipLaunchKernelGGL(rescore_kernel, dim3((Q + RS_WAVES - 1) / RS_WAVES), dim3(64 * RS_WAVES), 0, s, qn, qnorm, sb.qbad, Q,
                     q_count, db, dnorm, N, rmap, sb.recs, n_slots, sb.ovf_cnt, sb.ovf, sb.ovf_cap, sdb.dmax, (const float*)sb.tau, sdb.spread, idx1, d1, d2,
                     sb.stats, sdb.zero_idx, sdb.zero_d1, sdb.zero_d2);
  if (sb.ev) hipEventRecord(sb.ev[5], s);
}

}  // namespace mh
