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
//  (2) the closeness tests (canopy i against every earlier canopy) are independent:
//      all wavefronts compute them as bit rows, 64 canopies at a time.  One wavefront
//      then walks the list over those bits.  Every write of step i stores the value
//      i, so the only order dependence inside a step is whether an earlier canopy j'
//      (whose pointer is j) has already redirected j before j reads its own pointer.
//      That predicate ("ov") only matters when j's old target is not itself written in
//      the step; in that case it is resolved exactly by a short relaxation over the
//      pointer chains inside the step (:122-132);
//  (3) the folds (:134-148) -- sequential weighted-mean updates per target -- run one
//      thread per target, in rounds over the depth of the merge forest.
//
// One workgroup per point set (per model); all state in LDS; <= MS_CAP points.
#include "steps.h"

MH_TRACE_TU()

namespace mh {

namespace {

#ifdef MS_PROF   // phase timing build (make EXTRA=-DMS_PROF): cycles of thread 0 per phase, summed over calls
__device__ unsigned long long g_ms_prof[8];
__device__ unsigned long long g_ms_stat[8];   // walk: rows, rows with outward members, -, rows with a hit, relaxation rounds, cycles per row class
#ifdef MS_STATS   // (perturbs the timing: every update is a global read-modify-write)
#define MS_STAT(k, v) do { if (lane == 0) g_ms_stat[k] += (v); } while (0)
#else
#define MS_STAT(k, v) do { } while (0)
#endif
#define MS_T(k) do { if (tid == 0) { const unsigned long long now_ = clock64(); g_ms_prof[k] += now_ - t_prof; t_prof = now_; } } while (0)
#else
#define MS_T(k) do { } while (0)
#define MS_STAT(k, v) do { } while (0)
#endif

constexpr int MS_THREADS = 1024;   // one workgroup per point set owns a CU (LDS): use its 16 wavefronts
constexpr int MS_WAVES = MS_THREADS / 64;
constexpr int MS_ROWS = 64;            // outer canopies per block of the merge-marking phase
constexpr int MS_WORDS = MS_CAP / 64;  // 64-bit words of one row of the closeness matrix

// All per-canopy state is kept by LIST POSITION (canopiesRemaining order) and compacted in
// place when canopies are erased, so the inner loops read consecutive, wave-uniform addresses.
template <int ND>
struct MsLds {
  float C[ND][MS_CAP];    // canopy centre
  int S[MS_CAP];          // boundPointsSize
  int ID[MS_CAP];         // canopy id (= its first point): index of head/tail
  int mp[MS_CAP];         // merges: position this one merges into (self = none)
  int head[MS_CAP];       // boundPoints as a linked list over point ids, by canopy id
  int tail[MS_CAP];
  int next[MS_CAP];       // by point
  int nrem;
  int merged_any;
  int again;
  int ntargets;
  unsigned long long pmask[MS_WORDS];   // parallel marking's folds: positions that are targets with sources not folded yet
  // ---- from here to the end of the workgroup's LDS: dead between the closeness tests of an iteration and its folds;
  // the event matrices of the parallel merge marking live there (mark_parallel) ----
  float a[ND][MS_CAP];    // touchPtsAggregate
  int flag[MS_CAP];       // per-step stamps
  int st[MS_CAP];         // ov relaxation state of the members of the current step (0 outside it)
  int pending[MS_CAP];    // fold phase: 1 = target with sources not folded yet, 2 = folded this round
  int tlist[MS_CAP];      // fold phase: the targets
  unsigned long long rowbits[MS_ROWS][MS_WORDS];  // closeness of the block's outer canopies to the earlier ones
  unsigned long long tbw[MS_WORDS];   // positions that have been merge targets in this iteration (earlier blocks)
  unsigned long long neblock[2];      // rows of the current block that have members (by block parity)
  unsigned long long slowblock[2];    // rows of the current block that need the list walk
};
constexpr int MS_LDS_BYTES = 160 * 1024 - 1024;   // dynamic LDS of every mean-shift kernel: the CU's 160 KB less the kernels' few static words
static_assert(sizeof(MsLds<3>) <= MS_LDS_BYTES, "MsLds must fit the CU's LDS");

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// One wavefront folds every source of target position t into it, in list order (:136-143):
// centre = (centre * size + c.centre * c.size) / (size + c.size), splice c's points after t's.
// The sources' data are fetched 64 at a time; the running centre is wave-uniform arithmetic
// in the reference's operation order; the splices of one batch are independent of each other.
template <int ND>
__device__ __forceinline__ void fold_target(MsLds<ND>& L, int t, int lane) {
  float tC[ND];
#pragma unroll
  for (int x = 0; x < ND; ++x) tC[x] = L.C[x][t];
  int tsz = L.S[t];  // == boundPoints.size() of the target
  const int idt = L.ID[t];
  int carry_tail = L.tail[idt];
  const int npass = (t + 63) >> 6;
  for (int ps = 0; ps < npass; ++ps) {
    const int pj = ps * 64 + lane;
    const bool src = pj < t && L.mp[pj] == t;
    const unsigned long long mask = __ballot(src);
    if (mask == 0ull) continue;
    float sC[ND];
    int ssz = 0, sh = -1, stl = -1;
#pragma unroll
    for (int x = 0; x < ND; ++x) sC[x] = 0.f;
    if (src) {
#pragma unroll
      for (int x = 0; x < ND; ++x) sC[x] = L.C[x][pj];
      ssz = L.S[pj];
      const int id = L.ID[pj];
      sh = L.head[id];
      stl = L.tail[id];
    }
    const unsigned long long below = mask & ((1ull << lane) - 1ull);
    const int prev_lane = below ? 63 - __builtin_clzll(below) : 0;
    const int prev_tail_l = __shfl(stl, prev_lane);
    if (src) L.next[below ? prev_tail_l : carry_tail] = sh;
    carry_tail = __builtin_amdgcn_readlane(stl, __builtin_amdgcn_readfirstlane(63 - __builtin_clzll(mask)));
    for (unsigned long long m = mask; m; m &= m - 1ull) {
      const int l = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
      const int csz = __builtin_amdgcn_readlane(ssz, l);
      const int nsz = tsz + csz;
#pragma unroll
      for (int x = 0; x < ND; ++x) {
        const float cv = readlane_f(sC[x], l);
        const float v = __fadd_rn(__fmul_rn(tC[x], (float)tsz), __fmul_rn(cv, (float)csz));
        tC[x] = __fdiv_rn(v, (float)nsz);
      }
      tsz = nsz;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int x = 0; x < ND; ++x) L.C[x][t] = tC[x];
    L.S[t] = tsz;
    L.tail[idt] = carry_tail;
  }
}

// (2b) for point sets of up to 64*NP canopies: the list walk of one block with the pointers of
// the whole list in registers (lane l holds positions l, l+64, ...).  Only rows with members
// are visited.  Per row: the row's closeness words arrive as wave-uniform loads, the only
// dependent LDS access of the common step is the "is my pointer's target a member" bit test.
// Rows with outward members use LDS stamps for the cross-lane parts (flag: targets of
// members; st: positions written indirectly).
template <int ND, int NP>
__device__ __forceinline__ int walk_block_impl(MsLds<ND>& L, int i0, int nrem, unsigned long long ne, int lane,
                                               int stamp) {   // the stamp counter by value: by reference it lives in scratch
  int mpr[NP];
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int pos = u * 64 + lane;
    mpr[u] = pos < nrem ? L.mp[pos] : pos;
  }
  for (unsigned long long rows = ne; rows; rows &= rows - 1ull) {
    const int r = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rows));
    const int i = i0 + r;
#ifdef MS_STATS
    const unsigned long long t_row = clock64();
#endif
    // loads are unconditional (clamped indices) so that each batch is one LDS round trip
    unsigned long long W[NP], TW[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) W[u] = L.rowbits[r][u];
#pragma unroll
    for (int u = 0; u < NP; ++u) TW[u] = L.rowbits[r][(mpr[u] >> 6) & (MS_WORDS - 1)];
    bool mem[NP], out[NP];
    bool any_out_l = false;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int pos = u * 64 + lane;
      const int t = mpr[u];
      mem[u] = pos < i && ((W[u] >> lane) & 1ull);           // words past the row's end are stale: pos < i masks them
      out[u] = mem[u] && t != pos && !((TW[u] >> (t & 63)) & 1ull);
      any_out_l |= out[u];
    }
    MS_STAT(0, 1);
#ifdef MS_STATS
    int row_class = 5;
#endif
    if (__ballot(any_out_l) != 0ull) {
#ifdef MS_STATS
      row_class = 6;
#endif
      MS_STAT(1, 1);
      // is an outward member the target of another member?
      ++stamp;
#pragma unroll
      for (int u = 0; u < NP; ++u)
        if (mem[u] && mpr[u] != u * 64 + lane) L.flag[mpr[u]] = stamp;
      int fl[NP];
#pragma unroll
      for (int u = 0; u < NP; ++u) fl[u] = L.flag[u * 64 + lane];
      bool hit = false;
#pragma unroll
      for (int u = 0; u < NP; ++u) hit |= out[u] && fl[u] == stamp;
      int stv[NP];   // 1 = member & not ov, 2 = member & ov
#pragma unroll
      for (int u = 0; u < NP; ++u) stv[u] = mem[u] ? 1 : 0;
      if (__ballot(hit) != 0ull) {
        MS_STAT(3, 1);
#ifdef MS_STATS
        row_class = 7;
#endif
        // ov relaxation: ov(j) = exists j' in S_i, !ov(j'), m[j'] == j (j' != j).  The sweep above
        // (every member flags its pointer) was its first round.
#pragma unroll
        for (int u = 0; u < NP; ++u)
          if (mem[u] && fl[u] == stamp) stv[u] = 2;
        for (int round = 1; round < MS_CAP; ++round) {
          ++stamp;
#pragma unroll
          for (int u = 0; u < NP; ++u)
            if (stv[u] == 1 && mpr[u] != u * 64 + lane) L.flag[mpr[u]] = stamp;  // a canopy is not its own predecessor
#pragma unroll
          for (int u = 0; u < NP; ++u) fl[u] = L.flag[u * 64 + lane];
          bool ch = false;
#pragma unroll
          for (int u = 0; u < NP; ++u)
            if (stv[u] != 0) {
              const int s1 = (fl[u] == stamp) ? 2 : 1;
              ch |= s1 != stv[u];
              stv[u] = s1;
            }
          MS_STAT(4, 1);
          if (__ballot(ch) == 0ull) break;
        }
      }
      // indirect writes of step i: the old targets of the members that were not pre-empted.  The flags of the last
      // sweep mark exactly those positions: that sweep was made by the members with stv == 1 (all members when there
      // was no hit; the final, unchanged state when the relaxation ran), and a flagged position that is the target of
      // a non-outward member is itself a member and gets i below anyway -- no sweep of its own (it was two more
      // dependent LDS round trips per row).
#pragma unroll
      for (int u = 0; u < NP; ++u)
        if (u * 64 + lane < nrem && fl[u] == stamp) mpr[u] = i;
    }
    // direct writes: every member points at i
#pragma unroll
    for (int u = 0; u < NP; ++u)
      if (mem[u]) mpr[u] = i;
    MS_STAT(row_class, clock64() - t_row);
  }
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int pos = u * 64 + lane;
    if (pos < nrem) L.mp[pos] = mpr[u];
  }
  return stamp;
}
// Inlined for the sizes a frame's models usually have; the large ones stay functions of their own (inlined, their
// register arrays push the whole kernel over the 128 VGPRs a 1024-thread workgroup allows), at the price of the
// call and of one dynamic-LDS table lookup per LDS base address.
template <int ND, int NP>
__device__ __forceinline__ int walk_block_regs(MsLds<ND>& L, int i0, int nrem, unsigned long long ne, int lane, int stamp) {
  return walk_block_impl<ND, NP>(L, i0, nrem, ne, lane, stamp);
}
template <int ND, int NP>
__device__ __noinline__ int walk_block_regs_call(MsLds<ND>& L, int i0, int nrem, unsigned long long ne, int lane, int stamp) {
  return walk_block_impl<ND, NP>(L, i0, nrem, ne, lane, stamp);
}

// (2') Merge marking without the list walk, for point sets whose event matrices fit the LDS left of the position
// arrays (up to 640 canopies in 2-D, 576 in 3-D; larger sets take the walk above until they have shrunk).
//
// The reference's loop (:122-132) is `for i: for j < i close to i: m[m[j]] = i; m[j] = i`.  Every store of step i
// writes the value i, so "m[p] at step i" is simply the LAST STEP BEFORE i THAT WROTE p (p itself if none), and the
// whole loop is described by its write events.  Position p is written at step i > p
//   directly   iff p is close to i                                  D[p][i]: known up front (the closeness matrix), and
//   indirectly iff some member q of step i (q close to i, q != p) that has not itself been redirected earlier in the
//              step still points at p when its turn comes:           IT[p][i] = exists q: D[q][i] & !IT[q][i] & last(q, i) = p
// with last(q, i) = the latest event of q before i in D[q] | IT[q].  (A member q that an earlier member has
// redirected -- IT[q][i] -- reads i as its pointer, and `m[i] = i` is a no-op: that is the order dependence of the
// step, and it is the same predicate as the indirect write.)  An event at (p, i) depends only on events at earlier
// steps and, within step i, on events of earlier positions (last(q, i) = p implies q < p): the system is causal in
// (i, p) order, so it has exactly one solution -- the sequential loop's -- and iterating IT <- F(IT) from IT = 0 reaches it
// (the entries settle in causal order; 5-10 rounds on real keypoint layouts, 2 on one dense blob) and stops there
// (F(IT) = IT).  Afterwards m[p] = the last event of p.
//
// A round is embarrassingly parallel from the positions' point of view: the thread that owns 64 steps of position p
// (one word of its row) walks the events of that word in ascending order, carries `last` along and, at every direct
// event that is not also an indirect one, reports (last, i) with one LDS atomic.  Rows are stored from their diagonal
// word on (groups of 64 rows with W - g words each); D never changes, so a thread keeps its words of it in registers.
constexpr int MS_PAR_UNITS = 4;   // row words a thread may own: 4096 words per matrix, 640 canopies
template <int ND>
__device__ __forceinline__ bool mark_parallel_fits(int nrem) {
  const int W = (nrem + 63) >> 6;
  const int T = 32 * W * (W + 1);   // words of one matrix: 64 rows x (W - g) words for g = 0..W-1
  const int region = MS_LDS_BYTES - (int)offsetof(MsLds<ND>, a);
  const int a_bytes = (int)sizeof(float) * ND * MS_CAP;
  const int off_d = max(a_bytes, 16 * T);   // D is built from `a`: it must not lie on it; the two IT matrices may
  return T <= MS_PAR_UNITS * MS_THREADS && off_d + 8 * T <= region;
}
__device__ __forceinline__ int ms_src_off(int t) {   // first word of row t of a lower-triangular matrix (words 0 .. t >> 6)
  const int g = t >> 6;
  return 32 * __mul24(g, g + 1) + __mul24(t & 63, g + 1);
}
__device__ __forceinline__ int ms_row_off(int p, int W) {   // first word (the diagonal word) of row p
  const int g = p >> 6;
  return 32 * __mul24(g, 2 * W - g + 1) + __mul24(p & 63, W - g);   // (v_mul_lo_u32 is a quarter-rate instruction)
}
template <int ND>
__device__ __forceinline__ const unsigned long long* mark_parallel(MsLds<ND>& L, int nrem, float sq_merge
#ifdef MS_PROF
                                              , unsigned long long& t_prof
#endif
) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W = (nrem + 63) >> 6;
  const int T = 32 * W * (W + 1);
  unsigned long long* const base = reinterpret_cast<unsigned long long*>(&L.a[0][0]);
  unsigned long long* it_old = base;
  unsigned long long* it_new = base + T;
  unsigned long long* const D = base + max((int)sizeof(float) * ND * MS_CAP / 8, 2 * T);
  // ---- D: row p = the later canopies close to p (this = the later one, other = p: :124-125).  (Word-major with the
  // rows as wave-uniform reads, several in flight, is no faster: the phase is bound by its instruction count with four
  // wavefronts per SIMD, not by LDS latency.) ----
  for (int p = wave; p < nrem; p += MS_WAVES) {
    const int g = p >> 6;
    float ap[ND];
#pragma unroll
    for (int x = 0; x < ND; ++x) ap[x] = L.a[x][p];
    unsigned long long mine = 0ull;
    for (int w = g; w < W; ++w) {
      const int j = w * 64 + lane;
      bool close = false;
      if (j > p && j < nrem) {
        float dist = 0.f;
#pragma unroll
        for (int x = 0; x < ND; ++x) {
          const float d = __fsub_rn(ap[x], L.a[x][j]);   // sqEuclDist(other): other - this
          dist = __fadd_rn(dist, __fmul_rn(d, d));
        }
        close = dist < sq_merge;
      }
      const unsigned long long b = __ballot(close);
      if (lane == w - g) mine = b;
    }
    if (lane < W - g) D[ms_row_off(p, W) + lane] = mine;
  }
  __syncthreads();   // `a` is dead from here on
  MS_T(2);
  for (int x = tid; x < 2 * T; x += MS_THREADS) base[x] = 0ull;
  // ---- this thread's row words ----
  int up[MS_PAR_UNITS], uw[MS_PAR_UNITS];
  unsigned long long ud[MS_PAR_UNITS];
#pragma unroll
  for (int k = 0; k < MS_PAR_UNITS; ++k) {
    const int x = tid + k * MS_THREADS;
    up[k] = -1;
    uw[k] = 0;
    ud[k] = 0ull;
    if (x < T) {
      int g = 0, gb = 0;
      while (x >= gb + 64 * (W - g)) {
        gb += 64 * (W - g);
        ++g;
      }
      const int r = x - gb, wd = W - g;
      const int q = r / wd;
      const int p = 64 * g + q;
      if (p < nrem) {
        up[k] = p;
        uw[k] = g + (r - q * wd);
        ud[k] = D[x];
      }
    }
  }
  __syncthreads();
  MS_T(7);
  // ---- rounds ----
  for (;;) {
#pragma unroll
    for (int k = 0; k < MS_PAR_UNITS; ++k) {
      const int p = up[k];
      if (p < 0) continue;
      const int x = tid + k * MS_THREADS;
      const unsigned long long dw = ud[k], iw = it_old[x];
      unsigned long long ev = dw | iw;
      if (ev == 0ull) continue;
      const int w = uw[k], g = p >> 6;
      int last = p;   // the latest event of p before this word
      for (int b = 1; b <= w - g; ++b) {
        const unsigned long long e = D[x - b] | it_old[x - b];
        if (e) {
          last = 64 * (w - b) + 63 - __builtin_clzll(e);
          break;
        }
      }
      do {
        const int b = __builtin_ctzll(ev);
        const unsigned long long bit = 1ull << b;
        ev &= ev - 1ull;
        if ((dw & bit) && !(iw & bit) && last != p)
          atomicOr(&it_new[ms_row_off(last, W) + (w - (last >> 6))], bit);
        last = 64 * w + b;
      } while (ev);
    }
    __syncthreads();
    int changed = 0;
    for (int x = tid; x < T; x += MS_THREADS) {
      changed |= it_old[x] != it_new[x];
      it_old[x] = 0ull;
    }
    unsigned long long* const t = it_old;
    it_old = it_new;
    it_new = t;
    if (!__syncthreads_or(changed)) break;
  }
  // (Leaving the step words before the first change of a round alone in the next one -- events are causal in time --
  // was built and bought nothing: the late rounds that it shortens are cheap already.  So was one thread per
  // (position, step) PAIR instead of per word -- a wavefront waits for the lane with the most events in its word, 9-14
  // where the average is 3-5: pairs listed by a scan over the words' counts and kept 16 per thread in a rotating register
  // array (sixteen unrolled bodies spill 150 registers, a run-time index puts the array into scratch).  Exact, the rounds
  // 30% shorter at n = 500 and 586, but the list costs 35 k cycles per iteration of the mean shift and the rotation as
  // much as the work it balances: n = 500 -8%, n = 379 +8%, n = 150 +17% on the kernel.  Dropped.)
  // ---- m[p] = the last event of p; SM[t] = the sources of t (row t: words 0 .. t >> 6), in the IT buffer that the last
  // round left empty; L.pmask = the targets ----
  unsigned long long* const SM = it_new;
  if (tid < MS_WORDS) L.pmask[tid] = 0ull;
  __syncthreads();
  for (int p = tid; p < nrem; p += MS_THREADS) {
    const int g = p >> 6, ro = ms_row_off(p, W);
    int best = p;
    for (int w = W - 1; w >= g; --w) {
      const unsigned long long e = D[ro + w - g] | it_old[ro + w - g];
      if (e) {
        best = 64 * w + 63 - __builtin_clzll(e);
        break;
      }
    }
    L.mp[p] = best;
    if (best != p) {
      atomicOr(&SM[ms_src_off(best) + g], 1ull << (p & 63));
      atomicOr(&L.pmask[best >> 6], 1ull << (best & 63));
    }
  }
  __syncthreads();
  return SM;
}

// (3') the folds of (3) with the sources of a target as a bit row (mark_parallel's SM) and the unfinished targets as
// a bit mask: a target is ready when none of its sources is an unfinished target -- two ANDs instead of a sweep over
// the pointers of all earlier positions per target and round.
template <int ND>
__device__ __forceinline__ void fold_parallel(MsLds<ND>& L, const unsigned long long* SM, int nrem) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W = (nrem + 63) >> 6;
  for (;;) {
    const unsigned long long pm = lane < W ? L.pmask[lane] : 0ull;   // lane u: word u of the unfinished targets
    if (__ballot(pm != 0ull) == 0ull) break;   // (the same in every wavefront)
    __syncthreads();   // everybody has its snapshot
    int k = 0;
    for (int u = 0; u < W; ++u) {
      unsigned long long tw = readlane64(pm, u);
      for (; tw; tw &= tw - 1ull, ++k) {
        if ((k & (MS_WAVES - 1)) != wave) continue;
        const int t = u * 64 + __builtin_amdgcn_readfirstlane(__builtin_ctzll(tw));
        const unsigned long long* row = SM + ms_src_off(t);
        const unsigned long long sw = lane <= u ? row[lane] : 0ull;   // lane v: the sources of t among positions [64 v, 64 v + 64)
        if (__ballot((sw & pm) != 0ull) != 0ull) continue;            // a source still waits for its own sources
        float tC[ND];
#pragma unroll
        for (int x = 0; x < ND; ++x) tC[x] = L.C[x][t];
        int tsz = L.S[t];
        const int idt = L.ID[t];
        int carry_tail = L.tail[idt];
        for (int v = 0; v <= u; ++v) {
          const unsigned long long mask = readlane64(sw, v);
          if (mask == 0ull) continue;
          const int pj = v * 64 + lane;
          const bool src = (mask >> lane) & 1ull;
          float sC[ND];
          int ssz = 0, sh = -1, stl = -1;
#pragma unroll
          for (int x = 0; x < ND; ++x) sC[x] = 0.f;
          if (src) {
#pragma unroll
            for (int x = 0; x < ND; ++x) sC[x] = L.C[x][pj];
            ssz = L.S[pj];
            const int id = L.ID[pj];
            sh = L.head[id];
            stl = L.tail[id];
          }
          const unsigned long long below = mask & ((1ull << lane) - 1ull);
          const int prev_lane = below ? 63 - __builtin_clzll(below) : 0;
          const int prev_tail_l = __shfl(stl, prev_lane);
          if (src) L.next[below ? prev_tail_l : carry_tail] = sh;
          carry_tail = __builtin_amdgcn_readlane(stl, __builtin_amdgcn_readfirstlane(63 - __builtin_clzll(mask)));
          for (unsigned long long m = mask; m; m &= m - 1ull) {
            const int l = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
            const int csz = __builtin_amdgcn_readlane(ssz, l);
            const int nsz = tsz + csz;
#pragma unroll
            for (int x = 0; x < ND; ++x) {
              const float cv = readlane_f(sC[x], l);
              const float vv = __fadd_rn(__fmul_rn(tC[x], (float)tsz), __fmul_rn(cv, (float)csz));
              tC[x] = __fdiv_rn(vv, (float)nsz);
            }
            tsz = nsz;
          }
        }
        if (lane == 0) {
#pragma unroll
          for (int x = 0; x < ND; ++x) L.C[x][t] = tC[x];
          L.S[t] = tsz;
          L.tail[idt] = carry_tail;
          atomicAnd(&L.pmask[u], ~(1ull << (t & 63)));
        }
      }
    }
    __syncthreads();
  }
}

template <int ND>
__device__ __forceinline__ void meanshift_body(MsLds<ND>& L, const float* __restrict__ pts, int pts_stride, int n,
                               float radius, float merge, int min_pts, int max_iter,
                               int32_t* __restrict__ members_out, int32_t member_base,
                               int32_t* __restrict__ cl_start_out, int32_t* __restrict__ ncl_out,
                               int32_t* __restrict__ label_out, int32_t* __restrict__ iters_out) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float sq_radius = __fmul_rn(radius, radius);
  const float sq_merge = __fmul_rn(merge, merge);
#ifdef MS_PROF
  unsigned long long t_prof = clock64();
#endif

  for (int i = tid; i < n; i += MS_THREADS) {
#pragma unroll
    for (int x = 0; x < ND; ++x) L.C[x][i] = pts[(size_t)i * pts_stride + x];
    L.S[i] = 1;
    L.ID[i] = i;
    L.head[i] = i;
    L.tail[i] = i;
    L.next[i] = -1;
    L.flag[i] = 0;
    L.st[i] = 0;
  }
  if (tid == 0) L.nrem = n;
  __syncthreads();

  int stamp = 0;  // used by wavefront 0 only
  int it = 0;
  for (; it < max_iter; ++it) {
    const int nrem = L.nrem;
    MS_T(0);
    if (tid < MS_WORDS) L.tbw[tid] = 0ull;
    if (tid < 2) {
      L.neblock[tid] = 0ull;
      L.slowblock[tid] = 0ull;
    }
    // ---- (1) weighted mean of the canopies within Radius (:102-120) ------------------
    // One thread per canopy, the others in list order.  Every thread walks the same list, so
    // a wavefront fetches 64 canopies at a time (one per lane) and broadcasts them one after
    // the other through scalar registers.
    for (int pb = wave * 64; pb < nrem; pb += MS_THREADS) {   // wave-uniform trip count
      const int p = pb + lane;
      const bool live = p < nrem;
      float cc[ND], agg[ND];
      const int sz = live ? L.S[p] : 1;
      const float fsz = (float)sz;
#pragma unroll
      for (int x = 0; x < ND; ++x) {
        cc[x] = live ? L.C[x][p] : 0.f;
        agg[x] = __fmul_rn(cc[x], fsz);
      }
      int touch = sz;
      for (int p0 = 0; p0 < nrem; p0 += 64) {
        const int q = p0 + lane;
        float oc[ND], ow[ND];
        int os = 0;
#pragma unroll
        for (int x = 0; x < ND; ++x) oc[x] = ow[x] = 0.f;
        if (q < nrem) {
          os = L.S[q];
#pragma unroll
          for (int x = 0; x < ND; ++x) {
            oc[x] = L.C[x][q];
            ow[x] = __fmul_rn(oc[x], (float)os);   // center * boundPointsSize
          }
        }
        const int cnt = min(64, nrem - p0);
        for (int l = 0; l < cnt; ++l) {
          float bc[ND], bw[ND];
#pragma unroll
          for (int x = 0; x < ND; ++x) {
            bc[x] = readlane_f(oc[x], l);
            bw[x] = readlane_f(ow[x], l);
          }
          const int bs = __builtin_amdgcn_readlane(os, l);
          float dist = 0.f;
#pragma unroll
          for (int x = 0; x < ND; ++x) {
            const float d = __fsub_rn(bc[x], cc[x]);
            dist = __fadd_rn(dist, __fmul_rn(d, d));
          }
          if (dist < sq_radius && p0 + l != p) {
            touch += bs;
#pragma unroll
            for (int x = 0; x < ND; ++x) agg[x] = __fadd_rn(agg[x], bw[x]);
          }
        }
      }
      if (live) {
#pragma unroll
        for (int x = 0; x < ND; ++x) L.a[x][p] = __fdiv_rn(agg[x], (float)touch);
        L.mp[p] = p;
      }
    }
    __syncthreads();
    MS_T(1);

    // ---- (2) merge marking (:122-132), in blocks of MS_ROWS outer canopies ------------
    // The reference walks the list: for canopy i, every EARLIER canopy j whose mean is within
    // Merge does `m[m[j]] = i; m[j] = i` in list order.  Every store of step i writes the value
    // i, so step i is: all members of S_i = {j < i close to i} get i, and the old target t_j of
    // member j gets i unless an earlier member j' whose pointer was j had already redirected j
    // ("ov": j then reads i, and m[i] = i is a no-op).  ov only matters for "outward" members
    // (t_j neither j nor a member: members get i anyway).
    //  (2a) all wavefronts: closeness bits of the block's rows against the earlier positions;
    //  (2b) wavefront 0 walks the block's rows.  `tb` = every position that has been a merge
    //       target in this iteration (pointers only ever hold such positions): when all of
    //       them are members of S_i no member can be outward and the step is just "members
    //       get i"; otherwise the outward members are looked at, and only if one of them is
    //       another member's target is the relaxation needed.
    //       A block none of whose rows fails that test needs no walk at all: position j simply
    //       ends up pointing at the last row of the block it is close to.
    constexpr int RPW = MS_ROWS / MS_WAVES;   // rows per wavefront
#ifdef MS_NO_PAR   // (experiment build: the list walk for every size)
    const bool par_mark = false;
#else
    const bool par_mark = mark_parallel_fits<ND>(nrem);
#endif
    const unsigned long long* par_sm = nullptr;
    if (par_mark) {
#ifdef MS_PROF
      par_sm = mark_parallel<ND>(L, nrem, sq_merge, t_prof);
#else
      par_sm = mark_parallel<ND>(L, nrem, sq_merge);
#endif
      MS_T(3);
    }
    for (int i0 = 0; i0 < nrem && !par_mark; i0 += MS_ROWS) {
      const int ps0 = i0 >> 6;                // the block's rows are the positions of word ps0
      const int par = ps0 & 1;                // the flag words alternate: they are reset a block later
      unsigned long long mine[RPW];
#pragma unroll
      for (int u = 0; u < RPW; ++u) {
        mine[u] = 0ull;
        const int r = wave * RPW + u;
        const int i = i0 + r;
        if (i >= nrem || i == 0) continue;
        float ai[ND];
#pragma unroll
        for (int x = 0; x < ND; ++x) ai[x] = L.a[x][i];
        const int npass = (i + 63) >> 6;
        for (int ps = 0; ps < npass; ++ps) {
          const int pj = ps * 64 + lane;
          bool close = false;
          if (pj < i) {
            float dist = 0.f;
#pragma unroll
            for (int x = 0; x < ND; ++x) {
              const float d = __fsub_rn(L.a[x][pj], ai[x]);  // sqEuclDist(other) : other - this
              dist = __fadd_rn(dist, __fmul_rn(d, d));
            }
            close = dist < sq_merge;
          }
          const unsigned long long b = __ballot(close);
          if (lane == ps) mine[u] = b;
        }
        if (lane < npass) L.rowbits[r][lane] = mine[u];
        if (__ballot(mine[u] != 0ull) != 0ull && lane == 0) atomicOr(&L.neblock[par], 1ull << r);
      }
      __syncthreads();
      {
        const unsigned long long ne = L.neblock[par];
        const unsigned long long tprev = lane < MS_WORDS ? L.tbw[lane] : 0ull;
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
          const int r = wave * RPW + u;
          if (!((ne >> r) & 1ull)) continue;
          unsigned long long tbv = tprev;
          if (lane == ps0) tbv |= ne & ((1ull << r) - 1ull);
          if (__ballot((tbv & ~mine[u]) != 0ull) != 0ull && lane == 0) atomicOr(&L.slowblock[par], 1ull << r);
        }
      }
      __syncthreads();
      MS_T(2);
      const unsigned long long ne = L.neblock[par];
      if (L.slowblock[par] == 0ull) {
        // no walk: position j points at the last row of the block that holds it
        const int jend = min(nrem, i0 + MS_ROWS);
        for (int j = tid; j < jend; j += MS_THREADS) {
          unsigned long long m = ne;
          if (j >= i0) m &= ~((2ull << (j - i0)) - 1ull);   // rows after j only
          while (m) {
            const int r = 63 - __builtin_clzll(m);
            if ((L.rowbits[r][j >> 6] >> (j & 63)) & 1ull) {
              L.mp[j] = i0 + r;
              break;
            }
            m &= ~(1ull << r);
          }
        }
      } else if (n <= 64 * 16) {
        if (wave == 0) {
          if (n <= 64 * 2) stamp = walk_block_regs<ND, 2>(L, i0, nrem, ne, lane, stamp);
          else if (n <= 64 * 4) stamp = walk_block_regs<ND, 4>(L, i0, nrem, ne, lane, stamp);
          else if (n <= 64 * 6) stamp = walk_block_regs<ND, 6>(L, i0, nrem, ne, lane, stamp);
          else if (n <= 64 * 8) stamp = walk_block_regs<ND, 8>(L, i0, nrem, ne, lane, stamp);
          else if (n <= 64 * 12) stamp = walk_block_regs_call<ND, 12>(L, i0, nrem, ne, lane, stamp);
          else stamp = walk_block_regs_call<ND, 16>(L, i0, nrem, ne, lane, stamp);
        }
      } else if (wave == 0) {
        unsigned long long tb = lane < MS_WORDS ? L.tbw[lane] : 0ull;   // lane ps: target bits of positions [64 ps, 64 ps + 64)
        for (int r = (i0 == 0 ? 1 : 0); r < MS_ROWS; ++r) {
          const int i = i0 + r;
          if (i >= nrem) break;
          const int npass = (i + 63) >> 6;
          const unsigned long long w = lane < npass ? L.rowbits[r][lane] : 0ull;
          const unsigned long long active = __ballot(w != 0ull);   // passes that hold members
          if (active == 0ull) continue;
          if (__ballot((tb & ~w) != 0ull) != 0ull) {
            // outward members: pointer neither self nor a member of S_i
            unsigned long long outw = 0ull;   // lane ps: outward bits of pass ps
            bool any_out = false;
            for (unsigned long long m = active; m; m &= m - 1ull) {
              const int ps = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
              const bool mem = (readlane64(w, ps) >> lane) & 1ull;
              const int pj = ps * 64 + lane;
              bool out = false;
              if (mem) {
                const int t = L.mp[pj];
                if (t != pj) out = !((L.rowbits[r][t >> 6] >> (t & 63)) & 1ull);
              }
              const unsigned long long ob = __ballot(out);
              if (lane == ps) outw = ob;
              any_out |= ob != 0ull;
            }
            if (any_out) {
              // is an outward member the target of another member?
              ++stamp;
              for (unsigned long long m = active; m; m &= m - 1ull) {
                const int ps = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                const bool mem = (readlane64(w, ps) >> lane) & 1ull;
                const int pj = ps * 64 + lane;
                if (mem) {
                  const int t = L.mp[pj];
                  if (t != pj) L.flag[t] = stamp;
                }
              }
              bool need = false;
              for (unsigned long long m = active; m; m &= m - 1ull) {
                const int ps = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                const bool out = (readlane64(outw, ps) >> lane) & 1ull;
                const int pj = ps * 64 + lane;
                const bool hit = out && L.flag[pj] == stamp;
                need |= __ballot(hit) != 0ull;
              }
              if (need) {
                // ov relaxation: ov(j) = exists j' in S_i, !ov(j'), m[j'] == j (j' != j).
                // st: 1 = member & not ov, 2 = member & ov.
                for (unsigned long long m = active; m; m &= m - 1ull) {
                  const int ps = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                  if ((readlane64(w, ps) >> lane) & 1ull) L.st[ps * 64 + lane] = 1;
                }
                for (int round = 0; round < MS_CAP; ++round) {
                  ++stamp;
                  for (unsigned long long m = active; m; m &= m - 1ull) {
                    const int ps = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                    const int pj = ps * 64 + lane;
                    if (((readlane64(w, ps) >> lane) & 1ull) && L.st[pj] == 1) {
                      const int t = L.mp[pj];
                      if (t != pj) L.flag[t] = stamp;  // a canopy is not its own predecessor
                    }
                  }
                  bool changed = false;
                  for (unsigned long long m = active; m; m &= m - 1ull) {
                    const int ps = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                    const int pj = ps * 64 + lane;
                    bool ch = false;
                    if ((readlane64(w, ps) >> lane) & 1ull) {
                      const int s0 = L.st[pj];
                      const int s1 = (L.flag[pj] == stamp) ? 2 : 1;
                      ch = s1 != s0;
                      L.st[pj] = s1;
                    }
                    changed |= __ballot(ch) != 0ull;
                  }
                  if (!changed) break;
                }
              }
              // indirect writes of step i: the old targets of the outward members that were not pre-empted
              for (unsigned long long m = active; m; m &= m - 1ull) {
                const int ps = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                const bool out = (readlane64(outw, ps) >> lane) & 1ull;
                const int pj = ps * 64 + lane;
                if (out && !(need && L.st[pj] == 2)) L.mp[L.mp[pj]] = i;
              }
              if (need)
                for (unsigned long long m = active; m; m &= m - 1ull) {
                  const int ps = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                  if ((readlane64(w, ps) >> lane) & 1ull) L.st[ps * 64 + lane] = 0;
                }
            }
          }
          // direct writes: every member points at i
          for (unsigned long long m = active; m; m &= m - 1ull) {
            const int ps = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
            if ((readlane64(w, ps) >> lane) & 1ull) L.mp[ps * 64 + lane] = i;
          }
          if (lane == (i >> 6)) tb |= 1ull << (i & 63);
        }
      }
      __syncthreads();
      if (tid == 0) {
        L.tbw[ps0] |= ne;
        L.neblock[par] = 0ull;
        L.slowblock[par] = 0ull;
      }
      MS_T(3);
    }

    // ---- (3) fold marked canopies into their targets (:134-148) ------------------------
    // The reference folds in list order.  A target always sits later in the list than its
    // sources, so what matters is (a) a canopy is complete (has absorbed its own sources)
    // before it is folded, (b) the sources of one target are folded in list order.  Targets
    // whose sources are all complete are independent: one wavefront per target, in rounds
    // over the depth of the merge forest; very deep forests finish in list order.
    if (par_mark) {
      const unsigned long long pm = lane < MS_WORDS ? L.pmask[lane] : 0ull;
      if (__ballot(pm != 0ull) == 0ull) {   // nothing merges: done (:96, `done` stays true)
        ++it;
        break;
      }
      fold_parallel<ND>(L, par_sm, nrem);
    } else {
      for (int p = tid; p < nrem; p += MS_THREADS) L.pending[p] = 0;
      if (tid == 0) {
        L.merged_any = 0;
        L.ntargets = 0;
      }
      __syncthreads();
    for (int p = tid; p < nrem; p += MS_THREADS) {
      const int t = L.mp[p];
      if (t != p) {
        if (atomicExch(&L.pending[t], 1) == 0) L.tlist[atomicAdd(&L.ntargets, 1)] = t;
        L.merged_any = 1;
      }
    }
    __syncthreads();
    if (L.merged_any == 0) {
      ++it;
      break;
    }
    const int nt = L.ntargets;
    for (int round = 0;; ++round) {
      if (tid == 0) L.again = 0;
      __syncthreads();
      if (round >= 8) {
        if (wave == 0)
          for (int t = 1; t < nrem; ++t)
            if (L.pending[t] == 1) {
              fold_target<ND>(L, t, lane);
              if (lane == 0) L.pending[t] = 0;
            }
        __syncthreads();
        break;
      }
      for (int k = wave; k < nt; k += MS_WAVES) {
        const int t = L.tlist[k];
        if (L.pending[t] != 1) continue;
        bool blocked = false;   // a source that still waits for its own sources
        const int npass = (t + 63) >> 6;
        for (int ps = 0; ps < npass; ++ps) {
          const int pj = ps * 64 + lane;
          const bool b = pj < t && L.mp[pj] == t && L.pending[pj] != 0;
          blocked |= __ballot(b) != 0ull;
        }
        if (blocked) {
          if (lane == 0) L.again = 1;
          continue;
        }
        fold_target<ND>(L, t, lane);
        if (lane == 0) L.pending[t] = 2;   // complete; visible as such from the next round on
      }
      __syncthreads();
      for (int t = tid; t < nrem; t += MS_THREADS)
        if (L.pending[t] == 2) L.pending[t] = 0;
      const bool again = L.again != 0;
      __syncthreads();
      if (!again) break;
    }
    }   // (list-walk path)
    MS_T(4);
    // erase the merged canopies (:145): in-place stable compaction of the position arrays
    if (wave == 0) {
      int base = 0;
      for (int ps = 0; ps * 64 < nrem; ++ps) {
        const int p = ps * 64 + lane;
        bool keep = false;
        float cv[ND];
        int sv = 0, iv = 0;
#pragma unroll
        for (int x = 0; x < ND; ++x) cv[x] = 0.f;
        if (p < nrem) {
          keep = L.mp[p] == p;
#pragma unroll
          for (int x = 0; x < ND; ++x) cv[x] = L.C[x][p];
          sv = L.S[p];
          iv = L.ID[p];
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
          const int dst = base + __popcll(m & ((1ull << lane) - 1ull));   // dst <= p
#pragma unroll
          for (int x = 0; x < ND; ++x) L.C[x][dst] = cv[x];
          L.S[dst] = sv;
          L.ID[dst] = iv;
        }
        base += __popcll(m);
      }
      if (lane == 0) L.nrem = base;
    }
    __syncthreads();
    MS_T(5);
  }
  if (tid == 0 && iters_out) *iters_out = it;

  // ---- emit clusters of size >= MinPts in list order (:151-157) -----------------
  // cluster numbers and offsets by one wavefront; every point's place in its cluster's list
  // by pointer jumping over the linked lists (distance to the list's tail), all threads.
  __syncthreads();
  const int nrem = L.nrem;
  int* const clno = L.mp;        // by position: cluster number or -1
  int* const clbase = L.st;      // by position: first output slot of the cluster
  int* const nx = L.pending;     // by point
  int* const dist = L.flag;      // by point: hops to the tail
  int* const last = L.tlist;     // by point: node reached so far
  int* const owner = reinterpret_cast<int*>(&L.rowbits[0][0]);  // by (tail) point: position of its canopy
  if (wave == 0) {
    int ncl = 0, w = 0;
    for (int ps = 0; ps * 64 < nrem; ++ps) {
      const int p = ps * 64 + lane;
      const int sz = p < nrem ? L.S[p] : 0;
      const bool ok = p < nrem && sz >= min_pts;
      const unsigned long long m = __ballot(ok);
      int incl = ok ? sz : 0;   // inclusive prefix sum of the kept sizes over the lanes
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
      }
      const int k = ncl + __popcll(m & ((1ull << lane) - 1ull));
      const int start = w + incl - (ok ? sz : 0);
      if (p < nrem) {
        clno[p] = ok ? k : -1;
        clbase[p] = start;
        owner[L.tail[L.ID[p]]] = p;
      }
      if (ok && cl_start_out) cl_start_out[k] = start;
      ncl += __popcll(m);
      w += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) {
      if (cl_start_out) cl_start_out[ncl] = w;
      *ncl_out = ncl;
    }
  }
  for (int pt = tid; pt < n; pt += MS_THREADS) {
    const int nn = L.next[pt];
    nx[pt] = nn;
    dist[pt] = nn >= 0 ? 1 : 0;
    last[pt] = pt;
  }
  __syncthreads();
  for (int span = 1; span < n; span <<= 1) {
    int nn[(MS_CAP + MS_THREADS - 1) / MS_THREADS], dd[(MS_CAP + MS_THREADS - 1) / MS_THREADS],
        ll[(MS_CAP + MS_THREADS - 1) / MS_THREADS];
#pragma unroll
    for (int u = 0; u < (MS_CAP + MS_THREADS - 1) / MS_THREADS; ++u) {
      const int pt = tid + u * MS_THREADS;
      nn[u] = -1;
      if (pt < n) {
        const int q = nx[pt];
        if (q >= 0) {
          nn[u] = q;
          dd[u] = dist[q];
          ll[u] = last[q];
          nn[u] = nx[q] >= 0 ? nx[q] : -2;   // -2: jumped onto the tail
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < (MS_CAP + MS_THREADS - 1) / MS_THREADS; ++u) {
      const int pt = tid + u * MS_THREADS;
      if (pt < n && nn[u] != -1) {
        dist[pt] += dd[u];
        last[pt] = ll[u];
        nx[pt] = nn[u] == -2 ? -1 : nn[u];
      }
    }
    __syncthreads();
  }
  for (int pt = tid; pt < n; pt += MS_THREADS) {
    const int p = owner[last[pt]];
    const int k = clno[p];
    if (k >= 0 && members_out) members_out[clbase[p] + L.S[p] - 1 - dist[pt]] = member_base + pt;
    if (label_out) label_out[pt] = k;
  }
  MS_T(6);
}

// Frame form: one workgroup per model, points = uv of the model's matches.  The last
// workgroup to finish lays the per-model cluster lists out as the frame's flat cluster
// table in (model, emission) order -- the order POSE walks `clusters[model]`
// (POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:276-280) -- and publishes the counts.
__global__ __launch_bounds__(MS_THREADS) void meanshift_models_kernel(
    const mh_corr* __restrict__ corr0, const int32_t* __restrict__ model_off0, int n_models, float radius,
    float merge, int min_pts, int max_iter, int32_t* members0, int32_t* cl_start0, int32_t* ncl0,
    int max_clusters, int32_t* __restrict__ cl_model0, int32_t* __restrict__ cl_begin0,
    int32_t* __restrict__ cl_count0, int32_t* __restrict__ n_clusters_out0, int32_t* __restrict__ snap0,
    FrameCounts* counts0, unsigned int* ticket0, int models_div, FrameBatch fbx, int32_t* __restrict__ feedback) {
  MH_TRACE_SCOPE(mh::TK_CLUSTER);
  // models_div > 1: the "models" are (model, image) pairs in (model, image) order -- MeanShift runs per image
  // (CLUSTER_MEAN_SHIFT_CPU.hpp:194-195) -- and the cluster table names the real model
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  MsLds<2>& L = *reinterpret_cast<MsLds<2>*>(smem);
  // A workgroup of this kernel asks for a whole compute unit (its LDS), and one that finds nothing to cluster still has
  // to wait for a unit to drain -- behind the MATCH kernels of the other frames in flight.  So the grid is ONE row of a
  // few workgroups for all frames of the launch (about as many as the launches before found models to cluster): every
  // workgroup walks the frames, lists the models that have at least MinPts matches (a bit per model) and takes those
  // whose number -- counted through the frames -- is its own modulo the grid.  The workgroup that finishes a frame's
  // last model lays the frame's cluster table out (a frame with nothing to cluster: workgroup frame mod grid).
  __shared__ unsigned long long busy[MS_WAVES];
  __shared__ int lay_s[MS_WAVES];
  const int G = (int)gridDim.x;
  const int n_frames = fbx.n > 1 ? fbx.n : 1;
  int rank = 0;   // models with work, counted through the frames of the launch
  for (int f = 0; f < n_frames; ++f) {
    const unsigned long long a = (unsigned long long)f * fbx.arena;
    const mh_corr* corr = frame_ptr(corr0, a);
    const int32_t* model_off = frame_ptr(model_off0, a);
    int32_t* members = frame_ptr(members0, a);
    int32_t* cl_start = frame_ptr(cl_start0, a);
    int32_t* ncl = frame_ptr(ncl0, a);
    FrameCounts* counts = frame_ptr(counts0, a);
    unsigned int* ticket = frame_ptr(ticket0, a);
    int n_busy = 0, mine = 0;
    for (int c0 = 0; c0 < n_models; c0 += MS_THREADS) {   // 1024 models at a time
      {
        const int m = c0 + threadIdx.x;
        int n = 0;
        if (m < n_models) {
          n = model_off[m + 1] - model_off[m];
          if (n <= 0 || n < min_pts) n = 0;   // fewer points than MinPts: no canopy can reach the emission threshold (:151-157)
        }
        const unsigned long long bl = __ballot(n > 0);
        if ((threadIdx.x & 63) == 0) busy[threadIdx.x >> 6] = bl;
      }
      __syncthreads();
      for (int wd = 0; wd < MS_WAVES; ++wd) {
        for (unsigned long long bits = busy[wd]; bits; bits &= bits - 1ull, ++rank, ++n_busy) {
          if (rank % G != (int)blockIdx.x) continue;
          ++mine;
          const int m = c0 + wd * 64 + __builtin_ctzll(bits);
          const int b = model_off[m];
          int n = model_off[m + 1] - b;
          if (n > MS_CAP) {
            if (threadIdx.x == 0) atomicOr(&counts->error, ERR_MS_CAP);
            n = MS_CAP;
          }
          __syncthreads();   // the previous model's LDS is done with
          // cl_start needs n+1 slots inside a region of n: the final offset of the last
          // cluster is implied by the region, so write starts only (see cluster table).
          meanshift_body<2>(L, reinterpret_cast<const float*>(corr + b), sizeof(mh_corr) / sizeof(float), n,
                            radius, merge, min_pts, max_iter, members + b, b, cl_start + b + m, ncl + m,
                            nullptr, nullptr);
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
    const int k0 = layout_cluster_table(n_models, model_off, ncl, cl_start, models_div, max_clusters, frame_ptr(cl_model0, a),
                                        frame_ptr(cl_begin0, a), frame_ptr(cl_count0, a), lay_s,
                                        [min_pts](int nn) { return nn > 0 && nn >= min_pts; });   // fewer points than MinPts: never clustered
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


}  // namespace

#ifdef MS_PROF
extern "C" int mh_debug_ms_stat(unsigned long long out[8], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ms_stat), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ms_stat), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int mh_debug_ms_prof(unsigned long long out[8], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ms_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ms_prof), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif

void launch_meanshift_models(const mh_corr* corr, const int32_t* model_off, int n_models,
                             float radius, float merge, int min_pts, int max_iter, int32_t* members,
                             int32_t* cl_start, int32_t* ncl, int max_clusters, int32_t* cl_model,
                             int32_t* cl_begin, int32_t* cl_count, int32_t* n_clusters_out, int32_t* snap,
                             FrameCounts* counts, unsigned int* ticket, hipStream_t s, int models_div, int grid,
                             const FrameBatch* batch, int32_t* feedback) {
  static DynLds attr;
  attr.ensure(meanshift_models_kernel, MS_LDS_BYTES);
  // one row of workgroups for all frames of the launch; an empty database still gets one: it publishes "0 clusters"
  const long n_frames = batch && batch->n > 1 ? batch->n : 1;
  const long all = std::max(1L, (long)n_models * n_frames);
  const int wgs = (int)std::max(1L, grid > 0 ? std::min((long)grid, all) : std::min(all, 256L));
  hipLaunchKernelGGL(meanshift_models_kernel, dim3(wgs), dim3(MS_THREADS), MS_LDS_BYTES, s,
                     corr, model_off, n_models, radius, merge, min_pts, max_iter, members, cl_start, ncl,
                     max_clusters, cl_model, cl_begin, cl_count, n_clusters_out, snap, counts, ticket,
                     models_div > 0 ? models_div : 1, batch ? *batch : FrameBatch(), feedback);
}

void launch_meanshift_single(const float* pts, int n, int dim, float radius, float merge,
                             int min_pts, int max_iter, int32_t* members, int32_t* cl_start,
                             int32_t* ncl, int32_t* label, int32_t* iters, hipStream_t s) {
  static DynLds attr2, attr3;
  attr2.ensure(meanshift_single_kernel<2>, MS_LDS_BYTES);
  attr3.ensure(meanshift_single_kernel<3>, MS_LDS_BYTES);
  if (dim == 3)
    hipLaunchKernelGGL(meanshift_single_kernel<3>, dim3(1), dim3(MS_THREADS), MS_LDS_BYTES, s,
                       pts, n, radius, merge, min_pts, max_iter, members, cl_start, ncl, label, iters);
  else
    hipLaunchKernelGGL(meanshift_single_kernel<2>, dim3(1), dim3(MS_THREADS), MS_LDS_BYTES, s,
                       pts, n, radius, merge, min_pts, max_iter, members, cl_start, ncl, label, iters);
}

void launch_meanshift_batch(const float* pts, const int32_t* off, int n_problems, int dim, float radius,
                            float merge, int min_pts, int max_iter, int32_t* members, int32_t* cl_start,
                            int32_t* ncl, int32_t* label, hipStream_t s) {
  if (n_problems <= 0) return;
  static DynLds attr2, attr3;
  attr2.ensure(meanshift_batch_kernel<2>, MS_LDS_BYTES);
  attr3.ensure(meanshift_batch_kernel<3>, MS_LDS_BYTES);
  if (dim == 3)
    hipLaunchKernelGGL(meanshift_batch_kernel<3>, dim3(n_problems), dim3(MS_THREADS), MS_LDS_BYTES, s,
                       pts, off, radius, merge, min_pts, max_iter, members, cl_start, ncl, label);
  else
    hipLaunchKernelGGL(meanshift_batch_kernel<2>, dim3(n_problems), dim3(MS_THREADS), MS_LDS_BYTES, s,
                       pts, off, radius, merge, min_pts, max_iter, members, cl_start, ncl, label);
}

}  // namespace mh
