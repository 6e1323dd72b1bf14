// Two-stage MATCH (match_screen.hip): what the rest of the library needs to know about it.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mh {

// Per-database part of the screen, built at upload.
struct ScreenDb {
  const _Float16* dbh = nullptr;   // [rows padded to 128][128] f16 image of the (normalised) descriptors
  const float* dneg = nullptr;     // [tiles][192]: -dot(d,d)/2 of the tile's 128 rows (-inf on padding rows), then per 32-row block the largest and the smallest of them
  float dmax = 0.f;                // max_r |d_r| (sqrt of the largest norm term)
  float spread = __builtin_inff(); // largest (max - min) of -dd/2 inside a 32-row block of real rows
  bool usable = false;             // every norm term finite and non-negative, every coordinate inside f16's range
  // the answer for an all-zero query (its distance to row r is the row's norm term, exactly): a zero query ties with
  // every row of a normalised DB on the screen -- its candidate lists would overflow and it would go to brute force
  int32_t zero_idx = -1;
  float zero_d1 = __builtin_inff(), zero_d2 = __builtin_inff();
};

// Per-context scratch of the screen (frames of one context are stream ordered).
struct ScreenBufs {
  _Float16* qh = nullptr;          // [q_pad][128]
  uint8_t* qbad = nullptr;         // [q_pad] queries the screen does not vouch for (brute force in pass C)
  float2* part = nullptr;          // [screen_max_splits_a()][q_pad] pass A's per-split top-2 values
  float* tau = nullptr;            // [q_pad] the queries' thresholds for pass B
  uint2* recs = nullptr;           // [q_pad][screen_rec_slots()] candidate records, all empty between frames
  int32_t* ovf_cnt = nullptr;      // [q_pad] records in the query's overflow list; zero between frames
  uint2* ovf = nullptr;            // [q_pad][ovf_cap]
  int ovf_cap = 0, q_pad = 0;
  hipEvent_t* ev = nullptr;        // optional [6]: recorded around prepare / pass A / thresholds / pass B / pass C (mh_match_timing)
  // optional lane (mh_lane_create): passes A and B -- the kernels whose workgroups need whole compute units -- run on
  // `big` (a stream confined to a CU mask, or of lower priority) between two event hand-offs; the small kernels stay on
  // the context's stream
  hipStream_t big = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  unsigned int* stats = nullptr;   // optional [q_pad][3] per-query tallies: candidate rows, brute-force searches, searches
};

// ---- what a candidate record says about its block (pass B writes it, pass C prunes by it) -------------------------
// Pass B works on the 16 dot products a lane holds of a 32-row block.  `top` = the largest of them, `thr` = tau(q) - the
// block's largest -dd/2: the record carries f16(top - thr) = the block's largest screen value above tau, seen through
// the block's largest norm term (an UPPER bound of the largest screen value of the record's rows).
__host__ __device__ inline unsigned short screen_record_value(float top, float thr) {
  const _Float16 h = (_Float16)(top - thr);
  unsigned short b;
  __builtin_memcpy(&b, &h, 2);
  return b;
}
// Bounds of the largest screen value among the record's rows: hi >= it always; lo <= it when the block is 32 real
// rows (`spread` = the DB's largest in-block spread of -dd/2; a block with padding rows gives no lower bound).  Pass C
// drops a record whose `hi` lies below (second largest `lo` over the query's records) - screen_margin: two records hold
// different rows, so that second largest lower bound is a lower bound of the DB's second largest screen value, and a row
// of the exact top-2 lies at most 2E below THAT (the header of match_screen.hip).  tests/test_screen_bound_cpu.py checks
// both inequalities in host arithmetic (mh_screen_record_bounds).
__host__ __device__ inline void screen_record_bounds(unsigned short value_bits, unsigned row0, float tau_q, float spread, int N,
                                                     float dmax, float& lo, float& hi) {
  _Float16 h;
  __builtin_memcpy(&h, &value_bits, 2);
  const float dv = (float)h;
  // nearest f16: 2^-11 dv (2^-25 below the normals), doubled; + the f32 roundings on the way
  const float eps = fabsf(dv) * 0.001f + 1e-6f + 4e-7f * (fabsf(tau_q) + 0.5f * dmax * dmax);
  const bool fin = fabsf(dv) < 6.0e4f && fabsf(tau_q) < 1e30f;   // inf / nan: no information
  const bool whole = (int)(row0 | 31u) < N;                       // the block's 32 rows are all real rows
  hi = fin ? tau_q + dv + eps : __builtin_inff();
  lo = fin && whole && spread < 1e30f ? tau_q + dv - spread - eps : -__builtin_inff();
}

constexpr int SCREEN_OVF_CAP = 64;   // records in a query's overflow list before the query falls back to brute force
size_t screen_rec_slots();           // record slots per query

float screen_margin_host(float qq, float dmax);   // tau = T - margin (the error model, for tests)
// The screen values themselves, as the matrix pipe computes them (pass A's arithmetic: f16 operands, accumulator seeded
// with -dd/2, the block's MFMAs in ascending k -- eight v_mfma_f32_32x32x16_f16 or four k-steps of v_mfma_f32_16x16x32_f16):
// out[q][r] for q < Q, r < n_rows (both multiples of 32; qh =
// the queries' f16 image).  For tests of the error model against the HARDWARE's accumulation (mh_screen_values).
// shape: 0 = the instruction the large launches use, 1 = v_mfma_f32_32x32x16_f16, 2 = v_mfma_f32_16x16x32_f16
void launch_screen_values(const _Float16* qh, int Q, const ScreenDb& sdb, int n_rows, float* out, hipStream_t s, int shape);
void launch_screen_prepare(const float* qn, const float* qnorm, int Q, int q_pad, _Float16* qh, uint8_t* qbad, hipStream_t s);
size_t screen_db_half_elems(int N);
size_t screen_dneg_elems(int N);   // floats of the -dd/2 array: 192 per 128-row tile (rows, row blocks' extrema)
// f16 image + statistics {bits of max dd, bits of max |x|, non-finite flag, bits of the largest in-block spread,
// row / norm term / second norm term of an all-zero query's answer} (8 words, zeroed by the caller)
void launch_db_to_half(const float* db, const float* dnorm, int N, _Float16* dbh, float* dneg, unsigned int* stats,
                       hipStream_t s);
int screen_q_pad(int Q);
int screen_max_splits_a();
// mode: -1 = decide by size, 0 = never, 1 = whenever the DB has an f16 image (mh_match_set_mode)
bool screen_wanted(int q_expected, int N, int mode = -1);
// Same contract as launch_match (match.hip): exact (idx1, d1, d2) per query.
void launch_match_screen(const float* qn, const float* qnorm, int Q, const float* db, const float* dnorm, int N,
                         const RowMap& rmap, const ScreenDb& sdb, const ScreenBufs& sb, int32_t* idx1, float* d1,
                         float* d2, hipStream_t s, const int32_t* q_count, int q_expected);

}  // namespace mh
