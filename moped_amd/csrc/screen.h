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
  unsigned int* stats = nullptr;   // optional [q_pad][3] per-query tallies: candidate rows, brute-force searches, searches
};

constexpr int SCREEN_OVF_CAP = 64;   // records in a query's overflow list before the query falls back to brute force
size_t screen_rec_slots();           // record slots per query

float screen_margin_host(float qq, float dmax);   // tau = T - margin (the error model, for tests)
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
                         int32_t index_base, const ScreenDb& sdb, const ScreenBufs& sb, int32_t* idx1, float* d1,
                         float* d2, hipStream_t s, const int32_t* q_count, int q_expected);

}  // namespace mh
