// Shared declarations of libmoped_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/moped_hip.h"

namespace mh {

constexpr int DIM = MH_DESC_DIM;

// ---- match ------------------------------------------------------------------
// Per-query local result of one DB split / shard.
struct Top2 {
  float d1;     // best squared distance
  float d2;     // second best
  int32_t i1;   // row of the best (index_base applied), -1 = none
  int32_t pad;
};

// A1: normalise rows in place + norm term of the normalised rows.
void launch_normalize(float* desc, float* norm_out, int n, hipStream_t s);
// dot(d,d) chain for every DB row.
void launch_row_norms(const float* desc, float* norm_out, int n, hipStream_t s);
// Scratch (in Top2 units) the match kernel needs for Q queries.
size_t match_scratch_elems(int Q, int N);
// Local top-2 of Q queries vs N rows.
// Floats of packed-query scratch launch_match needs for Q queries.
size_t match_pack_floats(int Q);
void launch_match(const float* qn, const float* qnorm, int Q, const float* db, const float* dnorm,
                  int N, int32_t index_base, Top2* scratch, float* pack, int32_t* idx1, float* d1,
                  float* d2, hipStream_t s);
void launch_match_merge(const int32_t* idx1_s, const float* d1_s, const float* d2_s, int S, int Q,
                        int32_t* idx1, float* d1, float* d2, hipStream_t s);

// ---- group ------------------------------------------------------------------
struct FrameCounts {
  int32_t n_matches;
  int32_t n_clusters;
  int32_t n_objects;     // objects currently in the list
  int32_t n_obj_pose1;   // after POSE
  int32_t n_obj_filter1; // after FILTER
  int32_t n_obj_pose2;   // objects added by POSE2
  int32_t error;         // sticky capacity/overflow flags
  int32_t pad;
};

}  // namespace mh
