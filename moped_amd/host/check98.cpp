// Syntax check of the STEP plugin headers at the reference's language level
// (-std=gnu++98, tr1::shared_ptr as include/moped.hpp:80-81 uses).
#include <tr1/memory>
namespace std { using tr1::shared_ptr; }
#include "moped_types.hpp"
#include "FEAT_SIFT_HIP.hpp"
#include "MATCH_BRUTE_HIP.hpp"
#include "CLUSTER_MEAN_SHIFT_HIP.hpp"
#include "POSE_RANSAC_P3P_HIP.hpp"
#include "FILTER_PROJECTION_HIP.hpp"
#include "FRAME_RESIDENT_HIP.hpp"
int main() { return 0; }
