// Syntax check of the moped3d depth step header at the reference's language level.
#include <tr1/memory>
namespace std { using tr1::shared_ptr; }
#define MOPED_AMD_WITH_DEPTH
#include "moped_types.hpp"
#include "DEPTH_FILL_EXACT_HIP.hpp"
#include "MATCH_ADAPTIVE_BRUTE_HIP.hpp"
#include "CLUSTER_LINKAGE_HIP.hpp"
#include "POSE_RANSAC_P3P_DEPTH_HIP.hpp"
int main() { return 0; }
