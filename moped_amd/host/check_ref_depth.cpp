// moped3d's STEP plugin headers against moped3d's REAL include/moped.hpp (Image_Type, Image::getDepth / getProb are the
// reference's own; `make check_ref`), util.hpp's part from the mirror (see check_ref.cpp).
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <moped.hpp>
#define MOPED_AMD_WITH_DEPTH 1
#include "moped_util_mirror.hpp"
#include "DEPTH_FILL_EXACT_HIP.hpp"
#include "MATCH_ADAPTIVE_BRUTE_HIP.hpp"
#include "CLUSTER_LINKAGE_HIP.hpp"
#include "POSE_RANSAC_P3P_DEPTH_HIP.hpp"
int main() { return 0; }
