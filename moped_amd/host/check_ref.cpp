// The STEP plugin headers compiled against the reference's REAL include/moped.hpp (Pt, Pose, Model, Image, Object,
// TransformMatrix, toString / fromString are the reference's own; `make check_ref REF=/root/reference/moped2/libmoped`),
// at the reference's language level.  Only src/util.hpp's part (FrameData, MopedAlg, the pipeline) comes from the mirror:
// util.hpp includes OpenCV (util.hpp:51-52), which this image lacks.  Skipped where the reference is absent.
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <moped.hpp>
#include "moped_util_mirror.hpp"
#include "FEAT_SIFT_HIP.hpp"
#include "MATCH_BRUTE_HIP.hpp"
#include "CLUSTER_MEAN_SHIFT_HIP.hpp"
#include "POSE_RANSAC_P3P_HIP.hpp"
#include "FILTER_PROJECTION_HIP.hpp"
#include "FRAME_RESIDENT_HIP.hpp"

// what the plugins and the C ABI assume about the real types' layout (mh_object / mh_cam are packed from these)
typedef char pt2_is_two_floats[sizeof(MopedNS::Pt<2>) == 2 * sizeof(float) ? 1 : -1];
typedef char pt3_is_three_floats[sizeof(MopedNS::Pt<3>) == 3 * sizeof(float) ? 1 : -1];
typedef char pose_is_seven_floats[sizeof(MopedNS::Pose) == 7 * sizeof(float) ? 1 : -1];

int main() {
  // the steps construct with the reference's argument lists (config.hpp:83-120) and sit in a pipeline
  MopedNS::MopedPipeline pipeline;
  (void)pipeline;
  MopedNS::Pose p;
  p.rotation.init(0.f, 0.f, 0.f, 1.f);   // (x, y, z, w), include/moped.hpp:136-164
  p.translation.init(0.f, 0.f, 0.5f);
  return p.rotation[3] == 1.f ? 0 : 1;
}
