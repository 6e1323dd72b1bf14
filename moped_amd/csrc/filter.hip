// FILTER: FILTER_PROJECTION_CPU::process on gfx950
// (moped2/libmoped/src/filter/FILTER_PROJECTION_CPU.hpp:80-162).
//
//  F1  per object: project every match of its model, in-cluster flag where the
//      squared error < FeatureDistance, score = sum 1/(err+1) accumulated in the
//      reference's order and precision (Float += double, :102-112); then claim
//      each in-cluster keypoint with a 64-bit atomicMax on (score, first object)
//      -- the reference's "point.first < score" sweep in (model, list) order
//      (:117-129) keeps the FIRST object with the maximal score.
//  F2  per object: count the matches of its model whose keypoint it owns (:136-145)
//  F3  one thread: erase objects with too few points or too low a score, keep list
//      order, build the rewritten cluster table (:150-160)
//  F4  per kept object: ordered member list
// One launch: F1 in every workgroup (one wavefront per object), F2..F4 in the last workgroup to finish.  The frame paths
// do not launch this kernel: their FILTER is fused into the POSE launch before it -- every object's F1 by the wavefront
// that refined it (filter_score_wave), F2..F4 by the workgroup that closes the frame (pose.hip, pose_close_frame).
#include <cstdlib>

#include "filter_dev.h"

namespace mh {

namespace {

// The whole step in one launch.  Every workgroup scores objects (F1, grid-stride); the
// last one to finish does F2..F4 for all objects -- they are cheap sweeps over the few
// hundred matches of each object's model -- re-arms the claim table for the next FILTER
// of the frame and, if asked, packs the frame's result block.
__global__ __launch_bounds__(FT) void filter_kernel(FilterBuffers fb, DevCam cam, float feature_distance,
                                                    int min_points, float min_score,
                                                    int32_t* n_slots_dev, int32_t* n_clusters_dev,
                                                    FrameCounts* counts, FilterTail tail) {
  __shared__ FilterLds S;
  const int n_slots = *n_slots_dev;
  filter_score(S, fb, cam, feature_distance, n_slots, blockIdx.x, gridDim.x);
  if (!last_workgroup(tail.ticket)) return;
  filter_finish(S, fb, min_points, min_score, n_slots, n_slots_dev, n_clusters_dev, counts, tail);
}

}  // namespace

void launch_filter(const FilterBuffers& fb, const DevCam& cam, int min_points,
                   float feature_distance, float min_score, int32_t* n_slots_dev,
                   int32_t* n_clusters_dev, FrameCounts* counts, const FilterTail& tail, hipStream_t s) {
  const int grid_cap = tail.grid > 0 ? (tail.grid < FILTER_GRID ? tail.grid : FILTER_GRID) : FILTER_GRID;
  const int grid = fb.max_objects < grid_cap ? (fb.max_objects > 0 ? fb.max_objects : 1) : grid_cap;
  hipLaunchKernelGGL(filter_kernel, dim3(grid), dim3(FT), 0, s, fb, cam, feature_distance, min_points,
                     min_score, n_slots_dev, n_clusters_dev, counts, tail);
}

}  // namespace mh
