// Kernels behind the CLUSTER / POSE / FILTER steps and the device-resident frame.
#pragma once
#include "common.h"

namespace mh {

constexpr int MS_CAP = 2048;         // points per mean-shift problem (LDS resident)
constexpr int POSE_MAX_PTS = 2048;   // correspondences of one cluster cached in LDS

enum : int32_t {
  ERR_MS_CAP = 1,        // a model had more than MS_CAP matches (truncated)
  ERR_CLUSTER_CAP = 2,   // more clusters than reserved
  ERR_OBJECT_CAP = 4,    // more objects than reserved
  ERR_POSE_CAP = 8,      // a cluster had more than POSE_MAX_PTS points (truncated)
  ERR_EXCHANGE = 16,     // the ranks' blocks of a frame exchange carry different tags: collectives issued in different orders
};

// ---- the frames of a batch in ONE launch per stage ------------------------------------
// A dependent launch costs the pipeline about as much whatever is in it (the rest chain of a frame is four of them), so
// the B frames of mh_frame_enqueue_batch go through group / CLUSTER / POSE / POSE2 together: blockIdx.y = frame.  The
// frames' working arrays are copies of one arena at a fixed distance (FrameState), their top-2 arrays and keypoints lie
// frame after frame, their result blocks too.  n = 1 (the default) is a launch for one frame: nothing moves.
struct FrameBatch {
  unsigned long long arena = 0;   // bytes from a frame's working arrays to the next frame's
  int q = 0;                      // queries per frame: distance of the frames' top-2 arrays / keypoints (in elements)
  int result_bytes = 0;           // bytes from a frame's result block to the next
  int n = 1;                      // frames (gridDim.y)
  unsigned long long seed[MH_MAX_BATCH] = {};   // per frame (n > 1)
};
#ifdef __HIPCC__
template <typename T>
__device__ __forceinline__ T* frame_ptr(T* p, unsigned long long bytes) {   // (optional arrays stay nullptr)
  return p ? reinterpret_cast<T*>(reinterpret_cast<unsigned long long>(p) + bytes) : p;
}
#endif

// The depth maps of the frames of a merged batch (mh_frame_set_depth_image_batch): frame f's map and fill-distance map;
// all of the size of the context's DepthImage.  Frame 0's are the DepthImage's own.
struct DepthMaps {
  const float4* img[MH_MAX_BATCH] = {};
  const float* fill[MH_MAX_BATCH] = {};
};

// ---- moped3d depth rules (depth.hip; applied inside group_kernel) --------------------
struct DepthRules {
  // MATCH_ADAPTIVE_FLANN_CPU's ratio: per model (maxRatioDepth, minRatioDepth, ratioLow, ratioHigh)
  const float4* ratio_table = nullptr;   // nullptr = fixed ratio
  float max_depth = 4.f, default_depth = 1.f, cauchy_scale = 0.1f;
  // DEPTHFILTER (features): per query keep flag, nullptr = none
  const uint8_t* keep1 = nullptr;
  // DEPTHFILTER2 (matches of each model): nullptr inv_size = none
  const double* inv_size = nullptr;      // per patch 1.0 / sizeMap
  int32_t* cnt = nullptr;                // [n_models][pw*ph], zero between frames
  int patch = 64, pw = 0, ph = 0;
  float filter2 = 0.f;
};
#ifdef __HIPCC__
// patch of an image coordinate (:187: ((int) location) / PatchSize), clamped to the patch grid
// (the reference indexes out of bounds for coordinates outside the depth map)
__device__ __forceinline__ int patch_of(float u, float v, int patch, int pw, int ph) {
  int px = ((int)u) / patch, py = ((int)v) / patch;
  px = px < 0 ? 0 : (px >= pw ? pw - 1 : px);
  py = py < 0 ? 0 : (py >= ph ? ph - 1 : py);
  return py * pw + px;
}
// dilate (:76-113): the maximum over the patch and its existing 8 neighbours, of the values before dilation
template <typename F>
__device__ __forceinline__ float dilated_by(F&& value_at, int p, int pw, int ph) {
  const int px = p % pw, py = p / pw;
  float best = value_at(p);
  for (int dy = -1; dy <= 1; ++dy) {
    const int yp = py + dy;
    if (yp < 0 || yp >= ph) continue;
    for (int dx = -1; dx <= 1; ++dx) {
      const int xp = px + dx;
      if (xp < 0 || xp >= pw) continue;
      const float v = value_at(yp * pw + xp);
      if (v > best) best = v;
    }
  }
  return best;
}
__device__ __forceinline__ float dilated(const float* val, int p, int pw, int ph) {
  return dilated_by([&](int i) { return val[i]; }, p, pw, ph);
}
// countMap[p] after n features: n times `countMap[p] += 1.0 / sizeMap[p]` on a Float (:189)
__device__ __forceinline__ float density_replay(int n, double inv) {
  float c = 0.f;
  for (int k = 0; k < n; ++k) c = (float)((double)c + inv);
  return c;
}
#endif
// Patch maps of the frame's depth image: inv_size[pw*ph] (pw = ceil(w / patch), ...).
// (maps, n_frames > 1: the frames of a batch in one launch, frame f's patch map at inv_size + f pw ph)
void launch_depth_patches(const DepthImage& dimg, const float K[4], int patch, double* inv_size, hipStream_t s,
                          const DepthMaps* maps = nullptr, int n_frames = 1);
// DEPTHFILTER on the detected features: keep[q] for q < min(Q, *q_count).
void launch_feature_density(const float* q_uv, int Q, const int32_t* q_count, int patch, int pw, int ph,
                            const double* inv_size, float filter, uint8_t* keep, hipStream_t s,
                            int n_frames = 1 /* > 1: q_uv / keep / inv_size frame after frame (Q, Q, pw ph apart) */);

// ---- group -------------------------------------------------------------------
// Ratio test + grouping by model in ascending query order (MATCH_ANN_CPU.hpp:165-176).
// Rows the context's row map does not hold (RowMap, common.h) belong to another shard and are dropped.
// First launch of a frame's CLUSTER..FILTER2 part: also resets *counts and *n_slots, clears
// best[0, n_matches) and -- when `gathered` ([n_shards][3][Q] words, exchange 1) is given --
// first merges the shards' top-2 into idx1/d1/d2.
void launch_group(const int32_t* gathered, int n_shards, int32_t* idx1, float* d1, float* d2, int Q,
                  float ratio, const float* q_uv, const int32_t* db_model, const float* db_xyz, int N,
                  const RowMap& rmap, int n_models, int max_m, int32_t* acc_q, int32_t* acc_model,
                  int32_t* m_q, int32_t* m_model, mh_corr* m_corr, int32_t* m_rep,
                  int32_t* model_off, const mh_depth* q_depth, mh_depth* m_depth, const DepthImage& dimg,
                  FrameCounts* counts, int32_t* n_slots, unsigned long long* best, hipStream_t s,
                  const DepthRules& rules = DepthRules(), int shard_stride = 0 /* words between shard blocks; 0 = 3 Q */,
                  int plane_stride = 0 /* words between the idx / d1 / d2 planes of a block; 0 = Q */,
                  const FrameBatch* batch = nullptr,
                  const int32_t* tags = nullptr /* optional: shard k's two tag words at tags + k * shard_stride (comm.hip); all
                                                   shards must carry the same, else counts->error |= ERR_EXCHANGE */,
                  const DepthMaps* maps = nullptr /* batch with a depth map per frame: frame f reads maps->img[f]; its
                                                     rule buffers (keep1, inv_size, cnt) lie frame after frame */);
void launch_rep(const mh_corr* corr, int M, int32_t* rep, hipStream_t s, const int32_t* img = nullptr);
void launch_accept(const int32_t* idx1, const float* d1, const float* d2, int Q, float ratio,
                   int32_t* out_idx, hipStream_t s);

// ---- mean shift ----------------------------------------------------------------
// CLUSTER of a frame, or of the frames of a batch (`batch`): ONE row of `grid` workgroups shares the models that have
// matches, counted through the frames (0 = one workgroup per model); the one that finishes a frame's last model also
// writes the frame's flat
// cluster table in (model, emission) order, *n_clusters_out, counts->n_clusters and
// snap[0..1] = (matches, clusters).  *ticket: zero-initialised device word (last_workgroup).
void launch_meanshift_models(const mh_corr* corr, const int32_t* model_off, int n_models,
                             float radius, float merge, int min_pts, int max_iter, int32_t* members,
                             int32_t* cl_start, int32_t* ncl, int max_clusters, int32_t* cl_model,
                             int32_t* cl_begin, int32_t* cl_count, int32_t* n_clusters_out, int32_t* snap,
                             FrameCounts* counts, unsigned int* ticket, hipStream_t s, int models_div = 1, int grid = 0,
                             const FrameBatch* batch = nullptr,
                             int32_t* feedback = nullptr /* optional, host-visible: [frame] = models this launch had to cluster */);
// Frames with several images, between group and CLUSTER: m_img[i] = image of match i; m_rep redone with the
// image in the key (FILTER's bestPoints map is keyed by (coord2D, image), FILTER_PROJECTION_CPU.hpp:89); and the
// matches once more in (model, image, query) order -- mi_corr / mi_img, off2[n_models * n_images + 1] -- the point
// sets CLUSTER_MEAN_SHIFT_CPU::process hands to MeanShift image by image (:189-195).
void launch_image_split(const mh_corr* m_corr, const int32_t* m_q, const int32_t* m_model, const int32_t* model_off,
                        int n_models, const int32_t* q_img, int n_images, const FrameCounts* counts, int32_t* m_img,
                        int32_t* m_rep, mh_corr* mi_corr, int32_t* mi_img, int32_t* off2, hipStream_t s);
void launch_meanshift_single(const float* pts, int n, int dim, float radius, float merge,
                             int min_pts, int max_iter, int32_t* members, int32_t* cl_start,
                             int32_t* ncl, int32_t* label, int32_t* iters, hipStream_t s);
// n_problems independent point sets in one launch: points of problem p are rows
// [off[p], off[p+1]) of pts; members/label hold problem-local indices at the same rows,
// cl_start of problem p lives at cl_start + off[p] + p (n_p + 1 entries), ncl[p] = clusters.
void launch_meanshift_batch(const float* pts, const int32_t* off, int n_problems, int dim, float radius,
                            float merge, int min_pts, int max_iter, int32_t* members, int32_t* cl_start,
                            int32_t* ncl, int32_t* label, hipStream_t s);

// ---- linkage (moped3d CLUSTER_LINKAGE_CPU; linkage.hip) -----------------------------------
constexpr int LK_CAP = 1024;   // matches of one model
struct LinkageParams {         // the constructor arguments used (config.hpp:45)
  float cutoff = 0.1f;
  int min_pts = 7;             // clusters need MORE than this many members
  int use3d_filter = 2;        // 0 none, 1 add, 2 multiply the model/world distance-consistency kernel
  float sigma2d = -1.f, sigma3d = -1.f;   // -1: average nearest-neighbour distance
  int linkage_type = 1;        // 0 minimum, 1 average (config.hpp:45), 2 maximum linkage (CLUSTER_LINKAGE_CPU.hpp:506-526)
};
// Floats of scratch the models kernel needs for match lists of the given sizes: 3 n^2 each.
// depth4: per match (wx, wy, wz, weight) aligned with corr.  Outputs as launch_meanshift_models.
void launch_linkage_models(const mh_corr* corr, const float* depth4, const int32_t* model_off, int n_models,
                           const DepthImage& dimg, const LinkageParams& prm, float* scratch, size_t scratch_floats,
                           int32_t* members, int32_t* cl_start, int32_t* ncl, int max_clusters, int32_t* cl_model,
                           int32_t* cl_begin, int32_t* cl_count, int32_t* n_clusters_out, int32_t* snap,
                           FrameCounts* counts, unsigned int* ticket, hipStream_t s, int grid = 0,
                           const FrameBatch* batch = nullptr, const DepthMaps* maps = nullptr /* frame f's depth map */,
                           int32_t* feedback = nullptr /* [frames]: models that had something to cluster */);
void launch_linkage_batch(const mh_corr* corr, const float* depth4, const int32_t* off, int n_problems,
                          const DepthImage& dimg, const LinkageParams& prm, float* scratch, size_t scratch_floats,
                          int32_t* members, int32_t* cl_start, int32_t* ncl, int32_t* label, hipStream_t s);

// ---- pose ----------------------------------------------------------------------
struct DevCam {
  float K[4];
  float Rc[9];  // camera rotation (row-major), TransformMatrix::init of cameraPose
  float tc[3];
};
DevCam make_devcam(const mh_cam& cam);
// Frames with several images: a device table of the cameras and, aligned with the correspondence array a
// kernel works on, the image of every correspondence.  img_of == nullptr: one camera (the DevCam argument).
struct PoseImages {
  const DevCam* cams = nullptr;
  const int32_t* img_of = nullptr;
  int n_images = 1;
};

// Work of the last workgroup of a POSE launch inside a frame (ticket == nullptr: none):
// *n_slots = min(max_objects, *obj_base_dev + n_clusters * R); optionally the number of
// valid objects in [0, *n_slots) -> *snap_valid.
// POSE as two launches (round 4): pose_kernel stops at a task's winning hypothesis and leaves it here, pose_refine_kernel
// refines it -- one wavefront per task.  hyp [frame arena][object slot]; pts / list = per-frame scratch for clusters
// whose points do not fit a wavefront's LDS cache (pts [max_m][9], list [4][max_m]).  hyp == nullptr: one launch.
struct PoseHyp {
  float pose[12];     // R (row major), t of the winner (world frame)
  int32_t n_best;     // its inlier count; 0 = the task has no winner
  int32_t flags;      // 1 = near miss (pose_task)
  int32_t pad[2];
};
struct PoseSplit {
  PoseHyp* hyp = nullptr;
  float* pts = nullptr;
  int32_t* list = nullptr;
  int max_m = 0;
};
struct PoseTail {
  unsigned int* ticket;
  int32_t* n_slots;
  int32_t* snap_valid;
  int grid;   // workgroups to launch (0 = default cap) for ALL frames of the launch; more tasks than that are looped over
  int32_t* feedback = nullptr;   // optional, host-visible: [frame] = (cluster, replica) tasks this launch found (the next grids' guess)
};
struct FilterBuffers;
struct FilterTail;
// The FILTER step that follows a POSE launch in a frame, run by the POSE launch's last workgroup instead of a launch
// of its own (every dependent launch of a frame costs the pipeline ~5% of its throughput, whatever is in it).
struct FilterFuse {
  const FilterBuffers* fb = nullptr;   // host pointers: copied into the kernel's arguments by launch_pose
  const FilterTail* tail = nullptr;
  float feature_distance = 0.f, min_score = 0.f;
  int min_points = 0;
  int32_t* n_clusters_dev = nullptr;
  // where the step's arguments wait on the device (FilterFuseArgs, below) and the host's copy of what is there:
  // launch_pose stores them (a one-thread launch) only when they have changed -- for a context's frames, never again
  struct FilterFuseArgs* dev = nullptr;
  struct FilterFuseArgs* shadow = nullptr;
  bool* shadow_valid = nullptr;
};

// One workgroup per (cluster, replica).  Object slots: obj_base + cluster*R + replica.
// n_clusters_dev: device count (grid is launched for max_clusters).
// depth4 (optional): per-match (wx,wy,wz,cauchyWeight) aligned with corr; depth_kind
// 1 = back-projection residuals, 2 = reprojection+depth residuals (moped3d), 0 = moped2.
void launch_pose(const mh_corr* corr, const float* depth4, int depth_kind, float alpha,
                 const int32_t* members, const int32_t* cl_model,
                 const int32_t* cl_begin, const int32_t* cl_count, const int32_t* n_clusters_dev,
                 int max_clusters, const DevCam& cam, const mh_pose_params& prm, uint64_t seed,
                 const int32_t* obj_base_dev, int max_objects, int32_t* obj_model, float* obj_pose,
                 int32_t* obj_ninl, float* obj_err, int32_t* obj_cluster, int32_t* obj_valid,
                 FrameCounts* counts, const PoseTail& tail, hipStream_t s, const PoseImages& images = PoseImages(),
                 const FilterFuse* fuse = nullptr, const FrameBatch* batch = nullptr, const PoseSplit* split = nullptr);
// pose_kernel<kind>'s registers / LDS / threads / resident workgroups per compute unit / spill bytes (mh_pose_kernel_info)
int pose_kernel_info(int kind, int32_t out[8]);
void launch_project_test(const float* pose7, const mh_corr* corr, int n, const DevCam& cam,
                         float thr, uint8_t* inlier, float* err2, int32_t* n_inliers,
                         hipStream_t s);

// ---- filter ---------------------------------------------------------------------
struct FilterBuffers {
  // inputs
  const mh_corr* corr;        // matches in (model, query) order
  const int32_t* m_rep;       // first match with the same (u,v) [and image]
  const int32_t* m_img;       // image of every match, nullptr = one image
  const DevCam* cams;         // device table of the frame's cameras (m_img != nullptr)
  int n_images;
  const int32_t* model_off;
  int n_models;
  int max_m;
  // object list (slots [0, n_slots), valid flag)
  int32_t* obj_model;
  float* obj_pose;
  float* obj_score;
  float* obj_score_raw;       // score of every slot before the erase/compaction
  int32_t* obj_valid;
  int32_t* obj_npts;
  int max_objects;
  // scratch
  unsigned long long* best;   // [max_m] packed (score bits, ~object)
  int32_t* obj_clsize;        // [2*max_objects]: cluster size, then old slot of each kept object
  // outputs: compacted objects + cluster table 2
  int32_t* new_members;       // [max_objects-bounded] CSR members (sorted match index)
  int32_t* cl_model;
  int32_t* cl_begin;
  int32_t* cl_count;
  int max_clusters;
};
struct FilterTail {
  unsigned int* ticket;     // zero-initialised device word (last_workgroup); required
  int32_t* snap_kept;       // optional: number of kept objects
  unsigned char* result;    // optional: packed result block {int32 n; int32 pad[3]; mh_object[]}
  int grid;                 // workgroups to launch (0 = default cap)
  // optional (with `result`): the frame's head, counters and first objects ALSO into host memory the device can write
  // (FrameHostBlock in page-locked memory, one frame alone) -- mh_frame_fetch then needs a synchronisation and no copy
  struct FrameHostBlock* host = nullptr;
  const int32_t* snap_all = nullptr;   // the frame's four counters (snap[0..3]) for the host block
  unsigned int* host_seq = nullptr;    // with `host`: device word counting the tails that wrote the block (FrameHostBlock::seq)
};
constexpr int FRAME_HOST_OBJECTS = 32;
struct FrameHostBlock {
  int32_t head[4];                       // n objects, error flags, -, -  (as the result block's)
  mh_object objects[FRAME_HOST_OBJECTS];
  int32_t snap[4];
  int32_t error;                         // FrameCounts::error
  // Written by the kernel after everything above: the number of FILTER2 tails that have written this block so far (a
  // device counter).  The host counts its armed enqueues; a block whose seq is not that count is a stale one -- a frame
  // whose launches failed, or were replayed without frame_rest -- and mh_frame_fetch falls back to the copies.
  volatile uint32_t seq;
};
// The fused FILTER step's arguments as the POSE kernel reads them: from device memory, at the one place that needs them
// (the closing workgroup of a frame), instead of ~70 scalar registers' worth of kernel arguments held -- and spilled --
// through the RANSAC code.
struct FilterFuseArgs {
  FilterBuffers fb;
  FilterTail tail;
  float feature_distance, min_score;
  int min_points;
  int32_t* n_clusters_dev;
};

// n_slots_dev: number of object slots in use; after the call the kept objects are
// compacted to slots [0, kept) in list order, *n_slots_dev = kept, and the cluster
// table holds their rewritten clusters.  fb.best[0, n_matches) must be zero on entry and
// is zero again on exit.
void launch_filter(const FilterBuffers& fb, const DevCam& cam, int min_points,
                   float feature_distance, float min_score, int32_t* n_slots_dev,
                   int32_t* n_clusters_dev, FrameCounts* counts, const FilterTail& tail, hipStream_t s);

}  // namespace mh
