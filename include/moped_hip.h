/* include/moped_hip.h -- C ABI of libmoped_hip.so
 *
 * MI355X (gfx950) implementation of libmoped's per-frame hot path
 *   MATCH (brute-force 2-NN + ratio) -> CLUSTER (mean shift) -> POSE (RANSAC + LM)
 *   [-> FILTER -> POSE2 -> FILTER2]
 * behind plain C entry points: opaque context, plain pointers and sizes, int
 * status (0 = ok, < 0 = error, text via mh_last_error), never aborts, never
 * throws.  One context per pipeline; calls on one context are serialised by the
 * caller (libmoped calls its steps from one thread, src/moped.cpp:184-191).
 *
 * What each entry point replaces (paths relative to moped2/libmoped/):
 *   mh_db_upload        MATCH_ANN_CPU::Update            src/match/MATCH_ANN_CPU.hpp:72-109
 *   mh_normalize        MATCH_ANN_CPU::norm              src/match/MATCH_ANN_CPU.hpp:54-57
 *   mh_match            MATCH_ANN_CPU::process search    src/match/MATCH_ANN_CPU.hpp:155-165
 *                       (= MATCH_FLANN_CPU::process      src/match/MATCH_FLANN_CPU.hpp:133-191)
 *   mh_match_local /    the same search on one model shard + the cross-shard
 *   mh_match_merge      top-2 merge (new: the reference has one kd-tree over all
 *                       models, MATCH_ANN_CPU.hpp:76-107)
 *   mh_meanshift        CLUSTER_MEAN_SHIFT_CPU::MeanShift src/cluster/CLUSTER_MEAN_SHIFT_CPU.hpp:80-158
 *   mh_pose_ransac      POSE_RANSAC_LM_DIFF_REPROJECTION_CPU::RANSAC/process
 *                                                        src/pose/POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:76-211,264-307
 *   mh_pose_ransac_depth  POSE_RANSAC_LM_DIFF_BACKPROJECTION_DEPTH_CPU / ..._REPROJECTION_DEPTH_CPU
 *                       moped3d/libmoped/src/pose/POSE_RANSAC_LM_DIFF_BACKPROJECTION_DEPTH_CPU.hpp:108-325,
 *                       moped3d/libmoped/src/pose/POSE_RANSAC_LM_DIFF_REPROJECTION_DEPTH_CPU.hpp:106-216
 *   mh_project_test     testAllPoints / project()        …REPROJECTION_CPU.hpp:166-180, include/moped.hpp:330-354
 *   mh_filter           FILTER_PROJECTION_CPU::process   src/filter/FILTER_PROJECTION_CPU.hpp:80-162
 *   mh_frame_*          the per-frame loop over those steps, MopedPimpl::processImages
 *                                                        src/moped.cpp:166-194 (device-resident form)
 *
 * Pointer arguments named *_host are host memory; *_dev are device (HBM)
 * pointers on the context's device.  Poses are (qx,qy,qz,qw,tx,ty,tz) as in
 * MopedNS::Pose (include/moped.hpp:136-164); intrinsics K = (fx,fy,cx,cy) as in
 * Image::intrinsicLinearCalibration (include/moped.hpp:233).
 */
#ifndef MOPED_HIP_H
#define MOPED_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_DESC_DIM 128
#define MH_MAX_BATCH 32  /* frames one context can carry through one MATCH launch (mh_frame_enqueue_batch / _sharded_batch) */
#define MH_MAX_IMAGES 8    /* images (cameras) of one frame (mh_frame_set_images, mh_*_images) */
#define MH_MAX_MODELS 8192   /* models a context's frames can address (sharded: the global count); more -> MH_ERR_CAPACITY */
#define MH_OK 0
#define MH_ERR_ARG (-1)
#define MH_ERR_HIP (-2)
#define MH_ERR_CAPACITY (-3)
#define MH_ERR_NODEVICE (-4)

typedef struct mh_ctx mh_ctx;

/* ---- context ------------------------------------------------------------ */

/* Selects `device`, checks it is a gfx950 part, creates the context's stream.
 * Failure is the signal for a STEP plugin to set capable = false
 * (src/util.hpp:151-159). */
int mh_create(int device, mh_ctx** out);
void mh_destroy(mh_ctx* ctx);
const char* mh_last_error(const mh_ctx* ctx);
/* Run subsequent work on a caller-owned hipStream_t (e.g. the stream a
 * framework uses); NULL restores the context's own stream. */
int mh_set_stream(mh_ctx* ctx, void* hip_stream);
int mh_synchronize(mh_ctx* ctx);
/* Capacities of the per-frame device buffers (defaults: 16384 queries, 16384
 * matches, 1024 clusters, 4096 objects). Call before the first frame. */
int mh_reserve(mh_ctx* ctx, int max_queries, int max_clusters, int max_objects);
/* The same for a context that will carry BATCHES: `frames` <= MH_MAX_BATCH frames of up to `queries_per_frame` queries
 * each per mh_frame_enqueue_batch / _sharded_batch / _rest_frames call.  The MATCH buffers are sized for all
 * frames x queries_per_frame queries of a launch, the working arrays of CLUSTER..FILTER2 once per frame for
 * queries_per_frame matches each (mh_reserve would size every one of them for the whole launch), so that no enqueue
 * ever has to stop the stream and reallocate. */
int mh_reserve_batch(mh_ctx* ctx, int queries_per_frame, int frames, int max_clusters, int max_objects);

/* ---- model database (A2) -------------------------------------------------- */

/* desc_host: N x 128 row-major, ALREADY L2-normalised by the caller (the
 * reference normalises model descriptors in place in Update(), :94);
 * model_of_host[N]: model index of each row, 0..n_models-1 (a shard passes global model
 * ids and the global n_models); a value outside that range -> MH_ERR_ARG;
 * xyz_host: N x 3 model coordinates.  index_base: value added to row numbers in
 * every index this context reports (global row id of row 0 when the DB is a
 * shard).  Re-upload after modelsUpdated(). */
int mh_db_upload(mh_ctx* ctx, const float* desc_host, const int32_t* model_of_host,
                 const float* xyz_host, int N, int n_models, int32_t index_base);
/* The same for a shard whose rows are SEVERAL runs of global rows -- a rank that owns models m, m + W, m + 2 W, ...
 * of the flattened database (round-robin assignment: SURVEY 8(e) "interleave model -> GPU to balance POSE"; the
 * one-tree order it must preserve is MATCH_ANN_CPU::Update's, src/match/MATCH_ANN_CPU.hpp:76-107).  The N rows are
 * the blocks one after the other; block b holds global rows [block_global_row[b], + block_rows[b]); blocks in
 * ascending global order, so a tie between two rows breaks the same way as in the unsharded database.  Every index
 * the context reports is a global row.  normalize != 0: L2-normalise on the device (A1). */
int mh_db_upload_blocks(mh_ctx* ctx, const float* desc_host, const int32_t* model_of_host, const float* xyz_host, int N,
                        int n_models, const int32_t* block_global_row, const int32_t* block_rows, int n_blocks,
                        int normalize);
int mh_db_size(const mh_ctx* ctx, int* N, int* n_models);
/* Let `dst` use the database `src` holds: same device, no copy -- one upload and one HBM copy per GPU
 * however many contexts (frames in flight) work against it.  The analogue of the ONE kd-tree every
 * frame of the reference searches (MATCH_ANN_CPU.hpp:70,106).  The store lives as long as its last
 * user; a later mh_db_upload into either context gives that context a new store of its own and leaves
 * the other one's untouched.  Frames already enqueued on `dst` finish on its old database first. */
int mh_db_share(mh_ctx* dst, mh_ctx* src);

/* ---- MATCH ---------------------------------------------------------------- */

/* A1: in-place L2 normalisation of n descriptors on the device, bit-identical
 * to MATCH_ANN_CPU::norm.  Host round trip. */
int mh_normalize(mh_ctx* ctx, float* desc_host, int n);

/* A3: exact 2-NN of Q normalised queries against the uploaded DB, squared L2.
 * nn_idx[i] = row (index_base applied) of the nearest descriptor if
 * d1/d2 < ratio, else -1; nn_raw (optional) = the nearest row regardless of the
 * ratio test; d1/d2 (optional) = best / second-best squared distance. */
int mh_match(mh_ctx* ctx, const float* q_host, int Q, float ratio,
             int32_t* nn_idx, int32_t* nn_raw, float* d1, float* d2);
/* mh_normalize + mh_match in one call, as MATCH_ANN_CPU::process does both to a frame's descriptors
 * (moped2/libmoped/src/match/MATCH_ANN_CPU.hpp:155-165: norm() in place, then the search): q_host [Q][128] goes up
 * raw, is normalised on the device and comes back normalised; results as mh_match.  One upload and one
 * synchronisation instead of two of each (what MATCH_BRUTE_HIP calls). */
int mh_normalize_match(mh_ctx* ctx, float* q_host, int Q, float ratio, int32_t* nn_idx, int32_t* nn_raw, float* d1,
                       float* d2);

/* How MATCH ran on this context since the last reset: stats[0] = candidate rows the exact stage evaluated,
 * stats[1] = queries searched by brute force inside it (candidate list overflow, or a query the f16
 * screen cannot represent), stats[2] = queries through the two-stage path, stats[3] = 1 if a MATCH of
 * `Q` queries against the current DB takes the two-stage path (f16 screen + exact rescoring), 0 if it
 * takes the exact kernels.  Synchronises the context's stream. */
int mh_match_stats(mh_ctx* ctx, int Q, uint32_t stats[4], int reset);
/* Which kernels search: -1 (default) = the two-stage path when the work is large enough (Q x N >= 8e6,
 * N >= 4096) and every DB row is finite and inside f16's range, 0 = always the exact f32 kernels (the
 * VALU kernel below 1536 queries, the f32 matrix-pipe kernel from there), 1 = the two-stage path whenever
 * the DB allows, 2 = the exact VALU kernel (match_kernel) whatever the query count, 3 = the exact f32
 * matrix-pipe kernel (match_mfma_kernel) whatever the query count.  The results are the same bits in every
 * mode (tests/test_gpu_match_kernels.py compares them). */
int mh_match_set_mode(mh_ctx* ctx, int mode);
/* POSE / POSE2 of the device-resident frame paths as two launches (1, the default: the hypotheses of every (cluster,
 * replica) task, then the refines on one wavefront each) or as one (0: the workgroup that found a task's winner refines it
 * on one of its four wavefronts).  The objects are the same bits either way (tests/test_gpu_frame.py); the per-step entry
 * points (mh_pose_ransac*) and frames with stage timing always run the one-launch form. */
int mh_pose_set_split(mh_ctx* ctx, int on);
/* MATCH launch sequences this context has issued, by the kernel that searched: out[0] = match_kernel (VALU),
 * out[1] = match_mfma_kernel (f32 matrix pipe), out[2] = the two-stage path.  Host counters, no synchronisation;
 * for tests that must know which kernel produced a result. */
int mh_match_launches(mh_ctx* ctx, uint32_t out[3]);
/* The two-stage path's error model (host arithmetic, no device needed): a row can be one of a query's
 * two nearest only if its f16 screen value exceeds T - mh_screen_margin(dot(q,q), max row norm), T =
 * any lower bound of the second largest screen value.  Exposed so that tests can check the bound. */
float mh_screen_margin(float qq, float dmax);
/* More of the same, for tests (none of these is on a frame's path):
 *   mh_screen_values        the screen value of every (query, row) pair for Q queries (a multiple of 32; host, [Q][128],
 *                           as the caller normalised them) against the first n_rows rows (a multiple of 32) of the
 *                           uploaded DB, computed on the device by pass A's arithmetic -- f16 operands, accumulator
 *                           seeded with -dot(d,d)/2, the block's f16 MFMAs in ascending k (either shape) -- so that the error
 *                           model can be held against what the HARDWARE's matrix pipe accumulates, not an emulation of
 *                           it; *dmax / *spread (optional) = the DB statistics the thresholds use.  The DB needs an
 *                           f16 image (>= 4096 rows).
 *   mh_screen_record_value  the 16-bit value a candidate record of pass B carries for a block whose largest dot
 *                           product is `top` under the threshold `thr` = tau - the block's largest -dd/2
 *   mh_screen_record_bounds what pass C concludes from it: *hi >= the largest screen value among the record's rows,
 *                           *lo <= it (or -inf); it drops the record when hi < (second largest lo) - mh_screen_margin.
 *                           Host arithmetic, the same inline function the kernel runs. */
int mh_screen_values(mh_ctx* ctx, const float* q_host, int Q, int n_rows, float* out_host, float* dmax, float* spread,
                     int shape /* 0 = the MFMA shape the large launches use, 1 = 32x32x16, 2 = 16x16x32 */);
uint16_t mh_screen_record_value(float top, float thr);
void mh_screen_record_bounds(uint16_t value_bits, uint32_t row0, float tau, float spread, int N, float dmax, float* lo,
                             float* hi);

/* Device-pointer forms for a model-sharded DB: local top-2 of this shard
 * (idx carries index_base; -1 when the shard is empty), then the merge of S
 * shards' results laid out [S][Q] (e.g. the output of an all-gather). All
 * pointers are device memory; work is enqueued on the context's stream. */
int mh_match_local_dev(mh_ctx* ctx, const float* qn_dev, const float* qnorm_dev, int Q,
                       int32_t* idx1_dev, float* d1_dev, float* d2_dev);
int mh_match_merge_dev(mh_ctx* ctx, const int32_t* idx1_s_dev, const float* d1_s_dev,
                       const float* d2_s_dev, int n_shards, int Q,
                       int32_t* idx1_dev, float* d1_dev, float* d2_dev);
/* Normalise on device: q_dev [Q x 128] in place, qnorm_dev[Q] = dot(q,q) of the
 * normalised rows (the norm term of the distance). */
int mh_normalize_dev(mh_ctx* ctx, float* q_dev, float* qnorm_dev, int Q);

/* ---- CLUSTER (A6) ----------------------------------------------------------- */

/* Mean shift of n points (dim 2 or 3) exactly as MeanShift<T,N>: label[i] =
 * cluster number in the reference's emission order, -1 for points whose canopy
 * ended below min_pts; order[] (optional, n entries) = point indices grouped by
 * cluster, in the reference's within-cluster (splice) order; n_clusters out. */
int mh_meanshift(mh_ctx* ctx, const float* pts_host, int n, int dim, float radius,
                 float merge, int min_pts, int max_iter, int32_t* label,
                 int32_t* order, int32_t* n_clusters);
/* The same for n_problems independent point sets in one call -- the per-model loop of
 * CLUSTER_MEAN_SHIFT_CPU::process (src/cluster/CLUSTER_MEAN_SHIFT_CPU.hpp:182-199): one
 * upload, one launch (a workgroup per problem), one download.  pts_host = the problems'
 * points concatenated, off[n_problems+1] = first point of each problem (off[0] = 0);
 * label / order (optional) are laid out like pts and hold problem-local indices;
 * n_clusters[n_problems]. */
int mh_meanshift_batch(mh_ctx* ctx, const float* pts_host, const int32_t* off, int n_problems,
                       int dim, float radius, float merge, int min_pts, int max_iter,
                       int32_t* label, int32_t* order, int32_t* n_clusters);

/* ---- POSE (A8-A13) ---------------------------------------------------------- */

typedef struct {
  float u, v;    /* image coordinates of the matched keypoint */
  float x, y, z; /* model coordinates of the matched point */
} mh_corr;

typedef struct {
  float K[4];    /* fx, fy, cx, cy */
  float cam[7];  /* camera pose (qx,qy,qz,qw,tx,ty,tz) */
} mh_cam;

typedef struct {
  int n_hypotheses;        /* P3P hypotheses per (cluster, replica); 0 -> 1024 */
  int max_objects_per_cluster; /* replicas per cluster (reference: 4) */
  int n_pts_align;         /* distinct 2-D points a cluster needs (5 POSE / 6 POSE2) */
  int min_n_pts_object;    /* a hypothesis needs MORE inliers than this (6 / 8) */
  float error_threshold;   /* squared pixel error of an inlier (10 / 5) */
  int lm_iters_l2;         /* LM iterations on plain residuals (default 2: a warm start; with depth residuals,
                              depth_kind 1 / 2, values 1..9 are raised to 10; 0 = no plain phase) */
  int lm_iters_l4;         /* then on the reference's squared residuals */
} mh_pose_params;

typedef struct {
  float pose[7];
  int32_t cluster;     /* input cluster this object came from */
  int32_t n_inliers;   /* inliers of the winning hypothesis */
  float err;           /* final sum of squared residuals of the refine */
} mh_pose_out;

/* RANSAC on n_clusters clusters: corr_host[cluster_off[c] .. cluster_off[c+1]).
 * out_host has capacity n_clusters * max_objects_per_cluster; *n_out = objects
 * found (cluster-major, replica order).  Deterministic for a given seed. */
int mh_pose_ransac(mh_ctx* ctx, const mh_corr* corr_host, const int32_t* cluster_off,
                   int n_clusters, const mh_cam* cam, const mh_pose_params* prm,
                   uint64_t seed, mh_pose_out* out_host, int32_t* n_out);
/* The same with every correspondence in its own image (LmData::image, …REPROJECTION_CPU.hpp:213-237; POSE2's
 * clusters, rewritten by FILTER, mix images): image_of_host[i] in [0, n_images), cams[n_images].  A minimal
 * sample is drawn inside one image; inliers and the refinement use every point's own camera; two
 * correspondences with the same (image, coord2D) are never sampled together (:76-98). */
int mh_pose_ransac_images(mh_ctx* ctx, const mh_corr* corr_host, const int32_t* image_of_host,
                          const int32_t* cluster_off, int n_clusters, const mh_cam* cams, int n_images,
                          const mh_pose_params* prm, uint64_t seed, mh_pose_out* out_host, int32_t* n_out);

/* moped3d (Kinect) variants: every correspondence also carries the camera-frame point
 * read from the depth map and its Cauchy weight (Match.depthData.coord3D and
 * getCauchyWeight(fillDistance), moped3d/libmoped/src/util.hpp:73-84,
 * ...BACKPROJECTION_DEPTH_CPU.hpp:194-197). */
typedef struct {
  float wx, wy, wz; /* world3D: camera-frame xyz of the keypoint from the depth map */
  float w;          /* cauchyWeight */
} mh_depth;

#define MH_DEPTH_NONE 0          /* moped2 residuals (u,v only) */
#define MH_DEPTH_BACKPROJECTION 1 /* POSE_RANSAC_LM_DIFF_BACKPROJECTION_DEPTH_CPU (moped3d default) */
#define MH_DEPTH_REPROJECTION 2   /* POSE_RANSAC_LM_DIFF_REPROJECTION_DEPTH_CPU */

/* As mh_pose_ransac, with depth_host[i] beside corr_host[i]; kind = MH_DEPTH_*, alpha =
 * the class's Alpha (0.5 in moped3d/libmoped/src/config.hpp:46,48).  Hypotheses and the
 * inlier test are the 2-D ones (the reference's testAllPoints is unchanged); the refine
 * minimises the depth-aware residuals of the chosen class. */
int mh_pose_ransac_depth(mh_ctx* ctx, const mh_corr* corr_host, const mh_depth* depth_host,
                         const int32_t* cluster_off, int n_clusters, const mh_cam* cam,
                         const mh_pose_params* prm, int kind, float alpha, uint64_t seed,
                         mh_pose_out* out_host, int32_t* n_out);

/* testAllPoints: inlier_host[i] = squared reprojection error < thr; err2_host
 * (optional) the squared error (FLT_MAX-based for z < 0.001 like project()). */
int mh_project_test(mh_ctx* ctx, const float pose[7], const mh_corr* corr_host, int n,
                    const mh_cam* cam, float thr, uint8_t* inlier_host, float* err2_host,
                    int32_t* n_inliers);

/* ---- FILTER (N1) ------------------------------------------------------------- */

/* FILTER_PROJECTION_CPU::process for one image.  corr_host: all matches in
 * (model, query) order, model_off[n_models+1]; objects (model, pose) in list
 * order.  score[n_obj], keep[n_obj]; the rewritten clusters of kept objects in
 * (model, list) order: out_order[kept] object indices, cl_off[kept+1] /
 * cl_members (match index inside its model). */
/* The same for a frame whose matches come from n_images images (FILTER_PROJECTION_CPU.hpp:100-141: every match
 * is projected through *images[match.imageIdx], the ownership map is keyed by (coord2D, image)):
 * image_of_host[i] = image of match i (same order as corr_host), cams[n_images].  n_images == 1: mh_filter. */
int mh_filter_images(mh_ctx* ctx, const mh_corr* corr_host, const int32_t* image_of_host, const int32_t* model_off,
                     int n_models, const int32_t* obj_model, const float* obj_pose, int n_obj, const mh_cam* cams,
                     int n_images, int min_points, float feature_distance, float min_score, float* score,
                     uint8_t* keep, int32_t* out_order, int32_t* cl_members, int32_t* cl_off, int32_t* n_kept);
int mh_filter(mh_ctx* ctx, const mh_corr* corr_host, const int32_t* model_off, int n_models,
              const int32_t* obj_model, const float* obj_pose, int n_obj, const mh_cam* cam,
              int min_points, float feature_distance, float min_score,
              float* score, uint8_t* keep, int32_t* out_order, int32_t* cl_members,
              int32_t* cl_off, int32_t* n_kept);

/* ---- whole frame, device resident --------------------------------------------- */

typedef struct {
  float ratio;               /* 0.8  (config.hpp:83) */
  float ms_radius, ms_merge; /* 200, 20 (config.hpp:101) */
  int ms_min_pts, ms_max_iter; /* 7, 100 */
  mh_pose_params pose1;      /* (.., 4, 5, 6, 10)  config.hpp:110 */
  int f1_min_points; float f1_feature_distance, f1_min_score; /* 5, 4096, 2  config.hpp:115 */
  mh_pose_params pose2;      /* (.., 4, 6, 8, 5)   config.hpp:118 */
  int f2_min_points; float f2_feature_distance, f2_min_score; /* 7, 4096, 3  config.hpp:120 */
  int run_stage2;            /* 0: stop after POSE (objects unscored) */
} mh_frame_params;

void mh_frame_default_params(mh_frame_params* p);

typedef struct {
  int32_t model;   /* model index (local to the context) */
  float pose[7];
  float score;
  int32_t n_points; /* size of the object's final cluster */
} mh_object;

/* One frame from device-resident inputs: q_desc_dev [Q x 128] raw descriptors
 * (normalised in place, as the reference mutates detectedFeatures), q_uv_dev
 * [Q x 2].  Everything is enqueued on the context's stream; no host
 * synchronisation.  Results stay on the device until mh_frame_fetch. */
int mh_frame_enqueue(mh_ctx* ctx, float* q_desc_dev, const float* q_uv_dev, int Q,
                     const mh_cam* cam, const mh_frame_params* prm, uint64_t seed);
/* B <= MH_MAX_BATCH frames at once: their descriptors / keypoints one after the other ([B Q][128], [B Q][2]); ONE
 * MATCH launch sequence over all B Q queries (the DB passes the chip once per batch), then CLUSTER .. FILTER2 frame
 * by frame on the context's stream; frame f leaves its objects in result slot f (mh_frame_fetch_slot), the same
 * objects, bit for bit, as mh_frame_enqueue(…, seeds[f]) gives it alone.  Per-query depth attributes
 * (mh_frame_set_depth) are then [B Q] like the queries, and so is the per-query image index of frames with several
 * images (mh_frame_set_images: q_image_dev [B Q]; one camera table for the batch); depth maps (and the depth rules that
 * read them) come one per frame through mh_frame_set_depth_image_batch. */
/* The same frame from HOST memory, objects back on return: the body of the frame loop of
 * MopedPimpl::processImages (src/moped.cpp:183-191 -- MATCH_SIFT, CLUSTER, POSE, FILTER, POSE2, FILTER2 on one
 * FrameData) as one call.  q_desc_host [Q][128] raw descriptors (written back L2-normalised if write_back, as
 * MATCH_ANN_CPU.hpp:157 leaves them), q_uv_host [Q][2], q_image_host [Q] or NULL (FrameData::DetectedFeature::imageIdx,
 * renumbered into cams[]; needed when n_images > 1), cams[n_images] (1 <= n_images <= MH_MAX_IMAGES).  Results as
 * mh_frame_fetch.  What the STEP plugins move over PCIe between the steps (matches, clusters, objects, twice) stays on
 * the device; FRAME_RESIDENT_HIP (moped_amd/host) is the MopedAlg that calls this. */
int mh_frame_run_host(mh_ctx* ctx, float* q_desc_host, const float* q_uv_host, const int32_t* q_image_host, int Q,
                      const mh_cam* cams, int n_images, const mh_frame_params* prm, uint64_t seed, int write_back,
                      mh_object* objects_host, int max_objects, int32_t* n_objects, int32_t* counts);
/* The same in two halves, for a host that has work of its own to do while the frame runs (FRAME_RESIDENT_HIP copies the
 * normalised descriptors back into FrameData::detectedFeatures meanwhile): mh_frame_run_host_begin uploads and enqueues
 * the frame and returns; mh_frame_wait_descriptors returns when q_desc_host holds the normalised descriptors (write_back
 * was set; they are through right after the frame's first kernel); mh_frame_fetch ends the frame as usual.  The host
 * buffers must stay untouched by the caller until mh_frame_wait_descriptors (q_desc_host) / mh_frame_fetch (the others). */
int mh_frame_run_host_begin(mh_ctx* ctx, float* q_desc_host, const float* q_uv_host, const int32_t* q_image_host, int Q,
                            const mh_cam* cams, int n_images, const mh_frame_params* prm, uint64_t seed, int write_back);
int mh_frame_wait_descriptors(mh_ctx* ctx);
/* Page-locked host memory for the buffers a host hands to mh_frame_run_host (and to the other host-pointer entry points):
 * from pageable memory the frame's 1.5 MB of descriptors cross PCIe through the driver's staging copies (~0.1 ms each
 * way), from here at the link's rate and asynchronously.  FRAME_RESIDENT_HIP packs FrameData::detectedFeatures
 * (src/util.hpp:70-79) straight into such a block.  mh_host_free(NULL) is a no-op. */
int mh_host_alloc(mh_ctx* ctx, size_t bytes, void** out);
int mh_host_free(mh_ctx* ctx, void* p);

/* ---- the six slots one call each, the frame resident between them (the per-step plugins' hand-over) ----------------
 * The reference runs its steps strictly one after the other on one FrameData (src/moped.cpp:183-191, the step list of
 * src/config.hpp:83-120); every step's contract is what it leaves in FrameData for the next (src/util.hpp:68-110).  These
 * entry points keep that contract -- each returns what its slot writes into FrameData -- but the step that follows does
 * not upload it again: the frame's match lists, clusters and objects stay in the context's working arrays and each
 * call launches only its own slot's kernels behind the previous call's.  Order: mh_step_match, mh_step_match_fetch,
 * mh_step_cluster, mh_step_pose(1), mh_step_filter(1), mh_step_pose(2), mh_step_filter(2).  A call out of that order, or
 * after any other call that uses the context's frame arrays (mh_frame_*, mh_filter*, mh_pose_ransac*), is refused with
 * MH_ERR_ARG -- the plugin then takes the upload path of its slot (mh_normalize_match / mh_meanshift_batch /
 * mh_pose_ransac_images / mh_filter_images), which is always valid.  One camera per frame.  Objects are bit for bit
 * those of mh_frame_run_host with the same constants and seeds (POSE: seed, POSE2: seed ^ 0x5DEECE66D there).
 *
 * mh_step_match: MATCH_ANN_CPU::process (src/match/MATCH_ANN_CPU.hpp:136-178) -- uploads the frame's features
 * (q_desc_host [Q][128] raw, q_uv_host [Q][2]; page-locked memory from mh_host_alloc crosses PCIe asynchronously),
 * normalises (:157), searches, applies the ratio test and leaves matches[model] (:165-176) on the device; returns
 * without waiting.  write_back: the normalised descriptors come back into q_desc_host on a stream of their own
 * (mh_frame_wait_descriptors).  mh_step_match_fetch waits and returns the lists: model_off_host[n_models + 1],
 * match (model_off[m] + k) = k-th element of matches[m] = {query index, {u, v, x, y, z}}; cap >= Q always suffices. */
int mh_step_match(mh_ctx* ctx, float* q_desc_host, const float* q_uv_host, int Q, const mh_cam* cam, float ratio,
                  int write_back);
int mh_step_match_fetch(mh_ctx* ctx, int32_t* model_off_host, int32_t* match_query, mh_corr* match_pts, int cap,
                        int32_t* n_matches);
/* CLUSTER_MEAN_SHIFT_CPU::process (src/cluster/CLUSTER_MEAN_SHIFT_CPU.hpp:182-199) on the resident lists: cluster c
 * belongs to model cl_model_host[c] and holds members_host[cl_off_host[c] .. cl_off_host[c + 1]) = indices into
 * matches[model], in the reference's emission and member order; clusters in (model, emission) order. */
int mh_step_cluster(mh_ctx* ctx, float radius, float merge, int min_pts, int max_iter, int32_t* cl_model_host,
                    int32_t* cl_off_host /* cap_clusters + 1 */, int32_t* members_host, int cap_clusters, int cap_members,
                    int32_t* n_clusters);
/* POSE_RANSAC_LM_DIFF_REPROJECTION_CPU::process (src/pose/...REPROJECTION_CPU.hpp:264-307) on the resident clusters
 * (which = 1: CLUSTER's; 2: the clusters FILTER rewrote): the objects this step APPENDS to FrameData::objects (:299), in
 * task order. */
typedef struct {
  int32_t model;
  float pose[7];   /* qx qy qz qw tx ty tz */
} mh_step_object;
int mh_step_pose(mh_ctx* ctx, int which, const mh_pose_params* prm, uint64_t seed, mh_step_object* out, int cap,
                 int32_t* n_out);
/* FILTER_PROJECTION_CPU::process (src/filter/FILTER_PROJECTION_CPU.hpp:80-162) on the resident objects (which = 1:
 * FILTER, 2: FILTER2).  n_objects = length of the host's object list (checked against the device's).  Outputs as
 * mh_filter_images, indexed by LIST position: score[i], keep[i]; kept object k was list element out_order[k] and now
 * owns cl_members[cl_off[k] .. cl_off[k + 1]) = indices into its model's match list. */
int mh_step_filter(mh_ctx* ctx, int which, int min_points, float feature_distance, float min_score, int n_objects,
                   float* score, uint8_t* keep, int32_t* out_order, int32_t* cl_members, int32_t* cl_off /* n_objects + 1 */,
                   int cap_members, int32_t* n_kept);
int mh_frame_enqueue_batch(mh_ctx* ctx, float* q_desc_dev, const float* q_uv_dev, int Q, int B, const mh_cam* cam,
                           const mh_frame_params* prm, const uint64_t* seeds);
/* Frames with several images (FrameData::images; every DetectedFeature carries its imageIdx, src/util.hpp:70-79):
 * q_image_dev[Q] = image of every query of the frames enqueued from now on (device memory, read when a frame
 * runs), cams[n_images] their cameras.  CLUSTER then runs per (model, image) in image order
 * (CLUSTER_MEAN_SHIFT_CPU.hpp:189-195), POSE / FILTER project every correspondence through its own image; the
 * `cam` argument of mh_frame_enqueue* is ignored.  n_images <= 1 or NULL: back to one image.  With batches
 * (mh_frame_enqueue_batch, mh_frame_enqueue_sharded_batch) q_image_dev holds [B Q] entries, frame after frame like the
 * queries.  Not together with the moped3d depth steps: those are single-camera in the reference itself (DEPTHMAP_PROP
 * looks every match up in the ONE depth map of the frame whatever its image,
 * moped3d/libmoped/src/depthprop/DEPTHMAP_PROP_CPU.hpp:86-112). */
int mh_frame_set_images(mh_ctx* ctx, const int32_t* q_image_dev, const mh_cam* cams, int n_images);

/* Per-query depth attributes for the next frames (device pointer, [Q] mh_depth, in query
 * order; NULL switches back to the 2-D residuals).  POSE and POSE2 of the frame then use
 * the MH_DEPTH_* residuals `kind` with `alpha`. */
int mh_frame_set_depth(mh_ctx* ctx, const mh_depth* q_depth_dev, int kind, float alpha);
/* The same from the depth map itself, as moped3d holds it (moped3d/moped3d.cpp:279-333):
 * depth_xyzn_dev [height][width][4] floats (camera-frame x, y, z, norm or negative = invalid),
 * fill_distance_dev [height][width] or NULL (DEPTH_FILL's ".distance" map; NULL = -1 everywhere).
 * Each accepted match looks its pixel up on the device -- DEPTHMAP_PROP_CPU::process
 * (moped3d/libmoped/src/depthprop/DEPTHMAP_PROP_CPU.hpp:101-134): truncated coordinates, no
 * interpolation -- and gets weight getCauchyWeight(fillDistance) with `cauchy_scale` (0.1 in
 * POSE_..._BACKPROJECTION_DEPTH_CPU.hpp:66, 25 in ..._REPROJECTION_DEPTH_CPU.hpp:66).
 * NULL image switches depth off. */
int mh_frame_set_depth_image(mh_ctx* ctx, const float* depth_xyzn_dev, const float* fill_distance_dev,
                             int width, int height, int kind, float alpha, float cauchy_scale);
/* One depth map (and distance map) per frame of the batches enqueued from now on (mh_frame_enqueue_batch with
 * B = n_frames): pointer arrays of n_frames <= MH_MAX_BATCH device images of the same size.  The batch's frames share
 * their launches (round 4): every frame's map must stay valid until the batch's results are fetched. */
int mh_frame_set_depth_image_batch(mh_ctx* ctx, const float* const* depth_xyzn_dev, const float* const* fill_distance_dev,
                                   int n_frames, int width, int height, int kind, float alpha, float cauchy_scale);
/* The same for hosts that hold the maps in host memory (the STEP plugins): copies them into
 * context-owned device buffers (4.9 MB + 1.2 MB for 640x480) and sets them. */
int mh_frame_set_depth_image_host(mh_ctx* ctx, const float* depth_xyzn_host, const float* fill_distance_host,
                                  int width, int height, int kind, float alpha, float cauchy_scale);

/* moped3d's DEPTHFILL step: DEPTH_FILL_EXACT_CPU::process / fillInScaled
 * (moped3d/libmoped/src/depthfill/DEPTH_FILL_EXACT_CPU.hpp:268-349, 416-428; config.hpp:39 ships
 * DEPTH_FILL_EXACT_CPU(8, false)).  depth_xyzn_dev [height][width][4] floats (x, y, z, norm; z < 0 = hole) is filled
 * IN PLACE on the context's stream: every hole takes the depth the reference's FIFO wavefront over the map downscaled
 * by scale_factor gives it (nearest-neighbour upsampling with the reference's late row advance, or bilinear != 0:
 * its bilinear one), then x, y and norm of the filled pixels are recomputed from K = the depth map's
 * intrinsicLinearCalibration (:249-273).  fill_distance_dev [height][width] receives the distance map the step
 * appends to the frame ("<name>.distance": 0 on valid pixels), the one mh_frame_set_depth_image takes.
 * scale_factor = -1 chooses the factor from the share of holes as the reference does (:283-296; one small
 * reduction and a 4-byte read back); with a factor of 1 the reference never hands the filled depths back
 * (:336-338) and neither does this: only the distance map is written.  *scale_used (optional) = the factor.
 * Bit-identical to the reference's arithmetic (tests/test_gpu_depthfill.py against the oracle's restatement).
 * Limits: the downscaled map holds at most 8192 pixels (640 x 480 from factor 8 on) -> MH_ERR_CAPACITY.
 * Asynchronous; mh_depth_fill_status synchronises the stream and reports MH_ERR_CAPACITY if the fill's queue
 * outgrew its 16384-entry ring in a call since the last status (no map of a real sensor comes near it). */
int mh_depth_fill(mh_ctx* ctx, float* depth_xyzn_dev, int width, int height, int scale_factor, int bilinear,
                  const float K[4], float* fill_distance_dev, int* scale_used);
int mh_depth_fill_status(mh_ctx* ctx);
/* The same on host maps (what the DEPTH_FILL_EXACT_HIP plugin calls): upload, fill, both maps back, status. */
int mh_depth_fill_host(mh_ctx* ctx, float* depth_xyzn_host, int width, int height, int scale_factor, int bilinear,
                       const float K[4], float* fill_distance_host, int* scale_used);
/* moped3d's rules on which features and matches reach CLUSTER, applied on the device inside the
 * frame (they need the depth map: mh_frame_set_depth_image):
 *  - DEPTHFILTER_CPU (moped3d/libmoped/src/depthfilter/DEPTHFILTER_CPU.hpp:117-254), ToFilter = 1
 *    on the detected features (`feature_density`, config.hpp: 0.05) and ToFilter = 2 on every
 *    model's matches (`match_density`, 0.01): density per square metre of scene surface over
 *    PatchSize x PatchSize pixel patches, dilated 3x3, must exceed Density.  K = the depth map's
 *    intrinsicLinearCalibration.  A negative density switches that filter off.
 *  - MATCH_ADAPTIVE_FLANN_CPU's ratio (.../match/MATCH_ADAPTIVE_FLANN_CPU.hpp:193-215,361-376,
 *    457-467): ratio_table[m] = (maxRatioDepth, minRatioDepth, ratioLow, ratioHigh) of model m as
 *    its Update() derives them (:144-177; the host plugin MATCH_ADAPTIVE_BRUTE_HIP does);
 *    features deeper than maximum_depth never match; NULL = mh_frame_params.ratio for all.
 * Coordinates outside the depth map are clamped to it (the reference reads out of bounds).
 * rules == NULL switches all of it off. */
typedef struct mh_depth_rules {
  int32_t patch_size;
  float feature_density;
  float match_density;
  const float* ratio_table; /* host, [n_models][4]; copied by the call */
  int32_t n_models;
  float maximum_depth;      /* 4.0  (MATCH_ADAPTIVE_FLANN_CPU.hpp:107) */
  float default_depth;      /* 1.0  (:108) */
  float cauchy_scale;       /* 0.1  (:109) */
} mh_depth_rules;
int mh_frame_set_depth_rules(mh_ctx* ctx, const mh_depth_rules* rules, const float K[4]);
/* moped3d's default clusterer instead of mean shift: CLUSTER_LINKAGE_CPU
 * (moped3d/libmoped/src/cluster/CLUSTER_LINKAGE_CPU.hpp:573-704) as config.hpp:45 constructs it --
 * per model a similarity matrix over its matches (Gaussian kernels on image / camera-frame
 * distances with sigma = average nearest-neighbour distance, depth-discontinuity kernel along the
 * image line between two matches, model/world distance-consistency kernel, fill-distance weighted
 * sum) and average-linkage agglomeration down to `cutoff`.  Needs the depth map
 * (mh_frame_set_depth_image); at most 1024 matches per model.  LinkageType 1 (average) only. */
typedef struct mh_linkage_params {
  float cutoff;         /* 0.1 */
  int32_t min_pts;      /* 7: clusters need MORE than this many members (:535) */
  int32_t use3d_filter; /* 2: multiply (1: add, 0: skip) the distance-consistency kernel */
  float sigma2d;        /* -1: average nearest-neighbour distance */
  float sigma3d;        /* -1 */
  int32_t linkage_type; /* 1: average linkage (config.hpp:45); 0 minimum, 2 maximum (CLUSTER_LINKAGE_CPU.hpp:506-526) */
} mh_linkage_params;
/* CLUSTER of the following frames: linkage with *prm, or mean shift again when prm == NULL. */
int mh_frame_set_cluster_linkage(mh_ctx* ctx, const mh_linkage_params* prm);
/* The clusterer's scratch is three n x n similarity matrices per (model, frame) problem (CLUSTER_LINKAGE_CPU.hpp:573-704
 * keeps the same three as vector<vector<Float>>): sized for the worst case of what the context has reserved -- 3 x 1024 x
 * max matches floats per frame of a batch, 37 MB at 3 000 -- and bounded: a frame or batch that would need more than
 * `bytes` (0 = the default, 4 GiB) fails with MH_ERR_CAPACITY before anything is allocated or launched. */
int mh_set_linkage_scratch_limit(mh_ctx* ctx, size_t bytes);
/* The step on its own, n_problems point sets (one per model) in one call: corr_host / depth_host
 * = the matches (image point + model point / camera-frame point from the depth map), problem p =
 * rows [off[p], off[p+1]).  label[i] = cluster of row i within its problem or -1; order (optional)
 * = rows of each problem's clusters in the reference's member order, problem-local, -1 padded;
 * n_clusters[p].  Uses the depth map set by mh_frame_set_depth_image. */
int mh_cluster_linkage(mh_ctx* ctx, const mh_corr* corr_host, const mh_depth* depth_host, const int32_t* off,
                       int n_problems, const mh_linkage_params* prm, int32_t* label, int32_t* order,
                       int32_t* n_clusters);
/* The two halves around exchange 1 when the DB is sharded over ranks.
 *   mh_frame_enqueue_match_local : normalise + this shard's top-2 -> top2_dev, a
 *       caller-owned device block of [3][Q] 32-bit words {idx1 (global row, int32),
 *       bits of d1, bits of d2}: the send buffer of the all-gather
 *   mh_frame_enqueue_rest        : gathered_dev = the all-gather's receive buffer,
 *       [n_shards][3][Q] words in rank order; merges the shards' top-2, keeps the
 *       matches of models this context owns, then CLUSTER..FILTER2 as above. */
int mh_frame_enqueue_match_local(mh_ctx* ctx, float* q_desc_dev, int Q, int32_t* top2_dev);
int mh_frame_enqueue_rest(mh_ctx* ctx, const float* q_uv_dev, int Q, const int32_t* gathered_dev,
                          int n_shards, const mh_cam* cam, const mh_frame_params* prm,
                          uint64_t seed);
/* The same with the shards' blocks `shard_stride_words` (>= 3 Q) 32-bit words apart in the
 * gathered buffer: whatever rides behind each [3][Q] block (e.g. the previous frame's result
 * block, so that exchange 2 needs no collective of its own) is ignored here. */
int mh_frame_enqueue_rest_strided(mh_ctx* ctx, const float* q_uv_dev, int Q, const int32_t* gathered_dev,
                                  int n_shards, int shard_stride_words, const mh_cam* cam,
                                  const mh_frame_params* prm, uint64_t seed);
/* Frames in batches (small shards: one MATCH launch and one exchange for B frames).  The B frames'
 * descriptors lie one after the other, so mh_frame_enqueue_match_local(ctx, q_desc, B * Q, top2) is
 * the batched MATCH as it is: top2 = [3][B Q] words.  After the exchange, frame f of the batch is
 *   mh_frame_enqueue_rest_batch(ctx, q_uv + 2 f Q, Q, gathered + f Q, W, shard_stride, B Q, f, ...)
 * -- `plane_stride_words` = the distance between the idx / d1 / d2 planes of a shard's block (B Q),
 * `slot` = f < MH_MAX_BATCH selects the result block the frame writes.  The frames of a batch run
 * one after the other on the context's stream.  mh_frame_fetch_slot / mh_frame_result_copy_slots_dev
 * are the per-slot forms of mh_frame_fetch / mh_frame_result_copy_dev (the latter packs the heads
 * of slots 0..n_slots-1 one after the other). */
int mh_frame_enqueue_rest_batch(mh_ctx* ctx, const float* q_uv_dev, int Q, const int32_t* gathered_dev, int n_shards,
                                int shard_stride_words, int plane_stride_words, int slot, const mh_cam* cam,
                                const mh_frame_params* prm, uint64_t seed);
/* All B frames of such a batch at once -- the B calls above for f = 0..B-1 (q_uv_dev / gathered_dev name frame 0's,
 * the others lie Q rows / Q words further on; results in slots 0..B-1), but with ONE launch per stage for the B frames
 * where the frames allow it (no per-frame depth state): same objects, a quarter to an eighth of the dependent launches. */
int mh_frame_enqueue_rest_frames(mh_ctx* ctx, const float* q_uv_dev, int Q, const int32_t* gathered_dev, int n_shards,
                                 int shard_stride_words, int plane_stride_words, int B, const mh_cam* cam,
                                 const mh_frame_params* prm, const uint64_t* seeds);
int mh_frame_fetch_slot(mh_ctx* ctx, int slot, mh_object* objects_host, int max_objects, int32_t* n_objects,
                        int32_t* counts);
int mh_frame_result_copy_slots_dev(mh_ctx* ctx, void* dst_dev, int n_slots, int max_objects);
/* Enqueues a device copy of the head of the context's result block {int32 n; int32 pad[3];
 * mh_object[max_objects]} -- the result of the last frame enqueued on this context, n = 0
 * before the first -- to dst_dev (16 + 40 max_objects bytes). */
int mh_frame_result_copy_dev(mh_ctx* ctx, void* dst_dev, int max_objects);
/* Synchronises the stream and copies the frame's objects out (capacity
 * max_objects); *n_objects = count.  counts (optional, 4 ints): accepted
 * matches, clusters, objects after POSE, objects after FILTER. */
int mh_frame_fetch(mh_ctx* ctx, mh_object* objects_host, int max_objects,
                   int32_t* n_objects, int32_t* counts);
/* ---- delivery: the objects of EVERY frame of a batch to the host, without stopping the stream -------------------
 *
 * The reference's frame loop hands each frame's list<SP_Object> to its caller (MopedPimpl::processImages,
 * src/moped.cpp:166-194; the test node prints them, moped2/moped_test.cpp:205-207).  For a host that keeps batches in
 * flight the per-frame fetches above cost a stream synchronisation each; these two move a whole batch in ONE
 * stream-ordered operation into a block of PINNED host memory the caller owns (hipHostMalloc / hipHostRegister /
 * torch's pin_memory), and a wait that touches only that delivery's event:
 *
 *   mh_frame_enqueue_batch(ctx, ..., B, ...);                      // batch k on this context
 *   mh_frame_fetch_batch_async(ctx, B, cap, block, tag);           // behind it, on the same stream: no host wait
 *   ... other contexts' batches ...
 *   mh_frame_fetch_wait(ctx, &flags);                              // before the context's next batch: usually done long ago
 *   for f < B: head f = (const mh_frame_head*)((char*)block + f * mh_frame_block_stride(cap)); objects follow it
 *
 * Block layout: B records of mh_frame_block_stride(max_objects) bytes, record f = mh_frame_head {n_objects, flags,
 * counts[4] (accepted matches, clusters, objects after POSE, after FILTER; -1 where unknown), tag, frame} followed
 * by mh_object[max_objects] of which the first min(n_objects, max_objects) are written (model ids as
 * mh_frame_fetch gives them).  `tag` is the caller's: a block that still holds an older delivery is recognisable.
 * A block the device can address (pinned) is written by the delivering kernel itself over PCIe; any other host
 * memory goes through a device staging buffer and one hipMemcpyAsync (pageable memory makes that copy synchronous).
 * The next batch on the context may be enqueued at once -- the delivery is ordered before it on the stream -- but the
 * BLOCK must not be handed to another delivery before mh_frame_fetch_wait returned.  One delivery per context in
 * flight: a second async call before the wait -> MH_ERR_ARG. */
typedef struct {
  int32_t n_objects;
  int32_t flags;       /* capacity / exchange flags of the frame; 0 = clean */
  int32_t counts[4];
  uint32_t tag;
  int32_t frame;       /* 0 .. B-1 */
} mh_frame_head;
size_t mh_frame_block_stride(int max_objects);   /* sizeof(mh_frame_head) + max_objects * sizeof(mh_object) */
int mh_frame_fetch_batch_async(mh_ctx* ctx, int B, int max_objects, void* host_block, uint32_t tag);
/* The same for a SHARDED context (mh_frame_enqueue_sharded[_batch]): the objects of ALL ranks, in rank order, of
 * the B frames this context ran BEFORE the current batch, as they arrived with the current batch's exchange
 * (mh_frame_previous_objects without its stream synchronisation).  counts[3] = n_objects, the others -1; flags = the
 * ranks' flags or-ed, bit 30 set when a rank had more than MH_EX2_OBJECTS objects (its surplus did not travel). */
int mh_frame_fetch_previous_async(mh_ctx* ctx, int max_objects, void* host_block, uint32_t tag);
/* Blocks until the context's pending delivery has landed (returns at once if there is none).  *flags_or (optional)
 * = the or of the B heads' flags; any set -> MH_ERR_CAPACITY with the text mh_frame_fetch_slot gives.
 * mh_frame_fetch_query: MH_OK when the delivery has landed (or none is pending), 1 while it is still in flight. */
int mh_frame_fetch_wait(mh_ctx* ctx, int32_t* flags_or);
int mh_frame_fetch_query(mh_ctx* ctx);

/* The frame's accepted matches after MATCH (+ the depth rules): query index and model of each,
 * sorted by (model, query) = the reference's `matches[model]` lists one after the other
 * (MATCH_ANN_CPU.hpp:165-176).  Synchronises the stream; *n_matches = their number. */
int mh_frame_fetch_matches(mh_ctx* ctx, int32_t* query_host, int32_t* model_host, int cap, int32_t* n_matches);
/* The same matches as correspondences {u, v, x, y, z} (FrameData::Match's coord2D and coord3D, src/util.hpp:81-88), in the
 * same order, of the last frame: what a host needs to fill `frameData.matches` the way MATCH_ANN_CPU::process does. */
int mh_frame_fetch_match_points(mh_ctx* ctx, mh_corr* corr_host, int cap, int32_t* n_matches);
/* The same for frame `slot` of the last batch (mh_frame_enqueue_batch / _rest_frames / _sharded_batch): the B frames of a
 * batch that shared their launches keep their lists side by side; of frames that went through the steps one after the
 * other (depth maps per frame, stage timing) only the last one's remain -> MH_ERR_ARG for the others. */
int mh_frame_fetch_matches_slot(mh_ctx* ctx, int slot, int32_t* query_host, int32_t* model_host, int cap,
                                int32_t* n_matches);
/* Device address of the frame's packed result block {int32 n; mh_object[cap]}
 * for exchange 2 (gather of per-rank objects); *bytes = its size. */
int mh_frame_result_dev(mh_ctx* ctx, void** block_dev, int64_t* bytes);

/* ---- N > 1: the DB sharded over ranks, the frame's exchanges inside the library ----------------
 *
 * Rank r of W holds the models [r n/W, (r+1) n/W) (mh_db_upload with index_base = its first global row) and every
 * rank sees every frame.  A sharded frame = this shard's top-2 per query, ONE all-gather of [3][Q] words per rank
 * (exchange 1; behind them rides the result block of the context's previous frame = exchange 2, so a frame costs
 * one collective), merge, then CLUSTER .. FILTER2 for the models the rank owns -- all on the context's stream.
 * The objects of a sharded frame are bit-identical to the single-context frame's (tests/test_gpu_comm.py).
 * The loop this serves: MopedPimpl::processImages, moped2/libmoped/src/moped.cpp:166-194.
 *
 * Transport: RCCL (ncclAllGather over xGMI), resolved at run time -- the copy already in the process, else
 * librccl.so.1 from the loader path or /opt/rocm/lib, else $MH_RCCL_PATH -- or a host callback.
 *   mh_comm_create      one process per GPU: rank 0 makes the id (mh_comm_unique_id), the host's launcher hands
 *                       it to the others (MPI_Bcast, a file, torch.distributed ...), every rank calls this
 *   mh_comm_create_all  one process that owns W devices (ncclCommInitAll): comms[r] belongs to ctxs[r]
 *   mh_comm_create_host bring-your-own transport: fn(user, send, recv, bytes) must fill recv[W][bytes] with every
 *                       rank's send block in rank order; it is called on the host with the stream drained
 *                       (ranks that share a device, MPI-only hosts, test rigs)
 * A communicator may serve several contexts of its device (frames in flight); like any NCCL communicator it wants
 * its collectives issued in the same order on every rank. */
#define MH_COMM_ID_BYTES 128
#define MH_EX2_OBJECTS 62   /* objects per rank and frame that ride on the next frame's exchange */
typedef struct mh_comm mh_comm;
typedef int (*mh_allgather_fn)(void* user, const void* send_host, void* recv_host, size_t bytes_per_rank);
int mh_comm_unique_id(unsigned char id[MH_COMM_ID_BYTES]);
int mh_comm_create(mh_ctx* ctx, const unsigned char id[MH_COMM_ID_BYTES], int rank, int world, mh_comm** comm);
int mh_comm_create_all(mh_ctx* const* ctxs, int world, mh_comm** comms);
int mh_comm_create_host(mh_ctx* ctx, int rank, int world, mh_allgather_fn fn, void* user, mh_comm** comm);
int mh_comm_destroy(mh_comm* comm);
int mh_comm_info(const mh_comm* comm, int* rank, int* world, int* is_rccl);
/* One frame / a batch of B <= MH_MAX_BATCH frames (descriptors and keypoints of the B frames one after the other:
 * one MATCH launch and one exchange for all of them; frame f leaves its objects in result slot f). */
int mh_frame_enqueue_sharded(mh_ctx* ctx, mh_comm* comm, float* q_desc_dev, const float* q_uv_dev, int Q,
                             const mh_cam* cam, const mh_frame_params* prm, uint64_t seed);
int mh_frame_enqueue_sharded_batch(mh_ctx* ctx, mh_comm* comm, float* q_desc_dev, const float* q_uv_dev, int Q, int B,
                                   const mh_cam* cam, const mh_frame_params* prm, const uint64_t* seeds);
/* The single-process form: rank r's copy of the frame inputs on its device in q_desc_dev[r] / q_uv_dev[r]; the W
 * collectives go out as one group. */
int mh_frame_enqueue_sharded_all(mh_ctx* const* ctxs, mh_comm* const* comms, int world, float* const* q_desc_dev,
                                 const float* const* q_uv_dev, int Q, int B, const mh_cam* cam,
                                 const mh_frame_params* prm, const uint64_t* seeds);
/* Objects of ALL ranks (rank order = model order) for the frame(s) this context ran BEFORE the current one, as
 * they arrived with the current exchange: frame_in_batch < B.  Synchronises the stream.  *n_objects may exceed
 * cap (the first cap are written).  A rank with more than MH_EX2_OBJECTS objects -> MH_ERR_CAPACITY (use
 * mh_frame_gather_objects for that frame). */
int mh_frame_previous_objects(mh_ctx* ctx, int frame_in_batch, mh_object* objects_host, int cap, int32_t* n_objects);
/* Exchange 2 on its own for the frame in result slot `slot` (the last frame of a stream, or any frame whose
 * objects are wanted at once): all ranks call it; all ranks get all objects. */
int mh_frame_gather_objects(mh_ctx* ctx, mh_comm* comm, int slot, mh_object* objects_host, int cap,
                            int32_t* n_objects);

/* ---- FEAT: SIFT extraction ---------------------------------------- */

/* FEAT_SIFT_CPU::process for one image (src/feat/FEAT_SIFT_CPU.hpp:78-112 over libsiftfast 1.1's
 * GetKeypoints, libs.tgz -> libsiftfast-1.1-src/libsiftfast.cpp:301-361): gray = height x width
 * bytes; double_size = the ScaleOrigin "-1" setting (config.hpp:69).  Keypoints come out in the
 * reference's (single-thread) list order: xy = coord2D = (col, row), scale_ori (optional) =
 * (scale, orientation), desc = 128 floats, unit length.  `cap` = capacity of the outputs; more
 * keypoints than that -> MH_ERR_CAPACITY (the first `cap` are still written). */
int mh_sift_extract(mh_ctx* ctx, const uint8_t* gray_host, int width, int height, int double_size,
                    float* xy_host, float* scale_ori_host, float* desc_host, int cap,
                    int32_t* n_keypoints);
/* The same with everything on the device and no host synchronisation: desc_dev [cap][128] and
 * xy_dev [cap][2] can be handed to mh_frame_enqueue once *n_dev (device int32) has been read. */
int mh_sift_extract_dev(mh_ctx* ctx, const uint8_t* gray_dev, int width, int height, int double_size,
                        float* desc_dev, float* xy_dev, float* scale_ori_dev, int cap, int32_t* n_dev);

/* FEAT + the whole frame, image in, objects out, nothing through the host: SIFT of the device image
 * (as mh_sift_extract_dev) straight into the frame's query buffers, then mh_frame_enqueue's launch
 * list.  The keypoint count stays on the device: every launch is sized for `max_keypoints` and the
 * kernels read the count (query blocks beyond it leave at once); more keypoints than that -> the
 * first max_keypoints in list order are used.  Results: mh_frame_fetch; mh_frame_keypoints = the
 * keypoint count of the frame last fetched; mh_frame_features_dev = the device buffers
 * (descriptors already L2-normalised like MATCH_ANN_CPU.hpp:157 leaves them). */
int mh_frame_enqueue_image(mh_ctx* ctx, const uint8_t* gray_dev, int width, int height, int double_size,
                           int max_keypoints, const mh_cam* cam, const mh_frame_params* prm, uint64_t seed);
/* B <= MH_MAX_BATCH images at once: FEAT image by image (every image's keypoints at a stride of max_keypoints rows,
 * its count in a device word of its own), then ONE MATCH launch sequence over all of them -- an image's ~600
 * keypoints alone leave the matrix pipes a sixth as busy per query as a batch does --, then CLUSTER..FILTER2 image
 * after image into result slots 0 .. B-1 (mh_frame_fetch_slot).  Every image's objects are bit for bit those of
 * mh_frame_enqueue_image(..., seeds[f]) on it alone.  One camera; not combined with depth maps or image indices. */
int mh_frame_enqueue_image_batch(mh_ctx* ctx, const uint8_t* const* gray_dev, int n_images, int width, int height,
                                 int double_size, int max_keypoints, const mh_cam* cam, const mh_frame_params* prm,
                                 const uint64_t* seeds);
int mh_frame_features_dev(mh_ctx* ctx, float** desc_dev, float** uv_dev, int32_t** n_dev);
int mh_frame_keypoints(mh_ctx* ctx, int32_t* n_keypoints);

/* ---- model files ------------------------------------------------ */

/* Host-side set of models: parsed from `.moped.xml` files the way Moped::addModel(sXML&)
 * reads them (src/moped.cpp:101-137 over include/sXML.hpp:53-118) or mapped from a packed
 * `.mopeddb` container.  Rows are flattened in model order like
 * MATCH_ANN_CPU::Update (src/match/MATCH_ANN_CPU.hpp:76-99).  Only points whose desc_type
 * equals the set's (default "SIFT") are kept; a point whose descriptor does not have 128
 * values is an error.  Adding a model whose name exists replaces it (moped.cpp:141-146). */
typedef struct mh_model_set mh_model_set;
int mh_models_create(mh_model_set** out, const char* desc_type /* NULL = "SIFT" */);
void mh_models_destroy(mh_model_set* s);
const char* mh_models_last_error(const mh_model_set* s);
int mh_models_add_xml(mh_model_set* s, const char* path);
int mh_models_add_xml_buffer(mh_model_set* s, const char* data, int64_t bytes);
int mh_models_count(const mh_model_set* s);
int64_t mh_models_rows(const mh_model_set* s);
const char* mh_models_name(const mh_model_set* s, int i);
/* rows [row_begin, row_begin + n_rows) of model i; bbox = min xyz, max xyz (moped.cpp:107-124) */
int mh_models_range(const mh_model_set* s, int i, int64_t* row_begin, int64_t* n_rows, float bbox[6]);
const float* mh_models_desc(const mh_model_set* s);   /* [rows][128], as parsed (not normalised) */
const float* mh_models_xyz(const mh_model_set* s);    /* [rows][3] */
/* Packed container: written once, then mapped read-only and uploaded as it lies. */
int mh_models_save(const mh_model_set* s, const char* path);
int mh_models_load(mh_model_set** out, const char* path);
/* Update() for models [first_model, first_model + n_models) of the set (a rank's shard):
 * uploads their rows, L2-normalises them on the device (A1) and sets index_base to the
 * first row, so row and model ids stay global. */
int mh_db_upload_models(mh_ctx* ctx, const mh_model_set* s, int first_model, int n_models);
/* mh_db_upload with the normalisation done on the device (normalize != 0). */
int mh_db_upload_raw(mh_ctx* ctx, const float* desc_host, const int32_t* model_of_host,
                     const float* xyz_host, int N, int n_models, int32_t index_base, int normalize);

/* ---- timing --------------------------------------------------------------------- */

/* Device-side counters of the last frame that ran on the context (synchronises its stream):
 * out[0] accepted matches, [1] clusters, [2..4] reserved, [5] (cluster, replica) tasks of POSE + POSE2 that
 * evaluated hypotheses, [6] capacity flags, [7] P3P hypotheses those tasks evaluated -- POSE stops after the
 * first 256 of a task's n_hypotheses when the inlier ratio they reached makes a better all-inlier sample
 * unlikely; n_hypotheses < 0 in mh_pose_params means "-n_hypotheses, all of them". */
int mh_frame_counters(mh_ctx* ctx, int32_t out[8]);
/* Resource use of the RANSAC kernel behind mh_pose_ransac* and the frame's POSE / POSE2 launches, as the HIP runtime
 * reports it for this device (occupancy of the RANSAC kernel next to every GPU figure; the reference's
 * step is POSE_RANSAC_LM_DIFF_REPROJECTION_CPU::process, src/pose/POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:264-307):
 * kind = MH_DEPTH_* (3 = frames with several images); out[0] VGPRs per lane, [1] LDS bytes per workgroup, [2] threads
 * per workgroup, [3] workgroups resident per compute unit, [4] scratch bytes per lane (spills), [5] wavefronts per
 * workgroup. */
int mh_pose_kernel_info(mh_ctx* ctx, int kind, int32_t out[8]);
/* GPU time of the kernels of the two-stage MATCH on this context, after mh_enable_timing(ctx, 1): the mean over
 * the launch sequences since the previous call (at most the last 32; synchronises the context's stream):
 * ms[0] query image (f16), [1] pass A (thresholds from a sample of the rows), [2] threshold merge,
 * [3] pass B (the f16 screen of every (query, row) pair: the dominant kernel), [4] pass C (canonical
 * f32 arithmetic on the candidates).  MH_ERR_ARG if no two-stage MATCH has run with timing on. */
int mh_match_timing(mh_ctx* ctx, float ms[5]);

typedef struct {
  float match_ms, group_ms, cluster_ms, pose1_ms, filter1_ms, pose2_ms, filter2_ms, total_ms;
} mh_times;
/* GPU time of each stage of the last frame (hipEvents on the context's stream;
 * collected only after mh_enable_timing(ctx, 1)). */
int mh_enable_timing(mh_ctx* ctx, int on);
int mh_timing(mh_ctx* ctx, mh_times* out);

#ifdef __cplusplus
}
#endif
#endif /* MOPED_HIP_H */
