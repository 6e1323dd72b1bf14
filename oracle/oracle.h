/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of libmoped's per-frame hot path (MATCH -> CLUSTER -> POSE
 * [-> FILTER]) in plain C++ with a C ABI, used as the parity checker for the
 * HIP path and as the "port" CPU baseline in bench.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (moped_amd/) never does.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/moped2/libmoped/).  Pinning status (see DESIGN.md "Oracle"):
 *   match      pinned against the reference's own ANN 1.1.1 library at eps=0
 *              (oracle/_ref, tests/golden/match_*.npz)
 *   project /  pinned against the reference's project()/TransformMatrix and
 *   LM refine  levmar slevmar_dif (oracle/_ref, tests/golden/pose_*.npz)
 *   mean shift PARITY UNPINNED: the algorithm lives wholly inside
 *              CLUSTER_MEAN_SHIFT_CPU.hpp, which cannot be compiled here
 *              (needs util.hpp -> OpenCV headers) and the reference holds no
 *              fixture for it; restated from the source text only.
 *   RANSAC skeleton / FILTER: restated from the source text; the numerical
 *              kernels they call are the pinned ones above.
 */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A1  MATCH_ANN_CPU::norm (match/MATCH_ANN_CPU.hpp:54-57): in-place L2
 * normalisation, sequential fp32 sum, scale = (float)(1.0 / sqrtf(sum)). */
void orc_normalize(float* desc, int n, int dim);

/* A3  exact 2-NN by squared L2 over the whole DB (MATCH_ANN_CPU.hpp:155-162 with
 * Quality = 0).  Canonical arithmetic shared bit-for-bit with the HIP kernel:
 *   dot(a,b) = fmaf chain over k = 0..dim-1 starting from 0
 *   dist(q,d) = max(0, fmaf(-2, dot(q,d), dot(q,q) + dot(d,d)))
 * ties on distance -> lower DB index.  idx1[i] = -1 when N == 0; d2 = +inf
 * when N < 2.  n_threads <= 0 -> all OpenMP threads. */
/* dot(d,d) chain of every row: the norm term of the canonical distance. */
void orc_row_norms(const float* desc, int n, int dim, float* out);
void orc_match_2nn(const float* db, int N, const float* q, int Q, int dim,
                   int32_t* idx1, float* d1, float* d2, int n_threads);

/* A3  ratio test + scatter (MATCH_ANN_CPU.hpp:165-176): accept query i when
 * d1[i] / d2[i] < ratio (squared distances, fp32 division); matches are
 * grouped by model_of[idx1] and kept in ascending query order inside a model.
 * out_q[m] = query index of the m-th match in (model, query) order,
 * model_off[n_models + 1] = CSR offsets.  Returns the match count. */
int orc_match_accept(const int32_t* idx1, const float* d1, const float* d2, int Q,
                     float ratio, const int32_t* model_of, int n_models,
                     int32_t* out_q, int32_t* model_off);

/* Exchange-1 merge for a model-sharded DB (new design; the
 * reference builds one kd-tree over all models, MATCH_ANN_CPU.hpp:76-107):
 * per shard s the local (idx1 [global index], d1, d2) of every query, laid out
 * [s][Q]; result = the global top-2. */
void orc_match_merge(const int32_t* idx1_s, const float* d1_s, const float* d2_s,
                     int n_shards, int Q, int32_t* idx1, float* d1, float* d2);

/* A6  CLUSTER_MEAN_SHIFT_CPU::MeanShift (cluster/CLUSTER_MEAN_SHIFT_CPU.hpp:80-158)
 * on n points of `dim` (2 or 3) floats.  members[] receives the point indices
 * of every emitted cluster (size >= min_pts) in emission order and, inside a
 * cluster, in the reference's splice order; cluster_off[] the CSR offsets
 * (capacity n + 1).  Returns the number of clusters; *n_iter (optional) the
 * iterations run. */
int orc_meanshift(const float* pts, int n, int dim, float radius, float merge,
                  int min_pts, int max_iter, int32_t* members, int32_t* cluster_off,
                  int* n_iter);

/* N4  CLUSTER_LINKAGE_CPU::process for one model (moped3d/libmoped/src/cluster/CLUSTER_LINKAGE_CPU.hpp:
 * 573-704; see linkage_oracle.cpp): uv = coord2D, model_xyz = coord3D, world_xyz =
 * depthData.coord3D of the model's n matches; depth_img [h][w][4], fill_img [h][w] or NULL.
 * members / cluster_off (capacity n + 1) as orc_meanshift; K_out (optional, n*n) = the final
 * similarity matrix.  Returns the number of clusters (those with MORE than min_pts members). */
int orc_cluster_linkage(const float* uv, const float* model_xyz, const float* world_xyz, int n,
                        const float* depth_img, int w, int h, const float* fill_img, float cutoff,
                        int min_pts, int use3d_filter, int linkage_type, float sigma2d, float sigma3d,
                        int32_t* members, int32_t* cluster_off, float* K_out);

/* A12 project() (include/moped.hpp:330-354) for n points: pose/cam are
 * (qx,qy,qz,qw,tx,ty,tz), K = (fx,fy,cx,cy); z < 0.001 -> (FLT_MAX,FLT_MAX). */
void orc_project(const float pose7[7], const float* xyz, int n, const float K[4],
                 const float cam[7], float* uv);

/* A10 lmFuncQuat (pose/POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:100-138):
 * hx[2n] squared pixel residuals, (-z + 10) twice when z < 0. */
void orc_residuals(const float pose7[7], const float* uv, const float* xyz, int n,
                   const float K[4], const float cam[7], float* hx);

/* A12 testAllPoints (…REPROJECTION_CPU.hpp:166-180): inlier[i] = squared
 * reprojection error < thr.  Returns the inlier count. */
int orc_test_all_points(const float pose7[7], const float* uv, const float* xyz, int n,
                        const float K[4], const float cam[7], float thr, uint8_t* inlier);

/* A11 optimizeCamera (…REPROJECTION_CPU.hpp:140-164): Levenberg-Marquardt with a
 * forward-difference Jacobian and Broyden rank-one updates, restating the
 * published algorithm of levmar 2.4 slevmar_dif (lm_core.c:427-825; defaults
 * lm.h:83-85).  Returns iterations (>= 0) or -1; pose7 updated with the
 * quaternion re-normalised; info[0] = initial ||e||^2, info[1] = final ||e||^2,
 * info[2] = stop reason. */
int orc_optimize_camera(float pose7[7], const float* uv, const float* xyz, int n,
                        const float K[4], const float cam[7], int itmax, float* info);

typedef struct {
  int max_ransac_tests;       /* 600 POSE / 100 POSE2 (config.hpp:110,118) */
  int max_lm_tests;           /* 200 / 500 */
  int max_objects_per_cluster;/* 4 */
  int n_pts_align;            /* 5 / 6 */
  int min_n_pts_object;       /* 6 / 8 */
  float error_threshold;      /* 10 / 5 (px^2) */
} orc_pose_params;

/* A8+A9+A13 RANSAC (…REPROJECTION_CPU.hpp:76-98,182-211) on one cluster of k
 * correspondences (single image), libc rand() as the reference uses.
 * Returns 1 and the refined pose when a hypothesis with more than
 * min_n_pts_object inliers was found, else 0. */
int orc_ransac(const float* uv, const float* xyz, int k, const float K[4],
               const float cam[7], const orc_pose_params* prm, float pose7[7]);
/* The same with the points' rank by address (randSample sorts pair<Float, LmData*>, :81-84: equal keys come out in
 * pointer order = ascending match index, :287-288); addr == NULL: position order.  orc_frame_rest passes the clusters'
 * match indices. */
int orc_ransac_addr(const float* uv, const float* xyz, const int32_t* addr, int k, const float K[4],
                    const float cam[7], const orc_pose_params* prm, float pose7[7]);
/* rand() of the RANSAC skeletons: libc's (fn == NULL, the default) or a stream a test injects. */
void orc_set_rand(int (*fn)(void));

/* A14 moped3d depth variants.  mode 1 = POSE_RANSAC_LM_DIFF_BACKPROJECTION_DEPTH_CPU
 * (moped3d/libmoped/src/pose/...BACKPROJECTION_DEPTH_CPU.hpp:108-190; 2 residuals per
 * point), mode 2 = POSE_RANSAC_LM_DIFF_REPROJECTION_DEPTH_CPU (...:106-216; 3 per
 * point).  world = camera-frame xyz from the depth map (Match.depthData.coord3D),
 * wgt = cauchyWeight (getCauchyWeight of fillDistance), alpha = the class's Alpha. */
void orc_residuals_depth(int mode, const float pose7[7], const float* uv, const float* xyz,
                         const float* world, const float* wgt, int n, const float K[4],
                         const float cam[7], float alpha, float* err);
int orc_optimize_camera_depth(int mode, float pose7[7], const float* uv, const float* xyz,
                              const float* world, const float* wgt, int n, const float K[4],
                              const float cam[7], float alpha, int itmax, float* info);
int orc_ransac_depth(int mode, const float* uv, const float* xyz, const float* world,
                     const float* wgt, int k, const float K[4], const float cam[7], float alpha,
                     const orc_pose_params* prm, float pose7[7]);

/* N1  FILTER_PROJECTION_CPU::process (filter/FILTER_PROJECTION_CPU.hpp:80-162)
 * for one image.  Matches in (model, query) order with CSR model_off; objects
 * given as (model, pose) in list order.  Outputs: score per object, keep flag
 * per object, and the rewritten clusters (match indices inside the model) as
 * CSR over kept objects in output order (model-major, list order inside).
 * Returns the number of kept objects. */
int orc_filter(const float* uv, const float* xyz, const int32_t* model_off, int n_models,
               const int32_t* obj_model, const float* obj_pose, int n_obj,
               const float K[4], const float cam[7],
               int min_points, float feature_distance, float min_score,
               float* score, uint8_t* keep, int32_t* out_order,
               int32_t* cl_members, int32_t* cl_off);

typedef struct {
  float ratio;                 /* 0.8 */
  float ms_radius, ms_merge;   /* 200, 20 */
  int ms_min_pts, ms_max_iter; /* 7, 100 */
  orc_pose_params pose1;       /* (600, 200, 4, 5, 6, 10) */
  int f1_min_points; float f1_feature_distance, f1_min_score; /* 5, 4096, 2 */
  orc_pose_params pose2;       /* (100, 500, 4, 6, 8, 5) */
  int f2_min_points; float f2_feature_distance, f2_min_score; /* 7, 4096, 3 */
  int run_stage2;
} orc_frame_params;

/* Everything after the nearest-neighbour search of one frame: the loop of
 * MopedPimpl::processImages (src/moped.cpp:184-191) over MATCH's ratio test +
 * scatter, CLUSTER, POSE, FILTER, POSE2, FILTER2 with the reference's
 * OpenMP structure.  Returns the number of final objects. */
int orc_frame_rest(const float* q_uv, const int32_t* idx1, const float* d1, const float* d2, int Q,
                   float ratio, const int32_t* model_of, const float* db_xyz, int n_models,
                   const float K[4], const float cam[7], const orc_frame_params* fp, int n_threads,
                   int32_t* obj_model, float* obj_pose, float* obj_score, int max_obj,
                   int32_t* counts);
/* orc_frame_rest + every final object's inlier set (testAllPoints of its final pose over its final cluster,
 * POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:166-180): inl_off [n_objects + 1], inl_q = rows of q_uv. */
int orc_frame_rest_inliers(const float* q_uv, const int32_t* idx1, const float* d1, const float* d2, int Q,
                           float ratio, const int32_t* model_of, const float* db_xyz, int n_models,
                           const float K[4], const float cam[7], const orc_frame_params* fp, int n_threads,
                           int32_t* obj_model, float* obj_pose, float* obj_score, int max_obj,
                           int32_t* counts, int32_t* inl_off, int32_t* inl_q, int inl_cap);

/* ---- frames with several images (cameras): every feature / match / correspondence carries its image.
 * Ks [n_images][4], cam_poses [n_images][7]; with one image these are the functions above. ---- */
int orc_project_test_images(const float pose7[7], const float* uv, const float* xyz, const int32_t* img, int n,
                            const float* Ks, const float* cam_poses, int n_images, float thr, uint8_t* inlier);
int orc_ransac_images(const float* uv, const float* xyz, const int32_t* img, int k, const float* Ks,
                      const float* cam_poses, int n_images, const orc_pose_params* prm, float pose7[7]);
int orc_filter_images(const float* uv, const int32_t* img, const float* xyz, const int32_t* model_off, int n_models,
                      const int32_t* obj_model, const float* obj_pose, int n_obj, const float* Ks,
                      const float* cam_poses, int n_images, int min_points, float feature_distance,
                      float min_score, float* score, uint8_t* keep, int32_t* out_order, int32_t* cl_members,
                      int32_t* cl_off);
int orc_frame_rest_images(const float* q_uv, const int32_t* q_img, const int32_t* idx1, const float* d1,
                          const float* d2, int Q, float ratio, const int32_t* model_of, const float* db_xyz,
                          int n_models, const float* Ks, const float* cam_poses, int n_images,
                          const orc_frame_params* fp, int32_t* obj_model, float* obj_pose, float* obj_score,
                          int max_obj, int32_t* counts);

/* N2  SIFT extraction as FEAT_SIFT_CPU runs it (feat/FEAT_SIFT_CPU.hpp:78-112 over libsiftfast
 * 1.1, plain-C arithmetic; see sift_oracle.cpp).  gray = h x w bytes; keypoints in the
 * reference's list order: xy[i] = (col, row), scale_ori[i] = (scale, orientation) (optional),
 * desc[i][128].  Returns the keypoint count (only `cap` are written). */
int orc_sift(const uint8_t* gray, int w, int h, int double_size, float* xy, float* scale_ori,
             float* desc, int cap);
/* One pyramid image of that run: kind 0 = Gaussian i, 1 = DoG i of `octave`. */
int orc_sift_image(const uint8_t* gray, int w, int h, int double_size, int octave, int kind, int i,
                   float* out, int* rows, int* cols);

/* moped3d's DEPTHFILL step, DEPTH_FILL_EXACT_CPU::fillInScaled
 * (moped3d/libmoped/src/depthfill/DEPTH_FILL_EXACT_CPU.hpp:268-349; see depthfill_oracle.cpp): depth [h][w][4]
 * (x, y, z, norm; z < 0 = hole) is filled in place, dist_out [h][w] gets the distance map the step appends to
 * the frame.  scale = the downscale factor (-1: chosen from the share of holes, :283-296); K = the depth map's
 * intrinsicLinearCalibration.  Returns the factor used, -1 on bad arguments. */
int orc_depth_fill(float* depth, int w, int h, int scale, int bilinear, const float K[4], float* dist_out);

#ifdef __cplusplus
}
#endif
