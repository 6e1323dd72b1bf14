// mh_ctx: device state kept across frames (the analogue of the kd-tree
// MATCH_ANN_CPU keeps between frames, MATCH_ANN_CPU.hpp:70) plus the per-frame
// device buffers.  Everything lives in HBM; host pointers only at the C ABI.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "screen.h"
#include "steps.h"

// The model database in HBM.  Contexts (= frames in flight) of one device may share one store
// (mh_db_share): one upload and one copy per GPU instead of one per frame slot, and one set of lines in
// the Infinity Cache for all frames in flight.  A store that is shared is never resized or rewritten:
// an upload into a context whose store has other users gives that context a fresh store.
struct DbStore {
  int device = 0;
  int N = 0, n_models = 0;
  int32_t index_base = 0;
  int n_blocks = 0;            // > 1: the rows are n_blocks runs of global rows (mh_db_upload_blocks; RowMap, common.h)
  int32_t* blk_glo = nullptr;  // device [n_blocks]: first global row of each block
  int32_t* blk_llo = nullptr;  // device [n_blocks + 1]: first local row of each block, N at the end
  float* desc = nullptr;       // [rows padded to 128][128], zero padding rows
  float* norm = nullptr;       // [padded] dot(d,d), +inf on padding rows
  float* xyz = nullptr;        // [padded][3]
  int32_t* model = nullptr;    // [padded]
  size_t cap = 0;              // rows allocated
  _Float16* desc_h = nullptr;  // f16 image for the screen (match_screen.hip)
  float* neg_h = nullptr;      // [tiles][192] -norm/2 per row + row-block extrema (screen.h)
  size_t cap_h = 0;            // elements allocated
  unsigned int* stats = nullptr;
  mh::ScreenDb screen;
  ~DbStore() {
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    hipSetDevice(device);
    for (void* p : {(void*)desc, (void*)norm, (void*)xyz, (void*)model, (void*)desc_h, (void*)neg_h, (void*)stats, (void*)blk_glo, (void*)blk_llo})
      if (p) hipFree(p);
    if (have) hipSetDevice(cur);
  }
};

// A lane: a few streams for the kernels whose workgroups take whole compute units (passes A and B of the two-stage
// MATCH), either confined to a CU mask that leaves `reserve` units of every XCD to everything else, or of lower priority
// than the contexts' own streams.  Shared by the contexts (= frames in flight) of a device (mh_lane_create).
struct mh_lane {
  int device = 0;
  std::vector<hipStream_t> streams;
  unsigned next = 0;   // round-robin hand-out to contexts
  int reserve = 0, low_priority = 0;
};

struct mh_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;

  // ---- model database (resident; the arrays belong to `store`, these are its pointers) ----
  std::shared_ptr<DbStore> store;
  int N = 0, n_models = 0;
  int32_t index_base = 0;
  mh::RowMap rmap;               // local row <-> global row of the store (one block at index_base unless uploaded in blocks)
  float* db_desc = nullptr;      // [N][128]
  float* db_norm = nullptr;      // [N]
  float* db_xyz = nullptr;       // [N][3]
  int32_t* db_model = nullptr;   // [N]
  mh::ScreenDb sdb;
  mh::ScreenBufs sbuf;           // the screen's per-frame scratch
  int match_mode = -1;           // mh_match_set_mode
  int pose_split = 1;            // mh_pose_set_split: POSE as two launches (hypotheses, one-wavefront refines) in the frame paths
  uint32_t match_launches[3] = {0, 0, 0};   // MATCH launch sequences by kernel: VALU, f32 matrix pipe, two-stage (mh_match_launches)
  struct mh_lane* lane = nullptr;   // mh_set_lane: where the chip-filling MATCH passes run (not owned)
  hipStream_t lane_stream = nullptr;
  hipEvent_t lane_in = nullptr, lane_out = nullptr;

  // ---- per-frame buffers ----
  int max_q = 0, max_clusters = 0, max_objects = 0;
  float* q_desc = nullptr;       // [max_q][128] staging for host-pointer entry points
  float* q_norm = nullptr;       // [max_q]
  float* q_uv = nullptr;         // [max_q][2]
  int32_t* nn_idx = nullptr;     // [max_q]
  float* nn_d1 = nullptr;        // [max_q]
  float* nn_d2 = nullptr;        // [max_q]
  mh::Top2* match_scratch = nullptr;
  size_t match_scratch_cap = 0;
  float* match_pack = nullptr;   // queries re-laid out for scalar loads (match.hip)
  size_t match_pack_cap = 0;

  // generic byte scratch for host-pointer entry points
  void* scratch = nullptr;
  size_t scratch_cap = 0;
  void* pinned = nullptr;
  size_t pinned_cap = 0;
  // mh_frame_run_host's write-back of the normalised descriptors: a stream of its own behind an event recorded right
  // after normalize_kernel, so that the 1.5 MB copy runs beside MATCH .. FILTER2 instead of after them
  hipStream_t wb_stream = nullptr;
  hipEvent_t wb_ev = nullptr;
  bool wb_want = false;   // the frame being enqueued records wb_ev after its normalisation
  bool wb_pending = false;   // a write-back is in flight on wb_stream (mh_frame_wait_descriptors)

  // mh_step_*: the frame's six slots one call each.  stage_lo / stage_hi: what the next frame_rest launches (0 .. 5 =
  // everything); step: where the resident frame stands and what the host needs to know to read its results back.
  int stage_lo = 0, stage_hi = 5;
  struct StepState {
    int done = -1;          // last stage run on the resident frame (-1: none / invalidated)
    int Q = 0, M = 0;       // queries, accepted matches
    int n_clusters = 0;     // clusters POSE / POSE2 will work on (CLUSTER's, then FILTER's)
    int n_slots = 0;        // object slots in use
    mh_cam cam;
    std::vector<int32_t> model_off;   // host copy of the per-model offsets into the match list
    std::vector<int32_t> valid;       // slots that hold an object, ascending = the host's list order
    std::vector<int32_t> valid_model; // ... and the model of each
  } step;

  // frame state (group / cluster / pose / filter); defined in frame.h
  struct FrameState* fs = nullptr;
  struct SiftState* sift = nullptr;   // pyramid + keypoint buffers of the SIFT extractor (api_sift.hip)
  int32_t* feat_count_dev = nullptr;  // frame enqueued from an image: device word with its keypoint count
  int32_t* img_counts = nullptr;      // mh_frame_enqueue_image_batch: [MH_MAX_BATCH] keypoint counts of the batch's images (device)
  int feat_expected = 0;              // keypoints of the last fetched image frame (sizes the next MATCH launch)
  int feat_last = -1;
  int batch_q0 = 0;                   // first query of the frame frame_rest works on (mh_frame_enqueue_batch)
  int batch_f = 0;                    // ... and its number in the batch: the frame's slice of per-query attributes that lie frame after frame (q_img)
  const float4* batch_img[MH_MAX_BATCH] = {};   // depth maps of the frames of a batch (mh_frame_set_depth_image_batch)
  const float* batch_fill[MH_MAX_BATCH] = {};
  int batch_imgs = 0;                 // how many of them are set (0: one depth map, one frame)
  int exchange_plane = 0;             // words between the idx / d1 / d2 planes of one shard's block (0 = Q)
  int exchange_stride = 0;            // words between the shards' blocks of the gathered exchange buffer (0 = 3 Q)
  const int32_t* exchange_tags = nullptr;   // comm.hip: shard 0's tag words in the gathered buffer (checked by the frame's first launch)

  // N > 1 (comm.hip): the send / receive blocks of the frame exchange, the flush buffer of the last frames
  struct Exchange {
    int32_t* local = nullptr;      // [3][B Q] top-2 words + B result heads
    int32_t* gathered = nullptr;   // world of those
    size_t cap_local = 0, cap_gather = 0, stride = 0;   // words
    int world = 0, batch = 0, bq = 0;
    unsigned char* flush = nullptr;
    size_t flush_bytes = 0;
    std::vector<int32_t> host;
  } ex;

  // frames with several images (mh_frame_set_images): image of every query + the cameras; n_images == 1 = off
  const int32_t* q_img = nullptr;     // device, [Q]
  int32_t* hf_img = nullptr;          // mh_frame_run_host's copy of the per-query image indices
  int hf_img_cap = 0;
  mh::DevCam* cams_dev = nullptr;     // device, [MH_MAX_IMAGES]
  int n_images = 1;

  // optional depth attributes of the current queries (moped3d residuals)
  const mh_depth* q_depth = nullptr;
  int depth_kind = 0;
  float depth_alpha = 0.5f;
  mh::DepthImage depth_img;      // or: the depth map itself, looked up per match on the device (DEPTHMAP_PROP)

  // moped3d depth rules (mh_frame_set_depth_rules): DEPTHFILTER / DEPTHFILTER2 / adaptive ratio
  struct DepthRuleState {
    bool on = false;
    int patch = 64;
    float feature_filter = -1.f, match_filter = -1.f;   // Density * 100 * 100; < 0 = off
    float K[4] = {0, 0, 0, 0};
    float max_depth = 4.f, default_depth = 1.f, cauchy_scale = 0.1f;
    float* ratio_table = nullptr;   // device [n_models][4] or nullptr
    int table_models = 0;
    double* inv_size = nullptr;     // device [patches]
    int patches_cap = 0;
    int32_t* cnt = nullptr;         // device [n_models][patches], zero between frames
    size_t cnt_cap = 0;
    uint8_t* keep1 = nullptr;       // device [max_q]
    int keep_cap = 0;
  } rules;

  // moped3d CLUSTER_LINKAGE instead of mean shift (mh_frame_set_cluster_linkage)
  bool linkage_on = false;
  mh::LinkageParams linkage;
  float* own_depth = nullptr;     // device copies of a host depth / distance map (mh_frame_set_depth_image_host)
  float* own_fill = nullptr;
  size_t own_depth_px = 0;
  float* lk_scratch = nullptr;
  unsigned char* df_buf = nullptr; // mh_depth_fill: [status words | downscaled depths | downscaled distances]
  size_t lk_scratch_floats = 0;
  size_t lk_scratch_limit = (size_t)4 << 30;   // bytes; mh_set_linkage_scratch_limit

  // mh_frame_fetch_batch_async / mh_frame_fetch_previous_async: delivery of a batch's objects into the caller's pinned block
  struct Delivery {
    hipEvent_t done = nullptr;          // recorded behind the delivery on the context's stream
    bool pending = false;
    unsigned char* stage = nullptr;     // device staging for blocks the device cannot write directly
    size_t stage_cap = 0;
    unsigned char* host_block = nullptr;   // the pending delivery's destination, as the caller knows it
    int B = 0, max_objects = 0;
    uint32_t tag = 0;
  } dlv;

  bool timing = false;
  hipEvent_t ev[10] = {};
  static constexpr int MEV_SETS = 32;
  hipEvent_t mev[MEV_SETS][6] = {};   // around the kernels of the two-stage MATCH, one set per launch sequence (mh_match_timing)
  int mev_next = 0, mev_used = 0;     // ring position, sets recorded since the last mh_match_timing
  bool ev_made = false;
};

namespace mh {

void free_exchange(mh_ctx* ctx);   // comm.hip

#define MH_HIP(ctx, call)                                                         \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);             \
      return MH_ERR_HIP;                                                          \
    }                                                                             \
  } while (0)

int use_stream(mh_ctx* ctx);  // make sure ctx->stream is valid (creates the own stream lazily)
int ensure_frame_buffers(mh_ctx* ctx, int Q);
int ensure_scratch(mh_ctx* ctx, size_t bytes);
int ensure_pinned(mh_ctx* ctx, size_t bytes);
int ensure_match_scratch(mh_ctx* ctx, int Q);
// MATCH of Q normalised queries against the context's DB on its stream: exact (idx1, d1, d2) per query, by the
// two-stage screen when it pays and the DB allows it, else by the exact kernels (bit-identical results either way).
int ctx_match(mh_ctx* ctx, const float* qn, const float* qnorm, int Q, int32_t* idx1, float* d1, float* d2,
              const int32_t* q_count = nullptr, int q_expected = 0);
// Delivery of a batch's heads into a caller's host block (api_steps.hip): delivery_begin -> where the delivering kernel
// writes (the block itself when the device can address it, else the context's staging buffer); delivery_end -> the copy
// out of the staging buffer if one is needed, and the event behind it all.
int delivery_begin(mh_ctx* ctx, void* host_block, size_t bytes, unsigned char** dst_dev);
int delivery_end(mh_ctx* ctx, void* host_block, size_t bytes, unsigned char* dst_dev, int B, int max_objects, uint32_t tag);
int sift_into(mh_ctx* ctx, const uint8_t* gray_dev, int width, int height, int double_size, int cap,
              float* desc_dev, float* xy_dev, int32_t** n_dev_out, int32_t* count_word = nullptr);
int sift_into_batch(mh_ctx* ctx, const uint8_t* const* gray_dev, int n, int width, int height, int double_size, int cap,
                    float* desc_dev, float* xy_dev, int32_t* count_words);

}  // namespace mh
