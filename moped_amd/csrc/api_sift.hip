// C ABI: SIFT extraction (FEAT_SIFT_CPU's job).
#include <algorithm>
#include <vector>

#include "context.h"
#include "sift.h"

using namespace mh;

struct SiftState {
  int width = 0, height = 0, double_size = -1, cap = 0, images = 1;
  SiftPlan plan;
  SiftBuffers B = {};
  uint8_t* gray = nullptr;       // staging for host-pointer calls
  float *desc = nullptr, *xy = nullptr, *scale_ori = nullptr;
  int32_t* n_dev = nullptr;
  unsigned int own_epoch = 0;    // B.own_epoch points here
};

namespace {

void free_sift(SiftState* st) {
  if (!st) return;
  void* ptrs[] = {st->B.pyramid, st->B.tmp, st->B.owner, st->B.cand, st->B.keys, st->B.counters,
                  st->gray,      st->desc,  st->xy,      st->scale_ori, st->n_dev};
  for (void* p : ptrs)
    if (p) hipFree(p);
  delete st;
}

template <typename T>
int alloc(mh_ctx* ctx, T*& p, size_t n) {
  MH_HIP(ctx, hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
  return MH_OK;
}

// images > 1: room for a batch's images side by side (mh_frame_enqueue_image_batch: one launch per stage for all of them)
int ensure_sift(mh_ctx* ctx, int width, int height, int double_size, int cap, int images = 1) {
  SiftState* st = ctx->sift;
  // internal keypoint room is never below 8192, so that a small output capacity still selects
  // the FIRST keypoints of the reference's list (which are the last ones generated)
  cap = std::max(cap, 8192);
  if (st && st->width == width && st->height == height && st->double_size == double_size && st->cap >= cap &&
      st->images >= images)
    return MH_OK;
  if (st) {
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    cap = std::max(cap, st->cap);
    images = std::max(images, st->images);
    free_sift(st);
    ctx->sift = nullptr;
  }
  st = new SiftState;
  ctx->sift = st;
  st->width = width;
  st->height = height;
  st->double_size = double_size;
  st->cap = cap;
  st->images = images;
  if (sift_plan(width, height, double_size, &st->plan) == 0) {
    ctx->err = "mh_sift: image too small (both sides must exceed 12 pixels after scaling)";
    free_sift(st);   // no half-built state: the next call with this geometry fails the same way
    ctx->sift = nullptr;
    return MH_ERR_ARG;
  }
  size_t owner = 0;
  for (int o = 0; o < st->plan.n_octaves; ++o) owner += (size_t)st->plan.rows[o] * st->plan.cols[o];
  int rc = 0;
  const size_t ni = (size_t)images;
  rc |= alloc(ctx, st->B.pyramid, st->plan.floats * ni);
  rc |= alloc(ctx, st->B.tmp, (size_t)st->plan.rows0 * st->plan.cols0 * ni);
  rc |= alloc(ctx, st->B.owner, owner * ni);
  st->B.owner_elems = owner;
  st->B.own_epoch = &st->own_epoch;
  st->B.cand_cap = 4 * cap;
  st->B.key_cap = cap;
  st->B.images = images;
  rc |= alloc(ctx, st->B.cand, (size_t)st->B.cand_cap * ni);
  rc |= alloc(ctx, st->B.keys, (size_t)cap * ni);
  rc |= alloc(ctx, st->B.counters, 4 * ni);
  rc |= alloc(ctx, st->gray, (size_t)width * height);
  rc |= alloc(ctx, st->desc, (size_t)cap * 128);
  rc |= alloc(ctx, st->xy, (size_t)cap * 2);
  rc |= alloc(ctx, st->scale_ori, (size_t)cap * 2);
  rc |= alloc(ctx, st->n_dev, 1);
  if (rc) {
    free_sift(st);
    ctx->sift = nullptr;
    return MH_ERR_HIP;
  }
  return MH_OK;
}

}  // namespace

namespace mh {

// SIFT of a device image into caller-chosen device buffers, on the context's stream;
// *n_dev_out = the context's device word holding the keypoint count afterwards.
int sift_into(mh_ctx* ctx, const uint8_t* gray_dev, int width, int height, int double_size, int cap,
              float* desc_dev, float* xy_dev, int32_t** n_dev_out, int32_t* count_word) {
  int rc = ensure_sift(ctx, width, height, double_size ? 1 : 0, cap);
  if (rc) return rc;
  SiftState* st = ctx->sift;
  int32_t* const n_dev = count_word ? count_word : st->n_dev;   // (a batch of images keeps every image's count)
  launch_sift(gray_dev, width, height, double_size ? 1 : 0, st->plan, st->B, cap, desc_dev, xy_dev, nullptr,
              n_dev, ctx->stream);
  MH_HIP(ctx, hipGetLastError());
  if (n_dev_out) *n_dev_out = n_dev;
  return MH_OK;
}

// n images of one size into the rows of a batch: image i's keypoints at desc_dev + i cap 128 / xy_dev + i cap 2, its count
// in count_words[i]; one launch per stage for all of them.
int sift_into_batch(mh_ctx* ctx, const uint8_t* const* gray_dev, int n, int width, int height, int double_size, int cap,
                    float* desc_dev, float* xy_dev, int32_t* count_words) {
  int rc = ensure_sift(ctx, width, height, double_size ? 1 : 0, cap, n);
  if (rc) return rc;
  SiftState* st = ctx->sift;
  launch_sift_batch(gray_dev, n, width, height, double_size ? 1 : 0, st->plan, st->B, cap, cap, desc_dev, xy_dev, nullptr,
                    count_words, 1, ctx->stream);
  MH_HIP(ctx, hipGetLastError());
  return MH_OK;
}

}  // namespace mh

extern "C" {

void mh_free_sift_state(mh_ctx* ctx) {
  free_sift(ctx->sift);
  ctx->sift = nullptr;
}

int mh_sift_extract_dev(mh_ctx* ctx, const uint8_t* gray_dev, int width, int height, int double_size,
                        float* desc_dev, float* xy_dev, float* scale_ori_dev, int cap, int32_t* n_dev) {
  if (!ctx || !gray_dev || width <= 0 || height <= 0 || !desc_dev || !xy_dev || cap <= 0 || !n_dev) {
    if (ctx) ctx->err = "mh_sift_extract_dev: bad argument";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  int rc = ensure_sift(ctx, width, height, double_size ? 1 : 0, cap);
  if (rc) return rc;
  SiftState* st = ctx->sift;
  launch_sift(gray_dev, width, height, double_size ? 1 : 0, st->plan, st->B, cap, desc_dev, xy_dev, scale_ori_dev,
              n_dev, ctx->stream);
  MH_HIP(ctx, hipGetLastError());
  return MH_OK;
}

int mh_sift_extract(mh_ctx* ctx, const uint8_t* gray_host, int width, int height, int double_size, float* xy_host,
                    float* scale_ori_host, float* desc_host, int cap, int32_t* n_keypoints) {
  if (!ctx || !gray_host || width <= 0 || height <= 0 || !xy_host || !desc_host || cap <= 0 || !n_keypoints) {
    if (ctx) ctx->err = "mh_sift_extract: bad argument";
    return MH_ERR_ARG;
  }
  *n_keypoints = 0;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc_stream = mh::use_stream(ctx)) return rc_stream;
  int rc = ensure_sift(ctx, width, height, double_size ? 1 : 0, cap);
  if (rc) return rc;
  SiftState* st = ctx->sift;
  hipStream_t s = ctx->stream;
  MH_HIP(ctx, hipMemcpyAsync(st->gray, gray_host, (size_t)width * height, hipMemcpyHostToDevice, s));
  launch_sift(st->gray, width, height, double_size ? 1 : 0, st->plan, st->B, st->cap, st->desc, st->xy, st->scale_ori,
              st->n_dev, s);
  MH_HIP(ctx, hipGetLastError());
  int32_t head[4] = {0, 0, 0, 0};
  int32_t n = 0;
  MH_HIP(ctx, hipMemcpyAsync(&n, st->n_dev, sizeof n, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipMemcpyAsync(head, st->B.counters, sizeof head, hipMemcpyDeviceToHost, s));
  MH_HIP(ctx, hipStreamSynchronize(s));
  const int take = std::min(n, cap);
  if (take > 0) {
    MH_HIP(ctx, hipMemcpyAsync(desc_host, st->desc, (size_t)take * 128 * sizeof(float), hipMemcpyDeviceToHost, s));
    MH_HIP(ctx, hipMemcpyAsync(xy_host, st->xy, (size_t)take * 2 * sizeof(float), hipMemcpyDeviceToHost, s));
    if (scale_ori_host)
      MH_HIP(ctx, hipMemcpyAsync(scale_ori_host, st->scale_ori, (size_t)take * 2 * sizeof(float), hipMemcpyDeviceToHost, s));
    MH_HIP(ctx, hipStreamSynchronize(s));
  }
  *n_keypoints = take;
  if (head[2] || n > cap) {
    ctx->err = "mh_sift_extract: more keypoints than the capacity given (" + std::to_string(head[1]) + " found)";
    return MH_ERR_CAPACITY;
  }
  return MH_OK;
}

}  // extern "C"
