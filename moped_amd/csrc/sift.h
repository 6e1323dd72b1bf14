// SIFT extraction (sift.hip): pyramid layout and launch entry.
#pragma once
#include "common.h"

namespace mh {

constexpr int SIFT_MAX_OCTAVES = 8;
constexpr int SIFT_IMAGES_PER_OCTAVE = 11;  // 6 Gaussian + 5 DoG (gradient magnitude / orientation: computed per sample)

struct SiftOctave {
  int rows, cols;
  float fscale;             // octave pixel -> original image pixel (0.5 for the doubled image, :318)
  float* gaus[6];
  float* dog[5];
  unsigned int* owner;      // per pixel: epoch prefix | generation key of the extremum that owns it (SiftBatch::own_prefix)
};

struct SiftPyramid {
  int n_octaves;
  SiftOctave oct[SIFT_MAX_OCTAVES];
};

struct SiftCandidate {      // a refined extremum (InterpKeyPoint's result)
  int octave, index;
  unsigned int key;         // (index-1) * rows*cols + scan position of the pixel it started from
  int r, c;                 // final pixel
  float x0, x1, x2;         // offsets in scale, row, column
};

struct SiftKey {            // one (extremum, orientation peak)
  int octave, index;
  unsigned long long order; // generation order: octave, scale index, scan position, histogram bin
  float fsize, frow, fcol, ori;
};

// The images of a batch through ONE launch per stage (blockIdx.y / z = image): image i's working arrays lie `*_step`
// elements behind image i - 1's (SiftBuffers holds room for `images` of them), its outputs `out_step` keypoints and
// `n_out_step` count words behind.
struct SiftBatch {
  unsigned long long pyr_step = 0, own_step = 0, tmp_step = 0;   // floats / words
  int cand_step = 0, key_step = 0;                               // candidates / keys (counters: 4 words per image)
  int out_step = 0, n_out_step = 0;
  int n = 1;
  // The owner map is never cleared between images: a claim is `own_prefix | key` with a prefix that goes DOWN from image
  // to image, claims go in with atomicMin, so every claim of this image lies below everything older; a word whose prefix
  // is not this image's is free.  (When the prefixes run out -- every 2^(32 - key bits) images -- the map is filled with
  // ones again.)
  unsigned int own_prefix = 0;
};
struct SiftImages {
  const uint8_t* gray[MH_MAX_BATCH];
};

struct SiftPlan {
  int n_octaves;
  int rows0, cols0;
  int rows[SIFT_MAX_OCTAVES], cols[SIFT_MAX_OCTAVES];
  float fscale[SIFT_MAX_OCTAVES];
  size_t floats;            // pyramid size
};

struct SiftBuffers {
  float* pyramid;           // plan.floats
  float* tmp;               // rows0 * cols0 (row-blurred image)
  unsigned int* owner;      // sum of rows*cols over the octaves
  size_t owner_elems;
  SiftCandidate* cand;
  int cand_cap;
  SiftKey* keys;
  int key_cap;
  int32_t* counters;        // [4]: candidates, keys, overflow flag, -
  int images;               // every array above holds this many images' worth, one after the other
  unsigned int* own_epoch;  // host word: the prefix counter of the owner map (SiftBatch::own_prefix); starts at 0 = fill first
};

// Octave sizes of GetKeypoints' loop (:344-348).  Returns the number of octaves.
int sift_plan(int width, int height, int double_size, SiftPlan* plan);

// Everything on stream s.  desc_out [out_cap][128], xy_out [out_cap][2] = (col,row),
// scale_ori_out (optional) [out_cap][2], *n_out = number of keypoints written, all on the device,
// in the reference's list order.
void launch_sift(const uint8_t* gray, int width, int height, int double_size, const SiftPlan& plan,
                 const SiftBuffers& B, int out_cap, float* desc_out, float* xy_out, float* scale_ori_out,
                 int32_t* n_out, hipStream_t s);
// The same for n <= B.images images of one size in ONE launch per stage: image i's keypoints at desc_out + i out_step
// 128 (xy_out, scale_ori_out: + i out_step 2), its count at n_out + i n_out_step; out_cap <= out_step.
void launch_sift_batch(const uint8_t* const* gray, int n, int width, int height, int double_size, const SiftPlan& plan,
                       const SiftBuffers& B, int out_cap, int out_step, float* desc_out, float* xy_out,
                       float* scale_ori_out, int32_t* n_out, int n_out_step, hipStream_t s);

}  // namespace mh
