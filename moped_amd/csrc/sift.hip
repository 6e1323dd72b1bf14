// FEAT: SIFT keypoints + descriptors on gfx950.
//
// Replaces FEAT_SIFT_CPU::process (moped2/libmoped/src/feat/FEAT_SIFT_CPU.hpp:78-112), i.e.
// libsiftfast 1.1's GetKeypoints (libs.tgz -> libsiftfast-1.1-src/libsiftfast.cpp; line
// numbers below are that file's) in the plain-C arithmetic MOPED builds it with (:39-41).
// Same operations in the same order as oracle/sift_oracle.cpp -- separable Gaussian taps
// added in ascending order in fp32, edge replication, the (ksize+1)-term kernel sum (:492-501),
// Gauss-Jordan with row pivoting for the 3x3 fit, per-bin accumulation in raster order of the
// samples -- so the only differences to the oracle come from the device's expf / atan2f /
// sinf / cosf / powf (a few ulp).  The Gaussian kernels are computed on the host with libm.
//
// Layout: the WHOLE pyramid stays resident (14 images per octave, all octaves: ~90 MB for a
// doubled 640x480 frame), so everything after the blur chain runs ONCE over all octaves:
//   prepare (u8 -> [0,1] + 2x upsample)            1 launch
//   per octave: 5 x (row blur, column blur + DoG)   sequential by construction
//               + 2:1 subsample for the next octave
//   grad_ori      all octaves x 3 scale indices     1 launch
//   detect        extrema + edge test + quadratic fit, first-claim of the final pixel
//   orient        one wavefront per surviving extremum: 36-bin histogram, peaks
//   describe      one wavefront per (extremum, peak): 4x4x8 descriptor
//   order         rank by generation key -> the reference's list order
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "sift.h"

namespace mh {

namespace {

constexpr float kPi = 3.141592654f;   // :65
constexpr float kSqrt2 = 1.4142136f;  // :66
constexpr int kScales = 3;            // :108
constexpr float kInitSigma = 1.6f;    // :109

// ---- prepare ----------------------------------------------------------------------------
// FEAT_SIFT_CPU.hpp:91 (pixel * 1./255. in double) and SiftDoubleSize (:363-380).
__device__ __forceinline__ float to_unit(uint8_t g) { return (float)((double)(float)g * 1. / 255.); }

// blockIdx.z = image of a batch (SiftImages: its pixels; its output `out_step` floats behind the image's before it)
__global__ void prepare_kernel(const uint8_t* __restrict__ gray, int w, int h, int double_size,
                               float* __restrict__ out, int orows, int ocols, SiftImages imgs, size_t out_step) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= ocols || r >= orows) return;
  if (blockIdx.z) {
    gray = imgs.gray[blockIdx.z];
    out += blockIdx.z * out_step;
  }
  if (!double_size) {
    out[(size_t)r * ocols + c] = to_unit(gray[(size_t)r * w + c]);
    return;
  }
  const int i = r >> 1, j = c >> 1;
  const float a = to_unit(gray[(size_t)i * w + j]);
  float v;
  if ((r & 1) == 0 && (c & 1) == 0) {
    v = a;
  } else if ((r & 1) == 1 && (c & 1) == 0) {
    v = __fmul_rn(0.5f, __fadd_rn(a, to_unit(gray[(size_t)(i + 1) * w + j])));
  } else if ((r & 1) == 0) {
    v = __fmul_rn(0.5f, __fadd_rn(a, to_unit(gray[(size_t)i * w + j + 1])));
  } else {
    const float b = to_unit(gray[(size_t)i * w + j + 1]);
    const float d = to_unit(gray[(size_t)(i + 1) * w + j]);
    const float e = to_unit(gray[(size_t)(i + 1) * w + j + 1]);
    v = __fmul_rn(0.25f, __fadd_rn(__fadd_rn(__fadd_rn(a, b), d), e));
  }
  out[(size_t)r * ocols + c] = v;
}

// ---- blur ---------------------------------------------------------------------------------
// ConvHorizontal / ConvVertical (:523-584): replicated edges, taps in ascending order.
constexpr int MAX_TAPS = 64;
struct Taps {
  float k[MAX_TAPS];
  int n;
};

__global__ void blur_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols,
                                 Taps t) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= cols) return;
  const float* row = src + (size_t)r * cols;
  const int w = t.n >> 1;
  float a = 0.f;
  for (int j = 0; j < t.n; ++j) {
    int x = c + j - w;
    x = x < 0 ? 0 : (x >= cols ? cols - 1 : x);
    a = __fadd_rn(a, __fmul_rn(row[x], t.k[j]));
  }
  dst[(size_t)r * cols + c] = a;
}

// column pass; optionally also writes dog = prev - blurred (SubtractImage, :440-466)
__global__ void blur_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols,
                                 Taps t, const float* __restrict__ prev, float* __restrict__ dog) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= cols) return;
  const int w = t.n >> 1;
  float a = 0.f;
  for (int j = 0; j < t.n; ++j) {
    int y = r + j - w;
    y = y < 0 ? 0 : (y >= rows ? rows - 1 : y);
    a = __fadd_rn(a, __fmul_rn(src[(size_t)y * cols + c], t.k[j]));
  }
  const size_t o = (size_t)r * cols + c;
  dst[o] = a;
  if (dog) dog[o] = __fsub_rn(prev[o], a);
}

// Both passes of one pyramid level in one launch: a workgroup blurs a 64 x 32 tile out of LDS -- the tile and a
// halo of w pixels (replicated edges applied while loading), the row pass over the tile's rows and the halo rows
// above and below, then the column pass.  Every output pixel sees the same products added in the same order as
// with the two kernels above (taps ascending, the intermediate row-blurred values are the same numbers).
constexpr int BT_X = 64, BT_Y = 32, BT_MAXW = 16, BT_THREADS = 256;
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(BT_THREADS) void blur_level_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                int rows, int cols, Taps t,
                                                                const float* __restrict__ prev, float* __restrict__ dog) {
  __shared__ float in_s[(BT_Y + 2 * BT_MAXW) * (BT_X + 2 * BT_MAXW)];
  __shared__ __attribute__((aligned(8))) float row_s[(BT_Y + 2 * BT_MAXW) * BT_X];
  const int tid = threadIdx.x;
  const int w = t.n >> 1;
  const int c0 = blockIdx.x * BT_X, r0 = blockIdx.y * BT_Y;
  const int in_w = BT_X + 2 * w, in_h = BT_Y + 2 * w;
  for (int e = tid; e < in_w * in_h; e += BT_THREADS) {
    const int yy = e / in_w, xx = e - yy * in_w;
    int y = r0 - w + yy, x = c0 - w + xx;
    y = y < 0 ? 0 : (y >= rows ? rows - 1 : y);
    x = x < 0 ? 0 : (x >= cols ? cols - 1 : x);
    in_s[e] = src[(size_t)y * cols + x];
  }
  __syncthreads();
  // two outputs per packed instruction (the products and sums round like the scalar ones): the row pass pairs
  // two rows of the same column, the column pass two neighbouring columns (one 8-byte LDS read per tap)
  for (int e = tid; e < (in_h >> 1) * BT_X; e += BT_THREADS) {   // in_h is even
    const int yy = 2 * (e / BT_X), x = e % BT_X;
    const float* r0p = in_s + yy * in_w + x;
    const float* r1p = r0p + in_w;
    v2f a = {0.f, 0.f};
    for (int j = 0; j < t.n; ++j) a = a + v2f{r0p[j], r1p[j]} * v2f{t.k[j], t.k[j]};
    row_s[yy * BT_X + x] = a.x;
    row_s[(yy + 1) * BT_X + x] = a.y;
  }
  __syncthreads();
  for (int e = tid; e < BT_Y * (BT_X / 2); e += BT_THREADS) {
    const int y = e / (BT_X / 2), x = 2 * (e % (BT_X / 2));
    const int r = r0 + y, c = c0 + x;
    if (r >= rows || c >= cols) continue;
    const float* col = row_s + y * BT_X + x;
    v2f a = {0.f, 0.f};
    for (int j = 0; j < t.n; ++j) a = a + *reinterpret_cast<const v2f*>(col + j * BT_X) * v2f{t.k[j], t.k[j]};
    const size_t o = (size_t)r * cols + c;
    dst[o] = a.x;
    if (dog) dog[o] = __fsub_rn(prev[o], a.x);
    if (c + 1 < cols) {
      dst[o + 1] = a.y;
      if (dog) dog[o + 1] = __fsub_rn(prev[o + 1], a.y);
    }
  }
}

// The same tile blur for up to two independent pyramid levels in one launch (a job list): the levels of an octave
// depend on each other, but octave o + 1 starts from level kScales of octave o, so levels kScales + 1, kScales + 2 of
// octave o and levels 1, 2 of octave o + 1 are pairwise independent -- 13 launches for four octaves instead of 24.
// A job whose source is the previous octave (`half`) reads it at every second row and column (HalfImageSize,
// :390-408), writes those pixels out as its octave's level 0 and blurs them: the same numbers the half kernel followed
// by the blur kernel produces.  dog = source - blurred (SubtractImage, :440-466) comes from the tile in LDS.
struct BlurJob {
  const float* src;
  float* dst;
  float* dog;
  float* half_dst;   // level 0 of the job's octave when `half`
  int rows, cols, src_cols, half, taps, tiles_x, tile_begin;
  size_t src_step, pyr_step;   // images of a batch (blockIdx.y): floats from an image's source / pyramid to the next one's
};
struct BlurJobs {
  BlurJob j[2];
  Taps t[2];
  int n;
};
// The tile blur with the tap count known at compile time: every thread keeps a window of the source in registers and
// produces several neighbouring outputs from it (8 along a row for two rows at once, 4 down a column for two columns),
// so a pixel is read from LDS once per 8 (4) outputs instead of once per output and tap -- the loop-over-taps version
// below is bound by its 46 k LDS reads per tile.  Per output the products and sums are the ones of ConvHorizontal /
// ConvVertical in the same (ascending tap) order.  The source tile sits in LDS with rows interleaved in pairs
// ([row / 2][column][row % 2]): one 16-byte read gives two columns of two rows, laid out as the operands of the packed
// multiply and add.
template <int N>
__device__ __forceinline__ void blur_tile_fixed(const BlurJob& job, const float* __restrict__ k, float* in_s, float* row_s,
                                                int r0, int c0, int tid) {
  constexpr int W = N / 2, IN_W = BT_X + 2 * W, IN_H = BT_Y + 2 * W, IN_WP = (IN_W + 3) & ~3;
  static_assert((IN_H & 1) == 0 && IN_H * IN_WP <= (BT_Y + 2 * BT_MAXW) * (BT_X + 2 * BT_MAXW), "tile buffer");
  const int rows = job.rows, cols = job.cols;
  const int step = job.half ? 2 : 1;
  // staging: 16 x 16 threads over rows x columns (replicated edges; a `half` job reads every second row and column
  // of the previous octave and writes its own pixels out as level 0)
  for (int yy = tid >> 4; yy < IN_H; yy += 16) {
    const int yu = r0 - W + yy;
    const int y = yu < 0 ? 0 : (yu >= rows ? rows - 1 : yu);
    const float* srow = job.src + (size_t)(step * y) * job.src_cols;
    float* drow = in_s + (yy >> 1) * (2 * IN_WP) + (yy & 1);
    for (int xx = tid & 15; xx < IN_W; xx += 16) {
      const int xu = c0 - W + xx;
      const int x = xu < 0 ? 0 : (xu >= cols ? cols - 1 : xu);
      const float v = srow[step * x];
      drow[2 * xx] = v;
      if (job.half && yu == y && xu == x && yy >= W && yy < W + BT_Y && xx >= W && xx < W + BT_X)
        job.half_dst[(size_t)y * cols + x] = v;
    }
  }
  __syncthreads();
  // row pass: rows 2p, 2p + 1, outputs x0 .. x0 + 7
  if (tid < (IN_H / 2) * 8) {
    const int p = tid >> 3, x0 = 8 * (tid & 7);
    const float4* src = reinterpret_cast<const float4*>(in_s + p * (2 * IN_WP) + 2 * x0);
    v2f win[N + 7];
#pragma unroll
    for (int i = 0; i < (N + 7) / 2; ++i) {
      const float4 q = src[i];
      win[2 * i] = v2f{q.x, q.y};
      win[2 * i + 1] = v2f{q.z, q.w};
    }
    v2f acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = v2f{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const v2f kj = v2f{k[j], k[j]};
#pragma unroll
      for (int o = 0; o < 8; ++o) acc[o] = acc[o] + win[o + j] * kj;
    }
    float* d0 = row_s + (2 * p) * BT_X + x0;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      d0[o] = acc[o].x;
      d0[BT_X + o] = acc[o].y;
    }
  }
  __syncthreads();
  // column pass: columns x, x + 1, outputs y0 .. y0 + 3
  {
    const int x = 2 * (tid & 31), y0 = 4 * (tid >> 5);
    const v2f* src = reinterpret_cast<const v2f*>(row_s + y0 * BT_X + x);
    v2f win[N + 3];
#pragma unroll
    for (int i = 0; i < N + 3; ++i) win[i] = src[i * (BT_X / 2)];
    v2f acc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o] = v2f{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const v2f kj = v2f{k[j], k[j]};
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] = acc[o] + win[o + j] * kj;
    }
    const int c = c0 + x;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int y = y0 + o, r = r0 + y;
      if (r >= rows || c >= cols) continue;
      const size_t at = (size_t)r * cols + c;
      const float* old = in_s + ((y + W) >> 1) * (2 * IN_WP) + 2 * (x + W) + ((y + W) & 1);   // the source pixel (and its right neighbour: + 2)
      job.dst[at] = acc[o].x;
      if (job.dog) job.dog[at] = __fsub_rn(old[0], acc[o].x);
      if (c + 1 < cols) {
        job.dst[at + 1] = acc[o].y;
        if (job.dog) job.dog[at + 1] = __fsub_rn(old[2], acc[o].y);
      }
    }
  }
}

__global__ __launch_bounds__(BT_THREADS) void blur_jobs_kernel(BlurJobs J) {
  __shared__ __attribute__((aligned(16))) float in_s[(BT_Y + 2 * BT_MAXW) * (BT_X + 2 * BT_MAXW)];
  __shared__ __attribute__((aligned(16))) float row_s[(BT_Y + 2 * BT_MAXW) * BT_X];
  const int tid = threadIdx.x;
  const int jn = (J.n > 1 && (int)blockIdx.x >= J.j[1].tile_begin) ? 1 : 0;
  BlurJob job = J.j[jn];
  if (blockIdx.y) {   // image of a batch
    job.src += blockIdx.y * job.src_step;
    job.dst += blockIdx.y * job.pyr_step;
    if (job.dog) job.dog += blockIdx.y * job.pyr_step;
    if (job.half_dst) job.half_dst += blockIdx.y * job.pyr_step;
  }
  const Taps& t = J.t[job.taps];
  const int rows = job.rows, cols = job.cols;
  const int tile = blockIdx.x - job.tile_begin;
  const int w = t.n >> 1;
  const int c0 = (tile % job.tiles_x) * BT_X, r0 = (tile / job.tiles_x) * BT_Y;
  // the tap counts of the shipped constants (InitSigma 1.6, 3 scales: 11, 13, 17, 21, 25) have kernels of their own
  switch (t.n) {
    case 11: return blur_tile_fixed<11>(job, t.k, in_s, row_s, r0, c0, tid);
    case 13: return blur_tile_fixed<13>(job, t.k, in_s, row_s, r0, c0, tid);
    case 17: return blur_tile_fixed<17>(job, t.k, in_s, row_s, r0, c0, tid);
    case 21: return blur_tile_fixed<21>(job, t.k, in_s, row_s, r0, c0, tid);
    case 25: return blur_tile_fixed<25>(job, t.k, in_s, row_s, r0, c0, tid);
    default: break;
  }
  const int in_w = BT_X + 2 * w, in_h = BT_Y + 2 * w;
  const int step = job.half ? 2 : 1;
  for (int e = tid; e < in_w * in_h; e += BT_THREADS) {
    const int yy = e / in_w, xx = e - yy * in_w;
    int y = r0 - w + yy, x = c0 - w + xx;
    const bool inside = y >= r0 && y < r0 + BT_Y && y < rows && x >= c0 && x < c0 + BT_X && x < cols;
    y = y < 0 ? 0 : (y >= rows ? rows - 1 : y);
    x = x < 0 ? 0 : (x >= cols ? cols - 1 : x);
    const float v = job.src[(size_t)(step * y) * job.src_cols + step * x];
    in_s[e] = v;
    if (job.half && inside) job.half_dst[(size_t)y * cols + x] = v;
  }
  __syncthreads();
  for (int e = tid; e < (in_h >> 1) * BT_X; e += BT_THREADS) {   // in_h is even
    const int yy = 2 * (e / BT_X), x = e % BT_X;
    const float* r0p = in_s + yy * in_w + x;
    const float* r1p = r0p + in_w;
    v2f a = {0.f, 0.f};
    for (int j = 0; j < t.n; ++j) a = a + v2f{r0p[j], r1p[j]} * v2f{t.k[j], t.k[j]};
    row_s[yy * BT_X + x] = a.x;
    row_s[(yy + 1) * BT_X + x] = a.y;
  }
  __syncthreads();
  for (int e = tid; e < BT_Y * (BT_X / 2); e += BT_THREADS) {
    const int y = e / (BT_X / 2), x = 2 * (e % (BT_X / 2));
    const int r = r0 + y, c = c0 + x;
    if (r >= rows || c >= cols) continue;
    const float* col = row_s + y * BT_X + x;
    v2f a = {0.f, 0.f};
    for (int j = 0; j < t.n; ++j) a = a + *reinterpret_cast<const v2f*>(col + j * BT_X) * v2f{t.k[j], t.k[j]};
    const size_t o = (size_t)r * cols + c;
    const float* old = in_s + (y + w) * in_w + (x + w);
    job.dst[o] = a.x;
    if (job.dog) job.dog[o] = __fsub_rn(old[0], a.x);
    if (c + 1 < cols) {
      job.dst[o + 1] = a.y;
      if (job.dog) job.dog[o + 1] = __fsub_rn(old[1], a.y);
    }
  }
}

// HalfImageSize (:390-408)
__global__ void half_kernel(const float* __restrict__ src, int scols, float* __restrict__ dst, int rows, int cols,
                            size_t pyr_step) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= cols || r >= rows) return;
  src += blockIdx.z * pyr_step;   // image of a batch
  dst += blockIdx.z * pyr_step;
  dst[(size_t)r * cols + c] = src[(size_t)(2 * r) * scols + 2 * c];
}

// The small octaves (<= SMALL_OCTAVE_PX pixels: 80x60 and below for a 640x480 frame) cost more in
// launches than in arithmetic: one workgroup runs the whole rest of the blur chain -- every level of
// every remaining octave and the subsampling between them -- out of LDS, with the same per-pixel
// loops (taps in ascending order, replicated edges).
constexpr int SMALL_OCTAVE_PX = 80 * 60;
struct Taps5 {
  Taps t[kScales + 2];
};
// Octaves that fit get their two images with replicated borders of SO_PAD pixels (left / right of the source
// image, above / below the row-blurred one), so the tap loops read straight through without clamping an index per
// tap; the others run the clamped loops.
constexpr int SO_PAD = 12, SO_CAP = 6912;
__global__ __launch_bounds__(1024) void small_octaves_kernel(SiftPyramid P, int o_first, Taps5 T, size_t pyr_step) {
  const size_t off = blockIdx.x * pyr_step;   // one workgroup per image of a batch
  __shared__ float cur[SO_CAP];   // image i - 1 of the octave: [px], or [rows][cols + 2 SO_PAD]
  __shared__ float tmp[SO_CAP];   // its row-blurred version:   [px], or [rows + 2 SO_PAD][cols]
  const int tid = threadIdx.x;
  int wmax = 0;
  for (int i = 0; i < kScales + 2; ++i) wmax = max(wmax, T.t[i].n >> 1);
  for (int o = o_first; o < P.n_octaves; ++o) {
    const SiftOctave& O = P.oct[o];
    const int rows = O.rows, cols = O.cols, px = rows * cols;
    const int cs = cols + 2 * SO_PAD;
    const bool padded = wmax <= SO_PAD && rows * cs <= SO_CAP && (rows + 2 * SO_PAD) * cols <= SO_CAP;
    // cur[at(r, c)], with its left / right borders when padded
    auto put_cur = [&](int r, int c, float v) {
      if (!padded) {
        cur[r * cols + c] = v;
        return;
      }
      float* row = cur + r * cs;
      row[SO_PAD + c] = v;
      if (c == 0)
        for (int h = 0; h < SO_PAD; ++h) row[h] = v;
      if (c == cols - 1)
        for (int h = 0; h < SO_PAD; ++h) row[SO_PAD + cols + h] = v;
    };
    const float* src0 = O.gaus[0] + off;
    if (o > o_first) {   // HalfImageSize (:390-408) of the previous octave's image `kScales`
      const SiftOctave& V = P.oct[o - 1];
      const float* src = V.gaus[kScales] + off;
      float* dst = O.gaus[0] + off;
      for (int e = tid; e < px; e += 1024) {
        const int r = e / cols, c = e - r * cols;
        const float v = src[(size_t)(2 * r) * V.cols + 2 * c];
        dst[e] = v;
        put_cur(r, c, v);
      }
    } else {
      for (int e = tid; e < px; e += 1024) {
        const int r = e / cols, c = e - r * cols;
        put_cur(r, c, src0[e]);
      }
    }
    __syncthreads();
    for (int i = 1; i < kScales + 3; ++i) {
      const Taps& t = T.t[i - 1];
      const int w = t.n >> 1;
      for (int e = tid; e < px; e += 1024) {
        const int r = e / cols, c = e - r * cols;
        float a = 0.f;
        if (padded) {
          const float* row = cur + r * cs + SO_PAD + c - w;
          for (int j = 0; j < t.n; ++j) a = __fadd_rn(a, __fmul_rn(row[j], t.k[j]));
          tmp[(r + SO_PAD) * cols + c] = a;
          if (r == 0)
            for (int h = 0; h < SO_PAD; ++h) tmp[h * cols + c] = a;
          if (r == rows - 1)
            for (int h = 0; h < SO_PAD; ++h) tmp[(rows + SO_PAD + h) * cols + c] = a;
        } else {
          const float* row = cur + r * cols;
          for (int j = 0; j < t.n; ++j) {
            int x = c + j - w;
            x = x < 0 ? 0 : (x >= cols ? cols - 1 : x);
            a = __fadd_rn(a, __fmul_rn(row[x], t.k[j]));
          }
          tmp[e] = a;
        }
      }
      __syncthreads();
      float* dst = O.gaus[i] + off;
      float* dog = O.dog[i - 1] + off;
      for (int e = tid; e < px; e += 1024) {
        const int r = e / cols, c = e - r * cols;
        float a = 0.f;
        if (padded) {
          const float* col = tmp + (r + SO_PAD - w) * cols + c;
          for (int j = 0; j < t.n; ++j) a = __fadd_rn(a, __fmul_rn(col[j * cols], t.k[j]));
        } else {
          for (int j = 0; j < t.n; ++j) {
            int y = r + j - w;
            y = y < 0 ? 0 : (y >= rows ? rows - 1 : y);
            a = __fadd_rn(a, __fmul_rn(tmp[y * cols + c], t.k[j]));
          }
        }
        const float old = padded ? cur[r * cs + SO_PAD + c] : cur[e];
        dst[e] = a;
        dog[e] = __fsub_rn(old, a);
        put_cur(r, c, a);   // source of the next level (nobody reads cur in this phase but the owner of a pixel)
      }
      __syncthreads();
    }
  }
}

// ---- gradient / orientation (GradOriImages, :959-992) ---------------------------------------
// The per-pixel kernels over all (octave, level) images run on ONE flat grid of 64 x 4 tiles: slice z = octave * kScales
// + level - 1 owns the blocks [begin[z], begin[z + 1]) -- a 3-D grid sized for octave 0 launched six times as many
// blocks as there are tiles (the small octaves' slices were almost all empty blocks).
struct SiftGrid {
  int begin[SIFT_MAX_OCTAVES * kScales + 1];
  int tiles_x[SIFT_MAX_OCTAVES];
  int n;   // slices
};
__device__ __forceinline__ bool sift_tile(const SiftGrid& G, int& o, int& index, int& bx, int& by) {
  const int b = blockIdx.x;
  int z = 0;
  while (z + 1 < G.n && b >= G.begin[z + 1]) ++z;
  o = z / kScales;
  index = 1 + z % kScales;
  const int t = b - G.begin[z];
  bx = t % G.tiles_x[o];
  by = t / G.tiles_x[o];
  return b < G.begin[G.n];
}

__global__ void grad_ori_kernel(SiftPyramid P, SiftGrid G, size_t pyr_step) {
  const size_t off = blockIdx.y * pyr_step;   // image of a batch
  int o, index, bx, by;
  if (!sift_tile(G, o, index, bx, by)) return;
  const SiftOctave& O = P.oct[o];
  const int rows = O.rows, cols = O.cols;
  const int j = bx * blockDim.x + threadIdx.x;
  const int i = by * blockDim.y + threadIdx.y;
  if (j >= cols || i >= rows) return;
  const float* im = O.gaus[index] + off;
  const float* p = im + (size_t)i * cols;
  float dc, dr;
  if (j == 0) dc = __fmul_rn(2.0f, __fsub_rn(p[1], p[0]));
  else if (j == cols - 1) dc = __fmul_rn(2.0f, __fsub_rn(p[j], p[j - 1]));
  else dc = __fsub_rn(p[j + 1], p[j - 1]);
  if (i == 0) dr = __fmul_rn(2.0f, __fsub_rn(p[j], p[cols + j]));
  else if (i == rows - 1) dr = __fmul_rn(2.0f, __fsub_rn(p[-cols + j], p[j]));
  else dr = __fsub_rn(p[-cols + j], p[cols + j]);
  const size_t at = (size_t)i * cols + j;
  (O.grad[index - 1] + off)[at] = sqrtf(__fadd_rn(__fmul_rn(dc, dc), __fmul_rn(dr, dr)));
  (O.ori[index - 1] + off)[at] = atan2f(dr, dc);
}

// ---- detection ------------------------------------------------------------------------------
__device__ __forceinline__ bool local_extremum(float v, const float* d, int cols, int r, int c) {
  for (int rr = r - 1; rr <= r + 1; ++rr) {
    const float* p = d + (size_t)rr * cols + c - 1;
    if (v > 0 ? (p[0] > v || p[1] > v || p[2] > v) : (v > p[0] || v > p[1] || v > p[2])) return false;
  }
  return true;
}

__device__ __forceinline__ bool not_on_edge(const float* d, int cols, int r, int c) {  // :1149-1162
  const float* p = d + (size_t)r * cols;
  const float f1 = __fadd_rn(__fsub_rn(p[-cols + c], __fmul_rn(p[c], 2.f)), p[cols + c]);
  const float f2 = __fadd_rn(__fsub_rn(p[c - 1], __fmul_rn(p[c], 2.f)), p[c + 1]);
  const float f3 = __fsub_rn(p[cols + c + 1], p[cols + c - 1]);
  const float f4 = __fsub_rn(p[-cols + c + 1], p[-cols + c - 1]);
  const float f5 = __fmul_rn(__fsub_rn(f3, f4), 0.25f);
  const float f6 = __fsub_rn(__fmul_rn(f1, f2), __fmul_rn(f5, f5));
  const float f8 = __fadd_rn(f1, f2);
  return __fmul_rn(__fmul_rn(f6, 11.f), 11.f) > __fmul_rn(__fmul_rn(f8, f8), 10.f);
}

__device__ void solve3(float* Y, float* H) {  // SolveLinearSystem (:1235-1272), dim = 3
  int best = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float fmax = -1.f;
    for (int j = i; j < 3; ++j) {
      float f = H[j * 3 + i];
      if (f < 0) f = -f;
      if (f > fmax) {
        fmax = f;
        best = j;
      }
    }
    if (best != i) {
      for (int j = 0; j < 3; ++j) {
        const float t = H[best * 3 + j];
        H[best * 3 + j] = H[i * 3 + j];
        H[i * 3 + j] = t;
      }
      const float t = Y[best];
      Y[best] = Y[i];
      Y[i] = t;
    }
    for (int j = i + 1; j < 3; ++j) {
      const float f = __fdiv_rn(H[j * 3 + i], H[i * 3 + i]);
      for (int k = i; k < 3; ++k) H[j * 3 + k] = __fsub_rn(H[j * 3 + k], __fmul_rn(f, H[i * 3 + k]));
      Y[j] = __fsub_rn(Y[j], __fmul_rn(Y[i], f));
    }
  }
  for (int i = 2; i >= 0; --i) {
    for (int j = 2; j > i; --j) Y[i] = __fsub_rn(Y[i], __fmul_rn(Y[j], H[i * 3 + j]));
    Y[i] = __fdiv_rn(Y[i], H[i * 3 + i]);
  }
}

__device__ float fit_quadratic(float* X, const float* p0, const float* p1, const float* p2, int cols, int r,
                               int c) {  // :1208-1231
  const size_t o = (size_t)r * cols + c;
  float Y[3], H[9];
  Y[0] = __fmul_rn(0.5f, __fsub_rn(p2[o], p0[o]));
  Y[1] = __fmul_rn(0.5f, __fsub_rn(p1[o + cols], p1[o - cols]));
  Y[2] = __fmul_rn(0.5f, __fsub_rn(p1[o + 1], p1[o - 1]));
  H[0] = __fadd_rn(__fsub_rn(p0[o], __fmul_rn(2.0f, p1[o])), p2[o]);
  H[4] = __fadd_rn(__fsub_rn(p1[o - cols], __fmul_rn(2.0f, p1[o])), p1[o + cols]);
  H[8] = __fadd_rn(__fsub_rn(p1[o - 1], __fmul_rn(2.0f, p1[o])), p1[o + 1]);
  H[3] = H[1] = __fmul_rn(0.25f, __fsub_rn(__fsub_rn(p2[o + cols], p2[o - cols]), __fsub_rn(p0[o + cols], p0[o - cols])));
  H[6] = H[2] = __fmul_rn(0.25f, __fsub_rn(__fsub_rn(p2[o + 1], p2[o - 1]), __fsub_rn(p0[o + 1], p0[o - 1])));
  H[7] = H[5] = __fmul_rn(0.25f, __fsub_rn(__fsub_rn(p1[o + cols + 1], p1[o + cols - 1]),
                                           __fsub_rn(p1[o - cols + 1], p1[o - cols - 1])));
  X[0] = -Y[0];
  X[1] = -Y[1];
  X[2] = -Y[2];
  solve3(X, H);
  const float dot = __fadd_rn(__fadd_rn(__fmul_rn(X[0], Y[0]), __fmul_rn(X[1], Y[1])), __fmul_rn(X[2], Y[2]));
  return __fadd_rn(p1[o], __fmul_rn(0.5f, dot));
}

// FindMaxMin's scan (:925-940) + InterpKeyPoint (:1164-1206).  A surviving extremum claims
// its FINAL pixel with atomicMin(generation key): the reference's s_MaxMinArray gives that
// pixel to the first survivor in (scale index, row, column) order.
__global__ void detect_kernel(SiftPyramid P, SiftGrid G, SiftCandidate* __restrict__ cand, int32_t* __restrict__ n_cand,
                              int cap, int32_t* __restrict__ overflow, SiftBatch Bt) {
  const size_t off = blockIdx.y * Bt.pyr_step;   // image of a batch: its pyramid, owner map, candidates, counters
  cand += (size_t)blockIdx.y * Bt.cand_step;
  n_cand += 4 * blockIdx.y;
  overflow += 4 * blockIdx.y;
  int o, index, bx, by;
  if (!sift_tile(G, o, index, bx, by)) return;
  const SiftOctave& O = P.oct[o];
  const int rows = O.rows, cols = O.cols;
  const int c = 5 + bx * blockDim.x + threadIdx.x;
  const int r = 5 + by * blockDim.y + threadIdx.y;
  if (c >= cols - 5 || r >= rows - 5) return;
  const float peak_thresh = 0.04f / (float)kScales;
  const float* d1 = O.dog[index] + off;
  const float v = d1[(size_t)r * cols + c];
  if (!(fabsf(v) > __fmul_rn(peak_thresh, 0.8f))) return;
  const float *d0 = O.dog[index - 1] + off, *d2 = O.dog[index + 1] + off;
  if (!local_extremum(v, d1, cols, r, c) || !local_extremum(v, d0, cols, r, c) ||
      !local_extremum(v, d2, cols, r, c) || !not_on_edge(d1, cols, r, c))
    return;
  int rr = r, cc = c;
  float X[3], val;
  for (int steps = 5;; --steps) {
    val = fit_quadratic(X, d0, d1, d2, cols, rr, cc);
    int nr = rr, nc = cc;
    if (X[1] > 0.6f && rr < rows - 3) nr++;
    if (X[1] < -0.6f && rr > 3) nr--;
    if (X[2] > 0.6f && cc < cols - 3) nc++;
    if (X[2] < -0.6f && cc > 3) nc--;
    if (steps > 0 && (nr != rr || nc != cc)) {
      rr = nr;
      cc = nc;
      continue;
    }
    break;
  }
  if (!(fabsf(X[0]) <= 1.5f && fabsf(X[1]) <= 1.5f && fabsf(X[2]) <= 1.5f && fabsf(val) >= peak_thresh)) return;
  const unsigned key = (unsigned)(index - 1) * (unsigned)(rows * cols) + (unsigned)(r * cols + c);
  atomicMin(&(O.owner + blockIdx.y * Bt.own_step)[(size_t)rr * cols + cc], key);
  const int at = atomicAdd(n_cand, 1);
  if (at >= cap) {
    *overflow = 1;
    return;
  }
  SiftCandidate k;
  k.octave = o;
  k.index = index;
  k.key = key;
  k.r = rr;
  k.c = cc;
  k.x0 = X[0];
  k.x1 = X[1];
  k.x2 = X[2];
  cand[at] = k;
}

// ---- orientation (AssignOriHist, :1274-1382) ---------------------------------------------------
// One wavefront per candidate.  Samples are visited 64 at a time in raster order; lane b < 36
// owns histogram bin b and adds the chunk's contributions to it in that order, so every bin
// sums exactly like the serial loop.
__global__ __launch_bounds__(64) void orient_kernel(SiftPyramid P, const SiftCandidate* __restrict__ cand,
                                                    const int32_t* __restrict__ n_cand, int cand_cap,
                                                    SiftKey* __restrict__ keys, int32_t* __restrict__ n_keys,
                                                    int key_cap, int32_t* __restrict__ overflow, SiftBatch Bt) {
  const int lane = threadIdx.x;
  const size_t off = blockIdx.y * Bt.pyr_step;   // image of a batch
  cand += (size_t)blockIdx.y * Bt.cand_step;
  keys += (size_t)blockIdx.y * Bt.key_step;
  n_cand += 4 * blockIdx.y;
  n_keys += 4 * blockIdx.y;
  overflow += 4 * blockIdx.y;
  int n = *n_cand;
  if (n > cand_cap) n = cand_cap;
  for (int ci = blockIdx.x; ci < n; ci += gridDim.x) {
    const SiftCandidate k = cand[ci];
    const SiftOctave& O = P.oct[k.octave];
    const int rows = O.rows, cols = O.cols;
    if ((O.owner + blockIdx.y * Bt.own_step)[(size_t)k.r * cols + k.c] != k.key) continue;  // another extremum got this pixel first
    const float* grad = O.grad[k.index - 1] + off;
    const float* orim = O.ori[k.index - 1] + off;
    const float fSize = __fmul_rn(kInitSigma, powf(2.0f, __fdiv_rn(__fadd_rn((float)k.index, k.x0), (float)kScales)));
    const float frow = __fadd_rn((float)k.r, k.x1), fcol = __fadd_rn((float)k.c, k.x2);
    const int rowstart = (int)__fadd_rn(frow, 0.5f), colstart = (int)__fadd_rn(fcol, 0.5f);
    const float fexpmult = __fdiv_rn(-1.0f, __fmul_rn(__fmul_rn(__fmul_rn(__fmul_rn(2.0f, 1.5f), 1.5f), fSize), fSize));
    const float fbinmult = 36.0f / (2 * kPi);
    const float fbinadd = (float)(kPi + 0.001f) * fbinmult;
    const int win = (int)__fmul_rn(__fmul_rn(fSize, 1.5f), 3.0f);
    const int side = 2 * win + 1, total = side * side;
    float h = 0.f;  // lane's bin
    // The window's samples 64 at a time, folded into the bins one after the other in raster order (the serial code's
    // order per bin).  All batches' gradient / orientation values are fetched FIRST -- batch by batch every fetch was
    // two dependent trips to L2 / HBM in front of a few hundred cycles of folding (0.048 ms per frame, most of it
    // waiting); windows of more than OR_MAXB batches (fSize > 3.9) finish with fetches of their own.
    constexpr int OR_MAXB = 20;
    float gv[OR_MAXB], ov[OR_MAXB];
#pragma unroll
    for (int b = 0; b < OR_MAXB; ++b) {
      gv[b] = 0.f;
      ov[b] = 0.f;
      const int s = b * 64 + lane;
      if (s < total) {
        const int r = rowstart - win + s / side, c = colstart - win + s % side;
        if (r >= 0 && r < rows - 2 && c >= 0 && c < cols - 2) {
          gv[b] = grad[(size_t)r * cols + c];
          ov[b] = orim[(size_t)r * cols + c];
        }
      }
    }
    auto fold = [&](int s, float g, float ori, bool fetched) {
      int bin = -1;
      float val = 0.f;
      if (s < total) {
        const int r = rowstart - win + s / side, c = colstart - win + s % side;
        if (r >= 0 && r < rows - 2 && c >= 0 && c < cols - 2) {
          if (!fetched) g = grad[(size_t)r * cols + c];
          if (g > 0) {
            const float dr = __fsub_rn((float)r, frow), dc = __fsub_rn((float)c, fcol);
            const float rad2 = __fadd_rn(__fmul_rn(dr, dr), __fmul_rn(dc, dc));
            if (__fadd_rn((float)(win * win), 0.5f) > rad2) {
              const float w = expf(__fmul_rn(rad2, fexpmult));
              if (!fetched) ori = orim[(size_t)r * cols + c];
              bin = (int)__fadd_rn(__fmul_rn(ori, fbinmult), fbinadd);
              if (bin > 36) bin = 0;
              if (bin == 36) bin = 35;
              val = __fmul_rn(g, w);
            }
          }
        }
      }
      unsigned long long m = __ballot(bin >= 0);
      while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int b = __builtin_amdgcn_readlane(bin, src);
        const float x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val), src));
        if (lane == b) h = __fadd_rn(h, x);
      }
    };
#pragma unroll
    for (int b = 0; b < OR_MAXB; ++b) {
      if (b * 64 >= total) break;
      fold(b * 64 + lane, gv[b], ov[b], true);
    }
    for (int base = OR_MAXB * 64; base < total; base += 64) fold(base + lane, 0.f, 0.f, false);
    // SmoothHistogram x 6 (:1395-1408): every new bin is (previous + own) + next of the OLD values (the serial
    // loop carries the old left neighbour along and has not reached the right one yet; bin 35 closes the ring
    // with the old bin 0 and its own, shorter, constant), so the 36 lanes smooth their bins side by side.
    const int lp = lane == 0 ? 35 : lane - 1, ln = lane >= 35 ? 0 : lane + 1;
    for (int it = 0; it < 6; ++it) {
      const float hp = __shfl(h, lp), hn = __shfl(h, ln);
      h = __fmul_rn(__fadd_rn(__fadd_rn(hp, h), hn), lane == 35 ? 0.3333333f : 0.33333333f);
    }
    float fmax = lane < 36 ? h : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) fmax = fmaxf(fmax, __shfl_xor(fmax, off));
    fmax = fmaxf(fmax, 0.f);
    fmax = __fmul_rn(fmax, 0.8f);
    const float foriadd = 0.5f * 2 * kPi / 36.0f - kPi, forimult = 2 * kPi / 36.0f;
    {
      const float hp = __shfl(h, lp), hn = __shfl(h, ln);
      const bool is_peak = lane < 36 && !(h <= hp || h <= hn || h < fmax);
      const unsigned long long pm = __ballot(is_peak);
      if (pm != 0ull) {
        int at0 = 0;
        if (lane == 0) at0 = atomicAdd(n_keys, __popcll(pm));   // slots in any order: order_kernel sorts by q.order
        at0 = __builtin_amdgcn_readfirstlane(at0);
        if (is_peak) {
          const int at = at0 + __popcll(pm & ((1ull << lane) - 1ull));
          if (at >= key_cap) {
            *overflow = 1;
          } else {
            float f0 = hp, f1 = h, f2 = hn;  // InterpPeak (:1384-1393)
            if (f1 < 0) {
              f0 = -f0;
              f1 = -f1;
              f2 = -f2;
            }
            const float peak = __fdiv_rn(__fmul_rn(0.5f, __fsub_rn(f0, f2)),
                                         __fadd_rn(__fsub_rn(f0, __fmul_rn(2.0f, f1)), f2));
            SiftKey q;
            q.octave = k.octave;
            q.index = k.index;
            q.order = ((unsigned long long)k.octave << 40) | ((unsigned long long)k.key << 8) | (unsigned)lane;
            q.fsize = fSize;
            q.frow = frow;
            q.fcol = fcol;
            q.ori = __fadd_rn(__fmul_rn(__fadd_rn((float)lane, peak), forimult), foriadd);
            keys[at] = q;
          }
        }
      }
    }
  }
}

// ---- descriptor (MakeKeypointSample / KeySample / AddSample / PlaceInIndex, :1424-1668) ---------
// One 256-thread workgroup per key.  The reference adds a sample's (up to eight) contributions to the descriptor
// entries it touches one sample after the other in raster order; the 128 entries are independent of each other, so
// what has to be kept is the ORDER OF THE ADDITIONS PER ENTRY (a chain of only ~S/16 additions), not the walk over all
// S samples.  The window is taken in chunks of 256 raster-ordered samples:
//   A. four wavefronts, 64 samples each, one per lane: the sample's weight, bilinear fractions and first row / column
//      / orientation bin (the arithmetic of KeySample / PlaceInIndex).  A sample feeds entry (cell, bin) when it
//      touches the cell (rows nr, nr + 1, columns nc, nc + 1) AND the bin is its own or the next one -- two
//      independent conditions, so 16 cell ballots + 8 bin ballots describe all 128 lists of the wavefront:
//      list (cell, bin) = lanes in cmask[cell] & bmask[bin], in lane = raster order.  Their sizes (two lists per
//      lane, a wave prefix sum) give every list its place in the wavefront's value array; every sample then writes
//      its up to eight products -- the reference's cg (1 - of) / cg of -- at offset + rank;
//   B. thread E < 128 owns entry E = 8 cell + bin and adds its list's values, wavefront after wavefront: the same
//      products in the same order as the serial code, and nothing else (no entry is visited that is not added).
// History (per 640x480 frame): every wavefront folding all samples of its key with readlanes, eight wavefronts sharing
// a key when keys are few 0.28 ms; per-cell lists that four lanes per cell filter by bin 0.14; this 0.0x.
constexpr int DESC_WAVES = 4;
struct DescLds {
  unsigned long long cmask[DESC_WAVES][16];
  unsigned long long bmask[DESC_WAVES][8];
  uint16_t off[DESC_WAVES][130];     // list (cell, bin) of the wavefront = val[off[8 cell + bin] .. off[8 cell + bin + 1])
  float val[DESC_WAVES][512];
  float d[128];
  float scal;
};
__global__ __launch_bounds__(64 * DESC_WAVES) void describe_kernel(SiftPyramid P, const SiftKey* __restrict__ keys,
                                                      const int32_t* __restrict__ n_keys, int key_cap,
                                                      float* __restrict__ desc_out /* [key][128] */,
                                                      float* __restrict__ geo_out /* [key][4] col,row,scale,ori */,
                                                      SiftBatch Bt) {
  const size_t off = blockIdx.y * Bt.pyr_step;   // image of a batch
  keys += (size_t)blockIdx.y * Bt.key_step;
  n_keys += 4 * blockIdx.y;
  desc_out += (size_t)blockIdx.y * Bt.key_step * 128;
  geo_out += (size_t)blockIdx.y * Bt.key_step * 4;
  __shared__ DescLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int n = *n_keys;
  if (n > key_cap) n = key_cap;
  const unsigned long long lt = (1ull << lane) - 1ull;
  // LDS written by one lane of a wavefront and read by another of the same wavefront
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  for (int ki = blockIdx.x; ki < n; ki += gridDim.x) {
    const SiftKey k = keys[ki];
    const SiftOctave& O = P.oct[k.octave];
    const int rows = O.rows, cols = O.cols;
    const float* grad = O.grad[k.index - 1] + off;
    const float* orim = O.ori[k.index - 1] + off;
    const float fSize = k.fsize, frow = k.frow, fcol = k.fcol, ang = k.ori;
    const int rowstart = (int)__fadd_rn(frow, 0.5f), colstart = (int)__fadd_rn(fcol, 0.5f);
    const float sinang = sinf(ang), cosang = cosf(ang);
    const float fdrow = __fsub_rn(frow, (float)rowstart), fdcol = __fsub_rn(fcol, (float)colstart);
    const float frealsize = __fmul_rn(3.0f, fSize), firealsize = __fdiv_rn(1.0f, __fmul_rn(3.0f, fSize));
    const int win = (int)__fadd_rn(__fmul_rn(__fmul_rn(__fmul_rn(frealsize, kSqrt2), 5.0f), 0.5f), 0.5f);
    const float fsr = __fmul_rn(sinang, firealsize), fcr = __fmul_rn(cosang, firealsize);
    const float fdrr = __fmul_rn(-fdrow, firealsize), fdcr = __fmul_rn(-fdcol, firealsize);
    const int side = 2 * win + 1, total = side * side;
    float acc = 0.f;
    for (int chunk = 0; chunk < total; chunk += 64 * DESC_WAVES) {
      // ---- A ----
      {
        const int s = chunk + tid;
        bool ok = false;
        float mag = 0.f, rf = 0.f, cf = 0.f, of = 0.f;
        int nr = 0, nc = 0, nb = 0;
        if (s < total) {
          const int row = s / side - win, col = s % side - win;
          const float fr = (float)row, fc = (float)col;
          const float rpos = __fadd_rn(__fadd_rn(__fmul_rn(fsr, fc), __fmul_rn(fcr, fr)), fdrr);
          const float cpos = __fadd_rn(__fsub_rn(__fmul_rn(fcr, fc), __fmul_rn(fsr, fr)), fdcr);
          const float rx = __fadd_rn(rpos, 2.0f - 0.5f), cx = __fadd_rn(cpos, 2.0f - 0.5f);
          const int r = rowstart + row, c = colstart + col;
          if (rx > -0.9999f && rx < 3.9999f && cx > -0.9999f && cx < 3.9999f && r >= 0 && r < rows && c >= 0 &&
              c < cols) {
            ok = true;
            const float e = expf(__fmul_rn(-0.125f, __fadd_rn(__fmul_rn(rpos, rpos), __fmul_rn(cpos, cpos))));
            mag = __fmul_rn(grad[(size_t)r * cols + c], e);
            float o = __fsub_rn(orim[(size_t)r * cols + c], ang);
            while (o > 2 * kPi) o = __fsub_rn(o, 2 * kPi);
            while (o < 0) o = __fadd_rn(o, 2 * kPi);
            const float oribin = __fmul_rn(o, 8.0f / (2 * (float)kPi));   // PlaceInIndex
            nr = rx < 0 ? (int)__fsub_rn(rx, 1.f) : (int)rx;
            rf = __fsub_rn(rx, (float)nr);
            nc = cx < 0 ? (int)__fsub_rn(cx, 1.f) : (int)cx;
            cf = __fsub_rn(cx, (float)nc);
            const int no = oribin < 0 ? (int)__fsub_rn(oribin, 1.f) : (int)oribin;
            of = __fsub_rn(oribin, (float)no);
            nb = no & 7;   // the bins wrap: orientation 2 pi falls into bin 8 = bin 0
          }
        }
        // the 24 masks of this wavefront: cell c in lane c, bin b in lane 16 + b
        unsigned long long keep = 0ull;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const int r_ = c >> 2, c_ = c & 3;
          const unsigned long long m = __ballot(ok && (nr == r_ - 1 || nr == r_) && (nc == c_ - 1 || nc == c_));
          if (lane == c) keep = m;
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const unsigned long long m = __ballot(ok && (nb == b || ((nb + 1) & 7) == b));
          if (lane == 16 + b) keep = m;
        }
        if (lane < 16) L.cmask[wave][lane] = keep;
        else if (lane < 24) L.bmask[wave][lane - 16] = keep;
        wave_sync();
        // sizes of lists 2 lane, 2 lane + 1 (cell lane / 4, bins 2 (lane % 4), + 1) and their places
        {
          const unsigned long long cm = L.cmask[wave][lane >> 2];
          const int c0 = __popcll(cm & L.bmask[wave][2 * (lane & 3)]);
          const int c1 = __popcll(cm & L.bmask[wave][2 * (lane & 3) + 1]);
          int incl = c0 + c1;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
          }
          const int excl = incl - (c0 + c1);
          L.off[wave][2 * lane] = (uint16_t)excl;
          L.off[wave][2 * lane + 1] = (uint16_t)(excl + c0);
          if (lane == 63) L.off[wave][128] = (uint16_t)incl;
        }
        wave_sync();
        if (ok) {
          const float rg0 = __fmul_rn(mag, __fsub_rn(1.f, rf)), rg1 = __fmul_rn(mag, rf);   // rows nr, nr + 1
          const float cfm = __fsub_rn(1.f, cf), ofm = __fsub_rn(1.f, of);
          const int b1 = (nb + 1) & 7;
          const unsigned long long bm0 = L.bmask[wave][nb] & lt, bm1 = L.bmask[wave][b1] & lt;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int r_ = nr + i;
            if ((unsigned)r_ >= 4u) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int c_ = nc + j;
              if ((unsigned)c_ >= 4u) continue;
              const int cell = 4 * r_ + c_;
              const float rg = i == 0 ? rg0 : rg1;
              const float cg = j == 0 ? __fmul_rn(rg, cfm) : __fmul_rn(rg, cf);
              const unsigned long long cm = L.cmask[wave][cell];
              L.val[wave][L.off[wave][8 * cell + nb] + __popcll(cm & bm0)] = __fmul_rn(cg, ofm);
              L.val[wave][L.off[wave][8 * cell + b1] + __popcll(cm & bm1)] = __fmul_rn(cg, of);
            }
          }
        }
      }
      __syncthreads();
      // ---- B ----
      if (tid < 128) {
#pragma unroll
        for (int w = 0; w < DESC_WAVES; ++w) {
          const int e0 = L.off[w][tid], e1 = L.off[w][tid + 1];
          const float* v = L.val[w];
          for (int e = e0; e < e1; e += 4) {
            // four at a time: the loads do not depend on the running sum (reads past the list's end stay inside val)
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = v[min(e + j, 511)];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (e + j < e1) acc = __fadd_rn(acc, x[j]);
          }
        }
      }
      __syncthreads();   // the lists are rewritten by the next chunk
    }
    if (tid < 128) L.d[tid] = acc;
    __syncthreads();
    // NormalizeVec, clamp at 0.2, NormalizeVec again if anything was clamped (:1497-1527)
    for (int pass = 0; pass < 2; ++pass) {
      if (tid == 0) {
        float a = 0.f;
        for (int i = 0; i < 128; ++i) a = __fadd_rn(a, __fmul_rn(L.d[i], L.d[i]));
        L.scal = __fdiv_rn(1.f, sqrtf(a));
      }
      __syncthreads();
      bool clamp = false;
      if (tid < 128) {
        float v = __fmul_rn(L.d[tid], L.scal);
        if (pass == 0 && v > 0.2f) {
          v = 0.2f;
          clamp = true;
        }
        L.d[tid] = v;
      }
      const int any = __syncthreads_or(clamp ? 1 : 0);
      if (!any) break;
    }
    if (tid < 128) desc_out[(size_t)ki * 128 + tid] = L.d[tid];
    if (tid == 0) {
      const float fscale = O.fscale;
      float* g = geo_out + (size_t)ki * 4;
      g[0] = __fmul_rn(fscale, fcol);   // coord2D = (col, row), FEAT_SIFT_CPU.hpp:103-104
      g[1] = __fmul_rn(fscale, frow);
      g[2] = __fmul_rn(fscale, fSize);
      g[3] = ang;
    }
    __syncthreads();
  }
}

// ---- order: the reference's list = generation order reversed (every key is pushed on the
// front of a linked list, :1432, :944-952) -------------------------------------------------------
__global__ void order_kernel(const SiftKey* __restrict__ keys, const int32_t* __restrict__ n_keys, int key_cap,
                             const float* __restrict__ desc_in, const float* __restrict__ geo_in, int out_cap,
                             float* __restrict__ desc_out, float* __restrict__ xy_out,
                             float* __restrict__ scale_ori_out, int32_t* __restrict__ n_out, SiftBatch Bt) {
  if (blockIdx.y) {   // image of a batch: its keys, its rows of the outputs (out_step keypoints apart), its count word
    keys += (size_t)blockIdx.y * Bt.key_step;
    n_keys += 4 * blockIdx.y;
    desc_in += (size_t)blockIdx.y * Bt.key_step * 128;
    geo_in += (size_t)blockIdx.y * Bt.key_step * 4;
    desc_out += (size_t)blockIdx.y * Bt.out_step * 128;
    xy_out += (size_t)blockIdx.y * Bt.out_step * 2;
    if (scale_ori_out) scale_ori_out += (size_t)blockIdx.y * Bt.out_step * 2;
    n_out += (size_t)blockIdx.y * Bt.n_out_step;
  }
  int n = *n_keys;
  if (n > key_cap) n = key_cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = n < out_cap ? n : out_cap;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    __shared__ int cnt_s;
    if (threadIdx.x == 0) cnt_s = 0;
    __syncthreads();
    const unsigned long long mine = keys[i].order;
    int c = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) c += keys[j].order > mine;  // keys are distinct
    if (c) atomicAdd(&cnt_s, c);
    __syncthreads();
    const int dst = cnt_s;  // number of keys generated after this one
    if (dst < out_cap) {
      for (int t = threadIdx.x; t < 128; t += blockDim.x) desc_out[(size_t)dst * 128 + t] = desc_in[(size_t)i * 128 + t];
      if (threadIdx.x == 0) {
        xy_out[2 * dst] = geo_in[4 * (size_t)i];
        xy_out[2 * dst + 1] = geo_in[4 * (size_t)i + 1];
        if (scale_ori_out) {
          scale_ori_out[2 * dst] = geo_in[4 * (size_t)i + 2];
          scale_ori_out[2 * dst + 1] = geo_in[4 * (size_t)i + 3];
        }
      }
    }
    __syncthreads();
  }
}

// GaussianBlur's kernel (:470-506), on the host with libm's expf like the reference.
Taps make_taps(float fblur) {
  Taps t;
  const float trunc = 4.0f;
  int ksize = (int)(2.0f * trunc * fblur + 1.0f);
  if (ksize < 3) ksize = 3;
  ksize += !(ksize & 1);
  if (ksize > MAX_TAPS - 1) ksize = MAX_TAPS - 1;
  const int width = ksize >> 1;
  float k[MAX_TAPS];
  double acc = 0;
  for (int i = 0; i <= ksize; ++i) {
    const float w = expf(-(float)(i - width) * (i - width) / (2.0f * fblur * fblur));
    acc += (double)w;
    k[i] = w;
  }
  for (int i = 0; i < ksize; ++i) t.k[i] = k[i] / (float)acc;
  for (int i = ksize; i < MAX_TAPS; ++i) t.k[i] = 0.f;
  t.n = ksize;
  return t;
}

}  // namespace

int sift_plan(int width, int height, int double_size, SiftPlan* plan) {
  int rows = double_size ? 2 * height - 2 : height;
  int cols = double_size ? 2 * width - 2 : width;
  plan->n_octaves = 0;
  plan->floats = 0;
  float fscale = double_size ? 0.5f : 1.0f;
  while (rows > 12 && cols > 12 && plan->n_octaves < SIFT_MAX_OCTAVES) {
    plan->rows[plan->n_octaves] = rows;
    plan->cols[plan->n_octaves] = cols;
    plan->fscale[plan->n_octaves] = fscale;
    plan->floats += (size_t)rows * cols * SIFT_IMAGES_PER_OCTAVE;
    ++plan->n_octaves;
    rows >>= 1;
    cols >>= 1;
    fscale += fscale;
  }
  plan->rows0 = double_size ? 2 * height - 2 : height;
  plan->cols0 = double_size ? 2 * width - 2 : width;
  return plan->n_octaves;
}

namespace {

// launch_sift / launch_sift_batch: n images (n = 1: `grays` holds the one) in one launch per stage
void launch_sift_images(const uint8_t* const* grays, int n, int width, int height, int double_size, const SiftPlan& plan,
                        const SiftBuffers& B, int out_cap, int out_step, float* desc_out, float* xy_out,
                        float* scale_ori_out, int32_t* n_out, int n_out_step, hipStream_t s) {
  const uint8_t* const gray = grays[0];
  SiftBatch Bt;
  Bt.pyr_step = plan.floats;
  Bt.own_step = B.owner_elems;
  Bt.tmp_step = (size_t)plan.rows0 * plan.cols0;
  Bt.cand_step = B.cand_cap;
  Bt.key_step = B.key_cap;
  Bt.out_step = out_step;
  Bt.n_out_step = n_out_step;
  Bt.n = n;
  SiftImages imgs;
  for (int i = 0; i < MH_MAX_BATCH; ++i) imgs.gray[i] = grays[i < n ? i : 0];
  const unsigned un = (unsigned)n;
  SiftPyramid P;
  memset(&P, 0, sizeof P);
  P.n_octaves = plan.n_octaves;
  float* p = B.pyramid;
  unsigned int* own = B.owner;
  for (int o = 0; o < plan.n_octaves; ++o) {
    SiftOctave& O = P.oct[o];
    O.rows = plan.rows[o];
    O.cols = plan.cols[o];
    O.fscale = plan.fscale[o];
    const size_t px = (size_t)O.rows * O.cols;
    for (int i = 0; i < kScales + 3; ++i, p += px) O.gaus[i] = p;
    for (int i = 0; i < kScales + 2; ++i, p += px) O.dog[i] = p;
    for (int i = 0; i < kScales; ++i, p += px) O.grad[i] = p;
    for (int i = 0; i < kScales; ++i, p += px) O.ori[i] = p;
    O.owner = own;
    own += px;
  }
  hipMemsetAsync(B.owner, 0xFF, B.owner_elems * sizeof(unsigned int) * n, s);
  hipMemsetAsync(B.counters, 0, 4 * sizeof(int32_t) * n, s);

  const SiftOctave& O0 = P.oct[0];
  const dim3 tb(256);
  auto grid_for = [un](int rows, int cols) { return dim3((cols + 255) / 256, rows, un); };
  const float fnew = double_size ? 1.0f : 0.5f;
  const bool init_blur = kInitSigma > fnew;   // :325-327
  const Taps t0 = init_blur ? make_taps(sqrtf(kInitSigma * kInitSigma - fnew * fnew)) : Taps{};
  const bool init_fused = init_blur && (t0.n >> 1) <= BT_MAXW;
  // the prepared image goes to the scratch image when the fused blur can write octave 0's first level from there
  hipLaunchKernelGGL(prepare_kernel, grid_for(O0.rows, O0.cols), tb, 0, s, gray, width, height, double_size,
                     init_fused ? B.tmp : O0.gaus[0], O0.rows, O0.cols, imgs, init_fused ? Bt.tmp_step : Bt.pyr_step);
  if (init_fused) {
    BlurJobs J;
    J.n = 1;
    BlurJob& b = J.j[0];
    b.src = B.tmp;
    b.dst = O0.gaus[0];
    b.dog = nullptr;
    b.half_dst = nullptr;
    b.rows = O0.rows;
    b.cols = b.src_cols = O0.cols;
    b.half = 0;
    b.taps = 0;
    b.tiles_x = (O0.cols + BT_X - 1) / BT_X;
    b.tile_begin = 0;
    b.src_step = Bt.tmp_step;
    b.pyr_step = Bt.pyr_step;
    J.j[1] = b;
    J.t[0] = J.t[1] = t0;
    hipLaunchKernelGGL(blur_jobs_kernel, dim3(b.tiles_x * ((O0.rows + BT_Y - 1) / BT_Y), un), dim3(BT_THREADS), 0, s, J);
  } else if (init_blur) {   // in place through the scratch image
    hipLaunchKernelGGL(blur_rows_kernel, grid_for(O0.rows, O0.cols), tb, 0, s, O0.gaus[0], B.tmp, O0.rows, O0.cols, t0);
    hipLaunchKernelGGL(blur_cols_kernel, grid_for(O0.rows, O0.cols), tb, 0, s, B.tmp, O0.gaus[0], O0.rows, O0.cols, t0,
                       (const float*)nullptr, (float*)nullptr);
  }
  const float fwidth = powf(2.0f, 1.0f / (float)kScales);
  const float fincsigma = sqrtf(fwidth * fwidth - 1.0f);
  // per-level kernels (they only depend on the level: sigma restarts at every octave, :410-438)
  Taps5 T5;
  {
    float sigma = kInitSigma;
    for (int i = 1; i < kScales + 3; ++i) {
      T5.t[i - 1] = make_taps(fincsigma * sigma);
      sigma *= fwidth;
    }
  }
  int o_small = plan.n_octaves;   // first octave the single-workgroup kernel takes over
  for (int o = 1; o < plan.n_octaves; ++o)
    if ((size_t)plan.rows[o] * plan.cols[o] <= (size_t)SMALL_OCTAVE_PX) {
      o_small = o;
      break;
    }
  bool jobs_ok = true;   // every level's kernel fits the tile kernel's halo
  for (int i = 0; i < kScales + 2; ++i) jobs_ok = jobs_ok && (T5.t[i].n >> 1) <= BT_MAXW;
  if (jobs_ok) {
    // the large octaves as a dependency-ordered list of launches, two independent levels per launch where there are two
    auto make_job = [&](int o, int i) {   // level i of octave o from level i - 1 (or, i == 1 and o > 0, from octave o - 1)
      const SiftOctave& O = P.oct[o];
      BlurJob b;
      b.rows = O.rows;
      b.cols = O.cols;
      b.dst = O.gaus[i];
      b.dog = O.dog[i - 1];
      b.taps = i - 1;
      b.tiles_x = (O.cols + BT_X - 1) / BT_X;
      b.tile_begin = 0;
      b.src_step = b.pyr_step = Bt.pyr_step;
      if (i == 1 && o > 0) {
        b.src = P.oct[o - 1].gaus[kScales];
        b.src_cols = P.oct[o - 1].cols;
        b.half = 1;
        b.half_dst = O.gaus[0];
      } else {
        b.src = O.gaus[i - 1];
        b.src_cols = O.cols;
        b.half = 0;
        b.half_dst = nullptr;
      }
      return b;
    };
    auto tiles_of = [&](const BlurJob& b) { return b.tiles_x * ((b.rows + BT_Y - 1) / BT_Y); };
    auto launch_jobs = [&](BlurJob a, const BlurJob* b2) {
      BlurJobs J;
      J.n = b2 ? 2 : 1;
      J.j[0] = a;
      J.t[0] = T5.t[a.taps];
      J.j[0].taps = 0;
      int total = tiles_of(a);
      if (b2) {
        J.j[1] = *b2;
        J.t[1] = T5.t[b2->taps];
        J.j[1].taps = 1;
        J.j[1].tile_begin = total;
        total += tiles_of(*b2);
      } else {
        J.j[1] = a;
        J.t[1] = J.t[0];
      }
      hipLaunchKernelGGL(blur_jobs_kernel, dim3(total, un), dim3(BT_THREADS), 0, s, J);
    };
    for (int o = 0; o < o_small; ++o) {
      // levels 1 (unless it went out with the previous octave's level kScales + 1), 2 .. kScales on their own
      for (int i = (o == 0 ? 1 : 3); i <= kScales; ++i) launch_jobs(make_job(o, i), nullptr);
      // levels kScales + 1, kScales + 2 together with levels 1, 2 of the next large octave
      const bool next = o + 1 < o_small;
      for (int d = 1; d <= 2; ++d) {
        const BlurJob a = make_job(o, kScales + d);
        if (next) {
          const BlurJob b = make_job(o + 1, d);
          launch_jobs(a, &b);
        } else {
          launch_jobs(a, nullptr);
        }
      }
      if (!next && o + 1 < plan.n_octaves) {   // the first small octave's level 0
        const SiftOctave& O = P.oct[o];
        const SiftOctave& N = P.oct[o + 1];
        hipLaunchKernelGGL(half_kernel, grid_for(N.rows, N.cols), tb, 0, s, (const float*)O.gaus[kScales], O.cols,
                           N.gaus[0], N.rows, N.cols, (size_t)Bt.pyr_step);
      }
    }
  } else
  for (int o = 0; o < o_small; ++o) {
    const SiftOctave& O = P.oct[o];
    for (int i = 1; i < kScales + 3; ++i) {   // OctaveKeypoints (:410-438)
      const Taps& t = T5.t[i - 1];
      if ((t.n >> 1) <= BT_MAXW) {
        hipLaunchKernelGGL(blur_level_kernel, dim3((O.cols + BT_X - 1) / BT_X, (O.rows + BT_Y - 1) / BT_Y), dim3(BT_THREADS), 0,
                           s, (const float*)O.gaus[i - 1], O.gaus[i], O.rows, O.cols, t, (const float*)O.gaus[i - 1],
                           O.dog[i - 1]);
        continue;
      }
      hipLaunchKernelGGL(blur_rows_kernel, grid_for(O.rows, O.cols), tb, 0, s, O.gaus[i - 1], B.tmp, O.rows, O.cols, t);
      hipLaunchKernelGGL(blur_cols_kernel, grid_for(O.rows, O.cols), tb, 0, s, B.tmp, O.gaus[i], O.rows, O.cols, t,
                         (const float*)O.gaus[i - 1], O.dog[i - 1]);
    }
    if (o + 1 < plan.n_octaves) {
      const SiftOctave& N = P.oct[o + 1];
      hipLaunchKernelGGL(half_kernel, grid_for(N.rows, N.cols), tb, 0, s, (const float*)O.gaus[kScales], O.cols,
                         N.gaus[0], N.rows, N.cols, (size_t)Bt.pyr_step);
    }
  }
  if (o_small < plan.n_octaves)
    hipLaunchKernelGGL(small_octaves_kernel, dim3(un), dim3(1024), 0, s, P, o_small, T5, (size_t)Bt.pyr_step);
  const dim3 tb2(64, 4);
  SiftGrid G;
  G.n = plan.n_octaves * kScales;
  G.begin[0] = 0;
  for (int o = 0; o < plan.n_octaves; ++o) {
    G.tiles_x[o] = (plan.cols[o] + 63) / 64;
    for (int i = 0; i < kScales; ++i)
      G.begin[o * kScales + i + 1] = G.begin[o * kScales + i] + G.tiles_x[o] * ((plan.rows[o] + 3) / 4);
  }
  const dim3 g2(G.begin[G.n], un);
  hipLaunchKernelGGL(grad_ori_kernel, g2, tb2, 0, s, P, G, (size_t)Bt.pyr_step);
  hipLaunchKernelGGL(detect_kernel, g2, tb2, 0, s, P, G, B.cand, B.counters + 0, B.cand_cap, B.counters + 2, Bt);
  // (the per-key kernels loop over the keys: a batch's images share the chip, fewer workgroups per image)
  const unsigned per = n > 4 ? 4 : 1;
  hipLaunchKernelGGL(orient_kernel, dim3(4096 / per, un), dim3(64), 0, s, P, (const SiftCandidate*)B.cand,
                     (const int32_t*)(B.counters + 0), B.cand_cap, B.keys, B.counters + 1, B.key_cap, B.counters + 2, Bt);
  hipLaunchKernelGGL(describe_kernel, dim3(2048 / per, un), dim3(64 * DESC_WAVES), 0, s, P, (const SiftKey*)B.keys,
                     (const int32_t*)(B.counters + 1), B.key_cap, B.desc_tmp, B.geo_tmp, Bt);
  hipLaunchKernelGGL(order_kernel, dim3(1024 / per, un), dim3(64), 0, s, (const SiftKey*)B.keys,
                     (const int32_t*)(B.counters + 1), B.key_cap, (const float*)B.desc_tmp, (const float*)B.geo_tmp,
                     out_cap, desc_out, xy_out, scale_ori_out, n_out, Bt);
}

}  // namespace

void launch_sift(const uint8_t* gray, int width, int height, int double_size, const SiftPlan& plan,
                 const SiftBuffers& B, int out_cap, float* desc_out, float* xy_out, float* scale_ori_out,
                 int32_t* n_out, hipStream_t s) {
  launch_sift_images(&gray, 1, width, height, double_size, plan, B, out_cap, out_cap, desc_out, xy_out, scale_ori_out,
                     n_out, 1, s);
}

void launch_sift_batch(const uint8_t* const* gray, int n, int width, int height, int double_size, const SiftPlan& plan,
                       const SiftBuffers& B, int out_cap, int out_step, float* desc_out, float* xy_out,
                       float* scale_ori_out, int32_t* n_out, int n_out_step, hipStream_t s) {
  if (n <= 0) return;
  // the tile blur takes every level of the shipped constants; kernels it does not take go image after image
  const float fnew = double_size ? 1.0f : 0.5f;
  bool tiles = !(kInitSigma > fnew) || (make_taps(sqrtf(kInitSigma * kInitSigma - fnew * fnew)).n >> 1) <= BT_MAXW;
  {
    const float fwidth = powf(2.0f, 1.0f / (float)kScales);
    const float fincsigma = sqrtf(fwidth * fwidth - 1.0f);
    float sigma = kInitSigma;
    for (int i = 1; i < kScales + 3; ++i) {
      tiles = tiles && (make_taps(fincsigma * sigma).n >> 1) <= BT_MAXW;
      sigma *= fwidth;
    }
  }
  if (n == 1 || n > B.images || !tiles) {
    for (int i = 0; i < n; ++i)
      launch_sift_images(gray + i, 1, width, height, double_size, plan, B, out_cap, out_step,
                         desc_out + (size_t)i * out_step * 128, xy_out + (size_t)i * out_step * 2,
                         scale_ori_out ? scale_ori_out + (size_t)i * out_step * 2 : nullptr, n_out + (size_t)i * n_out_step,
                         n_out_step, s);
    return;
  }
  launch_sift_images(gray, n, width, height, double_size, plan, B, out_cap, out_step, desc_out, xy_out, scale_ori_out,
                     n_out, n_out_step, s);
}

}  // namespace mh
