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
#include "sift_rows.h"

namespace mh {

namespace {

constexpr float kPi = 3.141592654f;   // :65
constexpr float kSqrt2 = 1.4142136f;  // :66
constexpr int kScales = 3;            // :108
constexpr float kInitSigma = 1.6f;    // :109

// ---- prepare ----------------------------------------------------------------------------
// FEAT_SIFT_CPU.hpp:91 (pixel * 1./255. in double) and SiftDoubleSize (:363-380).
__device__ __forceinline__ float to_unit(uint8_t g) { return (float)((double)(float)g * 1. / 255.); }

// pixel (r, c) of the image GetKeypoints starts from: the frame itself or SiftDoubleSize's (:363-380) 2x - 2 version of it
__device__ __forceinline__ float prepared_px(const uint8_t* __restrict__ gray, int w, int double_size, int r, int c) {
  if (!double_size) return to_unit(gray[(size_t)r * w + c]);
  const int i = r >> 1, j = c >> 1;
  const float a = to_unit(gray[(size_t)i * w + j]);
  if ((r & 1) == 0 && (c & 1) == 0) return a;
  if ((r & 1) == 1 && (c & 1) == 0) return __fmul_rn(0.5f, __fadd_rn(a, to_unit(gray[(size_t)(i + 1) * w + j])));
  if ((r & 1) == 0) return __fmul_rn(0.5f, __fadd_rn(a, to_unit(gray[(size_t)i * w + j + 1])));
  const float b = to_unit(gray[(size_t)i * w + j + 1]);
  const float d = to_unit(gray[(size_t)(i + 1) * w + j]);
  const float e = to_unit(gray[(size_t)(i + 1) * w + j + 1]);
  return __fmul_rn(0.25f, __fadd_rn(__fadd_rn(__fadd_rn(a, b), d), e));
}

// blockIdx.z = image of a batch (SiftImages: its pixels; its output `out_step` floats behind the image's before it)
__global__ void prepare_kernel(const uint8_t* __restrict__ gray, int w, int h, int double_size,
                               float* __restrict__ out, int orows, int ocols, SiftImages imgs, size_t out_step,
                               int32_t* __restrict__ counters) {
  // the image's counters (candidates, keys, overflow, -) start at zero: the first kernel of the chain clears them
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4) counters[4 * blockIdx.z + threadIdx.x] = 0;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= ocols || r >= orows) return;
  if (blockIdx.z) {
    gray = imgs.gray[blockIdx.z];
    out += blockIdx.z * out_step;
  }
  out[(size_t)r * ocols + c] = prepared_px(gray, w, double_size, r, c);
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
  // staging: 8 x 32 threads over rows x columns (replicated edges; a `half` job reads every second row and column
  // of the previous octave and writes its own pixels out as level 0).  ALL of a thread's pixels are requested before
  // the first one is used: written as one loop (load, store to LDS, next) every pixel was a round trip to L2 of its
  // own -- up to 24 in a row, most of a level's 12 us.
  {
    constexpr int NY = (IN_H + 7) / 8, NX = (IN_W + 31) / 32;
    float v[NY][NX];
#pragma unroll
    for (int iy = 0; iy < NY; ++iy) {
      const int yy = (tid >> 5) + 8 * iy;
      const int yu = r0 - W + yy;
      const int y = yu < 0 ? 0 : (yu >= rows ? rows - 1 : yu);
      const float* srow = job.src + (size_t)(step * y) * job.src_cols;
#pragma unroll
      for (int ix = 0; ix < NX; ++ix) {
        const int xx = (tid & 31) + 32 * ix;
        const int xu = c0 - W + xx;
        const int x = xu < 0 ? 0 : (xu >= cols ? cols - 1 : xu);
        v[iy][ix] = (yy < IN_H && xx < IN_W) ? srow[step * x] : 0.f;
      }
    }
#pragma unroll
    for (int iy = 0; iy < NY; ++iy) {
      const int yy = (tid >> 5) + 8 * iy;
      const int yu = r0 - W + yy;
      const int y = yu < 0 ? 0 : (yu >= rows ? rows - 1 : yu);
      float* drow = in_s + (yy >> 1) * (2 * IN_WP) + (yy & 1);
#pragma unroll
      for (int ix = 0; ix < NX; ++ix) {
        const int xx = (tid & 31) + 32 * ix;
        if (yy >= IN_H || xx >= IN_W) continue;
        const int xu = c0 - W + xx;
        const int x = xu < 0 ? 0 : (xu >= cols ? cols - 1 : xu);
        drow[2 * xx] = v[iy][ix];
        if (job.half && yu == y && xu == x && yy >= W && yy < W + BT_Y && xx >= W && xx < W + BT_X)
          job.half_dst[(size_t)y * cols + x] = v[iy][ix];
      }
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
      if (job.dst) job.dst[at] = acc[o].x;   // (an octave's last level is read by nobody: only its DoG is kept)
      if (job.dog) job.dog[at] = __fsub_rn(old[0], acc[o].x);
      if (c + 1 < cols) {
        if (job.dst) job.dst[at + 1] = acc[o].y;
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
    if (job.dst) job.dst += blockIdx.y * job.pyr_step;
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
    if (job.dst) job.dst[o] = a.x;
    if (job.dog) job.dog[o] = __fsub_rn(old[0], a.x);
    if (c + 1 < cols) {
      if (job.dst) job.dst[o + 1] = a.y;
      if (job.dog) job.dog[o + 1] = __fsub_rn(old[1], a.y);
    }
  }
}

// (Round 5 also tried the tile blur as persistent workgroups -- a workgroup walking its job's tiles over all images of a
// batch, the next tile's source pixels requested while the current tile's passes run: the prefetch registers beside the
// passes' windows took the kernel from 81 to 154 registers, three wavefronts per SIMD instead of six, and the chain from
// 636 to 811 us per batch of 16 images.  One tile per workgroup.)
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
constexpr int SMALL_OCTAVE_PX = 40 * 30;
struct Taps5 {
  Taps t[kScales + 2];
};
// Octaves that fit get their two images with replicated borders of SO_PAD pixels (left / right of the source
// image, above / below the row-blurred one), so the tap loops read straight through without clamping an index per
// tap; the others run the clamped loops.
constexpr int SO_PAD = 12, SO_CAP = 6912;
__device__ __forceinline__ void so_put_padded(float* cur, int cs, int cols, int r, int c, float v) {
  float* row = cur + r * cs;
  row[SO_PAD + c] = v;
  if (c == 0)
    for (int h = 0; h < SO_PAD; ++h) row[h] = v;
  if (c == cols - 1)
    for (int h = 0; h < SO_PAD; ++h) row[SO_PAD + cols + h] = v;
}
// One level of a padded octave with the tap count known at compile time: the taps sit in registers and the tap loops are
// unrolled, N independent LDS reads in flight in front of the chain of products and sums (ascending taps, like
// ConvHorizontal / ConvVertical).  The loop over a run-time tap count below did one scalar load of the tap from the
// kernel's arguments and one LDS read per tap, each waited for: 88 us per image for ~800 taps per thread.
template <int N>
__device__ __forceinline__ void so_level_padded(float* cur, float* tmp, const float* __restrict__ tk, int rows, int cols,
                                                float* __restrict__ dst, float* __restrict__ dog, int tid) {
  constexpr int W = N / 2;
  const int px = rows * cols, cs = cols + 2 * SO_PAD;
  float k[N];
#pragma unroll
  for (int j = 0; j < N; ++j) k[j] = tk[j];
  for (int e = tid; e < px; e += 1024) {
    const int r = e / cols, c = e - r * cols;
    const float* row = cur + r * cs + SO_PAD + c - W;
    float v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = row[j];
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) a = __fadd_rn(a, __fmul_rn(v[j], k[j]));
    tmp[(r + SO_PAD) * cols + c] = a;
    if (r == 0)
      for (int h = 0; h < SO_PAD; ++h) tmp[h * cols + c] = a;
    if (r == rows - 1)
      for (int h = 0; h < SO_PAD; ++h) tmp[(rows + SO_PAD + h) * cols + c] = a;
  }
  __syncthreads();
  for (int e = tid; e < px; e += 1024) {
    const int r = e / cols, c = e - r * cols;
    const float* col = tmp + (r + SO_PAD - W) * cols + c;
    float v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = col[j * cols];
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) a = __fadd_rn(a, __fmul_rn(v[j], k[j]));
    const float old = cur[r * cs + SO_PAD + c];
    dst[e] = a;
    dog[e] = __fsub_rn(old, a);
    so_put_padded(cur, cs, cols, r, c, a);   // source of the next level (nobody reads cur in this phase but the owner of a pixel)
  }
  __syncthreads();
}
__global__ __launch_bounds__(1024) void small_octaves_kernel(SiftPyramid P, int o_first, Taps5 T, size_t pyr_step) {
  const size_t off = blockIdx.x * pyr_step;   // one workgroup per image of a batch
  __shared__ float cur[SO_CAP];   // image i - 1 of the octave: [px], or [rows][cols + 2 SO_PAD]
  __shared__ float tmp[SO_CAP];   // its row-blurred version:   [px], or [rows + 2 SO_PAD][cols]
  __shared__ float taps_s[kScales + 2][MAX_TAPS];
  const int tid = threadIdx.x;
  int wmax = 0;
  for (int i = 0; i < kScales + 2; ++i) wmax = max(wmax, T.t[i].n >> 1);
  if (tid < (kScales + 2) * MAX_TAPS) taps_s[tid / MAX_TAPS][tid % MAX_TAPS] = T.t[tid / MAX_TAPS].k[tid % MAX_TAPS];
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
      so_put_padded(cur, cs, cols, r, c, v);
    };
    const float* src0 = O.gaus[0] + off;
    if (o > 0) {   // HalfImageSize (:390-408) of the previous octave's image `kScales` (the first small octave's too: no launch of its own)
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
      const int tn = T.t[i - 1].n;
      const float* tk = taps_s[i - 1];
      const int w = tn >> 1;
      float* dst = O.gaus[i] + off;
      float* dog = O.dog[i - 1] + off;
      if (padded) {   // the tap counts of the shipped constants
        bool done = true;
        switch (tn) {
          case 11: so_level_padded<11>(cur, tmp, tk, rows, cols, dst, dog, tid); break;
          case 13: so_level_padded<13>(cur, tmp, tk, rows, cols, dst, dog, tid); break;
          case 17: so_level_padded<17>(cur, tmp, tk, rows, cols, dst, dog, tid); break;
          case 21: so_level_padded<21>(cur, tmp, tk, rows, cols, dst, dog, tid); break;
          case 25: so_level_padded<25>(cur, tmp, tk, rows, cols, dst, dog, tid); break;
          default: done = false; break;
        }
        if (done) continue;
      }
      for (int e = tid; e < px; e += 1024) {
        const int r = e / cols, c = e - r * cols;
        float a = 0.f;
        if (padded) {
          const float* row = cur + r * cs + SO_PAD + c - w;
          for (int j = 0; j < tn; ++j) a = __fadd_rn(a, __fmul_rn(row[j], tk[j]));
          tmp[(r + SO_PAD) * cols + c] = a;
          if (r == 0)
            for (int h = 0; h < SO_PAD; ++h) tmp[h * cols + c] = a;
          if (r == rows - 1)
            for (int h = 0; h < SO_PAD; ++h) tmp[(rows + SO_PAD + h) * cols + c] = a;
        } else {
          const float* row = cur + r * cols;
          for (int j = 0; j < tn; ++j) {
            int x = c + j - w;
            x = x < 0 ? 0 : (x >= cols ? cols - 1 : x);
            a = __fadd_rn(a, __fmul_rn(row[x], tk[j]));
          }
          tmp[e] = a;
        }
      }
      __syncthreads();
      for (int e = tid; e < px; e += 1024) {
        const int r = e / cols, c = e - r * cols;
        float a = 0.f;
        if (padded) {
          const float* col = tmp + (r + SO_PAD - w) * cols + c;
          for (int j = 0; j < tn; ++j) a = __fadd_rn(a, __fmul_rn(col[j * cols], tk[j]));
        } else {
          for (int j = 0; j < tn; ++j) {
            int y = r + j - w;
            y = y < 0 ? 0 : (y >= rows ? rows - 1 : y);
            a = __fadd_rn(a, __fmul_rn(tmp[y * cols + c], tk[j]));
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
// gradient magnitude and orientation of pixel (i, j) of a Gaussian level, as GradOriImages stores them (:959-992):
// central differences, one-sided and doubled on the border rows / columns.  Computed where a key's window asks for them
// (orient_kernel, describe_kernel) -- until round 5 a kernel of its own wrote both for every pixel of three levels per
// octave: six of an octave's seventeen images, 59 MB per 640x480 frame, for the few hundred thousand samples the keys read.
struct GradTaps {   // the four neighbours' offsets and the border factor
  size_t a, b, u, d;
  bool edge_c, edge_r;
};
__device__ __forceinline__ GradTaps grad_taps(int rows, int cols, int i, int j) {
  GradTaps t;
  const int jm = j > 0 ? j - 1 : 0, jp = j < cols - 1 ? j + 1 : cols - 1;
  const int im = i > 0 ? i - 1 : 0, ip = i < rows - 1 ? i + 1 : rows - 1;
  t.a = (size_t)i * cols + jm;
  t.b = (size_t)i * cols + jp;
  t.u = (size_t)im * cols + j;
  t.d = (size_t)ip * cols + j;
  t.edge_c = j == 0 || j == cols - 1;
  t.edge_r = i == 0 || i == rows - 1;
  return t;
}
__device__ __forceinline__ void grad_of(float a, float b, float u, float d, bool edge_c, bool edge_r, float& dc, float& dr) {
  dc = __fsub_rn(b, a);   // p[j + 1] - p[j - 1]; on a border column 2 (p[1] - p[0]) / 2 (p[j] - p[j - 1])
  if (edge_c) dc = __fmul_rn(2.0f, dc);
  dr = __fsub_rn(u, d);   // p[i - 1] - p[i + 1]; on a border row 2 (p[i] - p[i + 1]) / 2 (p[i - 1] - p[i])
  if (edge_r) dr = __fmul_rn(2.0f, dr);
}
__device__ __forceinline__ float grad_mag(float dc, float dr) { return sqrtf(__fadd_rn(__fmul_rn(dc, dc), __fmul_rn(dr, dr))); }

// ---- detection ------------------------------------------------------------------------------
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
// A workgroup takes a 64 x 8 tile of an octave through ALL THREE scale indices: the five DoG levels' tiles (with a ring of
// one pixel) are staged in LDS first -- every pixel requested before the first is used -- and the threshold, the 27-point
// extremum test and the edge test read LDS.  (One pixel of one index per thread straight from global memory, as until
// round 5, paid for the extremum test row by row: up to nine dependent trips to L2 per level, 54 - 90 us per image, and
// read every level's tile three times.)  The few survivors (~3 000 per image) run the quadratic fit from global memory: it
// walks up to five pixels away.
constexpr int DT_X = 64, DT_Y = 8, DT_W = DT_X + 2, DT_H = DT_Y + 2, DT_THREADS = 256;
struct DetectGrid {
  int begin[SIFT_MAX_OCTAVES + 1];   // first block of octave o_first + i
  int tiles_x[SIFT_MAX_OCTAVES];
  int n, o_first;
};
__global__ __launch_bounds__(DT_THREADS) void detect_kernel(SiftPyramid P, DetectGrid G, SiftCandidate* __restrict__ cand,
                                                            int32_t* __restrict__ n_cand, int cap,
                                                            int32_t* __restrict__ overflow, SiftBatch Bt) {
  const size_t off = blockIdx.y * Bt.pyr_step;   // image of a batch: its pyramid, owner map, candidates, counters
  cand += (size_t)blockIdx.y * Bt.cand_step;
  n_cand += 4 * blockIdx.y;
  overflow += 4 * blockIdx.y;
  const int b = blockIdx.x;
  if (b >= G.begin[G.n]) return;
  int oi = 0;
  while (oi + 1 < G.n && b >= G.begin[oi + 1]) ++oi;
  const int o = G.o_first + oi;
  const SiftOctave& O = P.oct[o];
  const int rows = O.rows, cols = O.cols;
  const int t = b - G.begin[oi];
  const int c0 = 5 + (t % G.tiles_x[oi]) * DT_X, r0 = 5 + (t / G.tiles_x[oi]) * DT_Y;
  __shared__ float S[kScales + 2][DT_H][DT_W + 1];
  const int tid = threadIdx.x;
  {
    constexpr int PER = ((kScales + 2) * DT_H * DT_W + DT_THREADS - 1) / DT_THREADS;
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + i * DT_THREADS;
      const int l = e / (DT_H * DT_W), rem = e - l * (DT_H * DT_W);
      const int yy = rem / DT_W, xx = rem - yy * DT_W;
      // (the ring of a tile that ends at the scanned region's border lies inside the image; beyond it nobody looks)
      const int y = min(r0 - 1 + yy, rows - 1), x = min(c0 - 1 + xx, cols - 1);
      v[i] = l < kScales + 2 ? (O.dog[l] + off)[(size_t)y * cols + x] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + i * DT_THREADS;
      const int l = e / (DT_H * DT_W), rem = e - l * (DT_H * DT_W);
      const int yy = rem / DT_W, xx = rem - yy * DT_W;
      if (l < kScales + 2) S[l][yy][xx] = v[i];
    }
  }
  __syncthreads();
  const float peak_thresh = 0.04f / (float)kScales;
  const int x = tid & (DT_X - 1);
  const int c = c0 + x;
  if (c >= cols - 5) return;
#pragma unroll
  for (int index = 1; index <= kScales; ++index) {
#pragma unroll
    for (int k = 0; k < DT_Y / (DT_THREADS / DT_X); ++k) {
      const int y = (tid / DT_X) + k * (DT_THREADS / DT_X);
      const int r = r0 + y;
      if (r >= rows - 5) continue;
      const float v = S[index][y + 1][x + 1];
      if (!(fabsf(v) > __fmul_rn(peak_thresh, 0.8f))) continue;
      // local_extremum on the level, the one below and the one above: no neighbour beyond v on v's side
      float hi = v, lo = v;
#pragma unroll
      for (int l = index - 1; l <= index + 1; ++l)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float p = S[l][y + dy][x + dx];
            hi = fmaxf(hi, p);
            lo = fminf(lo, p);
          }
      if (v > 0 ? hi > v : v > lo) continue;
      {   // not on an edge (:1149-1162)
        const float(*p)[DT_W + 1] = S[index];
        const int yy = y + 1, xx = x + 1;
        const float f1 = __fadd_rn(__fsub_rn(p[yy - 1][xx], __fmul_rn(p[yy][xx], 2.f)), p[yy + 1][xx]);
        const float f2 = __fadd_rn(__fsub_rn(p[yy][xx - 1], __fmul_rn(p[yy][xx], 2.f)), p[yy][xx + 1]);
        const float f3 = __fsub_rn(p[yy + 1][xx + 1], p[yy + 1][xx - 1]);
        const float f4 = __fsub_rn(p[yy - 1][xx + 1], p[yy - 1][xx - 1]);
        const float f5 = __fmul_rn(__fsub_rn(f3, f4), 0.25f);
        const float f6 = __fsub_rn(__fmul_rn(f1, f2), __fmul_rn(f5, f5));
        const float f8 = __fadd_rn(f1, f2);
        if (!(__fmul_rn(__fmul_rn(f6, 11.f), 11.f) > __fmul_rn(__fmul_rn(f8, f8), 10.f))) continue;
      }
      const float *d0 = O.dog[index - 1] + off, *d1 = O.dog[index] + off, *d2 = O.dog[index + 1] + off;
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
      if (!(fabsf(X[0]) <= 1.5f && fabsf(X[1]) <= 1.5f && fabsf(X[2]) <= 1.5f && fabsf(val) >= peak_thresh)) continue;
      const unsigned key = (unsigned)(index - 1) * (unsigned)(rows * cols) + (unsigned)(r * cols + c);
      atomicMin(&(O.owner + blockIdx.y * Bt.own_step)[(size_t)rr * cols + cc], Bt.own_prefix | key);   // (SiftBatch::own_prefix)
      const int at = atomicAdd(n_cand, 1);
      if (at >= cap) {
        *overflow = 1;
        continue;
      }
      SiftCandidate q;
      q.octave = o;
      q.index = index;
      q.key = key;
      q.r = rr;
      q.c = cc;
      q.x0 = X[0];
      q.x1 = X[1];
      q.x2 = X[2];
      cand[at] = q;
    }
  }
}

// ---- orientation (AssignOriHist, :1274-1382) ---------------------------------------------------
// One wavefront per candidate.  Samples are visited 64 at a time in raster order and listed in LDS; lane b < 36
// owns histogram bin b and adds the list's contributions to it in that order, so every bin
// sums exactly like the serial loop.
constexpr int OR_CAP = 1728;   // samples per pass: the largest window of the shipped constants has 41 x 41 = 1 681
struct OrientLds {
  __attribute__((aligned(16))) float val[OR_CAP + 4];
  __attribute__((aligned(4))) unsigned char bin[OR_CAP + 4];
};
__global__ __launch_bounds__(64) void orient_kernel(SiftPyramid P, const SiftCandidate* __restrict__ cand,
                                                    const int32_t* __restrict__ n_cand, int cand_cap,
                                                    SiftKey* __restrict__ keys, int32_t* __restrict__ n_keys,
                                                    int key_cap, int32_t* __restrict__ overflow, SiftBatch Bt) {
  const int lane = threadIdx.x;
  __shared__ OrientLds L;
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
    if ((O.owner + blockIdx.y * Bt.own_step)[(size_t)k.r * cols + k.c] != (Bt.own_prefix | k.key)) continue;  // another extremum got this pixel first
    const float* im = O.gaus[k.index] + off;   // gradients of the key's Gaussian level (grad_of)
    const float fSize = __fmul_rn(kInitSigma, powf(2.0f, __fdiv_rn(__fadd_rn((float)k.index, k.x0), (float)kScales)));
    const float frow = __fadd_rn((float)k.r, k.x1), fcol = __fadd_rn((float)k.c, k.x2);
    const int rowstart = (int)__fadd_rn(frow, 0.5f), colstart = (int)__fadd_rn(fcol, 0.5f);
    const float fexpmult = __fdiv_rn(-1.0f, __fmul_rn(__fmul_rn(__fmul_rn(__fmul_rn(2.0f, 1.5f), 1.5f), fSize), fSize));
    const float fbinmult = 36.0f / (2 * kPi);
    const float fbinadd = (float)(kPi + 0.001f) * fbinmult;
    const int win = (int)__fmul_rn(__fmul_rn(fSize, 1.5f), 3.0f);
    const int side = 2 * win + 1, total = side * side;
    float h = 0.f;  // lane's bin
    // All batches' gradient / orientation values are fetched FIRST -- batch by batch every fetch was two dependent trips
    // to L2 / HBM in front of a few hundred cycles of work (0.048 ms per frame, most of it waiting); windows of more than
    // OR_MAXB batches (fSize > 3.9) finish with fetches of their own.
    constexpr int OR_MAXB = 12;
    float dcv[OR_MAXB], drv[OR_MAXB];   // a sample's column / row difference (zero: no sample)
    {
      float pa[OR_MAXB], pb[OR_MAXB], pu[OR_MAXB], pd[OR_MAXB];
      unsigned edges = 0;
#pragma unroll
      for (int b = 0; b < OR_MAXB; ++b) {
        pa[b] = pb[b] = pu[b] = pd[b] = 0.f;
        const int s = b * 64 + lane;
        if (s < total) {
          const int r = rowstart - win + s / side, c = colstart - win + s % side;
          if (r >= 0 && r < rows - 2 && c >= 0 && c < cols - 2) {
            const GradTaps t = grad_taps(rows, cols, r, c);
            pa[b] = im[t.a];
            pb[b] = im[t.b];
            pu[b] = im[t.u];
            pd[b] = im[t.d];
            edges |= (t.edge_c ? 1u : 0u) << (2 * b) | (t.edge_r ? 2u : 0u) << (2 * b);
          }
        }
      }
#pragma unroll
      for (int b = 0; b < OR_MAXB; ++b) grad_of(pa[b], pb[b], pu[b], pd[b], (edges >> (2 * b)) & 1u, (edges >> (2 * b)) & 2u, dcv[b], drv[b]);
    }
    // The window's samples 64 at a time: every lane's (bin, weighted magnitude), the ones that count packed into a list in
    // LDS in raster order (one ballot per 64); then lane b < 36 walks the whole list and adds the entries of bin b -- the
    // serial code's order per bin.  (Until round 5 the wavefront folded every sample into its bin one after the other with
    // two readlanes each: ~7 instructions per sample on all lanes against 4 here, 57 us at 2 700 candidates.)
    auto sample = [&](int s, float gdc, float gdr, bool fetched, int& bin, float& val) {
      bin = -1;
      val = 0.f;
      if (s < total) {
        const int r = rowstart - win + s / side, c = colstart - win + s % side;
        if (r >= 0 && r < rows - 2 && c >= 0 && c < cols - 2) {
          if (!fetched) {
            const GradTaps t = grad_taps(rows, cols, r, c);
            grad_of(im[t.a], im[t.b], im[t.u], im[t.d], t.edge_c, t.edge_r, gdc, gdr);
          }
          const float g = grad_mag(gdc, gdr);
          if (g > 0) {
            const float dr = __fsub_rn((float)r, frow), dc = __fsub_rn((float)c, fcol);
            const float rad2 = __fadd_rn(__fmul_rn(dr, dr), __fmul_rn(dc, dc));
            if (__fadd_rn((float)(win * win), 0.5f) > rad2) {
              const float w = expf(__fmul_rn(rad2, fexpmult));
              const float ori = atan2f(gdr, gdc);
              bin = (int)__fadd_rn(__fmul_rn(ori, fbinmult), fbinadd);
              if (bin > 36) bin = 0;
              if (bin == 36) bin = 35;
              val = __fmul_rn(g, w);
            }
          }
        }
      }
    };
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int s0 = 0; s0 < total; s0 += OR_CAP) {   // (windows of more than OR_CAP samples -- none with the shipped constants -- in parts)
      int n_list = 0;
      const int s1 = min(total, s0 + OR_CAP);
      auto push = [&](int bin, float val) {
        const unsigned long long m = __ballot(bin >= 0);
        if (bin >= 0) {
          const int at = n_list + __popcll(m & lt);
          L.val[at] = val;
          L.bin[at] = (unsigned char)bin;
        }
        n_list += __popcll(m);
      };
      if (s0 == 0) {
#pragma unroll
        for (int b = 0; b < OR_MAXB; ++b) {
          if (b * 64 >= s1) break;
          int bin;
          float val;
          sample(b * 64 + lane, dcv[b], drv[b], true, bin, val);
          push(bin, val);
        }
      }
      for (int base = s0 == 0 ? OR_MAXB * 64 : s0; base < s1; base += 64) {
        int bin;
        float val;
        sample(base + lane, 0.f, 0.f, false, bin, val);
        push(bin, val);
      }
      if (lane < 4) {   // the list's last group of four, padded with entries of no bin
        const int at = n_list + lane;
        L.val[at] = 0.f;
        L.bin[at] = 255;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int i = 0; i < n_list; i += 4) {
        const float4 v = *reinterpret_cast<const float4*>(&L.val[i]);
        const unsigned bb = *reinterpret_cast<const unsigned*>(&L.bin[i]);
        // (adding +0 where the entry belongs to another bin leaves h as it is: the magnitudes are positive)
        h = __fadd_rn(h, (int)(bb & 255u) == lane ? v.x : 0.f);
        h = __fadd_rn(h, (int)((bb >> 8) & 255u) == lane ? v.y : 0.f);
        h = __fadd_rn(h, (int)((bb >> 16) & 255u) == lane ? v.z : 0.f);
        h = __fadd_rn(h, (int)(bb >> 24) == lane ? v.w : 0.f);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
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
        if (lane == 0) at0 = atomicAdd(n_keys, __popcll(pm));   // slots in any order: rank_kernel places by q.order
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
// One workgroup of DESC_WAVES wavefronts per key.  The reference adds a sample's (up to eight) contributions to the
// descriptor entries it touches one sample after the other in raster order; the 128 entries are independent of each
// other, so what has to be kept is the ORDER OF THE ADDITIONS PER ENTRY (a chain of only ~S/16 additions), not the walk
// over all S samples:
//   0. the window's rows: KeySample keeps a sample only inside the rotated 5 x 5-cell square (rx, cx in (-0.9999,
//      3.9999)) -- half of the window's square.  Lane = row computes a CONSERVATIVE column interval (desc_row_interval:
//      the two linear conditions solved with a margin of 0.01 and a column on each side, tests/test_sift_rows_cpu.py),
//      a prefix sum over the rows numbers the samples inside the intervals in raster order, and the workgroup takes
//      them 64 DESC_WAVES at a time (every lane walks a row cursor of its own); the exact tests still run per sample;
//   A. every wavefront 64 samples, one per lane: weight, bilinear fractions, first row / column / orientation bin (the
//      arithmetic of KeySample / PlaceInIndex).  A sample feeds entry (cell, bin) when it touches the cell's row AND its
//      column AND the bin is its own or the next one -- three independent conditions, so 4 + 4 + 8 ballots describe all
//      128 lists of the wavefront: list (cell, bin) = lanes in rmask[cell / 4] & xmask[cell % 4] & bmask[bin], in lane
//      = raster order.  Their sizes (two lists per lane, a wave prefix sum) give every list its place in the wavefront's
//      value array; every sample writes its up to eight products -- the reference's cg (1 - of) / cg of -- at offset +
//      rank;
//   B. thread E < 128 owns entry E = 8 cell + bin and adds its list's values, wavefront after wavefront: the same
//      products in the same order as the serial code, and nothing else (no entry is visited that is not added).
// The lists are double-buffered -- step i + 1's A runs beside step i's B, ONE workgroup barrier per step -- and a
// sample's gradient / orientation values are requested a step before they are used.  The descriptor goes straight to
// its place in the reference's list: generation order reversed (every key is pushed on the front of a linked list, :1432,
// :944-952), so a key's place = the number of keys generated after it, which the key's workgroup counts itself (n / 256
// generation words per thread; until round 5 an order kernel ranked the keys -- one workgroup per key walking all keys --
// and copied every descriptor to its place: 49 us at 3 240 keys).
// History (per 640x480 frame of ~600 keys): every wavefront folding all samples of its key with readlanes 0.28 ms;
// per-cell lists that four lanes per cell filter by bin 0.14; 256 samples per step out of the window's whole square, 24
// ballots, two barriers per step 0.1 (197 us at 3 240 keys); one wavefront per key with the interval walk: slower (225
// us: ~2 us per step of dependent LDS / permute latency and three wavefronts per SIMD to hide it); this: DESIGN.md 7.
// inclusive prefix sum over the 64 lanes in six DPP additions (rows of 16: shifts by 1, 2, 4, 8; then lane 15 of rows 0 / 2
// onto rows 1 / 3 and lane 31 onto rows 2, 3) -- __shfl_up goes through the LDS crossbar, ~100 cycles a step
__device__ __forceinline__ int wave_incl_scan(int x) {
  int s = x;
  s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xf, 0xf, false);   // row_shr:1
  s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xf, 0xf, false);   // row_shr:2
  s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xf, false);   // row_shr:4
  s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xf, false);   // row_shr:8
  s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
  s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
  return s;
}
#ifdef SIFT_PROF   // phase timing build (make EXTRA=-DSIFT_PROF): cycles of lane 0 of the first ([0..9]) and of the last wavefront
// ([10..19]) of describe_kernel's workgroups, summed over keys: 0 setup + rows, 1 fetch, 2 sample arithmetic, 3 masks + sizes,
// 4 list writes, 5 waiting at the step's barrier, 6 B, 7 normalisation + output; [8] keys, [9] steps
__device__ unsigned long long g_sift_prof[32];
#define DPF(k) do { if (prof_on) { const unsigned long long now_ = clock64(); prof_acc[k] += now_ - prof_t; prof_t = now_; } } while (0)
#else
#define DPF(k) do { } while (0)
#endif
#ifndef MH_DESC_WAVES
#define MH_DESC_WAVES 4
#endif
#ifndef MH_DESC_B
#define MH_DESC_B 0
#endif
#ifndef MH_DESC_MINW
#define MH_DESC_MINW 1
#endif
constexpr int DESC_WAVES = MH_DESC_WAVES, DESC_MAX_SIDE = 128;
struct DescLds {
  unsigned long long mask[2][DESC_WAVES][16];   // 0..3 rows of cells, 4..7 columns of cells, 8..15 bins
  uint16_t off[2][DESC_WAVES][132];             // list (cell, bin) of the wavefront = val[off[8 cell + bin] .. off[8 cell + bin + 1])
  float val[2][DESC_WAVES][512];
  int rowbeg[DESC_MAX_SIDE + 1];                // samples in front of window row i
  short rowlo[DESC_MAX_SIDE];                   // first column (window coordinates) of row i's interval
  unsigned char chunkrow[DESC_MAX_SIDE * DESC_MAX_SIDE / 64];   // window row of sample 64 c
  float sq[128];
  float scal;
  int later[DESC_WAVES];                        // keys generated after this one, counted by wavefront w
};
__global__ __launch_bounds__(64 * DESC_WAVES, MH_DESC_MINW) void describe_kernel(SiftPyramid P, const SiftKey* __restrict__ keys,
                                                      const int32_t* __restrict__ n_keys, int key_cap, int out_cap,
                                                      int32_t* __restrict__ n_out,
                                                      float* __restrict__ desc_out /* [place][128] */,
                                                      float* __restrict__ xy_out /* [place][2] col,row */,
                                                      float* __restrict__ scale_ori_out /* [place][2] or null */,
                                                      SiftBatch Bt) {
  const size_t off = blockIdx.y * Bt.pyr_step;   // image of a batch
  keys += (size_t)blockIdx.y * Bt.key_step;
  n_keys += 4 * blockIdx.y;
  n_out += (size_t)blockIdx.y * Bt.n_out_step;
  desc_out += (size_t)blockIdx.y * Bt.out_step * 128;
  xy_out += (size_t)blockIdx.y * Bt.out_step * 2;
  if (scale_ori_out) scale_ori_out += (size_t)blockIdx.y * Bt.out_step * 2;
  __shared__ DescLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int n = *n_keys;
  if (n > key_cap) n = key_cap;
  if (blockIdx.x == 0 && tid == 0) *n_out = n < out_cap ? n : out_cap;
  const unsigned long long lt = (1ull << lane) - 1ull;
  // LDS written by one lane of a wavefront and read by another of the same wavefront
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
#ifdef SIFT_PROF
  const bool prof_on = lane == 0 && (wave == 0 || wave == DESC_WAVES - 1);
  unsigned long long prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_t = clock64();
#endif
  for (int ki = blockIdx.x; ki < n; ki += gridDim.x) {
    const SiftKey k = keys[ki];
    const SiftOctave& O = P.oct[k.octave];
    const int rows = O.rows, cols = O.cols;
    const float* im = O.gaus[k.index] + off;   // gradients of the key's Gaussian level (grad_of)
    const float fSize = k.fsize, frow = k.frow, fcol = k.fcol, ang = k.ori;
    const int rowstart = (int)__fadd_rn(frow, 0.5f), colstart = (int)__fadd_rn(fcol, 0.5f);
    const float sinang = sinf(ang), cosang = cosf(ang);
    const float fdrow = __fsub_rn(frow, (float)rowstart), fdcol = __fsub_rn(fcol, (float)colstart);
    const float frealsize = __fmul_rn(3.0f, fSize), firealsize = __fdiv_rn(1.0f, __fmul_rn(3.0f, fSize));
    const int win = (int)__fadd_rn(__fmul_rn(__fmul_rn(__fmul_rn(frealsize, kSqrt2), 5.0f), 0.5f), 0.5f);
    const float fsr = __fmul_rn(sinang, firealsize), fcr = __fmul_rn(cosang, firealsize);
    const float fdrr = __fmul_rn(-fdrow, firealsize), fdcr = __fmul_rn(-fdcol, firealsize);
    const int side = 2 * win + 1;
    // ---- 0: the rows' intervals (a window of more than DESC_MAX_SIDE rows -- none with the shipped constants: side <= 97
    // -- is walked whole, a sample's row and column by division)
    const bool compact = side <= DESC_MAX_SIDE;
    int total = side * side;
    {   // the key's place in the list (keys are distinct)
      int later = 0;
      for (int j = tid; j < n; j += 64 * DESC_WAVES) later += (int)(keys[j].order > k.order);
      later = wave_incl_scan(later);
      if (lane == 63) L.later[wave] = later;
    }
    {
      if (compact && wave == 0) {
        int sum = 0;
        for (int base = 0; base < side; base += 64) {
          const int rr = base + lane;
          int lo = 0, len = 0;
          if (rr < side) {
            int hi;
            desc_row_interval(fsr, fcr, fdrr, fdcr, rr - win, win, rowstart, colstart, rows, cols, lo, hi);
            len = hi - lo + 1;
          }
          const int incl = wave_incl_scan(len);
          if (rr < side) {
            L.rowbeg[rr] = sum + incl - len;
            L.rowlo[rr] = (short)lo;
          }
          sum += __builtin_amdgcn_readlane(incl, 63);
        }
        if (lane == 0) L.rowbeg[side] = sum;
        wave_sync();
        // the row of every 64th sample: a lane finds its sample's row from there without walking the rows (four
        // wavefronts share the walk, so a lane moves 256 samples = 6 - 12 rows per step: the walk -- dependent LDS
        // reads in a divergent loop -- was the longest phase of a step, SIFT_PROF)
        for (int base = 0; base < side; base += 64) {
          const int rr = base + lane;
          if (rr < side) {
            const int b0 = L.rowbeg[rr], b1 = L.rowbeg[rr + 1];
            for (int c = (b0 + 63) >> 6; (c << 6) < b1; ++c) L.chunkrow[c] = (unsigned char)rr;
          }
        }
      }
      __syncthreads();
      if (compact) total = L.rowbeg[side];
    }
    DPF(0);
    // sample t of the walk: its window row / column, and its two values when its pixel exists
    struct Sample {
      bool have, edge_c, edge_r;
      int row, col;
      float a, b, u, d;   // the pixel's left / right / upper / lower neighbour
    };
    auto fetch = [&](int t) -> Sample {
      Sample q{false, false, false, 0, 0, 0.f, 0.f, 0.f, 0.f};
      if (t >= total) return q;
      if (compact) {
        // the row of the chunk's first sample, then at most a few rows on (four bounds read at once; rows of fewer than 16
        // samples -- the rotated square's corners -- go on one by one)
        int rho = L.chunkrow[t >> 6];
        const int b1 = L.rowbeg[rho + 1], b2 = L.rowbeg[min(rho + 2, side)], b3 = L.rowbeg[min(rho + 3, side)],
                  b4 = L.rowbeg[min(rho + 4, side)];
        rho += (int)(t >= b1) + (int)(t >= b2) + (int)(t >= b3) + (int)(t >= b4);
        while (t >= L.rowbeg[rho + 1]) ++rho;
        q.row = rho - win;
        q.col = (int)L.rowlo[rho] + (t - L.rowbeg[rho]);
      } else {
        q.row = t / side - win;
        q.col = t % side - win;
      }
      const int r = rowstart + q.row, c = colstart + q.col;
      if (r >= 0 && r < rows && c >= 0 && c < cols) {
        q.have = true;
        const GradTaps t = grad_taps(rows, cols, r, c);
        q.a = im[t.a];
        q.b = im[t.b];
        q.u = im[t.u];
        q.d = im[t.d];
        q.edge_c = t.edge_c;
        q.edge_r = t.edge_r;
      }
      return q;
    };
    // ---- A: the wavefront's 64 samples into its lists of buffer `buf`
    auto phase_a = [&](const Sample& q, int buf) {
      unsigned long long* const mask = L.mask[buf][wave];
      uint16_t* const offs = L.off[buf][wave];
      float* const val = L.val[buf][wave];
      bool ok = false;
      float mag = 0.f, rf = 0.f, cf = 0.f, of = 0.f;
      int nr = 0, nc = 0, nb = 0;
      if (q.have) {
        const float fr = (float)q.row, fc = (float)q.col;
        const float rpos = __fadd_rn(__fadd_rn(__fmul_rn(fsr, fc), __fmul_rn(fcr, fr)), fdrr);
        const float cpos = __fadd_rn(__fsub_rn(__fmul_rn(fcr, fc), __fmul_rn(fsr, fr)), fdcr);
        const float rx = __fadd_rn(rpos, 2.0f - 0.5f), cx = __fadd_rn(cpos, 2.0f - 0.5f);
        if (rx > -0.9999f && rx < 3.9999f && cx > -0.9999f && cx < 3.9999f) {
          ok = true;
          const float e = expf(__fmul_rn(-0.125f, __fadd_rn(__fmul_rn(rpos, rpos), __fmul_rn(cpos, cpos))));
          float gdc, gdr;
          grad_of(q.a, q.b, q.u, q.d, q.edge_c, q.edge_r, gdc, gdr);
          mag = __fmul_rn(grad_mag(gdc, gdr), e);
          float oo = __fsub_rn(atan2f(gdr, gdc), ang);
          while (oo > 2 * kPi) oo = __fsub_rn(oo, 2 * kPi);
          while (oo < 0) oo = __fadd_rn(oo, 2 * kPi);
          const float oribin = __fmul_rn(oo, 8.0f / (2 * (float)kPi));   // PlaceInIndex
          nr = rx < 0 ? (int)__fsub_rn(rx, 1.f) : (int)rx;
          rf = __fsub_rn(rx, (float)nr);
          nc = cx < 0 ? (int)__fsub_rn(cx, 1.f) : (int)cx;
          cf = __fsub_rn(cx, (float)nc);
          const int no = oribin < 0 ? (int)__fsub_rn(oribin, 1.f) : (int)oribin;
          of = __fsub_rn(oribin, (float)no);
          nb = no & 7;   // the bins wrap: orientation 2 pi falls into bin 8 = bin 0
        }
      }
      DPF(2);
      // the 16 masks: rows of cells in lanes 0..3, columns in 4..7, bins in 8..15
      {
        const int nrk = ok ? nr : -100, nck = ok ? nc : -100;
        unsigned long long keep = 0ull;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned long long m = __ballot((unsigned)(i - nrk) < 2u);   // nr == i - 1 or nr == i
          if (lane == i) keep = m;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned long long m = __ballot((unsigned)(i - nck) < 2u);
          if (lane == 4 + i) keep = m;
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const unsigned long long m = __ballot(ok && (unsigned)((b - nb) & 7) < 2u);   // nb == b or nb + 1 == b (mod 8)
          if (lane == 8 + b) keep = m;
        }
        if (lane < 16) mask[lane] = keep;
      }
      wave_sync();
      // sizes of lists 2 lane, 2 lane + 1 (cell lane / 4, bins 2 (lane % 4), + 1) and their places
      {
        const int cell = lane >> 2;
        const unsigned long long cm = mask[cell >> 2] & mask[4 + (cell & 3)];
        const int c0 = __popcll(cm & mask[8 + 2 * (lane & 3)]);
        const int c1 = __popcll(cm & mask[9 + 2 * (lane & 3)]);
        const int incl = wave_incl_scan(c0 + c1);
        const int excl = incl - (c0 + c1);
        offs[2 * lane] = (uint16_t)excl;
        offs[2 * lane + 1] = (uint16_t)(excl + c0);
        if (lane == 63) offs[128] = (uint16_t)incl;
      }
      wave_sync();
      DPF(3);
      if (ok) {
        const float rg0 = __fmul_rn(mag, __fsub_rn(1.f, rf)), rg1 = __fmul_rn(mag, rf);   // rows nr, nr + 1
        const float cfm = __fsub_rn(1.f, cf), ofm = __fsub_rn(1.f, of);
        const int b1 = (nb + 1) & 7;
        const unsigned long long bm0 = mask[8 + nb] & lt, bm1 = mask[8 + b1] & lt;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r_ = nr + i;
          if ((unsigned)r_ >= 4u) continue;
          const unsigned long long rm = mask[r_];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c_ = nc + j;
            if ((unsigned)c_ >= 4u) continue;
            const int cell = 4 * r_ + c_;
            const float rg = i == 0 ? rg0 : rg1;
            const float cg = j == 0 ? __fmul_rn(rg, cfm) : __fmul_rn(rg, cf);
            const unsigned long long cm = rm & mask[4 + c_];
            val[offs[8 * cell + nb] + __popcll(cm & bm0)] = __fmul_rn(cg, ofm);
            val[offs[8 * cell + b1] + __popcll(cm & bm1)] = __fmul_rn(cg, of);
          }
        }
      }
      DPF(4);
    };
    const int per_step = 64 * DESC_WAVES;
    const int n_steps = (total + per_step - 1) / per_step;
    // step i's sample of this lane: t = i per_step + tid (raster order over the wavefronts, then the lanes)
    Sample cur = fetch(tid), nxt = fetch(per_step + tid);
    float acc = 0.f;
    DPF(1);
    if (n_steps > 0) phase_a(cur, 0);
    for (int i = 0; i < n_steps; ++i) {
      __syncthreads();   // step i's lists are written, step i - 1's are read: step i + 1 may overwrite those
      DPF(5);
      if (i + 1 < n_steps) {
        cur = nxt;
        nxt = fetch((i + 2) * per_step + tid);
      }
      DPF(1);
      // ---- B ----
      if (tid < 128) {
        const int buf = i & 1;
        int e0[DESC_WAVES], e1[DESC_WAVES];   // every wavefront's list bounds first: one trip to LDS, not one per list
#pragma unroll
        for (int w = 0; w < DESC_WAVES; ++w) {
          e0[w] = L.off[buf][w][tid];
          e1[w] = L.off[buf][w][tid + 1];
        }
#if MH_DESC_B > 0
        // the first MH_DESC_B values of every list requested before the first addition
        float x[DESC_WAVES][MH_DESC_B];
#pragma unroll
        for (int w = 0; w < DESC_WAVES; ++w)
#pragma unroll
          for (int j = 0; j < MH_DESC_B; ++j) x[w][j] = L.val[buf][w][min(e0[w] + j, 511)];
#pragma unroll
        for (int w = 0; w < DESC_WAVES; ++w) {
#pragma unroll
          for (int j = 0; j < MH_DESC_B; ++j)
            if (e0[w] + j < e1[w]) acc = __fadd_rn(acc, x[w][j]);
          for (int e = e0[w] + MH_DESC_B; e < e1[w]; ++e) acc = __fadd_rn(acc, L.val[buf][w][e]);
        }
#else
#pragma unroll
        for (int w = 0; w < DESC_WAVES; ++w) {
          const float* v = L.val[buf][w];
          for (int e = e0[w]; e < e1[w]; e += 4) {
            // four at a time: the loads do not depend on the running sum (reads past the list's end stay inside val)
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = v[min(e + j, 511)];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (e + j < e1[w]) acc = __fadd_rn(acc, x[j]);
          }
        }
#endif
      }
      DPF(6);
      if (i + 1 < n_steps) phase_a(cur, (i + 1) & 1);
    }
    // NormalizeVec, clamp at 0.2, NormalizeVec again if anything was clamped (:1497-1527); the squares are summed in the
    // entries' order by one thread
    float d = acc;
    for (int pass = 0; pass < 2; ++pass) {
      if (tid < 128) L.sq[tid] = __fmul_rn(d, d);
      __syncthreads();
      if (tid == 0) {
        float a = 0.f;
        const float4* q = reinterpret_cast<const float4*>(L.sq);
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
          const float4 v = q[i];
          a = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(a, v.x), v.y), v.z), v.w);
        }
        L.scal = __fdiv_rn(1.f, sqrtf(a));
      }
      __syncthreads();
      bool clamp = false;
      if (tid < 128) {
        d = __fmul_rn(d, L.scal);
        if (pass == 0 && d > 0.2f) {
          d = 0.2f;
          clamp = true;
        }
      }
      const int any = __syncthreads_or(clamp ? 1 : 0);
      if (!any) break;
    }
    int dst = 0;   // number of keys generated after this one
#pragma unroll
    for (int w = 0; w < DESC_WAVES; ++w) dst += L.later[w];
    if (dst < out_cap) {
      if (tid < 128) desc_out[(size_t)dst * 128 + tid] = d;
      if (tid == 0) {
        const float fscale = O.fscale;
        xy_out[2 * dst] = __fmul_rn(fscale, fcol);   // coord2D = (col, row), FEAT_SIFT_CPU.hpp:103-104
        xy_out[2 * dst + 1] = __fmul_rn(fscale, frow);
        if (scale_ori_out) {
          scale_ori_out[2 * dst] = __fmul_rn(fscale, fSize);
          scale_ori_out[2 * dst + 1] = ang;
        }
      }
    }
    __syncthreads();   // the next key rewrites the rows and the lists
    DPF(7);
#ifdef SIFT_PROF
    if (prof_on) {
      prof_acc[8] += 1;
      prof_acc[9] += n_steps;
    }
#endif
  }
#ifdef SIFT_PROF
  if (prof_on)
    for (int i = 0; i < 10; ++i) atomicAdd(&g_sift_prof[(wave == 0 ? 0 : 10) + i], prof_acc[i]);
#endif
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
    O.owner = own;
    own += px;
  }
  // (the counters are cleared by prepare_kernel; the owner map is not cleared at all: SiftBatch::own_prefix)
  {
    unsigned key_bits = 1;
    while (key_bits < 32 && (1ull << key_bits) < (unsigned long long)kScales * plan.rows[0] * plan.cols[0]) ++key_bits;
    const unsigned n_prefix = key_bits >= 31 ? 1u : 1u << (32 - key_bits);   // (a key is below kScales rows cols of octave 0)
    if (*B.own_epoch == 0 || n_prefix == 1) {
      hipMemsetAsync(B.owner, 0xFF, B.owner_elems * sizeof(unsigned int) * B.images, s);
      *B.own_epoch = n_prefix;
    }
    --*B.own_epoch;
    Bt.own_prefix = key_bits >= 32 ? 0u : *B.own_epoch << key_bits;
  }

  const SiftOctave& O0 = P.oct[0];
  const dim3 tb(256);
  auto grid_for = [un](int rows, int cols) { return dim3((cols + 255) / 256, rows, un); };
  const float fnew = double_size ? 1.0f : 0.5f;
  const bool init_blur = kInitSigma > fnew;   // :325-327
  const Taps t0 = init_blur ? make_taps(sqrtf(kInitSigma * kInitSigma - fnew * fnew)) : Taps{};
  const bool init_fused = init_blur && (t0.n >> 1) <= BT_MAXW;
  // the prepared image goes to the scratch image when the fused blur can write octave 0's first level from there.  (Round 5
  // tried the first blur on the frame's pixels directly -- prepared_px in its staging: the staging's loads no longer go
  // out together, every blur launch of the chain paid for it, 132 -> 175 us.)
  hipLaunchKernelGGL(prepare_kernel, grid_for(O0.rows, O0.cols), tb, 0, s, gray, width, height, double_size,
                     init_fused ? B.tmp : O0.gaus[0], O0.rows, O0.cols, imgs, init_fused ? Bt.tmp_step : Bt.pyr_step,
                     B.counters);
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
  // extrema of octaves [o_lo, o_hi) on stream st
  auto detect = [&](int o_lo, int o_hi, hipStream_t st) {
    if (o_hi <= o_lo) return;
    DetectGrid DG;
    DG.n = o_hi - o_lo;
    DG.o_first = o_lo;
    DG.begin[0] = 0;
    for (int o = o_lo; o < o_hi; ++o) {   // the scanned region: 5 pixels inside every border (:925)
      const int oi = o - o_lo, w = plan.cols[o] - 10, h = plan.rows[o] - 10;
      DG.tiles_x[oi] = w > 0 ? (w + DT_X - 1) / DT_X : 0;
      DG.begin[oi + 1] = DG.begin[oi] + (w > 0 && h > 0 ? DG.tiles_x[oi] * ((h + DT_Y - 1) / DT_Y) : 0);
    }
    if (DG.begin[DG.n] > 0)
      hipLaunchKernelGGL(detect_kernel, dim3(DG.begin[DG.n], un), dim3(DT_THREADS), 0, st, P, DG, B.cand, B.counters + 0,
                         B.cand_cap, B.counters + 2, Bt);
  };
  // The small octaves' blur chain, one workgroup per image.  (Round 5 tried it -- with the small octaves' gradients and
  // extrema -- on a stream of its own beside the last large octave's last two levels and the large octaves' gradients and
  // extrema, which need nothing of it: one image alone gained 13 us of the 58, the event hand-over between the streams
  // costs the rest, and sixteen contexts with two streams each share the hardware queues: 0.13 -> 0.26 ms per image with
  // four in flight.  One stream.)
  auto small_chain = [&]() {
    hipLaunchKernelGGL(small_octaves_kernel, dim3(un), dim3(1024), 0, s, P, o_small, T5, (size_t)Bt.pyr_step);
  };
  bool jobs_ok = true;   // every level's kernel fits the tile kernel's halo
  for (int i = 0; i < kScales + 2; ++i) jobs_ok = jobs_ok && (T5.t[i].n >> 1) <= BT_MAXW;
  if (jobs_ok) {
    // the large octaves as a dependency-ordered list of launches, two independent levels per launch where there are two
    auto make_job = [&](int o, int i) {   // level i of octave o from level i - 1 (or, i == 1 and o > 0, from octave o - 1)
      const SiftOctave& O = P.oct[o];
      BlurJob b;
      b.rows = O.rows;
      b.cols = O.cols;
      b.dst = i == kScales + 2 ? nullptr : O.gaus[i];   // (the last level: nobody reads it, its DoG is all that is kept)
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
    }
    if (o_small < plan.n_octaves) small_chain();
  } else {
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
    if (o_small < plan.n_octaves) small_chain();   // (the loop above wrote the half-size copy already: the same numbers again)
  }
  detect(0, plan.n_octaves, s);
  // (the per-key kernels loop over the keys: a batch's images share the chip, fewer workgroups per image)
  const unsigned per = n > 4 ? 4 : 1;
  hipLaunchKernelGGL(orient_kernel, dim3(4096 / per, un), dim3(64), 0, s, P, (const SiftCandidate*)B.cand,
                     (const int32_t*)(B.counters + 0), B.cand_cap, B.keys, B.counters + 1, B.key_cap, B.counters + 2, Bt);
  hipLaunchKernelGGL(describe_kernel, dim3(4096 / per, un), dim3(64 * DESC_WAVES), 0, s, P, (const SiftKey*)B.keys,
                     (const int32_t*)(B.counters + 1), B.key_cap, out_cap, n_out, desc_out, xy_out, scale_ori_out, Bt);
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

#ifdef SIFT_PROF
extern "C" int mh_debug_sift_prof(unsigned long long out[32], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mh::g_sift_prof), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mh::g_sift_prof), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif
