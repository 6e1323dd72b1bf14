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

__global__ void prepare_kernel(const uint8_t* __restrict__ gray, int w, int h, int double_size,
                               float* __restrict__ out, int orows, int ocols) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= ocols || r >= orows) return;
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

// HalfImageSize (:390-408)
__global__ void half_kernel(const float* __restrict__ src, int scols, float* __restrict__ dst, int rows, int cols) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= cols || r >= rows) return;
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
__global__ __launch_bounds__(1024) void small_octaves_kernel(SiftPyramid P, int o_first, Taps5 T) {
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
    const float* src0 = O.gaus[0];
    if (o > o_first) {   // HalfImageSize (:390-408) of the previous octave's image `kScales`
      const SiftOctave& V = P.oct[o - 1];
      const float* src = V.gaus[kScales];
      float* dst = O.gaus[0];
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
      float* dst = O.gaus[i];
      float* dog = O.dog[i - 1];
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
__global__ void grad_ori_kernel(SiftPyramid P) {
  const int o = blockIdx.z / kScales, index = 1 + blockIdx.z % kScales;
  if (o >= P.n_octaves) return;
  const SiftOctave& O = P.oct[o];
  const int rows = O.rows, cols = O.cols;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y * blockDim.y + threadIdx.y;
  if (j >= cols || i >= rows) return;
  const float* im = O.gaus[index];
  const float* p = im + (size_t)i * cols;
  float dc, dr;
  if (j == 0) dc = __fmul_rn(2.0f, __fsub_rn(p[1], p[0]));
  else if (j == cols - 1) dc = __fmul_rn(2.0f, __fsub_rn(p[j], p[j - 1]));
  else dc = __fsub_rn(p[j + 1], p[j - 1]);
  if (i == 0) dr = __fmul_rn(2.0f, __fsub_rn(p[j], p[cols + j]));
  else if (i == rows - 1) dr = __fmul_rn(2.0f, __fsub_rn(p[-cols + j], p[j]));
  else dr = __fsub_rn(p[-cols + j], p[cols + j]);
  const size_t at = (size_t)i * cols + j;
  O.grad[index - 1][at] = sqrtf(__fadd_rn(__fmul_rn(dc, dc), __fmul_rn(dr, dr)));
  O.ori[index - 1][at] = atan2f(dr, dc);
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
__global__ void detect_kernel(SiftPyramid P, SiftCandidate* __restrict__ cand, int32_t* __restrict__ n_cand,
                              int cap, int32_t* __restrict__ overflow) {
  const int o = blockIdx.z / kScales, index = 1 + blockIdx.z % kScales;
  if (o >= P.n_octaves) return;
  const SiftOctave& O = P.oct[o];
  const int rows = O.rows, cols = O.cols;
  const int c = 5 + blockIdx.x * blockDim.x + threadIdx.x;
  const int r = 5 + blockIdx.y * blockDim.y + threadIdx.y;
  if (c >= cols - 5 || r >= rows - 5) return;
  const float peak_thresh = 0.04f / (float)kScales;
  const float* d1 = O.dog[index];
  const float v = d1[(size_t)r * cols + c];
  if (!(fabsf(v) > __fmul_rn(peak_thresh, 0.8f))) return;
  const float *d0 = O.dog[index - 1], *d2 = O.dog[index + 1];
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
  atomicMin(&O.owner[(size_t)rr * cols + cc], key);
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
                                                    int key_cap, int32_t* __restrict__ overflow) {
  const int lane = threadIdx.x;
  int n = *n_cand;
  if (n > cand_cap) n = cand_cap;
  for (int ci = blockIdx.x; ci < n; ci += gridDim.x) {
    const SiftCandidate k = cand[ci];
    const SiftOctave& O = P.oct[k.octave];
    const int rows = O.rows, cols = O.cols;
    if (O.owner[(size_t)k.r * cols + k.c] != k.key) continue;  // another extremum got this pixel first
    const float* grad = O.grad[k.index - 1];
    const float* orim = O.ori[k.index - 1];
    const float fSize = __fmul_rn(kInitSigma, powf(2.0f, __fdiv_rn(__fadd_rn((float)k.index, k.x0), (float)kScales)));
    const float frow = __fadd_rn((float)k.r, k.x1), fcol = __fadd_rn((float)k.c, k.x2);
    const int rowstart = (int)__fadd_rn(frow, 0.5f), colstart = (int)__fadd_rn(fcol, 0.5f);
    const float fexpmult = __fdiv_rn(-1.0f, __fmul_rn(__fmul_rn(__fmul_rn(__fmul_rn(2.0f, 1.5f), 1.5f), fSize), fSize));
    const float fbinmult = 36.0f / (2 * kPi);
    const float fbinadd = (float)(kPi + 0.001f) * fbinmult;
    const int win = (int)__fmul_rn(__fmul_rn(fSize, 1.5f), 3.0f);
    const int side = 2 * win + 1, total = side * side;
    float h = 0.f;  // lane's bin
    for (int base = 0; base < total; base += 64) {
      const int s = base + lane;
      int bin = -1;
      float val = 0.f;
      if (s < total) {
        const int r = rowstart - win + s / side, c = colstart - win + s % side;
        if (r >= 0 && r < rows - 2 && c >= 0 && c < cols - 2) {
          const float g = grad[(size_t)r * cols + c];
          if (g > 0) {
            const float dr = __fsub_rn((float)r, frow), dc = __fsub_rn((float)c, fcol);
            const float rad2 = __fadd_rn(__fmul_rn(dr, dr), __fmul_rn(dc, dc));
            if (__fadd_rn((float)(win * win), 0.5f) > rad2) {
              const float w = expf(__fmul_rn(rad2, fexpmult));
              bin = (int)__fadd_rn(__fmul_rn(orim[(size_t)r * cols + c], fbinmult), fbinadd);
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
// Lane L of a wavefront owns descriptor entries 2L and 2L+1 (cell L/4 of the 4x4 grid, orientation
// bins 2(L%4) and 2(L%4)+1).  Samples are evaluated 64 at a time in raster order and then folded
// one after the other, each lane taking its share of the sample's up to eight contributions --
// the same products, added in the same order, as the serial code.
// Workgroups have eight wavefronts.  With many keys each wavefront describes its own key; with few
// (the chip would be mostly idle and the serial fold is what a key waits for) the eight wavefronts
// share ONE key, wavefront (r, h) folding only the samples that touch cell row r and cell columns
// 2h, 2h + 1 (about a quarter of them) on its first eight lanes: the entries are independent, so the
// per-entry addition order is untouched.  (Four wavefronts, a cell row each: 0.86 ms per frame instead
// of 0.83; sixteen, a cell each: 1.01 -- every wavefront evaluates all the samples.)
constexpr int DESC_WAVES = 8;
constexpr int DESC_SHARE_BELOW = 1536;   // keys; below this the wavefronts of a workgroup share a key
__global__ __launch_bounds__(64 * DESC_WAVES) void describe_kernel(SiftPyramid P, const SiftKey* __restrict__ keys,
                                                      const int32_t* __restrict__ n_keys, int key_cap,
                                                      float* __restrict__ desc_out /* [key][128] */,
                                                      float* __restrict__ geo_out /* [key][4] col,row,scale,ori */) {
  __shared__ float d_all[DESC_WAVES][128];
  __shared__ float scal_all[DESC_WAVES];
  __shared__ int any_all[DESC_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int n = *n_keys;
  if (n > key_cap) n = key_cap;
  const bool share = n < DESC_SHARE_BELOW;
  // shared key: wavefront `wave` owns cell row wave / 2, columns 2 (wave % 2) + {0, 1} on its first 8 lanes;
  // results meet in d_all[0]
  const int cell_r = share ? (wave >> 1) : (lane >> 4);
  const int col_lo = 2 * (wave & 1);                       // shared key: this wavefront's two cell columns
  const int cell_c = share ? col_lo + ((lane >> 2) & 1) : ((lane >> 2) & 3), ob0 = 2 * (lane & 3);
  const bool owner = !share || lane < 8;
  float* const d_s = share ? d_all[0] : d_all[wave];
  float& scal_s = share ? scal_all[0] : scal_all[wave];
  int& any_s = share ? any_all[0] : any_all[wave];
  // a key's 128 values belong to one wavefront, or (shared key) to the workgroup
  auto sync_scope = [&]() {
    if (share) {
      __syncthreads();
    } else {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };
  const int entry = share ? ((cell_r * 4 + cell_c) * 4 + (lane & 3)) : lane;   // this lane's pair of entries: 2 entry, 2 entry + 1
  const int k_first = share ? blockIdx.x : blockIdx.x * DESC_WAVES + wave;
  const int k_step = share ? gridDim.x : gridDim.x * DESC_WAVES;
  const int n_round = share ? n : ((n + DESC_WAVES - 1) / DESC_WAVES) * DESC_WAVES;   // whole workgroups reach the barriers
  for (int kb = k_first - (share ? 0 : wave); kb < n_round; kb += k_step) {
    const int ki = share ? kb : kb + wave;
    const bool live = ki < n;
    const SiftKey k = keys[live ? ki : 0];
    const SiftOctave& O = P.oct[k.octave];
    const int rows = O.rows, cols = O.cols;
    const float* grad = O.grad[k.index - 1];
    const float* orim = O.ori[k.index - 1];
    const float fSize = k.fsize, frow = k.frow, fcol = k.fcol, ang = k.ori;
    const int rowstart = (int)__fadd_rn(frow, 0.5f), colstart = (int)__fadd_rn(fcol, 0.5f);
    const float sinang = sinf(ang), cosang = cosf(ang);
    const float fdrow = __fsub_rn(frow, (float)rowstart), fdcol = __fsub_rn(fcol, (float)colstart);
    const float frealsize = __fmul_rn(3.0f, fSize), firealsize = __fdiv_rn(1.0f, __fmul_rn(3.0f, fSize));
    const int win = (int)__fadd_rn(__fmul_rn(__fmul_rn(__fmul_rn(frealsize, kSqrt2), 5.0f), 0.5f), 0.5f);
    const float fsr = __fmul_rn(sinang, firealsize), fcr = __fmul_rn(cosang, firealsize);
    const float fdrr = __fmul_rn(-fdrow, firealsize), fdcr = __fmul_rn(-fdcol, firealsize);
    const int side = 2 * win + 1, total = live ? side * side : 0;
    float acc0 = 0.f, acc1 = 0.f;
    for (int base = 0; base < total; base += 64) {
      const int s = base + lane;
      bool ok = false;
      float mag = 0.f, rf = 0.f, cf = 0.f, of = 0.f;
      int nr = 0, nc = 0, no = 0;
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
          no = oribin < 0 ? (int)__fsub_rn(oribin, 1.f) : (int)oribin;
          of = __fsub_rn(oribin, (float)no);
        }
      }
      // shared key: only the samples that reach this wavefront's cells (rows nr, nr + 1; columns nc, nc + 1)
      unsigned long long m = __ballot(ok && (!share || ((nr == cell_r - 1 || nr == cell_r) && nc >= col_lo - 1 && nc <= col_lo + 1)));
      while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
#define RL_F(x) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src))
        const int snr = __builtin_amdgcn_readlane(nr, src), snc = __builtin_amdgcn_readlane(nc, src);
        const int sno = __builtin_amdgcn_readlane(no, src);
        const float smag = RL_F(mag), srf = RL_F(rf), scf = RL_F(cf), sof = RL_F(of);
#undef RL_F
        const int i = cell_r - snr, j = cell_c - snc;
        if (owner && (unsigned)i < 2u && (unsigned)j < 2u) {
          const float rg = i == 0 ? __fmul_rn(smag, __fsub_rn(1.f, srf)) : __fmul_rn(smag, srf);
          const float cg = j == 0 ? __fmul_rn(rg, __fsub_rn(1.f, scf)) : __fmul_rn(rg, scf);
          const int k0 = (ob0 - sno) & 7, k1 = (ob0 + 1 - sno) & 7;
          if (k0 < 2) acc0 = __fadd_rn(acc0, k0 == 0 ? __fmul_rn(cg, __fsub_rn(1.f, sof)) : __fmul_rn(cg, sof));
          if (k1 < 2) acc1 = __fadd_rn(acc1, k1 == 0 ? __fmul_rn(cg, __fsub_rn(1.f, sof)) : __fmul_rn(cg, sof));
        }
      }
    }
    sync_scope();
    if (owner) {
      d_s[2 * entry] = acc0;
      d_s[2 * entry + 1] = acc1;
    }
    sync_scope();
    // NormalizeVec, clamp at 0.2, NormalizeVec again if anything was clamped (:1497-1527); one
    // wavefront works on the 128 values (its own key's, or the shared key's on wavefront 0)
    const bool norm_wave = !share || wave == 0;
    for (int pass = 0; pass < 2; ++pass) {
      if (norm_wave && lane == 0) {
        float a = 0.f;
        for (int i = 0; i < 128; ++i) a = __fadd_rn(a, __fmul_rn(d_s[i], d_s[i]));
        scal_s = __fdiv_rn(1.f, sqrtf(a));
      }
      sync_scope();
      bool clamp = false;
      if (norm_wave) {
        const float sc = scal_s;
        d_s[2 * lane] = __fmul_rn(d_s[2 * lane], sc);
        d_s[2 * lane + 1] = __fmul_rn(d_s[2 * lane + 1], sc);
        if (pass == 0) {
          if (d_s[2 * lane] > 0.2f) {
            d_s[2 * lane] = 0.2f;
            clamp = true;
          }
          if (d_s[2 * lane + 1] > 0.2f) {
            d_s[2 * lane + 1] = 0.2f;
            clamp = true;
          }
        }
        const bool any_w = __ballot(clamp) != 0ull;
        if (lane == 0) any_s = any_w ? 1 : 0;
      }
      sync_scope();
      const bool any = any_s != 0;
      sync_scope();
      if (!any) break;
    }
    if (live && norm_wave) {
      float* out = desc_out + (size_t)ki * 128;
      out[2 * lane] = d_s[2 * lane];
      out[2 * lane + 1] = d_s[2 * lane + 1];
      if (lane == 0) {
        const float fscale = O.fscale;
        float* g = geo_out + (size_t)ki * 4;
        g[0] = __fmul_rn(fscale, fcol);   // coord2D = (col, row), FEAT_SIFT_CPU.hpp:103-104
        g[1] = __fmul_rn(fscale, frow);
        g[2] = __fmul_rn(fscale, fSize);
        g[3] = ang;
      }
    }
    sync_scope();
  }
}

// ---- order: the reference's list = generation order reversed (every key is pushed on the
// front of a linked list, :1432, :944-952) -------------------------------------------------------
__global__ void order_kernel(const SiftKey* __restrict__ keys, const int32_t* __restrict__ n_keys, int key_cap,
                             const float* __restrict__ desc_in, const float* __restrict__ geo_in, int out_cap,
                             float* __restrict__ desc_out, float* __restrict__ xy_out,
                             float* __restrict__ scale_ori_out, int32_t* __restrict__ n_out) {
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

void launch_sift(const uint8_t* gray, int width, int height, int double_size, const SiftPlan& plan,
                 const SiftBuffers& B, int out_cap, float* desc_out, float* xy_out, float* scale_ori_out,
                 int32_t* n_out, hipStream_t s) {
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
  hipMemsetAsync(B.owner, 0xFF, B.owner_elems * sizeof(unsigned int), s);
  hipMemsetAsync(B.counters, 0, 4 * sizeof(int32_t), s);

  const SiftOctave& O0 = P.oct[0];
  const dim3 tb(256);
  auto grid_for = [](int rows, int cols) { return dim3((cols + 255) / 256, rows); };
  const float fnew = double_size ? 1.0f : 0.5f;
  const bool init_blur = kInitSigma > fnew;   // :325-327
  const Taps t0 = init_blur ? make_taps(sqrtf(kInitSigma * kInitSigma - fnew * fnew)) : Taps{};
  const bool init_fused = init_blur && (t0.n >> 1) <= BT_MAXW;
  // the prepared image goes to the scratch image when the fused blur can write octave 0's first level from there
  hipLaunchKernelGGL(prepare_kernel, grid_for(O0.rows, O0.cols), tb, 0, s, gray, width, height, double_size,
                     init_fused ? B.tmp : O0.gaus[0], O0.rows, O0.cols);
  if (init_fused) {
    hipLaunchKernelGGL(blur_level_kernel, dim3((O0.cols + BT_X - 1) / BT_X, (O0.rows + BT_Y - 1) / BT_Y), dim3(BT_THREADS), 0, s,
                       (const float*)B.tmp, O0.gaus[0], O0.rows, O0.cols, t0, (const float*)nullptr, (float*)nullptr);
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
                         N.gaus[0], N.rows, N.cols);
    }
  }
  if (o_small < plan.n_octaves)
    hipLaunchKernelGGL(small_octaves_kernel, dim3(1), dim3(1024), 0, s, P, o_small, T5);
  const dim3 tb2(64, 4);
  const dim3 g2((O0.cols + 63) / 64, (O0.rows + 3) / 4, plan.n_octaves * kScales);
  hipLaunchKernelGGL(grad_ori_kernel, g2, tb2, 0, s, P);
  hipLaunchKernelGGL(detect_kernel, g2, tb2, 0, s, P, B.cand, B.counters + 0, B.cand_cap, B.counters + 2);
  hipLaunchKernelGGL(orient_kernel, dim3(1024), dim3(64), 0, s, P, (const SiftCandidate*)B.cand,
                     (const int32_t*)(B.counters + 0), B.cand_cap, B.keys, B.counters + 1, B.key_cap, B.counters + 2);
  hipLaunchKernelGGL(describe_kernel, dim3(2048), dim3(64 * DESC_WAVES), 0, s, P, (const SiftKey*)B.keys,
                     (const int32_t*)(B.counters + 1), B.key_cap, B.desc_tmp, B.geo_tmp);
  hipLaunchKernelGGL(order_kernel, dim3(1024), dim3(64), 0, s, (const SiftKey*)B.keys,
                     (const int32_t*)(B.counters + 1), B.key_cap, (const float*)B.desc_tmp, (const float*)B.geo_tmp,
                     out_cap, desc_out, xy_out, scale_ori_out, n_out);
}

}  // namespace mh
