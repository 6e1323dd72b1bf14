// oracle/sift_oracle.cpp -- TEST INFRASTRUCTURE ONLY (part of liboracle.so).
//
// CPU restatement of the SIFT extractor behind FEAT_SIFT_CPU
// (moped2/libmoped/src/feat/FEAT_SIFT_CPU.hpp:78-112): libsiftfast 1.1
// (moped2/libmoped/libs/libs.tgz -> libsiftfast-1.1-src/libsiftfast.cpp) in the arithmetic
// MOPED actually builds: its copy of the library `#undef`s __SSE__/__SSE2__/__SSE3__
// (:39-41), so the plain-C branches run (ConvHorizontal / ConvVertical / GradOriImages with
// libm's atan2f, two-pass descriptor normalisation).
//
// PINNED: bit-identical -- keypoint count, order, position, scale, orientation and all 128
// descriptor values -- to the reference's own build of that library (oracle/_ref: ref_sift2,
// run with one OpenMP thread) on the five frames of moped2/test_data/timing.bag
// (tests/golden/sift_ref_frames.npz, tests/test_sift_cpu.py).  With more threads the
// reference's list ORDER depends on thread timing (rows are spliced into the list under
// `omp critical`, :944-952); the order here is the single-thread one.
//
// Line numbers below refer to libsiftfast.cpp.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

constexpr float kPi = 3.141592654f;   // :65
constexpr float kSqrt2 = 1.4142136f;  // :66
constexpr int kScales = 3;            // :108
constexpr float kInitSigma = 1.6f;    // :109

struct Img {
  int rows = 0, cols = 0;
  std::vector<float> px;
  Img() {}
  Img(int r, int c) : rows(r), cols(c), px((size_t)r * c, 0.f) {}
  float& at(int r, int c) { return px[(size_t)r * cols + c]; }
  float at(int r, int c) const { return px[(size_t)r * cols + c]; }
};

// GaussianBlur's kernel (:470-506): ksize odd, >= 3; weights expf(-(i-w)^2 / (2 s^2)) for
// i = 0..ksize INCLUSIVE summed in double, only the first ksize of them divided by the sum.
std::vector<float> gauss_kernel(float fblur) {
  const float trunc = 4.0f;
  int ksize = (int)(2.0f * trunc * fblur + 1.0f);
  if (ksize < 3) ksize = 3;
  ksize += !(ksize & 1);
  const int width = ksize >> 1;
  std::vector<float> k(ksize + 1);
  double acc = 0;
  for (int i = 0; i <= ksize; ++i) {
    const float w = expf(-(float)(i - width) * (i - width) / (2.0f * fblur * fblur));
    acc += (double)w;
    k[i] = w;
  }
  for (int i = 0; i < ksize; ++i) k[i] /= (float)acc;
  k.resize(ksize);
  return k;
}

// ConvHorizontal + ConvVertical (:523-584): edge pixels replicated, float accumulator,
// taps added in ascending order; rows first, then columns of the row-blurred image.
void blur(Img& dst, const Img& src, float fblur) {
  const std::vector<float> k = gauss_kernel(fblur);
  const int ks = (int)k.size(), w = ks >> 1, rows = src.rows, cols = src.cols;
  Img tmp(rows, cols);
  std::vector<float> buf((size_t)std::max(rows, cols) + ks);
  for (int r = 0; r < rows; ++r) {
    for (int j = 0; j < w; ++j) buf[j] = src.at(r, 0);
    for (int j = 0; j < cols; ++j) buf[w + j] = src.at(r, j);
    for (int j = 0; j < w; ++j) buf[cols + w + j] = src.at(r, cols - 1);
    for (int i = 0; i < cols; ++i) {
      float a = 0;
      for (int j = 0; j < ks; ++j) a += buf[i + j] * k[j];
      tmp.at(r, i) = a;
    }
  }
  if (dst.rows != rows || dst.cols != cols) dst = Img(rows, cols);
  for (int c = 0; c < cols; ++c) {
    for (int i = 0; i < w; ++i) buf[i] = tmp.at(0, c);
    for (int i = 0; i < rows; ++i) buf[w + i] = tmp.at(i, c);
    for (int i = 0; i < w; ++i) buf[rows + w + i] = tmp.at(rows - 1, c);
    for (int i = 0; i < rows; ++i) {
      float a = 0;
      for (int j = 0; j < ks; ++j) a += buf[i + j] * k[j];
      dst.at(i, c) = a;
    }
  }
}

// GradOriImages (:959-992)
void grad_ori(const Img& im, Img& grad, Img& ori) {
  const int rows = im.rows, cols = im.cols;
  grad = Img(rows, cols);
  ori = Img(rows, cols);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) {
      float dc, dr;
      if (j == 0) dc = 2.0f * (im.at(i, 1) - im.at(i, 0));
      else if (j == cols - 1) dc = 2.0f * (im.at(i, j) - im.at(i, j - 1));
      else dc = im.at(i, j + 1) - im.at(i, j - 1);
      if (i == 0) dr = 2.0f * (im.at(0, j) - im.at(1, j));
      else if (i == rows - 1) dr = 2.0f * (im.at(i - 1, j) - im.at(i, j));
      else dr = im.at(i - 1, j) - im.at(i + 1, j);
      grad.at(i, j) = sqrtf(dc * dc + dr * dr);
      ori.at(i, j) = atan2f(dr, dc);
    }
}

// LocalMaxMin (:1126-1147)
bool local_extremum(float v, const Img& d, int r, int c) {
  for (int rr = r - 1; rr <= r + 1; ++rr)
    for (int cc = c - 1; cc <= c + 1; ++cc) {
      const float o = d.at(rr, cc);
      if (v > 0 ? o > v : v > o) return false;
    }
  return true;
}

// NotOnEdge (:1149-1162)
bool not_on_edge(const Img& d, int r, int c) {
  const float f1 = d.at(r - 1, c) - d.at(r, c) * 2 + d.at(r + 1, c);
  const float f2 = d.at(r, c - 1) - d.at(r, c) * 2 + d.at(r, c + 1);
  const float f3 = d.at(r + 1, c + 1) - d.at(r + 1, c - 1);
  const float f4 = d.at(r - 1, c + 1) - d.at(r - 1, c - 1);
  const float f5 = (f3 - f4) * 0.25f;
  const float f6 = f1 * f2 - f5 * f5;
  const float f8 = f1 + f2;
  return f6 * 11 * 11 > f8 * f8 * 10;
}

// SolveLinearSystem (:1235-1272): Gaussian elimination with row pivoting, 3x3
void solve3(float* Y, float* H) {
  const int dim = 3;
  int best = 0;
  for (int i = 0; i < dim - 1; ++i) {
    float fmax = -1;
    for (int j = i; j < dim; ++j) {
      float f = H[j * dim + i];
      if (f < 0) f = -f;
      if (f > fmax) {
        fmax = f;
        best = j;
      }
    }
    if (best != i) {
      for (int j = 0; j < dim; ++j) std::swap(H[best * dim + j], H[i * dim + j]);
      std::swap(Y[best], Y[i]);
    }
    for (int j = i + 1; j < dim; ++j) {
      const float f = H[j * dim + i] / H[i * dim + i];
      for (int k = i; k < dim; ++k) H[j * dim + k] -= f * H[i * dim + k];
      Y[j] -= Y[i] * f;
    }
  }
  for (int i = dim - 1; i >= 0; --i) {
    for (int j = dim - 1; j > i; --j) Y[i] -= Y[j] * H[i * dim + j];
    Y[i] /= H[i * dim + i];
  }
}

// FitQuadratic (:1208-1231)
float fit_quadratic(float* X, const Img* dog, int index, int r, int c) {
  const Img &p0 = dog[index - 1], &p1 = dog[index], &p2 = dog[index + 1];
  float Y[3], H[9];
  Y[0] = 0.5f * (p2.at(r, c) - p0.at(r, c));
  Y[1] = 0.5f * (p1.at(r + 1, c) - p1.at(r - 1, c));
  Y[2] = 0.5f * (p1.at(r, c + 1) - p1.at(r, c - 1));
  H[0] = p0.at(r, c) - 2.0f * p1.at(r, c) + p2.at(r, c);
  H[4] = p1.at(r - 1, c) - 2.0f * p1.at(r, c) + p1.at(r + 1, c);
  H[8] = p1.at(r, c - 1) - 2.0f * p1.at(r, c) + p1.at(r, c + 1);
  H[3] = H[1] = 0.25f * ((p2.at(r + 1, c) - p2.at(r - 1, c)) - (p0.at(r + 1, c) - p0.at(r - 1, c)));
  H[6] = H[2] = 0.25f * ((p2.at(r, c + 1) - p2.at(r, c - 1)) - (p0.at(r, c + 1) - p0.at(r, c - 1)));
  H[7] = H[5] = 0.25f * ((p1.at(r + 1, c + 1) - p1.at(r + 1, c - 1)) - (p1.at(r - 1, c + 1) - p1.at(r - 1, c - 1)));
  X[0] = -Y[0];
  X[1] = -Y[1];
  X[2] = -Y[2];
  solve3(X, H);
  return p1.at(r, c) + 0.5f * (X[0] * Y[0] + X[1] * Y[1] + X[2] * Y[2]);
}

struct Key {
  float row, col, scale, ori;
  float desc[128];
};

// PlaceInIndex (:1609-1668)
void place(float* fdesc, float mag, float ori, float rx, float cx) {
  const float oribin = ori * (8.0f / (2 * (float)kPi));
  const int nr = rx < 0 ? (int)(rx - 1) : (int)rx;
  const float rf = rx - (float)nr;
  const int nc = cx < 0 ? (int)(cx - 1) : (int)cx;
  const float cf = cx - (float)nc;
  const int no = oribin < 0 ? (int)(oribin - 1) : (int)oribin;
  const float of = oribin - (float)no;
  for (int i = 0; i < 2; ++i) {
    if ((unsigned)(i + nr) >= 4) continue;
    const float rg = i == 0 ? mag * (1 - rf) : mag * rf;
    for (int j = 0; j < 2; ++j) {
      if ((unsigned)(j + nc) >= 4) continue;
      const float cg = j == 0 ? rg * (1 - cf) : rg * cf;
      float* cell = fdesc + 8 * (4 * (i + nr) + j + nc);
      for (int k = 0; k < 2; ++k) cell[(no + k) & 7] += k == 0 ? cg * (1 - of) : cg * of;
    }
  }
}

// NormalizeVec (:1519-1527)
void normalize_vec(float* pf, int num) {
  float acc = 0;
  for (int i = 0; i < num; ++i) acc += pf[i] * pf[i];
  acc = 1 / sqrtf(acc);
  for (int i = 0; i < num; ++i) pf[i] *= acc;
}

// MakeKeypoint / MakeKeypointSample / KeySample / AddSample (:1424-1607), plain-C branch
void make_key(std::vector<Key>& out, const Img& grad, const Img& orim, float fscale, float fSize, float frow,
              float fcol, float forient) {
  Key k;
  k.ori = forient;
  k.row = fscale * frow;
  k.col = fscale * fcol;
  k.scale = fscale * fSize;
  memset(k.desc, 0, sizeof k.desc);
  const int rowstart = (int)(frow + 0.5f), colstart = (int)(fcol + 0.5f);
  const float sinang = sinf(forient), cosang = cosf(forient);
  const float fdrow = frow - (float)rowstart, fdcol = fcol - (float)colstart;
  const float frealsize = 3.0f * fSize, firealsize = 1.0f / (3.0f * fSize);
  const int win = (int)(frealsize * kSqrt2 * 5.0f * 0.5f + 0.5f);
  const float fsr = sinang * firealsize, fcr = cosang * firealsize, fdrr = -fdrow * firealsize,
              fdcr = -fdcol * firealsize;
  for (int row = -win; row <= win; ++row) {
    const float fr = (float)row;
    float fc = -(float)win;
    for (int col = -win; col <= win; ++col, fc += 1) {
      const float rpos = fsr * fc + fcr * fr + fdrr;
      const float cpos = fcr * fc - fsr * fr + fdcr;
      const float rx = rpos + (2.0f - 0.5f), cx = cpos + (2.0f - 0.5f);
      if (!(rx > -0.9999f && rx < 3.9999f && cx > -0.9999f && cx < 3.9999f)) continue;
      const int r = rowstart + row, c = colstart + col;
      if (r < 0 || r >= grad.rows || c < 0 || c >= grad.cols) continue;
      const float g = grad.at(r, c) * expf(-0.125f * (rpos * rpos + cpos * cpos));
      float o = orim.at(r, c) - forient;
      while (o > 2 * kPi) o -= 2 * kPi;
      while (o < 0) o += 2 * kPi;
      place(k.desc, g, o, rx, cx);
    }
  }
  normalize_vec(k.desc, 128);
  bool again = false;
  for (int i = 0; i < 128; ++i)
    if (k.desc[i] > 0.2f) {
      k.desc[i] = 0.2f;
      again = true;
    }
  if (again) normalize_vec(k.desc, 128);
  out.push_back(k);
}

// SmoothHistogram (:1395-1408) incl. its different constants for the last bin
void smooth_hist(float* h, int n) {
  const float first = h[0];
  float prev = h[n - 1];
  for (int i = 0; i < n - 1; ++i) {
    const float org = h[i];
    h[i] = (prev + org + h[i + 1]) * 0.33333333f;
    prev = org;
  }
  h[n - 1] = (prev + h[n - 1] + first) * 0.3333333f;
}

// AssignOriHist (:1274-1382)
void assign_ori(std::vector<Key>& out, const Img& grad, const Img& orim, float fscale, float fSize, float frow,
                float fcol) {
  const int rowstart = (int)(frow + 0.5f), colstart = (int)(fcol + 0.5f);
  const int rows = grad.rows, cols = grad.cols;
  float hist[36];
  memset(hist, 0, sizeof hist);
  const float fexpmult = -1.0f / (2.0f * 1.5f * 1.5f * fSize * fSize);
  const float fbinmult = 36.0f / (2 * kPi);
  const float fbinadd = (float)(kPi + 0.001f) * fbinmult;
  const int win = (int)(fSize * 1.5f * 3.0f);
  for (int r = rowstart - win; r <= rowstart + win; ++r) {
    if (r < 0 || r >= rows - 2) continue;
    for (int c = colstart - win; c <= colstart + win; ++c) {
      if (c < 0 || c >= cols - 2) continue;
      const float g = grad.at(r, c);
      if (!(g > 0)) continue;
      const float dr = (float)r - frow, dc = (float)c - fcol;
      const float rad2 = dr * dr + dc * dc;
      if (!((float)(win * win) + 0.5f > rad2)) continue;
      const float w = expf(rad2 * fexpmult);
      int bin = (int)(orim.at(r, c) * fbinmult + fbinadd);
      if (bin > 36) bin = 0;
      if (bin == 36) bin = 35;
      hist[bin] += g * w;
    }
  }
  for (int i = 0; i < 6; ++i) smooth_hist(hist, 36);
  float fmax = 0;
  for (int i = 0; i < 36; ++i)
    if (hist[i] > fmax) fmax = hist[i];
  fmax *= 0.8f;
  const float foriadd = 0.5f * 2 * kPi / 36.0f - kPi, forimult = 2 * kPi / 36.0f;
  for (int i = 0; i < 36; ++i) {
    const int prev = i == 0 ? 35 : i - 1, next = i == 35 ? 0 : i + 1;
    if (hist[i] <= hist[prev] || hist[i] <= hist[next] || hist[i] < fmax) continue;
    float f0 = hist[prev], f1 = hist[i], f2 = hist[next];   // InterpPeak (:1384-1393)
    if (f1 < 0) {
      f0 = -f0;
      f1 = -f1;
      f2 = -f2;
    }
    const float peak = 0.5f * (f0 - f2) / (f0 - 2.0f * f1 + f2);
    make_key(out, grad, orim, fscale, fSize, frow, fcol, (i + peak) * forimult + foriadd);
  }
}

// InterpKeyPoint (:1164-1206): the recursion as a loop
void interp_key(std::vector<Key>& out, const Img* dog, int index, int r, int c, const Img& grad, const Img& orim,
                std::vector<char>& taken, float fscale, float peak_thresh) {
  const int rows = dog[0].rows, cols = dog[0].cols;
  float X[3], val = 0;
  for (int steps = 5;; --steps) {
    val = fit_quadratic(X, dog, index, r, c);
    int nr = r, nc = c;
    if (X[1] > 0.6f && r < rows - 3) nr++;
    if (X[1] < -0.6f && r > 3) nr--;
    if (X[2] > 0.6f && c < cols - 3) nc++;
    if (X[2] < -0.6f && c > 3) nc--;
    if (steps > 0 && (nr != r || nc != c)) {
      r = nr;
      c = nc;
      continue;
    }
    break;
  }
  if (fabsf(X[0]) <= 1.5f && fabsf(X[1]) <= 1.5f && fabsf(X[2]) <= 1.5f && fabsf(val) >= peak_thresh) {
    char& t = taken[(size_t)r * cols + c];
    if (!t) {
      t = 1;
      const float fSize = kInitSigma * powf(2.0f, ((float)index + X[0]) / (float)kScales);
      assign_ori(out, grad, orim, fscale, fSize, (float)r + X[1], (float)c + X[2]);
    }
  }
}

struct Pyramid {
  std::vector<std::vector<Img>> gaus, dog;  // [octave][i]
};

// GetKeypoints / OctaveKeypoints / FindMaxMin (:301-361, :410-438, :891-957).  Keypoints come
// out in GENERATION order (octave, scale index, row, column, histogram peak ascending); the
// reference's linked list is that order reversed (every new key is pushed on the front).
void run_sift(const uint8_t* gray, int w, int h, int double_size, std::vector<Key>& keys, Pyramid* pyr) {
  const float peak_thresh = 0.04f / (float)kScales;
  Img org(h, w);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) org.at(y, x) = ((float)gray[(size_t)w * y + x]) * 1. / 255.;  // FEAT_SIFT_CPU.hpp:91
  Img cur;
  float fscale = 1.0f;
  if (double_size) {  // SiftDoubleSize (:363-380)
    cur = Img(2 * h - 2, 2 * w - 2);
    for (int i = 0; i < h - 1; ++i)
      for (int j = 0; j < w - 1; ++j) {
        cur.at(2 * i, 2 * j) = org.at(i, j);
        cur.at(2 * i + 1, 2 * j) = 0.5f * (org.at(i, j) + org.at(i + 1, j));
        cur.at(2 * i, 2 * j + 1) = 0.5f * (org.at(i, j) + org.at(i, j + 1));
        cur.at(2 * i + 1, 2 * j + 1) = 0.25f * (org.at(i, j) + org.at(i, j + 1) + org.at(i + 1, j) + org.at(i + 1, j + 1));
      }
    fscale = 0.5f;
  } else {
    cur = org;
  }
  const float fnew = double_size ? 1.0f : 0.5f;
  if (kInitSigma > fnew) {
    Img tmp;
    blur(tmp, cur, sqrtf(kInitSigma * kInitSigma - fnew * fnew));
    cur = tmp;
  }
  const float fwidth = powf(2.0f, 1.0f / (float)kScales);
  const float fincsigma = sqrtf(fwidth * fwidth - 1.0f);
  while (cur.rows > 12 && cur.cols > 12) {
    std::vector<Img> gaus(kScales + 3), dog(kScales + 2);
    gaus[0] = cur;
    float sigma = kInitSigma;
    for (int i = 1; i < kScales + 3; ++i) {
      blur(gaus[i], gaus[i - 1], fincsigma * sigma);
      dog[i - 1] = Img(cur.rows, cur.cols);
      for (size_t p = 0; p < dog[i - 1].px.size(); ++p) dog[i - 1].px[p] = gaus[i - 1].px[p] - gaus[i].px[p];
      sigma *= fwidth;
    }
    const int rows = cur.rows, cols = cur.cols;
    std::vector<char> taken((size_t)rows * cols, 0);
    for (int index = 1; index < kScales + 1; ++index) {
      Img grad, orim;
      grad_ori(gaus[index], grad, orim);
      for (int r = 5; r < rows - 5; ++r)
        for (int c = 5; c < cols - 5; ++c) {
          const float v = dog[index].at(r, c);
          if (fabsf(v) > peak_thresh * 0.8f && local_extremum(v, dog[index], r, c) &&
              local_extremum(v, dog[index - 1], r, c) && local_extremum(v, dog[index + 1], r, c) &&
              not_on_edge(dog[index], r, c))
            interp_key(keys, dog.data(), index, r, c, grad, orim, taken, fscale, peak_thresh);
        }
    }
    // HalfImageSize of gaus[Scales] (:390-408)
    Img half(rows >> 1, cols >> 1);
    for (int r = 0; r < half.rows; ++r)
      for (int c = 0; c < half.cols; ++c) half.at(r, c) = gaus[kScales].at(2 * r, 2 * c);
    if (pyr) {
      pyr->gaus.push_back(gaus);
      pyr->dog.push_back(dog);
    }
    cur = half;
    fscale += fscale;
  }
}

}  // namespace

extern "C" {

int orc_sift(const uint8_t* gray, int w, int h, int double_size, float* xy, float* scale_ori, float* desc, int cap) {
  std::vector<Key> keys;
  run_sift(gray, w, h, double_size, keys, nullptr);
  const int n = (int)keys.size();
  // list order of the reference = generation order reversed
  for (int i = 0; i < n && i < cap; ++i) {
    const Key& k = keys[n - 1 - i];
    xy[2 * i] = k.col;          // coord2D = (col, row), FEAT_SIFT_CPU.hpp:103-104
    xy[2 * i + 1] = k.row;
    if (scale_ori) {
      scale_ori[2 * i] = k.scale;
      scale_ori[2 * i + 1] = k.ori;
    }
    memcpy(desc + (size_t)i * 128, k.desc, sizeof k.desc);
  }
  return n;
}

// Pyramid images for stage-by-stage checks of the HIP path: kind 0 = Gaussian i (0..5),
// 1 = DoG i (0..4) of `octave`.  Returns rows*cols (0 if the octave does not exist) and the size.
int orc_sift_image(const uint8_t* gray, int w, int h, int double_size, int octave, int kind, int i, float* out,
                   int* rows, int* cols) {
  std::vector<Key> keys;
  Pyramid p;
  run_sift(gray, w, h, double_size, keys, &p);
  if (octave < 0 || octave >= (int)p.gaus.size()) return 0;
  const std::vector<Img>& v = kind == 0 ? p.gaus[octave] : p.dog[octave];
  if (i < 0 || i >= (int)v.size()) return 0;
  *rows = v[i].rows;
  *cols = v[i].cols;
  if (out) memcpy(out, v[i].px.data(), v[i].px.size() * sizeof(float));
  return (int)v[i].px.size();
}

}  // extern "C"
