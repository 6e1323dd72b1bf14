// oracle/oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// CPU restatement of libmoped's MATCH -> CLUSTER -> POSE (-> FILTER) path.
// Written from the reference's source text; no reference code is included or
// linked here.  Compile with -ffp-contract=off: fused multiply-adds appear only
// where fmaf() is written, so the arithmetic is the same on every host.
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <list>
#include <map>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------
// small geometry helpers restating include/moped.hpp
// ---------------------------------------------------------------------------

struct Mat34 {
  float m[3][4];
};

// TransformMatrix::init (moped.hpp:175-182): rotation from a *normalised*
// quaternion (x,y,z,w) plus translation.
void tm_init(Mat34& T, const float* q, const float* t) {
  T.m[0][0] = 1 - 2 * q[1] * q[1] - 2 * q[2] * q[2];
  T.m[0][1] = 2 * q[0] * q[1] - 2 * q[3] * q[2];
  T.m[0][2] = 2 * q[0] * q[2] + 2 * q[3] * q[1];
  T.m[0][3] = t[0];
  T.m[1][0] = 2 * q[0] * q[1] + 2 * q[3] * q[2];
  T.m[1][1] = 1 - 2 * q[0] * q[0] - 2 * q[2] * q[2];
  T.m[1][2] = 2 * q[1] * q[2] - 2 * q[3] * q[0];
  T.m[1][3] = t[1];
  T.m[2][0] = 2 * q[0] * q[2] - 2 * q[3] * q[1];
  T.m[2][1] = 2 * q[1] * q[2] + 2 * q[3] * q[0];
  T.m[2][2] = 1 - 2 * q[0] * q[0] - 2 * q[1] * q[1];
  T.m[2][3] = t[2];
}

// TransformMatrix::transform (moped.hpp:183-188)
void tm_apply(const Mat34& T, const float* x, float* y) {
  float a = x[0] * T.m[0][0] + x[1] * T.m[0][1] + x[2] * T.m[0][2] + T.m[0][3];
  float b = x[0] * T.m[1][0] + x[1] * T.m[1][1] + x[2] * T.m[1][2] + T.m[1][3];
  float c = x[0] * T.m[2][0] + x[1] * T.m[2][1] + x[2] * T.m[2][2] + T.m[2][3];
  y[0] = a;
  y[1] = b;
  y[2] = c;
}

// TransformMatrix::inverseTransform (moped.hpp:190-200)
void tm_apply_inv(const Mat34& T, const float* x, float* y) {
  float d0 = x[0] - T.m[0][3], d1 = x[1] - T.m[1][3], d2 = x[2] - T.m[2][3];
  float a = d0 * T.m[0][0] + d1 * T.m[1][0] + d2 * T.m[2][0];
  float b = d0 * T.m[0][1] + d1 * T.m[1][1] + d2 * T.m[2][1];
  float c = d0 * T.m[0][2] + d1 * T.m[1][2] + d2 * T.m[2][2];
  y[0] = a;
  y[1] = b;
  y[2] = c;
}

// Pt<N>::norm (moped.hpp:124): sum in float, sqrt resolves to the float overload,
// the quotient 1./sqrt is formed in double and rounded to float, then each
// component is scaled in float.
void quat_normalize(float* q) {
  float s = 0;
  for (int i = 0; i < 4; i++) s += q[i] * q[i];
  float d = (float)(1. / (double)sqrtf(s));
  for (int i = 0; i < 4; i++) q[i] *= d;
}

struct Camera {
  float K[4];
  Mat34 TM;
};

void camera_init(Camera& c, const float K[4], const float cam[7]) {
  memcpy(c.K, K, sizeof c.K);
  tm_init(c.TM, cam, cam + 4);  // image->TM.init(cameraPose), src/moped.cpp:168-169
}

// project() (moped.hpp:330-354), alternatePose == NULL branch.
void project_one(const Mat34& poseTM, const Camera& c, const float* X, float* uv) {
  float p[3];
  tm_apply(poseTM, X, p);
  tm_apply_inv(c.TM, p, p);
  uv[0] = FLT_MAX;
  uv[1] = FLT_MAX;
  if (p[2] < 0.001) return;
  uv[0] = p[0] / p[2] * c.K[0] + c.K[2];
  uv[1] = p[1] / p[2] * c.K[1] + c.K[3];
}

// lmFuncQuat (…REPROJECTION_CPU.hpp:100-138)
void residuals(const float* p7, const float* uv, const float* xyz, int n, const Camera& c,
               float* hx) {
  float q[4] = {p7[0], p7[1], p7[2], p7[3]};
  quat_normalize(q);
  Mat34 T;
  tm_init(T, q, p7 + 4);
  for (int i = 0; i < n; i++) {
    float p[3];
    tm_apply(T, xyz + 3 * i, p);
    tm_apply_inv(c.TM, p, p);
    float u = p[0] / p[2] * c.K[0] + c.K[2];
    float v = p[1] / p[2] * c.K[1] + c.K[3];
    if (p[2] < 0) {
      hx[2 * i] = -p[2] + 10;
      hx[2 * i + 1] = -p[2] + 10;
    } else {
      float du = u - uv[2 * i];
      float dv = v - uv[2 * i + 1];
      hx[2 * i] = du * du;
      hx[2 * i + 1] = dv * dv;
    }
  }
}

// ---------------------------------------------------------------------------
// Levenberg-Marquardt, forward differences + Broyden updates (levmar 2.4
// slevmar_dif as published: lm_core.c:427-825, misc_core.c:135-170)
// ---------------------------------------------------------------------------

// m x m dense solve by Gaussian elimination with implicitly scaled partial
// pivoting (the method of Axb_core.c:888-1031).  Returns false if singular.
bool solve_dense(const float* A_in, const float* b_in, float* x, int m) {
  std::vector<float> A(A_in, A_in + m * m), b(b_in, b_in + m), scale(m);
  std::vector<int> perm(m);
  for (int i = 0; i < m; i++) {
    float mx = 0;
    for (int j = 0; j < m; j++) mx = std::max(mx, std::fabs(A[i * m + j]));
    if (mx == 0) return false;
    scale[i] = 1.f / mx;
    perm[i] = i;
  }
  for (int k = 0; k < m; k++) {
    int piv = k;
    float best = -1;
    for (int i = k; i < m; i++) {
      float v = std::fabs(A[i * m + k]) * scale[i];
      if (v > best) {
        best = v;
        piv = i;
      }
    }
    if (A[piv * m + k] == 0) return false;
    if (piv != k) {
      for (int j = 0; j < m; j++) std::swap(A[piv * m + j], A[k * m + j]);
      std::swap(b[piv], b[k]);
      std::swap(scale[piv], scale[k]);
    }
    for (int i = k + 1; i < m; i++) {
      float f = A[i * m + k] / A[k * m + k];
      if (f == 0) continue;
      for (int j = k + 1; j < m; j++) A[i * m + j] -= f * A[k * m + j];
      b[i] -= f * b[k];
    }
  }
  for (int i = m - 1; i >= 0; i--) {
    float s = b[i];
    for (int j = i + 1; j < m; j++) s -= A[i * m + j] * x[j];
    x[i] = s / A[i * m + i];
  }
  return true;
}

// moped3d residuals (A14).  world = camera-frame point from the depth map,
// w = cauchyWeight; alpha = the 2-D/3-D trade-off.
// mode 1: POSE_RANSAC_LM_DIFF_BACKPROJECTION_DEPTH_CPU::lmFuncQuat
//         (moped3d/libmoped/src/pose/...BACKPROJECTION_DEPTH_CPU.hpp:108-190), 2 per point
// mode 2: POSE_RANSAC_LM_DIFF_REPROJECTION_DEPTH_CPU::lmFuncQuat
//         (moped3d/.../...REPROJECTION_DEPTH_CPU.hpp:106-216), 3 per point
void residuals_depth(int mode, const float* p7, const float* uv, const float* xyz, const float* world,
                     const float* wgt, int n, const Camera& c, float alpha, float* err) {
  float q[4] = {p7[0], p7[1], p7[2], p7[3]};
  quat_normalize(q);
  Mat34 T;
  tm_init(T, q, p7 + 4);
  for (int i = 0; i < n; i++) {
    float p[3];
    tm_apply(T, xyz + 3 * i, p);
    tm_apply_inv(c.TM, p, p);
    const float* W = world + 3 * i;
    const float w3 = (1 - alpha) * wgt[i];
    if (mode == 1) {
      float* e = err + 2 * i;
      if (p[2] < 0) {
        e[0] = -p[2] + 10;
        e[1] = -p[2] + 10;
      } else {
        float norm = std::sqrt(W[0] * W[0] + W[1] * W[1] + W[2] * W[2]);
        float nx = W[0] / norm, ny = W[1] / norm, nz = W[2] / norm;
        float dot = nx * p[0] + ny * p[1] + nz * p[2];
        float hx = nx * dot, hy = ny * dot, hz = nz * dot;   // projection of p on the ray
        // Pt::euclDist = sqrt(sqEuclDist), differences taken as (other - this)
        float d0 = hx - p[0], d1 = hy - p[1], d2 = hz - p[2];
        float dxy = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        float g0 = hx - W[0], g1 = hy - W[1], g2 = hz - W[2];
        float dz = std::sqrt(g0 * g0 + g1 * g1 + g2 * g2);
        e[0] = dxy * dxy;
        e[1] = dz * dz;
      }
      const float w2 = 1 - w3;
      e[0] *= w2;
      e[1] *= w3;
    } else {
      float* e = err + 3 * i;
      float u = p[0] / p[2] * c.K[0] + c.K[2];
      float v = p[1] / p[2] * c.K[1] + c.K[3];
      if (p[2] < 0) {
        e[0] = -p[2] + 10;
        e[1] = -p[2] + 10;
        e[2] = -p[2] + 10;
      } else {
        float dx = u - uv[2 * i], dy = v - uv[2 * i + 1];
        e[0] = dx * dx;
        e[1] = dy * dy;
      }
      float vtp = p[0] * W[0] + p[1] * W[1] + p[2] * W[2];
      float a0 = p[0] - p[0] * vtp, a1 = p[1] - p[1] * vtp, a2 = p[2] - p[2] * vtp;  // (p3D - projWorld)
      float de = std::sqrt(a0 * a0 + a1 * a1 + a2 * a2);
      e[2] = de * de;
      e[2] *= 50;
      e[0] *= (1 - w3);
      e[1] *= (1 - w3);
      e[2] *= w3;
    }
  }
}

// lmFuncQuat with every correspondence in its own image (LmData::image, …REPROJECTION_CPU.hpp:100-138,
// 213-237): cams[img[i]] is the camera of point i.  Same arithmetic as residuals() per point.
void residuals_images(const float* p7, const float* uv, const float* xyz, const int32_t* img, int n,
                      const Camera* cams, float* hx) {
  float q[4] = {p7[0], p7[1], p7[2], p7[3]};
  quat_normalize(q);
  Mat34 T;
  tm_init(T, q, p7 + 4);
  for (int i = 0; i < n; i++) {
    const Camera& c = cams[img[i]];
    float p[3];
    tm_apply(T, xyz + 3 * i, p);
    tm_apply_inv(c.TM, p, p);
    float u = p[0] / p[2] * c.K[0] + c.K[2];
    float v = p[1] / p[2] * c.K[1] + c.K[3];
    if (p[2] < 0) {
      hx[2 * i] = -p[2] + 10;
      hx[2 * i + 1] = -p[2] + 10;
    } else {
      float du = u - uv[2 * i];
      float dv = v - uv[2 * i + 1];
      hx[2 * i] = du * du;
      hx[2 * i + 1] = dv * dv;
    }
  }
}

struct LmProblem {
  const float* uv;
  const float* xyz;
  int npts;
  const Camera* cam;
  // depth variants (mode 0 = the moped2 residual)
  int mode;
  const float* world;
  const float* wgt;
  float alpha;
  // mode 0 with several images: img[i] selects the camera of point i out of cam[0..] (nullptr = one camera)
  const int32_t* img = nullptr;
  int per_item() const { return mode == 0 ? 2 : (mode == 1 ? 2 : 3); }
  void eval(const float* p, float* hx) const {
    if (mode == 0 && img)
      residuals_images(p, uv, xyz, img, npts, cam, hx);
    else if (mode == 0)
      residuals(p, uv, xyz, npts, *cam, hx);
    else
      residuals_depth(mode, p, uv, xyz, world, wgt, npts, *cam, alpha, hx);
  }
};

// Returns iterations or -1.  Target vector is zero, so e = -hx.
int lm_dif(const LmProblem& f, float* p, int itmax, float* info) {
  const int m = 7, n = f.per_item() * f.npts;
  if (n < m) return -1;
  const float tau = 1e-3f, eps1 = 1e-17f, eps2 = 1e-17f, eps3 = 1e-17f, delta = 1e-6f;
  const float eps2_sq = eps2 * eps2;
  const float EPSILON = 1e-12f;  // levmar's "almost singular" guard constant
  const int K = 10;              // Broyden updates before the Jacobian is recomputed
  std::vector<float> e(n), hx(n), hxx(n), e2(n), J((size_t)n * m), JtJ(m * m), Jte(m), diag(m),
      Dp(m), pDp(m);
  f.eval(p, &hx[0]);
  float p_eL2 = 0;
  for (int i = 0; i < n; i++) {
    e[i] = -hx[i];
    p_eL2 += e[i] * e[i];
  }
  const float init_eL2 = p_eL2;
  int stop = 0, k = 0, nu = 20, updjac = 0;
  bool updp = true, newjac = false;
  float mu = 0, jacTe_inf = 0, p_L2 = 0, Dp_L2 = FLT_MAX;
  if (!std::isfinite(p_eL2)) stop = 7;
  for (k = 0; k < itmax && !stop; ++k) {
    if (p_eL2 <= eps3) {
      stop = 6;
      break;
    }
    if ((updp && nu > 16) || updjac == K) {  // fresh forward-difference Jacobian
      for (int j = 0; j < m; j++) {
        float d = std::fabs(1e-4f * p[j]);
        if (d < delta) d = delta;
        float keep = p[j];
        p[j] += d;
        f.eval(p, &hxx[0]);
        p[j] = keep;
        d = 1.f / d;
        for (int i = 0; i < n; i++) J[(size_t)i * m + j] = (hxx[i] - hx[i]) * d;
      }
      nu = 2;
      updjac = 0;
      updp = false;
      newjac = true;
    }
    if (newjac) {
      newjac = false;
      std::fill(JtJ.begin(), JtJ.end(), 0.f);
      std::fill(Jte.begin(), Jte.end(), 0.f);
      for (int l = 0; l < n; l++) {
        const float* r = &J[(size_t)l * m];
        for (int i = 0; i < m; i++) {
          for (int j = 0; j <= i; j++) JtJ[i * m + j] += r[j] * r[i];
          Jte[i] += r[i] * e[l];
        }
      }
      for (int i = 0; i < m; i++)
        for (int j = i + 1; j < m; j++) JtJ[i * m + j] = JtJ[j * m + i];
      p_L2 = jacTe_inf = 0;
      for (int i = 0; i < m; i++) {
        jacTe_inf = std::max(jacTe_inf, std::fabs(Jte[i]));
        diag[i] = JtJ[i * m + i];
        p_L2 += p[i] * p[i];
      }
    }
    if (jacTe_inf <= eps1) {
      Dp_L2 = 0;
      stop = 1;
      break;
    }
    if (k == 0) {
      float mx = -FLT_MAX;
      for (int i = 0; i < m; i++) mx = std::max(mx, diag[i]);
      mu = tau * mx;
    }
    for (int i = 0; i < m; i++) JtJ[i * m + i] += mu;
    if (solve_dense(&JtJ[0], &Jte[0], &Dp[0], m)) {
      Dp_L2 = 0;
      for (int i = 0; i < m; i++) {
        pDp[i] = p[i] + Dp[i];
        Dp_L2 += Dp[i] * Dp[i];
      }
      if (Dp_L2 <= eps2_sq * p_L2) {
        stop = 2;
        break;
      }
      if (Dp_L2 >= (p_L2 + eps2) / (EPSILON * EPSILON)) {
        stop = 4;
        break;
      }
      f.eval(&pDp[0], &hxx[0]);
      float pDp_eL2 = 0;
      for (int i = 0; i < n; i++) {
        e2[i] = -hxx[i];
        pDp_eL2 += e2[i] * e2[i];
      }
      if (!std::isfinite(pDp_eL2)) {
        stop = 7;
        break;
      }
      float dF = p_eL2 - pDp_eL2;
      if (updp || dF > 0) {  // Broyden rank-one update of J
        for (int i = 0; i < n; i++) {
          float* r = &J[(size_t)i * m];
          float t = 0;
          for (int l = 0; l < m; l++) t += r[l] * Dp[l];
          t = (hxx[i] - hx[i] - t) / Dp_L2;
          for (int j = 0; j < m; j++) r[j] += t * Dp[j];
        }
        ++updjac;
        newjac = true;
      }
      float dL = 0;
      for (int i = 0; i < m; i++) dL += Dp[i] * (mu * Dp[i] + Jte[i]);
      if (dL > 0 && dF > 0) {
        float t = 2.f * dF / dL - 1.f;
        t = 1.f - t * t * t;
        mu *= (t >= (1.f / 3.f)) ? t : (1.f / 3.f);
        nu = 2;
        for (int i = 0; i < m; i++) p[i] = pDp[i];
        for (int i = 0; i < n; i++) {
          e[i] = e2[i];
          hx[i] = hxx[i];
        }
        p_eL2 = pDp_eL2;
        updp = true;
        continue;
      }
    }
    mu *= nu;
    int nu2 = nu << 1;
    if (nu2 <= nu) {
      stop = 5;
      break;
    }
    nu = nu2;
    for (int i = 0; i < m; i++) JtJ[i * m + i] = diag[i];
  }
  if (k >= itmax) stop = 3;
  if (info) {
    info[0] = init_eL2;
    info[1] = p_eL2;
    info[2] = (float)stop;
  }
  return (stop != 4 && stop != 7) ? k : -1;
}

// optimizeCamera (…REPROJECTION_CPU.hpp:140-164): returns final ||e||^2, or the
// (negative) LM return value on error, in which case pose is left untouched.
float optimize_camera(float* pose7, const float* uv, const float* xyz, int n, const Camera& cam,
                      int itmax, int* iters) {
  float p[7];
  memcpy(p, pose7, sizeof p);
  LmProblem f = {uv, xyz, n, &cam, 0, NULL, NULL, 0.f};
  float info[3];
  int ret = lm_dif(f, p, itmax, info);
  if (iters) *iters = ret;
  if (ret < 0) return (float)ret;
  quat_normalize(p);
  memcpy(pose7, p, sizeof p);
  return info[1];
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================

extern "C" {

void orc_normalize(float* desc, int n, int dim) {
  for (int i = 0; i < n; i++) {
    float* d = desc + (size_t)i * dim;
    float s = 0;
    for (int x = 0; x < dim; x++) s += d[x] * d[x];
    s = (float)(1. / sqrtf(s));
    for (int x = 0; x < dim; x++) d[x] *= s;
  }
}

static inline float dot_chain(const float* a, const float* b, int dim) {
  float s = 0.f;
  for (int k = 0; k < dim; k++) s = fmaf(a[k], b[k], s);
  return s;
}

void orc_row_norms(const float* desc, int n, int dim, float* out) {
  for (int i = 0; i < n; i++) out[i] = dot_chain(desc + (size_t)i * dim, desc + (size_t)i * dim, dim);
}

void orc_match_2nn(const float* db, int N, const float* q, int Q, int dim, int32_t* idx1,
                   float* d1, float* d2, int n_threads) {
  std::vector<float> nd(N);
#ifdef _OPENMP
  if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
  n_threads = 1;
#endif
#pragma omp parallel for num_threads(n_threads) schedule(static)
  for (int j = 0; j < N; j++) nd[j] = dot_chain(db + (size_t)j * dim, db + (size_t)j * dim, dim);
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 8)
  for (int i = 0; i < Q; i++) {
    const float* qi = q + (size_t)i * dim;
    const float nq = dot_chain(qi, qi, dim);
    float b1 = std::numeric_limits<float>::infinity(), b2 = b1;
    int32_t i1 = -1;
    for (int j = 0; j < N; j++) {
      float dot = dot_chain(qi, db + (size_t)j * dim, dim);
      float dist = fmaxf(0.f, fmaf(-2.f, dot, nq + nd[j]));
      if (dist < b1) {
        b2 = b1;
        b1 = dist;
        i1 = j;
      } else if (dist < b2) {
        b2 = dist;
      }
    }
    idx1[i] = i1;
    d1[i] = b1;
    d2[i] = b2;
  }
}

int orc_match_accept(const int32_t* idx1, const float* d1, const float* d2, int Q, float ratio,
                     const int32_t* model_of, int n_models, int32_t* out_q, int32_t* model_off) {
  std::vector<std::vector<int32_t> > per_model(n_models);
  for (int i = 0; i < Q; i++) {
    if (idx1[i] < 0) continue;
    if (d1[i] / d2[i] < ratio) per_model[model_of[idx1[i]]].push_back(i);
  }
  int m = 0;
  for (int k = 0; k < n_models; k++) {
    model_off[k] = m;
    for (size_t j = 0; j < per_model[k].size(); j++) out_q[m++] = per_model[k][j];
  }
  model_off[n_models] = m;
  return m;
}

void orc_match_merge(const int32_t* idx1_s, const float* d1_s, const float* d2_s, int n_shards,
                     int Q, int32_t* idx1, float* d1, float* d2) {
  const float inf = std::numeric_limits<float>::infinity();
  for (int i = 0; i < Q; i++) {
    float b1 = inf, b2 = inf;
    int32_t i1 = -1;
    // candidates: every shard's (d1, idx) and every shard's d2 (index irrelevant)
    for (int s = 0; s < n_shards; s++) {
      float c = d1_s[(size_t)s * Q + i];
      int32_t ci = idx1_s[(size_t)s * Q + i];
      if (ci < 0) continue;
      if (c < b1 || (c == b1 && ci < i1)) {
        b2 = b1;
        b1 = c;
        i1 = ci;
      } else if (c < b2) {
        b2 = c;
      }
    }
    for (int s = 0; s < n_shards; s++) {
      if (idx1_s[(size_t)s * Q + i] < 0) continue;
      float c = d2_s[(size_t)s * Q + i];
      if (c < b2) b2 = c;
    }
    idx1[i] = i1;
    d1[i] = b1;
    d2[i] = b2;
  }
}

// ---------------------------------------------------------------------------
// mean shift (CLUSTER_MEAN_SHIFT_CPU.hpp:80-158) -- PARITY UNPINNED
// ---------------------------------------------------------------------------
int orc_meanshift(const float* pts, int n, int dim, float radius, float merge, int min_pts,
                  int max_iter, int32_t* members, int32_t* cluster_off, int* n_iter) {
  struct Canopy {
    float center[3];
    float agg[3];
    std::list<int> bound;
    int size;
    int merges;  // index of the canopy this one merges into (self = none)
  };
  const float sq_radius = radius * radius, sq_merge = merge * merge;
  std::vector<Canopy> can(n);
  std::list<int> remaining;
  for (int i = 0; i < n; i++) {
    for (int x = 0; x < dim; x++) can[i].center[x] = pts[(size_t)i * dim + x];
    can[i].bound.push_back(i);
    can[i].size = 1;
    can[i].merges = i;
    remaining.push_back(i);
  }
  bool done = false;
  int it = 0;
  for (; !done && it < max_iter; it++) {
    done = true;
    // (1) flat-kernel weighted mean of the canopies within Radius (:102-120)
    for (std::list<int>::iterator c = remaining.begin(); c != remaining.end(); ++c) {
      Canopy& C = can[*c];
      for (int x = 0; x < dim; x++) C.agg[x] = C.center[x] * C.size;
      int touch = C.size;
      for (std::list<int>::iterator o = remaining.begin(); o != remaining.end(); ++o) {
        if (*o == *c) continue;
        const Canopy& O = can[*o];
        float dist = 0;
        for (int x = 0; x < dim; x++) {
          float d = O.center[x] - C.center[x];
          dist += d * d;
        }
        if (dist < sq_radius) {
          touch += O.size;
          for (int x = 0; x < dim; x++) C.agg[x] += O.center[x] * O.size;
        }
      }
      for (int x = 0; x < dim; x++) C.agg[x] /= touch;
    }
    // (2) order-dependent merge marking (:122-132)
    for (std::list<int>::iterator c = remaining.begin(); c != remaining.end(); ++c) {
      for (std::list<int>::iterator o = remaining.begin(); o != remaining.end(); ++o) {
        if (*o == *c) break;
        float dist = 0;
        for (int x = 0; x < dim; x++) {
          float d = can[*o].agg[x] - can[*c].agg[x];
          dist += d * d;
        }
        if (dist < sq_merge) {
          can[can[*o].merges].merges = *c;
          can[*o].merges = *c;
        }
      }
    }
    // (3) fold marked canopies into their targets, in list order (:134-148)
    for (std::list<int>::iterator c = remaining.begin(); c != remaining.end();) {
      Canopy& C = can[*c];
      if (C.merges != *c) {
        Canopy& T = can[C.merges];
        for (int x = 0; x < dim; x++)
          T.center[x] = T.center[x] * (float)T.bound.size() + C.center[x] * C.size;
        T.bound.splice(T.bound.end(), C.bound);
        T.size += C.size;
        for (int x = 0; x < dim; x++) T.center[x] /= T.size;
        c = remaining.erase(c);
        done = false;
      } else {
        ++c;
      }
    }
  }
  if (n_iter) *n_iter = it;
  int ncl = 0, w = 0;
  for (std::list<int>::iterator c = remaining.begin(); c != remaining.end(); ++c) {
    if (can[*c].size < min_pts) continue;
    cluster_off[ncl++] = w;
    for (std::list<int>::iterator b = can[*c].bound.begin(); b != can[*c].bound.end(); ++b)
      members[w++] = *b;
  }
  cluster_off[ncl] = w;
  return ncl;
}

// ---------------------------------------------------------------------------
// pose
// ---------------------------------------------------------------------------
void orc_project(const float pose7[7], const float* xyz, int n, const float K[4],
                 const float cam[7], float* uv) {
  Camera c;
  camera_init(c, K, cam);
  Mat34 T;
  tm_init(T, pose7, pose7 + 4);
  for (int i = 0; i < n; i++) project_one(T, c, xyz + 3 * i, uv + 2 * i);
}

void orc_residuals(const float pose7[7], const float* uv, const float* xyz, int n,
                   const float K[4], const float cam[7], float* hx) {
  Camera c;
  camera_init(c, K, cam);
  residuals(pose7, uv, xyz, n, c, hx);
}

static int test_all_points(const float* pose7, const float* uv, const float* xyz, int n,
                           const Camera& c, float thr, uint8_t* inlier) {
  Mat34 T;
  tm_init(T, pose7, pose7 + 4);
  int cnt = 0;
  for (int i = 0; i < n; i++) {
    float p[2];
    project_one(T, c, xyz + 3 * i, p);
    p[0] -= uv[2 * i];
    p[1] -= uv[2 * i + 1];
    float err = p[0] * p[0] + p[1] * p[1];
    bool in = err < thr;
    if (inlier) inlier[i] = in;
    cnt += in;
  }
  return cnt;
}

int orc_test_all_points(const float pose7[7], const float* uv, const float* xyz, int n,
                        const float K[4], const float cam[7], float thr, uint8_t* inlier) {
  Camera c;
  camera_init(c, K, cam);
  return test_all_points(pose7, uv, xyz, n, c, thr, inlier);
}

int orc_optimize_camera(float pose7[7], const float* uv, const float* xyz, int n,
                        const float K[4], const float cam[7], int itmax, float* info) {
  Camera c;
  camera_init(c, K, cam);
  float p[7];
  memcpy(p, pose7, sizeof p);
  LmProblem f = {uv, xyz, n, &c, 0, NULL, NULL, 0.f};
  float linfo[3] = {0, 0, 0};
  int ret = lm_dif(f, p, itmax, linfo);
  if (info) memcpy(info, linfo, sizeof linfo);
  if (ret < 0) return ret;
  quat_normalize(p);
  memcpy(pose7, p, sizeof p);
  return ret;
}

// libc rand() like the reference (:83, :184), or a stream a test injects (orc_set_rand: tests/test_witness_cpu.py drives the
// RANSAC skeleton with streams that force ties between the points' random keys).
static int (*g_rand_fn)(void) = NULL;
static inline int orc_rand_() { return g_rand_fn ? g_rand_fn() : rand(); }
void orc_set_rand(int (*fn)(void)) { g_rand_fn = fn; }

// addr (optional, NULL = position order): the points' rank by ADDRESS.  randSample sorts pair<Float, LmData*> (:81-84), so
// two points whose (Float)rand() keys are equal come out in pointer order -- &lmData[model][match] (:287-288): ascending
// match index, not the position inside the cluster.  Keys collide in ~1 sample in 1 500 at 150 points (a 31-bit rand() in
// a 24-bit mantissa).
int orc_ransac_addr(const float* uv, const float* xyz, const int32_t* addr, int k, const float K[4], const float cam[7],
                    const orc_pose_params* prm, float pose7[7]) {
  Camera c;
  camera_init(c, K, cam);
  std::vector<float> suv, sxyz;
  std::vector<uint8_t> inl(k);
  for (int it = 0; it < prm->max_ransac_tests; it++) {
    // randSample (:76-98): a random float key per point, sort, take from the
    // front skipping points whose 2-D coordinate was already taken.
    std::vector<std::pair<float, std::pair<int, int> > > keyed(k);
    for (int i = 0; i < k; i++) keyed[i] = std::make_pair((float)orc_rand_(), std::make_pair(addr ? addr[i] : i, i));
    std::sort(keyed.begin(), keyed.end());
    std::map<std::pair<float, float>, int> used;
    suv.clear();
    sxyz.clear();
    size_t pos = 0;
    while ((int)used.size() < prm->n_pts_align && pos < keyed.size()) {
      int i = keyed[pos++].second.second;
      std::pair<float, float> key(uv[2 * i], uv[2 * i + 1]);
      if (!used[key]++) {
        suv.push_back(uv[2 * i]);
        suv.push_back(uv[2 * i + 1]);
        sxyz.push_back(xyz[3 * i]);
        sxyz.push_back(xyz[3 * i + 1]);
        sxyz.push_back(xyz[3 * i + 2]);
      }
    }
    if ((int)used.size() != prm->n_pts_align) return 0;
    // initPose (:182-186)
    // (the reference draws the four inside ONE call's argument list, :184: their order is the compiler's; here x, y, z, w)
    for (int j = 0; j < 4; j++) pose7[j] = (float)((orc_rand_() & 255) / 256.);
    pose7[4] = 0.f;
    pose7[5] = 0.f;
    pose7[6] = 0.5f;
    int lm = (int)optimize_camera(pose7, &suv[0], &sxyz[0], (int)suv.size() / 2, c,
                                  prm->max_lm_tests, NULL);
    if (lm == -1) continue;
    int cnt = test_all_points(pose7, uv, xyz, k, c, prm->error_threshold, &inl[0]);
    if (cnt > prm->min_n_pts_object) {
      suv.clear();
      sxyz.clear();
      for (int i = 0; i < k; i++)
        if (inl[i]) {
          suv.push_back(uv[2 * i]);
          suv.push_back(uv[2 * i + 1]);
          sxyz.push_back(xyz[3 * i]);
          sxyz.push_back(xyz[3 * i + 1]);
          sxyz.push_back(xyz[3 * i + 2]);
        }
      optimize_camera(pose7, &suv[0], &sxyz[0], cnt, c, prm->max_lm_tests, NULL);
      return 1;
    }
  }
  return 0;
}

int orc_ransac(const float* uv, const float* xyz, int k, const float K[4], const float cam[7],
               const orc_pose_params* prm, float pose7[7]) {
  return orc_ransac_addr(uv, xyz, NULL, k, K, cam, prm, pose7);
}

// ---- moped3d depth variants (A14) -------------------------------------------------
void orc_residuals_depth(int mode, const float pose7[7], const float* uv, const float* xyz,
                         const float* world, const float* wgt, int n, const float K[4],
                         const float cam[7], float alpha, float* err) {
  Camera c;
  camera_init(c, K, cam);
  residuals_depth(mode, pose7, uv, xyz, world, wgt, n, c, alpha, err);
}

int orc_optimize_camera_depth(int mode, float pose7[7], const float* uv, const float* xyz,
                              const float* world, const float* wgt, int n, const float K[4],
                              const float cam[7], float alpha, int itmax, float* info) {
  Camera c;
  camera_init(c, K, cam);
  float p[7];
  memcpy(p, pose7, sizeof p);
  LmProblem f = {uv, xyz, n, &c, mode, world, wgt, alpha};
  float linfo[3] = {0, 0, 0};
  int ret = lm_dif(f, p, itmax, linfo);
  if (info) memcpy(info, linfo, sizeof linfo);
  if (ret < 0) return ret;
  quat_normalize(p);
  memcpy(pose7, p, sizeof p);
  return ret;
}

// RANSAC of the depth variants: same skeleton as orc_ransac; initPose starts the
// translation at the centroid of the sample's world3D
// (...BACKPROJECTION_DEPTH_CPU.hpp:265-283), the inlier test is unchanged.
int orc_ransac_depth(int mode, const float* uv, const float* xyz, const float* world,
                     const float* wgt, int k, const float K[4], const float cam[7], float alpha,
                     const orc_pose_params* prm, float pose7[7]) {
  Camera c;
  camera_init(c, K, cam);
  std::vector<uint8_t> inl(k);
  std::vector<int> pick;
  auto gather = [&](const std::vector<int>& idx, std::vector<float>& a, std::vector<float>& b,
                    std::vector<float>& w3, std::vector<float>& ww) {
    a.clear(); b.clear(); w3.clear(); ww.clear();
    for (size_t j = 0; j < idx.size(); j++) {
      const int i = idx[j];
      a.push_back(uv[2 * i]); a.push_back(uv[2 * i + 1]);
      for (int x = 0; x < 3; x++) { b.push_back(xyz[3 * i + x]); w3.push_back(world[3 * i + x]); }
      ww.push_back(wgt[i]);
    }
  };
  std::vector<float> suv, sxyz, sw3, sww;
  for (int it = 0; it < prm->max_ransac_tests; it++) {
    std::vector<std::pair<float, int> > keyed(k);
    for (int i = 0; i < k; i++) keyed[i] = std::make_pair((float)orc_rand_(), i);
    std::sort(keyed.begin(), keyed.end());
    std::map<std::pair<float, float>, int> used;
    pick.clear();
    size_t pos = 0;
    while ((int)used.size() < prm->n_pts_align && pos < keyed.size()) {
      int i = keyed[pos++].second;
      std::pair<float, float> key(uv[2 * i], uv[2 * i + 1]);
      if (!used[key]++) pick.push_back(i);
    }
    if ((int)used.size() != prm->n_pts_align) return 0;
    for (int j = 0; j < 4; j++) pose7[j] = (float)((orc_rand_() & 255) / 256.);
    float sum[3] = {0, 0, 0};
    for (size_t j = 0; j < pick.size(); j++)
      for (int x = 0; x < 3; x++) sum[x] += world[3 * pick[j] + x];
    const int ns = (int)pick.size();
    for (int x = 0; x < 3; x++) pose7[4 + x] = sum[x] / ns;
    gather(pick, suv, sxyz, sw3, sww);
    float p[7];
    memcpy(p, pose7, sizeof p);
    LmProblem f = {&suv[0], &sxyz[0], ns, &c, mode, &sw3[0], &sww[0], alpha};
    float info[3];
    int ret = lm_dif(f, p, prm->max_lm_tests, info);
    if (ret >= 0) {
      quat_normalize(p);
      memcpy(pose7, p, sizeof p);
    }
    const int lm = (ret < 0) ? ret : (int)info[1];
    if (lm == -1) continue;
    int cnt = test_all_points(pose7, uv, xyz, k, c, prm->error_threshold, &inl[0]);
    if (cnt > prm->min_n_pts_object) {
      pick.clear();
      for (int i = 0; i < k; i++)
        if (inl[i]) pick.push_back(i);
      gather(pick, suv, sxyz, sw3, sww);
      memcpy(p, pose7, sizeof p);
      LmProblem g = {&suv[0], &sxyz[0], cnt, &c, mode, &sw3[0], &sww[0], alpha};
      if (lm_dif(g, p, prm->max_lm_tests, info) >= 0) {
        quat_normalize(p);
        memcpy(pose7, p, sizeof p);
      }
      return 1;
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------
// FILTER_PROJECTION_CPU::process (filter/FILTER_PROJECTION_CPU.hpp:80-162)
// ---------------------------------------------------------------------------
int orc_filter(const float* uv, const float* xyz, const int32_t* model_off, int n_models,
               const int32_t* obj_model, const float* obj_pose, int n_obj, const float K[4],
               const float cam[7], int min_points, float feature_distance, float min_score,
               float* score, uint8_t* keep, int32_t* out_order, int32_t* cl_members,
               int32_t* cl_off) {
  Camera c;
  camera_init(c, K, cam);
  typedef std::pair<float, float> Key;
  std::map<Key, std::pair<float, int> > best;  // keypoint -> (score, object) ; object -1 = none
  for (int m = 0; m < n_models; m++) {
    for (int o = 0; o < n_obj; o++) {
      if (obj_model[o] != m) continue;
      Mat34 T;
      tm_init(T, obj_pose + 7 * o, obj_pose + 7 * o + 4);
      std::vector<int> cl;
      float sc = 0;
      for (int i = model_off[m]; i < model_off[m + 1]; i++) {
        float p[2];
        project_one(T, c, xyz + 3 * i, p);
        p[0] -= uv[2 * i];
        p[1] -= uv[2 * i + 1];
        float err = p[0] * p[0] + p[1] * p[1];
        if (err < feature_distance) {
          cl.push_back(i);
          sc += 1. / (err + 1.);
        }
      }
      score[o] = sc;
      for (size_t j = 0; j < cl.size(); j++) {
        Key key(uv[2 * cl[j]], uv[2 * cl[j] + 1]);
        std::map<Key, std::pair<float, int> >::iterator itb = best.find(key);
        if (itb == best.end()) itb = best.insert(std::make_pair(key, std::make_pair(0.f, -1))).first;
        if (itb->second.first < sc) itb->second = std::make_pair(sc, o);
      }
    }
  }
  std::vector<std::vector<int> > newcl(n_obj);
  for (int m = 0; m < n_models; m++)
    for (int i = model_off[m]; i < model_off[m + 1]; i++) {
      std::map<Key, std::pair<float, int> >::iterator itb = best.find(Key(uv[2 * i], uv[2 * i + 1]));
      if (itb == best.end()) continue;
      int o = itb->second.second;
      if (o >= 0 && obj_model[o] == m) newcl[o].push_back(i - model_off[m]);
    }
  int kept = 0, w = 0;
  for (int o = 0; o < n_obj; o++) keep[o] = 0;
  for (int m = 0; m < n_models; m++)
    for (int o = 0; o < n_obj; o++) {
      if (obj_model[o] != m) continue;
      if ((int)newcl[o].size() < min_points || score[o] < min_score) continue;
      keep[o] = 1;
      out_order[kept] = o;
      cl_off[kept++] = w;
      for (size_t j = 0; j < newcl[o].size(); j++) cl_members[w++] = newcl[o][j];
    }
  cl_off[kept] = w;
  return kept;
}


// ---------------------------------------------------------------------------
// Whole frame after the nearest-neighbour search: the loop of
// MopedPimpl::processImages (src/moped.cpp:184-191) over
//   MATCH tail (ratio + scatter) -> CLUSTER -> POSE -> FILTER -> POSE2 -> FILTER2
// with the reference's constants (config.hpp:83-120), single image.
// OpenMP over models (CLUSTER, :186) and over (model, cluster, replica) tasks
// (POSE, :282) as the reference does; rand() is shared like there.
// Returns the number of final objects; obj_model/obj_pose/obj_score (capacity
// max_obj).  counts[4] = matches, clusters, objects after POSE, after FILTER.
// ---------------------------------------------------------------------------
int orc_frame_rest(const float* q_uv, const int32_t* idx1, const float* d1, const float* d2, int Q,
                   float ratio, const int32_t* model_of, const float* db_xyz, int n_models,
                   const float K[4], const float cam[7], const orc_frame_params* fp, int n_threads,
                   int32_t* obj_model, float* obj_pose, float* obj_score, int max_obj,
                   int32_t* counts) {
  return orc_frame_rest_inliers(q_uv, idx1, d1, d2, Q, ratio, model_of, db_xyz, n_models, K, cam, fp, n_threads,
                                obj_model, obj_pose, obj_score, max_obj, counts, NULL, NULL, 0);
}

// The same frame, and with every final object its INLIER SET as the reference defines one
// (testAllPoints, POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:166-180): the correspondences of the object's final
// cluster (what FILTER2 left it, FILTER_PROJECTION_CPU.hpp:136-160) whose squared reprojection error under the
// object's final pose is below the last POSE stage's ErrorThreshold.  inl_off [n_objects + 1], inl_q = query
// indices (rows of q_uv), capacity inl_cap; both may be NULL.  This is the point set north_star's pose bar
// ("within 1 px mean reprojection error of the reference pose over the reference's inlier set") is taken over.
int orc_frame_rest_inliers(const float* q_uv, const int32_t* idx1, const float* d1, const float* d2, int Q,
                           float ratio, const int32_t* model_of, const float* db_xyz, int n_models,
                           const float K[4], const float cam[7], const orc_frame_params* fp, int n_threads,
                           int32_t* obj_model, float* obj_pose, float* obj_score, int max_obj,
                           int32_t* counts, int32_t* inl_off, int32_t* inl_q, int inl_cap) {
#ifdef _OPENMP
  if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
  n_threads = 1;
#endif
  std::vector<int32_t> out_q(Q > 0 ? Q : 1), model_off(n_models + 1);
  const int M = orc_match_accept(idx1, d1, d2, Q, ratio, model_of, n_models, &out_q[0], &model_off[0]);
  std::vector<float> uv(2 * (size_t)std::max(M, 1)), xyz(3 * (size_t)std::max(M, 1));
  for (int i = 0; i < M; i++) {
    const int q = out_q[i];
    uv[2 * i] = q_uv[2 * q];
    uv[2 * i + 1] = q_uv[2 * q + 1];
    const int r = idx1[q];
    xyz[3 * i] = db_xyz[3 * (size_t)r];
    xyz[3 * i + 1] = db_xyz[3 * (size_t)r + 1];
    xyz[3 * i + 2] = db_xyz[3 * (size_t)r + 2];
  }
  // CLUSTER
  std::vector<std::vector<std::vector<int> > > clusters(n_models);
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (int m = 0; m < n_models; m++) {
    const int b = model_off[m], n = model_off[m + 1] - b;
    if (n == 0) continue;
    std::vector<int32_t> members(n), off(n + 2);
    const int k = orc_meanshift(&uv[2 * (size_t)b], n, 2, fp->ms_radius, fp->ms_merge, fp->ms_min_pts,
                                fp->ms_max_iter, &members[0], &off[0], NULL);
    for (int c = 0; c < k; c++)
      clusters[m].push_back(std::vector<int>(members.begin() + off[c], members.begin() + off[c + 1]));
  }
  int ncl = 0;
  for (int m = 0; m < n_models; m++) ncl += (int)clusters[m].size();
  // POSE / POSE2 share this
  struct Obj {
    int model;
    float pose[7];
  };
  std::vector<Obj> objects;
  auto run_pose = [&](const orc_pose_params& pp) {
    std::vector<std::pair<int, int> > tasks;
    for (int m = 0; m < n_models; m++)
      for (size_t c = 0; c < clusters[m].size(); c++)
        for (int r = 0; r < pp.max_objects_per_cluster; r++) tasks.push_back(std::make_pair(m, (int)c));
    std::vector<Obj> found(tasks.size());
    std::vector<char> ok(tasks.size(), 0);
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
    for (int t = 0; t < (int)tasks.size(); t++) {
      const int m = tasks[t].first;
      const std::vector<int>& cl = clusters[m][tasks[t].second];
      std::vector<float> cuv(2 * cl.size()), cxyz(3 * cl.size());
      for (size_t i = 0; i < cl.size(); i++) {
        const int g = model_off[m] + cl[i];
        cuv[2 * i] = uv[2 * g];
        cuv[2 * i + 1] = uv[2 * g + 1];
        cxyz[3 * i] = xyz[3 * g];
        cxyz[3 * i + 1] = xyz[3 * g + 1];
        cxyz[3 * i + 2] = xyz[3 * g + 2];
      }
      found[t].model = m;
      ok[t] = (char)orc_ransac_addr(&cuv[0], &cxyz[0], &cl[0], (int)cl.size(), K, cam, &pp, found[t].pose);   // (cl: match indices = address order)
    }
    for (size_t t = 0; t < tasks.size(); t++)
      if (ok[t]) objects.push_back(found[t]);  // task order (the 1-thread order of :294-303)
  };
  auto run_filter = [&](int min_points, float fdist, float min_score, std::vector<float>& score_out) {
    const int n_obj = (int)objects.size();
    std::vector<int32_t> om(std::max(n_obj, 1)), order(std::max(n_obj, 1)), members(std::max(M, 1)), off(n_obj + 2);
    std::vector<float> op(7 * (size_t)std::max(n_obj, 1)), score(std::max(n_obj, 1));
    std::vector<uint8_t> keep(std::max(n_obj, 1));
    for (int o = 0; o < n_obj; o++) {
      om[o] = objects[o].model;
      memcpy(&op[7 * (size_t)o], objects[o].pose, 28);
    }
    const int kept = orc_filter(&uv[0], &xyz[0], &model_off[0], n_models, &om[0], &op[0], n_obj, K, cam,
                                min_points, fdist, min_score, &score[0], &keep[0], &order[0], &members[0], &off[0]);
    std::vector<Obj> nobj;
    score_out.clear();
    for (int m = 0; m < n_models; m++) clusters[m].clear();
    // list order is preserved by the erase; clusters[m] receive kept objects of model m in list order
    std::vector<int> kept_of(n_obj, -1);
    for (int k = 0; k < kept; k++) kept_of[order[k]] = k;
    for (int o = 0; o < n_obj; o++)
      if (keep[o]) {
        nobj.push_back(objects[o]);
        score_out.push_back(score[o]);
      }
    for (int k = 0; k < kept; k++)
      clusters[om[order[k]]].push_back(std::vector<int>(members.begin() + off[k], members.begin() + off[k + 1]));
    objects.swap(nobj);
    return kept;
  };
  run_pose(fp->pose1);
  const int n_pose1 = (int)objects.size();
  std::vector<float> scores;
  int n_f1 = n_pose1;
  if (fp->run_stage2) {
    n_f1 = run_filter(fp->f1_min_points, fp->f1_feature_distance, fp->f1_min_score, scores);
    run_pose(fp->pose2);
    run_filter(fp->f2_min_points, fp->f2_feature_distance, fp->f2_min_score, scores);
  } else {
    scores.assign(objects.size(), 0.f);
  }
  if (counts) {
    counts[0] = M;
    counts[1] = ncl;
    counts[2] = n_pose1;
    counts[3] = n_f1;
  }
  const int n_out = std::min((int)objects.size(), max_obj);
  for (int o = 0; o < n_out; o++) {
    obj_model[o] = objects[o].model;
    memcpy(obj_pose + 7 * (size_t)o, objects[o].pose, 28);
    obj_score[o] = scores[o];
  }
  if (inl_off && inl_q) {
    // after a filter stage clusters[m] holds one cluster per kept object of model m, in list order
    Camera c;
    camera_init(c, K, cam);
    const float thr = fp->run_stage2 ? fp->pose2.error_threshold : fp->pose1.error_threshold;
    std::vector<int> seen(n_models, 0);
    int w = 0;
    for (int o = 0; o < n_out; o++) {
      inl_off[o] = w;
      const int m = objects[o].model;
      if (!fp->run_stage2 || seen[m] >= (int)clusters[m].size()) continue;  // (no per-object clusters without a filter stage)
      const std::vector<int>& cl = clusters[m][seen[m]++];
      for (size_t i = 0; i < cl.size(); i++) {
        const int g = model_off[m] + cl[i];
        if (test_all_points(objects[o].pose, &uv[2 * (size_t)g], &xyz[3 * (size_t)g], 1, c, thr, NULL) && w < inl_cap)
          inl_q[w++] = out_q[g];
      }
    }
    inl_off[n_out] = w;
  }
  return (int)objects.size();
}


// ===========================================================================
// Frames with several images (cameras).  The reference carries the image with every
// feature / match / correspondence: CLUSTER runs MeanShift per image in image order
// (CLUSTER_MEAN_SHIFT_CPU.hpp:186-196), POSE projects every correspondence through its own
// image (…REPROJECTION_CPU.hpp:100-138, 213-237) and never samples two correspondences with the
// same (image, coord2D) (:76-98), FILTER projects every match through its image and keys the
// ownership map by (coord2D, image) (FILTER_PROJECTION_CPU.hpp:89-141).  With one image these
// functions are the single-image ones above, operation for operation.
// ===========================================================================
static int test_all_points_images(const float* pose7, const float* uv, const float* xyz, const int32_t* img,
                                  int n, const Camera* cams, float thr, uint8_t* inlier) {
  Mat34 T;
  tm_init(T, pose7, pose7 + 4);
  int cnt = 0;
  for (int i = 0; i < n; i++) {
    float p[2];
    project_one(T, cams[img[i]], xyz + 3 * i, p);
    p[0] -= uv[2 * i];
    p[1] -= uv[2 * i + 1];
    float err = p[0] * p[0] + p[1] * p[1];
    bool in = err < thr;
    if (inlier) inlier[i] = in;
    cnt += in;
  }
  return cnt;
}

static float optimize_camera_images(float* pose7, const float* uv, const float* xyz, const int32_t* img, int n,
                                    const Camera* cams, int itmax) {
  float p[7];
  memcpy(p, pose7, sizeof p);
  LmProblem f = {uv, xyz, n, cams, 0, NULL, NULL, 0.f};
  f.img = img;
  float info[3];
  int ret = lm_dif(f, p, itmax, info);
  if (ret < 0) return (float)ret;
  quat_normalize(p);
  memcpy(pose7, p, sizeof p);
  return info[1];
}

static std::vector<Camera> make_cameras(const float* Ks, const float* cam_poses, int n_images) {
  std::vector<Camera> cams(n_images > 0 ? n_images : 1);
  for (int i = 0; i < n_images; i++) camera_init(cams[i], Ks + 4 * i, cam_poses + 7 * i);
  return cams;
}

int orc_project_test_images(const float pose7[7], const float* uv, const float* xyz, const int32_t* img, int n,
                            const float* Ks, const float* cam_poses, int n_images, float thr, uint8_t* inlier) {
  std::vector<Camera> cams = make_cameras(Ks, cam_poses, n_images);
  return test_all_points_images(pose7, uv, xyz, img, n, &cams[0], thr, inlier);
}

int orc_ransac_images(const float* uv, const float* xyz, const int32_t* img, int k, const float* Ks,
                      const float* cam_poses, int n_images, const orc_pose_params* prm, float pose7[7]) {
  std::vector<Camera> cams = make_cameras(Ks, cam_poses, n_images);
  std::vector<float> suv, sxyz;
  std::vector<int32_t> simg;
  std::vector<uint8_t> inl(k);
  for (int it = 0; it < prm->max_ransac_tests; it++) {
    std::vector<std::pair<float, int> > keyed(k);
    for (int i = 0; i < k; i++) keyed[i] = std::make_pair((float)orc_rand_(), i);
    std::sort(keyed.begin(), keyed.end());
    std::map<std::pair<int, std::pair<float, float> >, int> used;   // (image, coord2D) (:79)
    suv.clear();
    sxyz.clear();
    simg.clear();
    size_t pos = 0;
    while ((int)used.size() < prm->n_pts_align && pos < keyed.size()) {
      int i = keyed[pos++].second;
      std::pair<int, std::pair<float, float> > key(img[i], std::make_pair(uv[2 * i], uv[2 * i + 1]));
      if (!used[key]++) {
        suv.push_back(uv[2 * i]);
        suv.push_back(uv[2 * i + 1]);
        for (int x = 0; x < 3; x++) sxyz.push_back(xyz[3 * i + x]);
        simg.push_back(img[i]);
      }
    }
    if ((int)used.size() != prm->n_pts_align) return 0;
    for (int j = 0; j < 4; j++) pose7[j] = (float)((orc_rand_() & 255) / 256.);
    pose7[4] = 0.f;
    pose7[5] = 0.f;
    pose7[6] = 0.5f;
    int lm = (int)optimize_camera_images(pose7, &suv[0], &sxyz[0], &simg[0], (int)simg.size(), &cams[0],
                                         prm->max_lm_tests);
    if (lm == -1) continue;
    int cnt = test_all_points_images(pose7, uv, xyz, img, k, &cams[0], prm->error_threshold, &inl[0]);
    if (cnt > prm->min_n_pts_object) {
      suv.clear();
      sxyz.clear();
      simg.clear();
      for (int i = 0; i < k; i++)
        if (inl[i]) {
          suv.push_back(uv[2 * i]);
          suv.push_back(uv[2 * i + 1]);
          for (int x = 0; x < 3; x++) sxyz.push_back(xyz[3 * i + x]);
          simg.push_back(img[i]);
        }
      optimize_camera_images(pose7, &suv[0], &sxyz[0], &simg[0], cnt, &cams[0], prm->max_lm_tests);
      return 1;
    }
  }
  return 0;
}

int orc_filter_images(const float* uv, const int32_t* img, const float* xyz, const int32_t* model_off, int n_models,
                      const int32_t* obj_model, const float* obj_pose, int n_obj, const float* Ks,
                      const float* cam_poses, int n_images, int min_points, float feature_distance,
                      float min_score, float* score, uint8_t* keep, int32_t* out_order, int32_t* cl_members,
                      int32_t* cl_off) {
  std::vector<Camera> cams = make_cameras(Ks, cam_poses, n_images);
  typedef std::pair<std::pair<float, float>, int> Key;   // (coord2D, image) (:89)
  std::map<Key, std::pair<float, int> > best;
  for (int m = 0; m < n_models; m++) {
    for (int o = 0; o < n_obj; o++) {
      if (obj_model[o] != m) continue;
      Mat34 T;
      tm_init(T, obj_pose + 7 * o, obj_pose + 7 * o + 4);
      std::vector<int> cl;
      float sc = 0;
      for (int i = model_off[m]; i < model_off[m + 1]; i++) {
        float p[2];
        project_one(T, cams[img[i]], xyz + 3 * i, p);
        p[0] -= uv[2 * i];
        p[1] -= uv[2 * i + 1];
        float err = p[0] * p[0] + p[1] * p[1];
        if (err < feature_distance) {
          cl.push_back(i);
          sc += 1. / (err + 1.);
        }
      }
      score[o] = sc;
      for (size_t j = 0; j < cl.size(); j++) {
        Key key(std::make_pair(uv[2 * cl[j]], uv[2 * cl[j] + 1]), img[cl[j]]);
        std::map<Key, std::pair<float, int> >::iterator itb = best.find(key);
        if (itb == best.end()) itb = best.insert(std::make_pair(key, std::make_pair(0.f, -1))).first;
        if (itb->second.first < sc) itb->second = std::make_pair(sc, o);
      }
    }
  }
  std::vector<std::vector<int> > newcl(n_obj);
  for (int m = 0; m < n_models; m++)
    for (int i = model_off[m]; i < model_off[m + 1]; i++) {
      std::map<Key, std::pair<float, int> >::iterator itb = best.find(Key(std::make_pair(uv[2 * i], uv[2 * i + 1]), img[i]));
      if (itb == best.end()) continue;
      int o = itb->second.second;
      if (o >= 0 && obj_model[o] == m) newcl[o].push_back(i - model_off[m]);
    }
  int kept = 0, w = 0;
  for (int o = 0; o < n_obj; o++) keep[o] = 0;
  for (int m = 0; m < n_models; m++)
    for (int o = 0; o < n_obj; o++) {
      if (obj_model[o] != m) continue;
      if ((int)newcl[o].size() < min_points || score[o] < min_score) continue;
      keep[o] = 1;
      out_order[kept] = o;
      cl_off[kept++] = w;
      for (size_t j = 0; j < newcl[o].size(); j++) cl_members[w++] = newcl[o][j];
    }
  cl_off[kept] = w;
  return kept;
}

// orc_frame_rest for a frame whose features come from n_images images: q_img[Q] = image of every
// feature, Ks [n_images][4], cam_poses [n_images][7].  counts as orc_frame_rest.
int orc_frame_rest_images(const float* q_uv, const int32_t* q_img, const int32_t* idx1, const float* d1,
                          const float* d2, int Q, float ratio, const int32_t* model_of, const float* db_xyz,
                          int n_models, const float* Ks, const float* cam_poses, int n_images,
                          const orc_frame_params* fp, int32_t* obj_model, float* obj_pose, float* obj_score,
                          int max_obj, int32_t* counts) {
  std::vector<int32_t> out_q(Q > 0 ? Q : 1), model_off(n_models + 1);
  const int M = orc_match_accept(idx1, d1, d2, Q, ratio, model_of, n_models, &out_q[0], &model_off[0]);
  std::vector<float> uv(2 * (size_t)std::max(M, 1)), xyz(3 * (size_t)std::max(M, 1));
  std::vector<int32_t> img(std::max(M, 1));
  for (int i = 0; i < M; i++) {
    const int q = out_q[i];
    uv[2 * i] = q_uv[2 * q];
    uv[2 * i + 1] = q_uv[2 * q + 1];
    img[i] = q_img[q];
    const int r = idx1[q];
    for (int x = 0; x < 3; x++) xyz[3 * i + x] = db_xyz[3 * (size_t)r + x];
  }
  // CLUSTER: per model, per image in image order; cluster members = match indices inside the model
  std::vector<std::vector<std::vector<int> > > clusters(n_models);
  for (int m = 0; m < n_models; m++) {
    const int b = model_off[m], n = model_off[m + 1] - b;
    for (int im = 0; im < n_images; im++) {
      std::vector<float> pts;
      std::vector<int> which;
      for (int k = 0; k < n; k++)
        if (img[b + k] == im) {
          pts.push_back(uv[2 * (size_t)(b + k)]);
          pts.push_back(uv[2 * (size_t)(b + k) + 1]);
          which.push_back(k);
        }
      const int np = (int)which.size();
      if (np == 0) continue;
      std::vector<int32_t> members(np), off(np + 2);
      const int k = orc_meanshift(&pts[0], np, 2, fp->ms_radius, fp->ms_merge, fp->ms_min_pts, fp->ms_max_iter,
                                  &members[0], &off[0], NULL);
      for (int c = 0; c < k; c++) {
        std::vector<int> cl;
        for (int j = off[c]; j < off[c + 1]; j++) cl.push_back(which[members[j]]);
        clusters[m].push_back(cl);
      }
    }
  }
  int ncl = 0;
  for (int m = 0; m < n_models; m++) ncl += (int)clusters[m].size();
  struct Obj {
    int model;
    float pose[7];
  };
  std::vector<Obj> objects;
  auto run_pose = [&](const orc_pose_params& pp) {
    for (int m = 0; m < n_models; m++)
      for (size_t c = 0; c < clusters[m].size(); c++)
        for (int r = 0; r < pp.max_objects_per_cluster; r++) {
          const std::vector<int>& cl = clusters[m][c];
          std::vector<float> cuv(2 * cl.size()), cxyz(3 * cl.size());
          std::vector<int32_t> cimg(cl.size());
          for (size_t i = 0; i < cl.size(); i++) {
            const int g = model_off[m] + cl[i];
            cuv[2 * i] = uv[2 * g];
            cuv[2 * i + 1] = uv[2 * g + 1];
            for (int x = 0; x < 3; x++) cxyz[3 * i + x] = xyz[3 * g + x];
            cimg[i] = img[g];
          }
          Obj o;
          o.model = m;
          if (orc_ransac_images(&cuv[0], &cxyz[0], &cimg[0], (int)cl.size(), Ks, cam_poses, n_images, &pp, o.pose))
            objects.push_back(o);
        }
  };
  auto run_filter = [&](int min_points, float fdist, float min_score, std::vector<float>& score_out) {
    const int n_obj = (int)objects.size();
    std::vector<int32_t> om(std::max(n_obj, 1)), order(std::max(n_obj, 1)), members(std::max(M, 1)), off(n_obj + 2);
    std::vector<float> op(7 * (size_t)std::max(n_obj, 1)), score(std::max(n_obj, 1));
    std::vector<uint8_t> keep(std::max(n_obj, 1));
    for (int o = 0; o < n_obj; o++) {
      om[o] = objects[o].model;
      memcpy(&op[7 * (size_t)o], objects[o].pose, 28);
    }
    const int kept = orc_filter_images(&uv[0], &img[0], &xyz[0], &model_off[0], n_models, &om[0], &op[0], n_obj, Ks,
                                       cam_poses, n_images, min_points, fdist, min_score, &score[0], &keep[0],
                                       &order[0], &members[0], &off[0]);
    std::vector<Obj> nobj;
    score_out.clear();
    for (int m = 0; m < n_models; m++) clusters[m].clear();
    for (int o = 0; o < n_obj; o++)
      if (keep[o]) {
        nobj.push_back(objects[o]);
        score_out.push_back(score[o]);
      }
    for (int k = 0; k < kept; k++)
      clusters[om[order[k]]].push_back(std::vector<int>(members.begin() + off[k], members.begin() + off[k + 1]));
    objects.swap(nobj);
    return kept;
  };
  run_pose(fp->pose1);
  const int n_pose1 = (int)objects.size();
  std::vector<float> scores;
  int n_f1 = n_pose1;
  if (fp->run_stage2) {
    n_f1 = run_filter(fp->f1_min_points, fp->f1_feature_distance, fp->f1_min_score, scores);
    run_pose(fp->pose2);
    run_filter(fp->f2_min_points, fp->f2_feature_distance, fp->f2_min_score, scores);
  } else {
    scores.assign(objects.size(), 0.f);
  }
  if (counts) {
    counts[0] = M;
    counts[1] = ncl;
    counts[2] = n_pose1;
    counts[3] = n_f1;
  }
  const int n_out = std::min((int)objects.size(), max_obj);
  for (int o = 0; o < n_out; o++) {
    obj_model[o] = objects[o].model;
    memcpy(obj_pose + 7 * (size_t)o, objects[o].pose, 28);
    obj_score[o] = scores[o];
  }
  return (int)objects.size();
}

}  // extern "C"
