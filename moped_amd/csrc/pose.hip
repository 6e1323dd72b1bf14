// POSE: many-hypotheses RANSAC + LM refine on gfx950.
//
// Replaces POSE_RANSAC_LM_DIFF_REPROJECTION_CPU::process / RANSAC
// (moped2/libmoped/src/pose/POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:76-211,264-307).
// The reference draws 5 points, starts levmar from a random quaternion and keeps
// the FIRST hypothesis with enough inliers (up to 600 x 200 LM iterations per
// replica).  Here every (cluster, replica) gets one 1024-thread workgroup that
// evaluates n_hypotheses minimal samples at once, one hypothesis per lane:
//   sample 3 (+1) points with distinct image coordinates (:76-98)
//   -> P3P (Grunert's quartic, built by polynomial arithmetic, fp64)
//   -> pick the root that best reprojects the 4th point
//   -> count inliers against the LDS-cached cluster with the reference's exact
//      project()/testAllPoints arithmetic (:166-180)
// then a workgroup arg-max, and one wavefront refines the winner on its inliers
// with Levenberg-Marquardt: first on plain pixel residuals (fast convergence),
// then on the reference's squared-pixel residuals (lmFuncQuat, :100-138) so the
// refined pose sits at the minimiser optimizeCamera (:140-164) converges to.
// Parity with the reference is therefore at final-pose level.
#include <algorithm>

#include <cstdlib>
#include <cstring>

#include "filter_dev.h"

MH_TRACE_TU()

namespace mh {

DevCam make_devcam(const mh_cam& cam) {
  DevCam d;
  for (int i = 0; i < 4; ++i) d.K[i] = cam.K[i];
  TM T;
  tm_from_pose(T, cam.cam, cam.cam + 4);  // image->TM.init(cameraPose), src/moped.cpp:168-169
  for (int i = 0; i < 9; ++i) d.Rc[i] = T.r[i];
  for (int i = 0; i < 3; ++i) d.tc[i] = T.t[i];
  return d;
}

namespace {

// 256 threads = one wavefront per SIMD: at 253 VGPRs two POSE workgroups then share a compute unit, and the POSE
// launches of the frames in flight hold half as many units as with 512 threads (two wavefronts per SIMD: the whole
// register file of the unit for 0.13 ms).  The MATCH workgroups of other frames need a unit to themselves (236 VGPRs,
// two wavefronts per SIMD) and queued behind them: +5% frames/s at config 1 and on the per-rank load of 8 shards.
// The first hypothesis stage is 256 lanes wide anyway.
#ifndef MH_POSE_THREADS
#define MH_POSE_THREADS 256
#endif
constexpr int POSE_THREADS = MH_POSE_THREADS;
// Wavefronts of this kernel that must fit a SIMD together = the register cap (512 / this).  Round 3: the kernel had grown
// to 260 registers (256 + 4 AGPRs) -- ONE wavefront per SIMD, one POSE workgroup per compute unit, where the 256-thread
// shape above was chosen so that two share a unit (mh_pose_kernel_info reports what the runtime sees).  Capped at 256
// (no spills): config 1 10 870 -> 12 020 frames/s, objects unchanged.
#ifndef MH_POSE_MIN_WAVES
#define MH_POSE_MIN_WAVES 2
#endif
// ... and of the hypothesis-only instantiation of a split launch (no LM code: 176 registers since round 5)
#ifndef MH_POSE_SPLIT_WAVES
#define MH_POSE_SPLIT_WAVES 2
#endif

__device__ __forceinline__ uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// ---- quartic / cubic roots (fp64) ------------------------------------------------
// Real roots of c4 x^4 + c3 x^3 + c2 x^2 + c1 x + c0 in four fixed slots
// (NaN = no root) so every access is statically indexed (registers, no scratch).
__device__ __attribute__((unused)) void solve_quartic(const double* c, double& r0, double& r1, double& r2, double& r3) {
  const double nan = __builtin_nan("");
  r0 = r1 = r2 = r3 = nan;
  if (!(fabs(c[4]) > 1e-300)) return;
  const double a = c[3] / c[4], b = c[2] / c[4], cc = c[1] / c[4], d = c[0] / c[4];
  const double a2 = a * a;
  const double p = b - 0.375 * a2;
  const double q = cc - 0.5 * a * b + 0.125 * a2 * a;
  const double r = d - 0.25 * a * cc + 0.0625 * a2 * b - (3.0 / 256.0) * a2 * a2;
  // resolvent cubic z^3 + 2p z^2 + (p^2 - 4r) z - q^2 = 0, take its largest real root
  const double A = 2.0 * p, B = p * p - 4.0 * r, C = -q * q;
  const double Q = (A * A - 3.0 * B) / 9.0;
  const double R = (2.0 * A * A * A - 9.0 * A * B + 27.0 * C) / 54.0;
  double z;
  if (R * R < Q * Q * Q) {
    const double sq = sqrt(Q);
    double ct = R / (sq * sq * sq);
    ct = fmin(1.0, fmax(-1.0, ct));
    const double th = acos(ct);
    const double z0 = -2.0 * sq * cos(th / 3.0) - A / 3.0;
    const double z1 = -2.0 * sq * cos((th + 6.283185307179586) / 3.0) - A / 3.0;
    const double z2 = -2.0 * sq * cos((th - 6.283185307179586) / 3.0) - A / 3.0;
    z = fmax(z0, fmax(z1, z2));
  } else {
    const double S = -copysign(cbrt(fabs(R) + sqrt(fmax(R * R - Q * Q * Q, 0.0))), R);
    const double T = (S != 0.0) ? Q / S : 0.0;
    z = S + T - A / 3.0;
  }
  const double shift = -0.25 * a;
  if (z > 1e-14 * (1.0 + fabs(p))) {
    const double s = sqrt(z);
    const double t1 = 0.5 * (p + z - q / s), t2 = 0.5 * (p + z + q / s);
    double disc = z - 4.0 * t1;  // y^2 + s y + t1
    if (disc >= 0.0) {
      const double sd = sqrt(disc);
      r0 = 0.5 * (-s + sd) + shift;
      r1 = 0.5 * (-s - sd) + shift;
    }
    disc = z - 4.0 * t2;         // y^2 - s y + t2
    if (disc >= 0.0) {
      const double sd = sqrt(disc);
      r2 = 0.5 * (s + sd) + shift;
      r3 = 0.5 * (s - sd) + shift;
    }
  } else {  // q ~ 0: biquadratic in y
    const double disc = p * p - 4.0 * r;
    if (disc >= 0.0) {
      const double sd = sqrt(disc);
      const double y2a = 0.5 * (-p + sd), y2b = 0.5 * (-p - sd);
      if (y2a >= 0.0) {
        r0 = sqrt(y2a) + shift;
        r1 = -sqrt(y2a) + shift;
      }
      if (y2b >= 0.0) {
        r2 = sqrt(y2b) + shift;
        r3 = -sqrt(y2b) + shift;
      }
    }
  }
  // two Newton steps on the original polynomial (NaN stays NaN)
  auto polish = [&](double x) {
    for (int it = 0; it < 2; ++it) {
      const double f = (((c[4] * x + c[3]) * x + c[2]) * x + c[1]) * x + c[0];
      const double df = ((4.0 * c[4] * x + 3.0 * c[3]) * x + 2.0 * c[2]) * x + c[1];
      if (fabs(df) > 1e-300) x -= f / df;
    }
    return x;
  };
  r0 = polish(r0);
  r1 = polish(r1);
  r2 = polish(r2);
  r3 = polish(r3);
}

struct Pose34 {
  float r[9];
  float t[3];
};

__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
__device__ __forceinline__ bool normalize3(double* a) {
  const double n = sqrt(dot3(a, a));
  if (!(n > 1e-12)) return false;
  a[0] /= n;
  a[1] /= n;
  a[2] /= n;
  return true;
}

// Orthonormal frame from two edge vectors (columns e1,e2,e3).
__device__ __attribute__((unused)) __forceinline__ bool frame_from(const double* v12, const double* v13, double* F) {
  double e1[3] = {v12[0], v12[1], v12[2]}, e3[3], e2[3];
  if (!normalize3(e1)) return false;
  cross3(e1, v13, e3);
  if (!normalize3(e3)) return false;
  cross3(e3, e1, e2);
  for (int i = 0; i < 3; ++i) {
    F[i * 3 + 0] = e1[i];
    F[i * 3 + 1] = e2[i];
    F[i * 3 + 2] = e3[i];
  }
  return true;
}

// P3P: model points X[3], unit bearings y[3] (camera frame).  Calls
// consider(R, t) for each of the up to 4 poses (model -> camera frame).
// Grunert's formulation: with s2 = u s1, s3 = v s1 the three law-of-cosines
// equations reduce to u = N(v)/D(v) and the quartic
//   N^2 - 2 cos(gamma) N D + E D^2 = 0   (lengths normalised by b = |X1 X3|).
template <typename F>
__device__ void p3p(const double (&X)[3][3], const double (&y)[3][3], F&& consider) {
  double d12[3], d13[3], d23[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    d12[i] = X[1][i] - X[0][i];
    d13[i] = X[2][i] - X[0][i];
    d23[i] = X[2][i] - X[1][i];
  }
  const double a2 = dot3(d23, d23), b2 = dot3(d13, d13), c2 = dot3(d12, d12);
  if (!(b2 > 1e-18) || !(a2 > 1e-18) || !(c2 > 1e-18)) return;
  const double ca = dot3(y[1], y[2]), cb = dot3(y[0], y[2]), cg = dot3(y[0], y[1]);
  const double A = a2 / b2, C = c2 / b2, k = C - A;
  // p(v) = 1 - 2 cb v + v^2 ; N = (v^2 - 1) + k p ; D = 2 (ca v - cg) ; E = 1 - C p
  const double n0 = -1.0 + k, n1 = -2.0 * k * cb, n2 = 1.0 + k;
  const double d0 = -2.0 * cg, d1 = 2.0 * ca;
  const double e0 = 1.0 - C, e1 = 2.0 * C * cb, e2 = -C;
  const double dd0 = d0 * d0, dd1 = 2 * d0 * d1, dd2 = d1 * d1;  // D^2
  double c[5];
  c[0] = n0 * n0 + e0 * dd0 - 2.0 * cg * (n0 * d0);
  c[1] = 2 * n0 * n1 + (e0 * dd1 + e1 * dd0) - 2.0 * cg * (n0 * d1 + n1 * d0);
  c[2] = 2 * n0 * n2 + n1 * n1 + (e0 * dd2 + e1 * dd1 + e2 * dd0) - 2.0 * cg * (n1 * d1 + n2 * d0);
  c[3] = 2 * n1 * n2 + (e1 * dd2 + e2 * dd1) - 2.0 * cg * (n2 * d1);
  c[4] = n2 * n2 + e2 * dd2;
  double rt[4];
  solve_quartic(c, rt[0], rt[1], rt[2], rt[3]);
  double Fx[9];
  if (!frame_from(d12, d13, Fx)) return;
  const double b = sqrt(b2);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double v = rt[i];
    if (!(v > 0.0)) continue;
    const double pv = 1.0 - 2.0 * cb * v + v * v;
    const double D = d0 + d1 * v;
    if (!(pv > 1e-14) || !(fabs(D) > 1e-12)) continue;
    const double u = (n0 + n1 * v + n2 * v * v) / D;
    if (!(u > 0.0)) continue;
    const double s1 = b / sqrt(pv), s2 = u * s1, s3 = v * s1;
    double p12[3], p13[3], P0[3], Fp[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      P0[j] = s1 * y[0][j];
      p12[j] = s2 * y[1][j] - P0[j];
      p13[j] = s3 * y[2][j] - P0[j];
    }
    if (!frame_from(p12, p13, Fp)) continue;
    double R[9], t[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cidx = 0; cidx < 3; ++cidx)
        R[r * 3 + cidx] = Fp[r * 3 + 0] * Fx[cidx * 3 + 0] + Fp[r * 3 + 1] * Fx[cidx * 3 + 1] +
                          Fp[r * 3 + 2] * Fx[cidx * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      t[r] = P0[r] - (R[r * 3 + 0] * X[0][0] + R[r * 3 + 1] * X[0][1] + R[r * 3 + 2] * X[0][2]);
    consider(R, t);
  }
}

// ---- the same P3P with fp64 only where it is needed (round 5) --------------------------------------------------------
// The fp64 form above spent ~60 k of a hypothesis round's 68 k cycles in software transcendentals and divisions
// (acos, three cos, cbrt, a dozen sqrt, two dozen divisions, all fp64: POSE_PROF, profiles/r04_stage_latency.txt) on
// inputs that are fp32 pixels with half a pixel of noise.  What needs fp64 is the ALGEBRA of the quartic -- its
// coefficients cancel -- and the roots' final digits; nothing else does:
//   * bearings and model points stay fp32 (their rounding is 1e-4 px of image noise);
//   * the coefficients are formed in fp64 from those values (exact conversions, ~80 multiply-adds);
//   * the roots are SEEDED by the closed form in fp32 (v_sqrt / v_rcp / cosf / acosf / cbrtf) and then polished by
//     three Newton steps on the fp64 polynomial -- the step's 1 / f' from an fp32 reciprocal: an inverse that is off by
//     1e-7 makes the iteration converge by that factor per step instead of quadratically, which is plenty -- and kept
//     only if the polynomial's value there is zero to 1e-8 of its terms' magnitude (a seed that the fp32 discriminants
//     invented polishes to nothing and is dropped);
//   * every pose that follows from a root (two frames, a 3 x 3 product, a translation) is fp32.
// Same formulation, same root selection by the fourth point, same scoring; the hypotheses agree with the fp64 form's to
// ~1e-6 in the pose, i.e. a few 1e-4 px -- which winner a round picks can differ where two hypotheses tie to a point on
// the threshold, never what the refine converges to (tests/tools/frame_stress.py, scripts/p3p_ab.py).
#ifndef MH_P3P_FP64
#define MH_P3P_FP64 0
#endif

__device__ __forceinline__ float rcp_f(float x) { return __builtin_amdgcn_rcpf(x); }

// Real roots of c4 x^4 + .. + c0 (fp64 coefficients): fp32 closed-form seeds, fp64 Newton polish; NaN = no root.
__device__ void solve_quartic_seeded(const double* c, double& r0, double& r1, double& r2, double& r3) {
  const float nanf_ = __builtin_nanf("");
  r0 = r1 = r2 = r3 = __builtin_nan("");
  const float c4f = (float)c[4];
  if (!(fabsf(c4f) > 1e-30f)) return;
  const float i4 = rcp_f(c4f);
  const float a = (float)c[3] * i4, b = (float)c[2] * i4, cc = (float)c[1] * i4, d = (float)c[0] * i4;
  const float a2 = a * a;
  const float p = b - 0.375f * a2;
  const float q = cc - 0.5f * a * b + 0.125f * a2 * a;
  const float r = d - 0.25f * a * cc + 0.0625f * a2 * b - (3.0f / 256.0f) * a2 * a2;
  const float A = 2.0f * p, B = p * p - 4.0f * r, C = -q * q;
  const float Q = (A * A - 3.0f * B) * (1.0f / 9.0f);
  const float R = (2.0f * A * A * A - 9.0f * A * B + 27.0f * C) * (1.0f / 54.0f);
  float z;
  if (R * R < Q * Q * Q) {
    const float sq = sqrtf(Q);
    float ct = R * rcp_f(sq * sq * sq);
    ct = fminf(1.0f, fmaxf(-1.0f, ct));
    const float th = acosf(ct) * (1.0f / 3.0f);
    const float A3 = A * (1.0f / 3.0f);
    const float z0 = -2.0f * sq * cosf(th) - A3;
    const float z1 = -2.0f * sq * cosf(th + 2.0943951f) - A3;
    const float z2 = -2.0f * sq * cosf(th - 2.0943951f) - A3;
    z = fmaxf(z0, fmaxf(z1, z2));
  } else {
    const float S = -copysignf(cbrtf(fabsf(R) + sqrtf(fmaxf(R * R - Q * Q * Q, 0.0f))), R);
    const float T = (S != 0.0f) ? Q * rcp_f(S) : 0.0f;
    z = S + T - A * (1.0f / 3.0f);
  }
  const float shift = -0.25f * a;
  float s0 = nanf_, s1 = nanf_, s2 = nanf_, s3 = nanf_;
  // (discriminants that fp32 puts a hair below zero are double roots to the seed's accuracy: taken as zero -- the polish
  //  and the residual test below decide whether a root is really there)
  if (z > 1e-6f * (1.0f + fabsf(p))) {
    const float s = sqrtf(z), qs = q * rcp_f(s);
    const float t1 = 0.5f * (p + z - qs), t2 = 0.5f * (p + z + qs);
    const float tol = 1e-4f * (fabsf(z) + fabsf(p) + fabsf(qs));
    float disc = z - 4.0f * t1;
    if (disc >= -tol) {
      const float sd = sqrtf(fmaxf(disc, 0.0f));
      s0 = 0.5f * (-s + sd) + shift;
      s1 = 0.5f * (-s - sd) + shift;
    }
    disc = z - 4.0f * t2;
    if (disc >= -tol) {
      const float sd = sqrtf(fmaxf(disc, 0.0f));
      s2 = 0.5f * (s + sd) + shift;
      s3 = 0.5f * (s - sd) + shift;
    }
  } else {
    const float disc = p * p - 4.0f * r;
    if (disc >= -1e-4f * (p * p + fabsf(r))) {
      const float sd = sqrtf(fmaxf(disc, 0.0f));
      const float y2a = 0.5f * (-p + sd), y2b = 0.5f * (-p - sd);
      if (y2a >= 0.0f) {
        s0 = sqrtf(y2a) + shift;
        s1 = -sqrtf(y2a) + shift;
      }
      if (y2b >= 0.0f) {
        s2 = sqrtf(y2b) + shift;
        s3 = -sqrtf(y2b) + shift;
      }
    }
  }
  auto polish = [&](float seed) -> double {
    if (!(seed == seed)) return __builtin_nan("");
    double x = (double)seed;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const double f = (((c[4] * x + c[3]) * x + c[2]) * x + c[1]) * x + c[0];
      const double df = ((4.0 * c[4] * x + 3.0 * c[3]) * x + 2.0 * c[2]) * x + c[1];
      const float dff = (float)df;
      if (fabsf(dff) > 1e-30f) x -= f * (double)rcp_f(dff);
    }
    // a root?  |f| against the size of its terms
    const double ax = fabs(x);
    const double f = (((c[4] * x + c[3]) * x + c[2]) * x + c[1]) * x + c[0];
    const double mag = (((fabs(c[4]) * ax + fabs(c[3])) * ax + fabs(c[2])) * ax + fabs(c[1])) * ax + fabs(c[0]);
    return fabs(f) <= 1e-8 * mag ? x : __builtin_nan("");
  };
  r0 = polish(s0);
  r1 = polish(s1);
  r2 = polish(s2);
  r3 = polish(s3);
}

__device__ __forceinline__ bool frame_from_f(const float* v12, const float* v13, float* F) {
  float e1[3] = {v12[0], v12[1], v12[2]}, e3[3], e2[3];
  float n = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
  if (!(n > 1e-24f)) return false;
  float inv = __builtin_amdgcn_rsqf(n);
  e1[0] *= inv; e1[1] *= inv; e1[2] *= inv;
  e3[0] = e1[1] * v13[2] - e1[2] * v13[1];
  e3[1] = e1[2] * v13[0] - e1[0] * v13[2];
  e3[2] = e1[0] * v13[1] - e1[1] * v13[0];
  n = e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2];
  if (!(n > 1e-24f)) return false;
  inv = __builtin_amdgcn_rsqf(n);
  e3[0] *= inv; e3[1] *= inv; e3[2] *= inv;
  e2[0] = e3[1] * e1[2] - e3[2] * e1[1];
  e2[1] = e3[2] * e1[0] - e3[0] * e1[2];
  e2[2] = e3[0] * e1[1] - e3[1] * e1[0];
  for (int i = 0; i < 3; ++i) {
    F[i * 3 + 0] = e1[i];
    F[i * 3 + 1] = e2[i];
    F[i * 3 + 2] = e3[i];
  }
  return true;
}

// P3P, inputs and poses fp32, the quartic fp64 (see above).  consider(R[9], t[3]) per pose, floats.
template <typename F>
__device__ void p3p_f(const float (&X)[3][3], const float (&y)[3][3], F&& consider) {
  float d12[3], d13[3], d23[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    d12[i] = X[1][i] - X[0][i];
    d13[i] = X[2][i] - X[0][i];
    d23[i] = X[2][i] - X[1][i];
  }
  auto dotd = [](const float* u, const float* w) { return (double)u[0] * (double)w[0] + (double)u[1] * (double)w[1] + (double)u[2] * (double)w[2]; };
  const double a2 = dotd(d23, d23), b2 = dotd(d13, d13), c2 = dotd(d12, d12);
  if (!(b2 > 1e-18) || !(a2 > 1e-18) || !(c2 > 1e-18)) return;
  const double ca = dotd(y[1], y[2]), cb = dotd(y[0], y[2]), cg = dotd(y[0], y[1]);
  const float b2f = (float)b2;
  const double ib2 = (double)rcp_f(b2f) * (2.0 - b2 * (double)rcp_f(b2f));   // 1 / b2: an fp32 reciprocal + one Newton step
  const double A = a2 * ib2, C = c2 * ib2, k = C - A;
  const double n0 = -1.0 + k, n1 = -2.0 * k * cb, n2 = 1.0 + k;
  const double d0 = -2.0 * cg, d1 = 2.0 * ca;
  const double e0 = 1.0 - C, e1 = 2.0 * C * cb, e2 = -C;
  const double dd0 = d0 * d0, dd1 = 2 * d0 * d1, dd2 = d1 * d1;
  double c[5];
  c[0] = n0 * n0 + e0 * dd0 - 2.0 * cg * (n0 * d0);
  c[1] = 2 * n0 * n1 + (e0 * dd1 + e1 * dd0) - 2.0 * cg * (n0 * d1 + n1 * d0);
  c[2] = 2 * n0 * n2 + n1 * n1 + (e0 * dd2 + e1 * dd1 + e2 * dd0) - 2.0 * cg * (n1 * d1 + n2 * d0);
  c[3] = 2 * n1 * n2 + (e1 * dd2 + e2 * dd1) - 2.0 * cg * (n2 * d1);
  c[4] = n2 * n2 + e2 * dd2;
  double rt[4];
  solve_quartic_seeded(c, rt[0], rt[1], rt[2], rt[3]);
  float Fx[9];
  if (!frame_from_f(d12, d13, Fx)) return;
  const float bf = sqrtf(b2f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double vd = rt[i];
    if (!(vd > 0.0)) continue;
    // (the two quantities that cancel -- p(v) and N(v) / D(v) -- from the fp64 root; fp32 from here)
    const float pv = (float)(1.0 - 2.0 * cb * vd + vd * vd);
    const float D = (float)(d0 + d1 * vd);
    if (!(pv > 1e-12f) || !(fabsf(D) > 1e-10f)) continue;
    const float u = (float)(n0 + n1 * vd + n2 * vd * vd) / D;
    if (!(u > 0.0f)) continue;
    const float v = (float)vd;
    const float s1 = bf * __builtin_amdgcn_rsqf(pv), s2 = u * s1, s3 = v * s1;
    float p12[3], p13[3], P0[3], Fp[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      P0[j] = s1 * y[0][j];
      p12[j] = s2 * y[1][j] - P0[j];
      p13[j] = s3 * y[2][j] - P0[j];
    }
    if (!frame_from_f(p12, p13, Fp)) continue;
    float R[9], t[3];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int cidx = 0; cidx < 3; ++cidx)
        R[rr * 3 + cidx] = Fp[rr * 3 + 0] * Fx[cidx * 3 + 0] + Fp[rr * 3 + 1] * Fx[cidx * 3 + 1] + Fp[rr * 3 + 2] * Fx[cidx * 3 + 2];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) t[rr] = P0[rr] - (R[rr * 3 + 0] * X[0][0] + R[rr * 3 + 1] * X[0][1] + R[rr * 3 + 2] * X[0][2]);
    consider(R, t);
  }
}

__device__ __forceinline__ void to_world_f(const DevCam& cam, const float* R, const float* t, Pose34& o) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c)
      o.r[r * 3 + c] = cam.Rc[r * 3 + 0] * R[0 * 3 + c] + cam.Rc[r * 3 + 1] * R[1 * 3 + c] + cam.Rc[r * 3 + 2] * R[2 * 3 + c];
    o.t[r] = cam.Rc[r * 3 + 0] * t[0] + cam.Rc[r * 3 + 1] * t[1] + cam.Rc[r * 3 + 2] * t[2] + cam.tc[r];
  }
}

// camera-frame pose (R,t) -> world pose: Rw = Rc R, tw = Rc t + tc
// (The camera constants' fp64 images are hoisted out of the hypothesis loops into VGPR pairs; pinning the conversions to
// their uses with an empty asm was tried in round 3 to free those registers: 12 spilled registers became 92.)
__device__ __attribute__((unused)) __forceinline__ void to_world(const DevCam& cam, const double* R, const double* t, Pose34& o) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c)
      o.r[r * 3 + c] = (float)((double)cam.Rc[r * 3 + 0] * R[0 * 3 + c] + (double)cam.Rc[r * 3 + 1] * R[1 * 3 + c] +
                               (double)cam.Rc[r * 3 + 2] * R[2 * 3 + c]);
    o.t[r] = (float)((double)cam.Rc[r * 3 + 0] * t[0] + (double)cam.Rc[r * 3 + 1] * t[1] +
                     (double)cam.Rc[r * 3 + 2] * t[2] + (double)cam.tc[r]);
  }
}

// rotation matrix -> quaternion (x,y,z,w), normalised
__device__ void rot_to_quat(const float* r, float* q) {
  const float tr = r[0] + r[4] + r[8];
  float x, y, z, w;
  if (tr > 0.f) {
    const float s = sqrtf(tr + 1.f) * 2.f;
    w = 0.25f * s;
    x = (r[7] - r[5]) / s;
    y = (r[2] - r[6]) / s;
    z = (r[3] - r[1]) / s;
  } else if (r[0] > r[4] && r[0] > r[8]) {
    const float s = sqrtf(1.f + r[0] - r[4] - r[8]) * 2.f;
    w = (r[7] - r[5]) / s;
    x = 0.25f * s;
    y = (r[1] + r[3]) / s;
    z = (r[2] + r[6]) / s;
  } else if (r[4] > r[8]) {
    const float s = sqrtf(1.f + r[4] - r[0] - r[8]) * 2.f;
    w = (r[2] - r[6]) / s;
    x = (r[1] + r[3]) / s;
    y = 0.25f * s;
    z = (r[5] + r[7]) / s;
  } else {
    const float s = sqrtf(1.f + r[8] - r[0] - r[4]) * 2.f;
    w = (r[3] - r[1]) / s;
    x = (r[2] + r[6]) / s;
    y = (r[5] + r[7]) / s;
    z = 0.25f * s;
  }
  const float n = 1.f / sqrtf(x * x + y * y + z * z + w * w);
  q[0] = x * n;
  q[1] = y * n;
  q[2] = z * n;
  q[3] = w * n;
}

// ---- LM refine, one wavefront ----------------------------------------------------
// 6-DoF local update (omega, dt): R <- exp(omega) R, t <- t + dt.
//
// KIND selects the residual model (what the reference class minimises):
//   0  POSE_RANSAC_LM_DIFF_REPROJECTION_CPU   (moped2 …REPROJECTION_CPU.hpp:100-138)
//   1  POSE_RANSAC_LM_DIFF_BACKPROJECTION_DEPTH_CPU (moped3d …BACKPROJECTION_DEPTH_CPU.hpp:108-190)
//   2  POSE_RANSAC_LM_DIFF_REPROJECTION_DEPTH_CPU   (moped3d …REPROJECTION_DEPTH_CPU.hpp:106-216)
// phase 0 minimises the plain (un-squared) errors for fast convergence, phase 1 the
// reference's squared errors so the result sits at the minimiser levmar converges to.
// A point is pts[PS*i ..] = u,v, x,y,z [, wx,wy,wz, cauchyWeight].
constexpr int MAX_ROWS = 4;
// KIND 3 = KIND 0's residuals with every correspondence in its own image (several cameras): a point carries
// its image number behind x,y,z and every projection goes through that image's camera (LmData::image,
// …REPROJECTION_CPU.hpp:213-237).
template <int KIND> struct PointStride { static constexpr int value = (KIND == 0) ? 5 : ((KIND == 3) ? 6 : 9); };
// the camera of a point: `cams` is ONE camera for KIND 0..2, the frame's camera table for KIND 3
template <int KIND>
__device__ __forceinline__ const DevCam& cam_of(const DevCam* cams, const float* p) {
  return KIND == 3 ? cams[__float_as_int(p[5])] : cams[0];
}

struct Accum {
  float H[21];  // upper triangle of J^T J
  float g[6];   // J^T r
};

// J row for a residual whose gradient w.r.t. the camera-frame point is gc:
// dc/dw = Rc^T, dw/d(dt) = I, dw/d(omega) = -[y]_x with y the rotated model point
// ->  translation part = Rc gc (world frame), rotation part = y x (Rc gc).
__device__ __forceinline__ void row_from_grad(const DevCam& cam, const float* y, const float* gc, float* J) {
  float gw[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) gw[k] = gc[0] * cam.Rc[k * 3 + 0] + gc[1] * cam.Rc[k * 3 + 1] + gc[2] * cam.Rc[k * 3 + 2];
  J[0] = y[1] * gw[2] - y[2] * gw[1];
  J[1] = y[2] * gw[0] - y[0] * gw[2];
  J[2] = y[0] * gw[1] - y[1] * gw[0];
  J[3] = gw[0];
  J[4] = gw[1];
  J[5] = gw[2];
}

// Residual rows of one correspondence.  Returns the row count; r[], and J[][6] when want_j.
template <int KIND>
__device__ __forceinline__ int residual_rows(const float* R, const float* t, const DevCam& cam, const float* p,
                                             float alpha, int phase, float* r, float (*J)[6], bool want_j) {
  const float X = p[2], Y = p[3], Z = p[4];
  float y[3];  // rotated model point
  y[0] = R[0] * X + R[1] * Y + R[2] * Z;
  y[1] = R[3] * X + R[4] * Y + R[5] * Z;
  y[2] = R[6] * X + R[7] * Y + R[8] * Z;
  const float wx = y[0] + t[0] - cam.tc[0], wy = y[1] + t[1] - cam.tc[1], wz = y[2] + t[2] - cam.tc[2];
  float c[3];  // camera-frame point
  c[0] = wx * cam.Rc[0] + wy * cam.Rc[3] + wz * cam.Rc[6];
  c[1] = wx * cam.Rc[1] + wy * cam.Rc[4] + wz * cam.Rc[7];
  c[2] = wx * cam.Rc[2] + wy * cam.Rc[5] + wz * cam.Rc[8];
  constexpr int NR = (KIND == 0 || KIND == 3) ? 2 : ((KIND == 1) ? 4 : 3);      // rows in phase 0
  const int nrows = (phase == 0) ? NR : ((KIND == 2) ? 3 : 2);   // the reference's row count in phase 1
  float w3 = 0.f, w2 = 1.f;
  if (KIND == 1 || KIND == 2) {
    w3 = (1.f - alpha) * p[8];
    w2 = 1.f - w3;
  }
  if (c[2] < 0.f || !(fabsf(c[2]) > 1e-9f)) {  // behind the camera: the reference's penalty, no gradient
    const float pen = -c[2] + 10.f;
    for (int a = 0; a < nrows; ++a) {
      float wgt = 1.f;
      if (KIND == 1) wgt = (a == nrows - 1) ? w3 : w2;
      if (KIND == 2) wgt = (a == 2) ? w3 : w2;
      r[a] = pen * ((phase == 0) ? sqrtf(wgt) : wgt);
      if (want_j)
        for (int k = 0; k < 6; ++k) J[a][k] = 0.f;
    }
    return nrows;
  }
  if (KIND == 0 || KIND == 2 || KIND == 3) {
    const float iz = 1.f / c[2];
    const float du = c[0] * iz * cam.K[0] + cam.K[2] - p[0];
    const float dv = c[1] * iz * cam.K[1] + cam.K[3] - p[1];
    const float gu[3] = {cam.K[0] * iz, 0.f, -cam.K[0] * c[0] * iz * iz};
    const float gv[3] = {0.f, cam.K[1] * iz, -cam.K[1] * c[1] * iz * iz};
    const float s0 = (phase == 0) ? sqrtf(w2) : w2;
    float g[3];
    if (phase == 0) {
      r[0] = s0 * du;
      r[1] = s0 * dv;
    } else {
      r[0] = s0 * du * du;
      r[1] = s0 * dv * dv;
    }
    if (want_j) {
      const float fu = (phase == 0) ? s0 : 2.f * s0 * du, fv = (phase == 0) ? s0 : 2.f * s0 * dv;
      for (int k = 0; k < 3; ++k) g[k] = fu * gu[k];
      row_from_grad(cam, y, g, J[0]);
      for (int k = 0; k < 3; ++k) g[k] = fv * gv[k];
      row_from_grad(cam, y, g, J[1]);
    }
    if (KIND == 2) {
      // depthError = | p (p.W) - p | = |p| |p.W - 1|  (…REPROJECTION_DEPTH_CPU.hpp:176-186), x50, weight3D
      const float* W = p + 5;
      const float pw = c[0] * W[0] + c[1] * W[1] + c[2] * W[2] - 1.f;
      const float n2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
      const float nn = sqrtf(n2);
      if (phase == 0) {
        const float s3 = sqrtf(50.f * w3);
        r[2] = s3 * nn * pw;
        if (want_j) {
          for (int k = 0; k < 3; ++k) g[k] = s3 * (c[k] / nn * pw + nn * W[k]);
          row_from_grad(cam, y, g, J[2]);
        }
      } else {
        r[2] = 50.f * w3 * n2 * pw * pw;
        if (want_j) {
          for (int k = 0; k < 3; ++k) g[k] = 50.f * w3 * (2.f * c[k] * pw * pw + n2 * 2.f * pw * W[k]);
          row_from_grad(cam, y, g, J[2]);
        }
      }
    }
    return nrows;
  }
  // KIND == 1: back-projection onto the ray through the depth-map point W
  {
    const float* W = p + 5;
    const float wn = sqrtf(W[0] * W[0] + W[1] * W[1] + W[2] * W[2]);
    const float n[3] = {W[0] / wn, W[1] / wn, W[2] / wn};
    const float sp = n[0] * c[0] + n[1] * c[1] + n[2] * c[2];
    const float e[3] = {c[0] - n[0] * sp, c[1] - n[1] * sp, c[2] - n[2] * sp};  // p - pHat
    const float ez = wn - sp;                                                     // along the ray
    float g[3];
    if (phase == 0) {
      const float s2 = sqrtf(w2), s3 = sqrtf(w3);
      for (int a = 0; a < 3; ++a) {
        r[a] = s2 * e[a];
        if (want_j) {
          for (int k = 0; k < 3; ++k) g[k] = s2 * ((a == k ? 1.f : 0.f) - n[a] * n[k]);
          row_from_grad(cam, y, g, J[a]);
        }
      }
      r[3] = s3 * ez;
      if (want_j) {
        for (int k = 0; k < 3; ++k) g[k] = -s3 * n[k];
        row_from_grad(cam, y, g, J[3]);
      }
    } else {
      r[0] = w2 * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);  // weight2D * dxy^2
      r[1] = w3 * ez * ez;                                    // weight3D * dz^2
      if (want_j) {
        for (int k = 0; k < 3; ++k) g[k] = 2.f * w2 * e[k];
        row_from_grad(cam, y, g, J[0]);
        for (int k = 0; k < 3; ++k) g[k] = -2.f * w3 * ez * n[k];
        row_from_grad(cam, y, g, J[1]);
      }
    }
    return nrows;
  }
}

// Wave-wide sum, the same value in every lane: four DPP steps inside the rows of 16 lanes
// (quad swaps, half-row mirror, row mirror), then the four row totals through scalar
// registers -- no LDS round trips (a __shfl_xor butterfly is six dependent ds_bpermutes).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);   // row_half_mirror
  v = dpp_add<0x140>(v);   // row_mirror
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}

// Sum of squared residuals over the inlier list at pose (R,t); all 64 lanes return it.
// KIND = the points' layout (PointStride, the camera of a point); RES = the residual model (default: KIND's own; 0 on a depth
// layout = plain reprojection residuals of the same points, the 2-D polish before the depth refine)
template <int KIND, int RES = KIND>
__device__ float lm_cost(const float* R, const float* t, const DevCam* cams, const float* pts,
                         const int* list, int n, float alpha, int phase, int lane) {
  constexpr int PS = PointStride<KIND>::value;
  float c = 0.f;
  for (int i = lane; i < n; i += 64) {
    float r[MAX_ROWS];
    const float* p = pts + PS * list[i];
    const int nr = residual_rows<RES>(R, t, cam_of<KIND>(cams, p), p, alpha, phase, r, nullptr, false);
    for (int a = 0; a < nr; ++a) c += r[a] * r[a];
  }
  return wave_sum(c);
}

// exp(omega) R
__device__ void rotate_left(const float* w, const float* R, float* out) {
  const float th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const float th = sqrtf(th2);
  float a, b;  // exp = I + a [w]x + b [w]x^2
  if (th < 1e-4f) {
    a = 1.f - th2 / 6.f;
    b = 0.5f - th2 / 24.f;
  } else {
    a = sinf(th) / th;
    b = (1.f - cosf(th)) / th2;
  }
  const float K[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
  float K2[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) K2[r * 3 + c] = K[r * 3] * K[c] + K[r * 3 + 1] * K[3 + c] + K[r * 3 + 2] * K[6 + c];
  float E[9];
  for (int i = 0; i < 9; ++i) E[i] = a * K[i] + b * K2[i];
  E[0] += 1.f;
  E[4] += 1.f;
  E[8] += 1.f;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out[r * 3 + c] = E[r * 3] * R[c] + E[r * 3 + 1] * R[3 + c] + E[r * 3 + 2] * R[6 + c];
}

// Solve (H + mu I) x = -g for symmetric positive H (6x6, upper triangle packed) by Cholesky.
__device__ bool solve6(const float* Hp, const float* g, float mu, float* x) {
  float A[6][6];
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      A[i][j] = Hp[k];
      A[j][i] = Hp[k];
      ++k;
    }
  for (int i = 0; i < 6; ++i) A[i][i] += mu;
  // L with the reciprocals of its diagonal kept beside it: one v_rsq per pivot, no divisions
  float Lm[6][6], inv[6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      float s = A[i][j];
      for (int m = 0; m < j; ++m) s -= Lm[i][m] * Lm[j][m];
      if (i == j) {
        if (!(s > 0.f)) return false;
        inv[i] = __frsqrt_rn(s);
        Lm[i][i] = s * inv[i];
      } else {
        Lm[i][j] = s * inv[j];
      }
    }
  float y[6];
  for (int i = 0; i < 6; ++i) {
    float s = -g[i];
    for (int m = 0; m < i; ++m) s -= Lm[i][m] * y[m];
    y[i] = s * inv[i];
  }
  for (int i = 5; i >= 0; --i) {
    float s = y[i];
    for (int m = i + 1; m < 6; ++m) s -= Lm[m][i] * x[m];
    x[i] = s * inv[i];
  }
  return true;
}

#ifdef POSE_PROF   // phase timing build (make EXTRA=-DPOSE_PROF): cycles of thread 0 per phase, summed over tasks;
__device__ unsigned long long g_pose_prof[32];   // [6] = LM iterations (accepted or not) over all refines;
// inside an iteration: [8] Jacobian pass, [9] the 27 wave sums, [10] solve + pose update, [11] cost of the new pose,
// [12] attempts, [13] the cost before the first iteration; [16 + 4 * phase + e] = refines of that phase that ended by
// e = 0 iteration cap, 1 eight rejections, 2 converged (or nothing promised), 3 zero gradient; [24 + phase] iterations, [26 + phase] attempts
#define LP_T(k) do { if (lane == 0) { const unsigned long long now_ = clock64(); atomicAdd(&g_pose_prof[k], now_ - t_lm); t_lm = now_; } } while (0)
#define LP_N(k) do { if (lane == 0) atomicAdd(&g_pose_prof[k], 1ull); } while (0)
#else
#define LP_T(k) do { } while (0)
#define LP_N(k) do { } while (0)
#endif

// Runs on one full wavefront; pose in/out is wave-uniform.
template <int KIND, int RES = KIND>
__device__ float lm_refine(float* R, float* t, const DevCam* cams, const float* pts, const int* list,
                           int n, float alpha, int phase, int iters, int lane) {
  constexpr int PS = PointStride<KIND>::value;
#ifdef POSE_PROF
  unsigned long long t_lm = clock64();
#endif
  float cost = lm_cost<KIND, RES>(R, t, cams, pts, list, n, alpha, phase, lane);
  LP_T(13);
  float mu = -1.f, nu = 2.f;
  // Phase 1's residuals are squares, r = d^2 (KIND 0 / 3: d = du, dv): the Hessian of sum r^2 = sum d^4 is
  // 12 d^2 grad d grad d^T, Gauss-Newton's 2 J^T J (J = 2 d grad d) is 8 d^2 grad d grad d^T -- two thirds of it, so every
  // Gauss-Newton step overshoots by half and the iteration converges linearly (error x -1/2 per step: 7.8 iterations
  // per refine measured, a third of them ending in eight rejected attempts).  With J^T J scaled by 3/2 the step is
  // Newton's for these residuals: same minimiser, quadratic convergence.  (The depth classes' phase-1 rows are sums of
  // squares of several terms; they keep Gauss-Newton.)
#ifndef LM_HS
#define LM_HS 1.5f
#endif
#ifndef LM_TOL0
#define LM_TOL0 1e-4f
#endif
#ifndef LM_TOL1
#define LM_TOL1 1e-5f
#endif
#ifndef LM_STOP
#define LM_STOP 1
#endif
#ifndef DRAW_MULHI
#define DRAW_MULHI 1
#endif
  const float hs = (phase == 1 && (RES == 0 || RES == 3)) ? LM_HS : 1.f;
  // Phase 0 only hands phase 1 a start: it stops at a relative gain of 1e-4 per step (1e-6 cost two more iterations per
  // refine and moved phase 1's end state by nothing); phase 1 at 1e-5, the resolution of its cost (below)
  // (the depth classes keep 1e-6 in both phases: their cost can hold a large constant, see the stop rule below)
  const float tol = (RES == 1 || RES == 2) ? 1e-6f : (phase == 0 ? LM_TOL0 : LM_TOL1);
  for (int it = 0; it < iters; ++it) {
#ifdef POSE_PROF
    if (lane == 0) atomicAdd(&g_pose_prof[6], 1ull);
    LP_N(24 + phase);
    t_lm = clock64();
#endif
    Accum acc;
    for (int i = 0; i < 21; ++i) acc.H[i] = 0.f;
    for (int i = 0; i < 6; ++i) acc.g[i] = 0.f;
    for (int i = lane; i < n; i += 64) {
      float r[MAX_ROWS], J[MAX_ROWS][6];
      const float* p = pts + PS * list[i];
      const int nr = residual_rows<RES>(R, t, cam_of<KIND>(cams, p), p, alpha, phase, r, J, true);
      for (int row = 0; row < nr; ++row) {
        int k = 0;
        for (int a = 0; a < 6; ++a) {
          for (int b = a; b < 6; ++b) acc.H[k++] += J[row][a] * J[row][b];
          acc.g[a] += J[row][a] * r[row];
        }
      }
    }
    LP_T(8);
    for (int i = 0; i < 21; ++i) acc.H[i] = hs * wave_sum(acc.H[i]);
    for (int i = 0; i < 6; ++i) acc.g[i] = wave_sum(acc.g[i]);
    LP_T(9);
    if (mu < 0.f) {  // tau * max diagonal (lm_core.c:672-676)
      float mx = 0.f;
      const int di[6] = {0, 6, 11, 15, 18, 20};
      for (int i = 0; i < 6; ++i) mx = fmaxf(mx, acc.H[di[i]]);
      mu = 1e-3f * mx;
    }
    float ginf = 0.f;
    for (int i = 0; i < 6; ++i) ginf = fmaxf(ginf, fabsf(acc.g[i]));
    if (!(ginf > 0.f)) { LP_N(16 + 4 * phase + 3); return cost; }
    bool accepted = false, converged = false;
    for (int attempt = 0; attempt < 8 && !accepted; ++attempt) {
      float dx[6];
#ifdef POSE_PROF
      if (lane == 0) atomicAdd(&g_pose_prof[12], 1ull);
      LP_N(26 + phase);
      t_lm = clock64();
#endif
      if (solve6(acc.H, acc.g, mu, dx)) {
        float Rn[9], tn[3];
        rotate_left(dx, R, Rn);
        for (int i = 0; i < 3; ++i) tn[i] = t[i] + dx[3 + i];
        LP_T(10);
        const float c2 = lm_cost<KIND, RES>(Rn, tn, cams, pts, list, n, alpha, phase, lane);
        LP_T(11);
        float dL = 0.f;
        for (int i = 0; i < 6; ++i) dL += dx[i] * (mu * dx[i] - acc.g[i]);
        const float dF = cost - c2;
        if (dF > 0.f && dL > 0.f) {
          for (int i = 0; i < 9; ++i) R[i] = Rn[i];
          for (int i = 0; i < 3; ++i) t[i] = tn[i];
          // nothing left at fp32 resolution: the next steps would only be rejected.  Only a step taken at the damping
          // the iteration started with says so: one that needed rejections first is short because mu grew, not
          // because the minimum is near (tests/tools/frame_stress.py: two objects in 600 frames stopped 1 degree short
          // of the optimum in a narrow valley, FILTER2 score 10-15% under the oracle's)
          converged = attempt == 0 && dF <= tol * cost;
          cost = c2;
          float tt = 2.f * dF / dL - 1.f;
          tt = 1.f - tt * tt * tt;
          mu *= fmaxf(tt, 1.f / 3.f);
          nu = 2.f;
          accepted = true;
          break;
        }
        // At the minimum, at fp32 resolution: the step of the damping this iteration started with was rejected, it is
        // itself below anything the 1 px bar could notice (2e-5 rad / 2e-5 m: 0.03 px at this geometry; at 2e-6, the pose's
        // own resolution, three refines in sixteen still burnt their eight attempts), and so is the UNDAMPED
        // Gauss-Newton step (one more 6 x 6 solve, once per refine).  Heavier damping only shortens the step: the seven
        // further attempts the loop used to make here -- a solve, a pose update and a pass over the points each, all
        // rejected -- were two thirds of a refine's time (POSE_PROF: 2.0 attempts per iteration, the last iteration of
        // most refines nothing but eight rejections).  The test is on the STEP, not on its promised gain relative to the
        // cost: a correspondence with a wrong depth attribute (or behind the camera) puts a constant into the depth
        // classes' cost that dwarfs everything the inliers can still gain, and a refine that stopped on "gain < 1e-6
        // of the cost" left frame 6 of the 50-model pool without one of its objects.
        auto tiny = [](const float* d) {
          float m = 0.f;
          for (int i = 0; i < 6; ++i) m = fmaxf(m, fabsf(d[i]));
          return m < 2e-5f;
        };
        if (LM_STOP && attempt == 0 && tiny(dx)) {
          float dx0[6], mxd = 0.f;
          const int di[6] = {0, 6, 11, 15, 18, 20};
          for (int i = 0; i < 6; ++i) mxd = fmaxf(mxd, acc.H[di[i]]);
          if (solve6(acc.H, acc.g, 1e-7f * mxd, dx0) && tiny(dx0)) {
            converged = true;
            break;
          }
        }
      }
      mu *= nu;
      nu *= 2.f;
    }
    if (converged) { LP_N(16 + 4 * phase + 2); return cost; }
    if (!accepted) { LP_N(16 + 4 * phase + 1); return cost; }
  }
  LP_N(16 + 4 * phase + 0);
  return cost;
}

#ifdef POSE_PROF
// (kept in registers and added to the totals only by a task that runs to the end of its refine: the totals then describe the
// tasks on the critical path of a launch, [7] counts them, [14] counts all tasks, [15] the cycles of those that ended early)
#define PP_T(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); pp_loc[k] += now_ - t_prof; t_prof = now_; } } while (0)
#else
#define PP_T(k) do { } while (0)
#endif

// The refine of one (cluster, replica) task on ONE wavefront: the winner's inliers, LM on plain then squared residuals,
// the inliers again, once more if the set changed; writes the slot's pose / count / error and marks it valid.  `pts`
// (PointStride<KIND> floats per cluster point) and `list` (k ints of scratch) may live in LDS or in global memory.
#ifdef POSE_PROF
#define PR_PROF_PARAMS , unsigned long long* pp_loc, unsigned long long* t_prof_p
#undef PP_T
#define PP_T(k) do { if (lane == 0 && pp_loc) { const unsigned long long now_ = clock64(); pp_loc[k] += now_ - *t_prof_p; *t_prof_p = now_; } } while (0)
#else
#define PR_PROF_PARAMS
#endif
template <int KIND>
__device__ void pose_refine(const float* pts, int* list, const int k, float* R, float* t, const int flags /* 1 near miss, 2 tight fit */,
                            const DevCam* cams, const mh_pose_params& prm, const float alpha, const int lane, const int slot,
                            float* __restrict__ obj_pose, int32_t* __restrict__ obj_ninl, float* __restrict__ obj_err,
                            int32_t* obj_valid, const FilterFuseArgs* __restrict__ fuse, const unsigned long long fa PR_PROF_PARAMS) {
  constexpr int PS = PointStride<KIND>::value;
  // The inliers of a pose, in point order, into list; `same` = the list already held exactly these points.
  auto collect = [&](bool& same, const float scale = 1.f) {
    int n = 0;
    bool eq = true;
    const float thr = prm.error_threshold * scale;
    for (int base = 0; base < k; base += 64) {
      const int i = base + lane;
      bool in = false;
      if (i < k) {
        const float* p = pts + PS * i;
        in = reproj_err2(R, t, cam_of<KIND>(cams, p), p[2], p[3], p[4], p[0], p[1]) < thr;
      }
      const unsigned long long m = __ballot(in);
      if (in) {
        const int at = n + __popcll(m & ((1ull << lane) - 1ull));
        if (list[at] != i) eq = false;
        list[at] = i;
      }
      n += __popcll(m);
    }
    same = __ballot(!eq) == 0ull;
    __builtin_amdgcn_wave_barrier();
    return n;
  };
  const bool near_miss = flags & 1, tight = flags & 2;
  bool same;
  int n_inl = collect(same);
  PP_T(3);
  const bool repass = prm.lm_iters_l2 >= 0;   // (launch_pose: MH_POSE_REPASS=0 hands the cap over negated = one pass, for A/B runs)
  int iters_l2 = prm.lm_iters_l2 >= 0 ? prm.lm_iters_l2 : -prm.lm_iters_l2;
  // The 2-iteration default of the plain phase is a warm start for a squared-residual phase that takes Newton steps
  // (LM_HS, RES 0 / 3).  The depth classes' second phase is plain Gauss-Newton at tol 1e-6: their first phase keeps the
  // ten iterations it had before that default came; 0 still means "no plain phase".
  if ((KIND == 1 || KIND == 2) && iters_l2 > 0 && iters_l2 < 10) iters_l2 = 10;
  // (a winner whose ranking count passed ":204" by a point that project()'s own arithmetic puts on the other side of
  //  the threshold is a near miss like any other)
  if (near_miss || n_inl <= prm.min_n_pts_object) {
    if (n_inl < 3) return;
    lm_refine<KIND>(R, t, cams, pts, list, n_inl, alpha, 0, 10, lane);   // (to convergence: the count is taken under this pose)
    n_inl = collect(same);
    if (n_inl <= prm.min_n_pts_object) {
      // Still short.  The reference's hypothesis is a least-squares fit of five or six points of the cluster whatever
      // they are (:194-199), and it tries hundreds of such samples: with ONE correspondence a few pixels off among four
      // or five good ones the fit gives way towards it, and a cluster that holds only MinNPtsObject points within the
      // threshold of the TRUE pose passes ":204" under the dragged one.  That is how the reference finds an object whose
      // matches mean shift has split into clusters too small to pass on their own (tests/tools/frame_stress.py scene
      // 241: 18 clean matches, none of its clusters with more than six of them) -- FILTER then hands the object all its
      // matches and POSE2 is not marginal at all.  The same here, for a near miss in a small cluster: samples of
      // n_pts_align points = all but one of them inliers of the near miss + one point within 25 thresholds (5 sigma) of it,
      // the fit from the near miss's pose, the strict count under the fitted pose; the first that passes is taken
      // (before the samples: the fit over all of those points at once).
      if (k > 64) return;                        // (wave-uniform; the slot stays invalid)
      bool in_strict = false, in_loose = false;
      if (lane < k) {
        const float* p = pts + PS * lane;
        const float e = reproj_err2(R, t, cam_of<KIND>(cams, p), p[2], p[3], p[4], p[0], p[1]);
        in_strict = e < prm.error_threshold;
        in_loose = e < 25.f * prm.error_threshold;   // (scene 241's correspondence lies 12 px off the true pose: 146 px^2 against the threshold's 10)
      }
      const unsigned long long S = __ballot(in_strict), X = __ballot(in_loose && !in_strict);
      const int nS = __popcll(S), nX = __popcll(X), want = prm.n_pts_align;
      if (nX == 0 || nS < want - 1 || want < 4 || want > 8) return;
      auto nth_bit = [](unsigned long long m, int n) {   // position of the n-th set bit (n < popcount)
        for (int i = 0; i < n; ++i) m &= m - 1ull;
        return __builtin_ctzll(m);
      };
      float R0[9], t0[3];
      for (int i = 0; i < 9; ++i) R0[i] = R[i];
      for (int i = 0; i < 3; ++i) t0[i] = t[i];
      bool found = false;
      const int attempts = 1 + min(16, nX * nS);
      for (int a0 = 0; a0 < attempts && !found; ++a0) {
        // attempt 0: the fit over ALL of them (what the reference's second optimizeCamera, :206, ends at); then the samples
        const int a = a0 - 1;
        int n_fit = want;
        if (a0 == 0) {
          n_fit = nS + nX;
          const unsigned long long all = S | X;
          if (lane < n_fit) list[lane] = nth_bit(all, lane);
        } else if (lane == 0) {
          list[0] = nth_bit(X, a % nX);
          for (int j = 0; j < want - 1; ++j) list[1 + j] = nth_bit(S, (a / nX + j) % nS);   // want - 1 consecutive inliers, rotating
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = 0; i < 9; ++i) R[i] = R0[i];
        for (int i = 0; i < 3; ++i) t[i] = t0[i];
        lm_refine<KIND, (KIND == 3 ? 3 : 0)>(R, t, cams, pts, list, n_fit, alpha, 0, 10, lane);
        n_inl = collect(same);
        found = n_inl > prm.min_n_pts_object;
      }
      if (!found) return;
    }
  }
  if (KIND == 1 || KIND == 2) {
    // The depth classes' refine works on 3-D residuals in metres, squared: ONE inlier whose depth attribute belongs to
    // something else (a clutter match 2.7 px from where the winner projects its model point, anywhere on its ray in
    // depth) outweighs the hundred true ones and drags the pose centimetres away -- and the winner is the hypothesis with
    // the MOST inliers, i.e. the one that reaches such borderline points (the reference takes the FIRST hypothesis over
    // MinNPtsObject, :204, an average one).  Frame 6 of the 50-model pool: 118 true inliers + 1, half of all (seed,
    // replica) results 12 px off where the oracle's are inside 1.6 px.  So the winner is first polished on plain
    // reprojection residuals -- the test its inliers are DEFINED by (:166-180) -- and the inliers are taken again
    // under the polished pose; the depth refine then starts from a least-squares pose and its own inlier set, like
    // the reference's does.
    lm_refine<KIND, 0>(R, t, cams, pts, list, n_inl, alpha, 0, 10, lane);   // (to convergence, whatever the plain phase's cap)
    n_inl = collect(same);
    if (n_inl <= prm.min_n_pts_object) return;   // (wave-uniform; the slot stays invalid)
  }
  lm_refine<KIND>(R, t, cams, pts, list, n_inl, alpha, 0, iters_l2, lane);
  PP_T(4);
  float err = lm_refine<KIND>(R, t, cams, pts, list, n_inl, alpha, 1, prm.lm_iters_l4, lane);
  // The reference refines on the inliers of a least-squares fit of 5-6 points (:199-207); a P3P pose of three noisy
  // points is a worse judge of which points belong to the object, so the inliers are taken again under the refined pose
  // and, if the set changed, the refine is repeated on it (once: tests/tools/frame_stress.py found objects whose FILTER2
  // score stayed 10-15% under the oracle's because a tenth of their points never entered the refine).
  // Odd replicas fit the inliers WELL INSIDE the threshold.  The refine minimises the squares of squared pixel errors
  // (:100-138): one correspondence that sits on the threshold outweighs twenty clean ones and bends the pose towards
  // itself -- every scene the stress tool lists with a FILTER2 score 7-15% under the oracle's is that picture: one match
  // with 5-15 px^2 under the planted pose that the max-consensus winner reaches, pulled to 3-4 px^2 while the clean
  // inliers' mean error goes from 0.34 to 0.57 px (profiles/r05_fixture_probe.txt).  The reference's four replicas per
  // cluster start from different random samples, some with that match among their inliers and some without, and FILTER
  // keeps the best-scoring of them (FILTER_PROJECTION_CPU.hpp:117-129); here all four end at the same consensus set, so
  // two of them are GIVEN the other choice: the last refine over the points within half the threshold.  The object still
  // needs more than MinNPtsObject inliers at the full threshold under the pose it ends with (:204).
  if (tight && repass) {
    float Rk[9], tk[3];
    for (int i = 0; i < 9; ++i) Rk[i] = R[i];
    for (int i = 0; i < 3; ++i) tk[i] = t[i];
    const int nk = n_inl;
    const float errk = err;
    const int nt = collect(same, 0.5f);
    bool kept = false;
    if (nt > prm.min_n_pts_object) {
      // (from the minimiser over the full set: the squared-residual phase alone, two to three Newton steps -- no warm start)
      err = lm_refine<KIND>(R, t, cams, pts, list, nt, alpha, 1, prm.lm_iters_l4, lane);
      const int nf = collect(same);
      if (nf > prm.min_n_pts_object) {
        n_inl = nf;
        kept = true;
      }
    }
    if (!kept) {
      for (int i = 0; i < 9; ++i) R[i] = Rk[i];
      for (int i = 0; i < 3; ++i) t[i] = tk[i];
      n_inl = nk;
      err = errk;
    }
  } else if (repass) {
    const int n0 = n_inl;
    const int n1 = collect(same);
    if (n1 > prm.min_n_pts_object && !(same && n1 == n0)) {
      n_inl = n1;
      lm_refine<KIND>(R, t, cams, pts, list, n_inl, alpha, 0, iters_l2, lane);
      err = lm_refine<KIND>(R, t, cams, pts, list, n_inl, alpha, 1, prm.lm_iters_l4, lane);
    } else if (!(same && n1 == n0)) {
      // too few points under the refined pose: keep the refined pose, report the first set's size
      n_inl = n0;
    }
  }
  PP_T(5);
  float q[4];
  rot_to_quat(R, q);
  if (lane == 0) {
    float* o = obj_pose + 7 * (size_t)slot;
    o[0] = q[0];
    o[1] = q[1];
    o[2] = q[2];
    o[3] = q[3];
    o[4] = t[0];
    o[5] = t[1];
    o[6] = t[2];
    obj_ninl[slot] = n_inl;
    obj_err[slot] = err;
    obj_valid[slot] = 1;
  }
  if (fuse) {
    // fused FILTER: the new object's F1 (score + keypoint claims, filter_dev.h) here, by the wavefront that made it -- the
    // frame's closing workgroup starts at F2
    FilterBuffers fb = fuse->fb;
    fb.corr = frame_ptr(fb.corr, fa);
    fb.m_rep = frame_ptr(fb.m_rep, fa);
    fb.model_off = frame_ptr(fb.model_off, fa);
    fb.obj_score = frame_ptr(fb.obj_score, fa);
    fb.best = frame_ptr(fb.best, fa);
    const int model = frame_ptr(fb.obj_model, fa)[slot];
    filter_score_wave(fb, cams[0], fuse->feature_distance, slot, model, q, t, lane);
  }
}
#ifdef POSE_PROF   // (back to the task-level form for pose_task below)
#undef PP_T
#define PP_T(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); pp_loc[k] += now_ - t_prof; t_prof = now_; } } while (0)
#endif

template <int KIND>
struct PoseLds {
  float pts[POSE_MAX_PTS * PointStride<KIND>::value];  // u,v,x,y,z[,wx,wy,wz,w] of the cluster
  int list[POSE_MAX_PTS];           // inlier list of the winner
  unsigned long long wave_best[POSE_THREADS / 64];
  float best_pose[12];
  int n_distinct;
  int n_inl;
  DevCam cams[MH_MAX_IMAGES];       // KIND 3: the frame's cameras
};

// One (cluster, replica) task, executed by a whole workgroup.  Every early exit is
// workgroup-uniform.
// `fa` = byte offset of the task's frame inside a batch's working arrays (FrameBatch): the pointers are the batch's
// first frame's and are shifted here, at the task's few uses of them, not in the kernel's loop over the frames (where
// thirty shifted pointers spilled).
// SPLIT: the task ends at its winning hypothesis (PoseHyp), the refine is pose_refine_kernel's -- the instantiation
// then carries no LM code at all
template <int KIND, bool SPLIT>
__device__ void pose_task(
    PoseLds<KIND>& L, const int cluster, const int replica, const unsigned long long fa,
    const mh_corr* __restrict__ corr0, const float4* __restrict__ depth0, float alpha,
    const int32_t* __restrict__ members0,
    const int32_t* __restrict__ cl_model0, const int32_t* __restrict__ cl_begin0,
    const int32_t* __restrict__ cl_count0, const DevCam& cam1, const DevCam* __restrict__ cam_table,
    const int32_t* __restrict__ img_of0, int n_images,
    const mh_pose_params& prm, uint64_t seed,
    const int obj_base, int max_objects,
    int32_t* __restrict__ obj_model0, float* __restrict__ obj_pose0, int32_t* __restrict__ obj_ninl0,
    float* __restrict__ obj_err0, int32_t* __restrict__ obj_cluster0, int32_t* obj_valid0,
    FrameCounts* counts0, PoseHyp* hyp_out0, const FilterFuseArgs* __restrict__ fuse) {
  PoseHyp* hyp_out = frame_ptr(hyp_out0, fa);
  const mh_corr* __restrict__ corr = frame_ptr(corr0, fa);
  const float4* __restrict__ depth = frame_ptr(depth0, fa);
  const int32_t* __restrict__ members = frame_ptr(members0, fa);
  const int32_t* __restrict__ cl_model = frame_ptr(cl_model0, fa);
  const int32_t* __restrict__ cl_begin = frame_ptr(cl_begin0, fa);
  const int32_t* __restrict__ cl_count = frame_ptr(cl_count0, fa);
  const int32_t* __restrict__ img_of = frame_ptr(img_of0, fa);
  int32_t* __restrict__ obj_model = frame_ptr(obj_model0, fa);
  float* __restrict__ obj_pose = frame_ptr(obj_pose0, fa);
  int32_t* __restrict__ obj_ninl = frame_ptr(obj_ninl0, fa);
  float* __restrict__ obj_err = frame_ptr(obj_err0, fa);
  int32_t* __restrict__ obj_cluster = frame_ptr(obj_cluster0, fa);
  int32_t* obj_valid = frame_ptr(obj_valid0, fa);
  FrameCounts* counts = frame_ptr(counts0, fa);
  constexpr int PS = PointStride<KIND>::value;
  const int R_ = prm.max_objects_per_cluster;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = obj_base + cluster * R_ + replica;
  if (slot >= max_objects) {
    if (tid == 0) atomicOr(&counts->error, ERR_OBJECT_CAP);
    return;
  }
#ifdef POSE_PROF
  unsigned long long t_prof = clock64();
  unsigned long long pp_loc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long t_task0 = t_prof;
  if (threadIdx.x == 0) atomicAdd(&g_pose_prof[14], 1ull);
  struct PpEnd {   // a task that leaves early: its cycles into [15]
    unsigned long long t0; bool done = false;
    __device__ ~PpEnd() { if (threadIdx.x == 0 && !done) atomicAdd(&g_pose_prof[15], clock64() - t0); }
  } pp_end{t_task0};
#endif
  int k = cl_count[cluster];
  const int begin = cl_begin[cluster];
  if (k > POSE_MAX_PTS) {
    if (tid == 0) atomicOr(&counts->error, ERR_POSE_CAP);
    k = POSE_MAX_PTS;
  }
  // the camera(s): one (KIND 0..2) or the frame's table in LDS (KIND 3)
  const DevCam* cams = &cam1;
  if (KIND == 3) {
    for (int i = tid; i < n_images * (int)(sizeof(DevCam) / 4); i += POSE_THREADS)
      reinterpret_cast<float*>(L.cams)[i] = reinterpret_cast<const float*>(cam_table)[i];
    cams = L.cams;
  }
  for (int i = tid; i < k; i += POSE_THREADS) {
    const int mi = members[begin + i];
    const mh_corr c = corr[mi];
    float* p = L.pts + PS * i;
    p[0] = c.u;
    p[1] = c.v;
    p[2] = c.x;
    p[3] = c.y;
    p[4] = c.z;
    if (KIND == 3) p[5] = __int_as_float(img_of[mi]);
    if (KIND == 1 || KIND == 2) {
      const float4 d = depth[mi];
      p[5] = d.x;
      p[6] = d.y;
      p[7] = d.z;
      p[8] = d.w;
    }
  }
  if (tid == 0) {
    L.n_distinct = 0;
    L.n_inl = 0;
  }
  __syncthreads();
  // randSample needs n_pts_align correspondences with distinct image coordinates (:76-98):
  // point i is a duplicate if an earlier point has its coordinates; the comparisons of one
  // point are split over POSE_THREADS / k threads (L.list serves as the duplicate flags)
  {
    for (int i = tid; i < k; i += POSE_THREADS) L.list[i] = 0;
    __syncthreads();
    int per = POSE_THREADS / (k > 0 ? k : 1);
    per = per < 1 ? 1 : (per > 32 ? 32 : per);
    for (int w = tid; w < k * per; w += POSE_THREADS) {
      const int i = w / per, part = w - i * per;
      const float ui = L.pts[PS * i], vi = L.pts[PS * i + 1];
      bool dup = false;
      for (int j = part; j < i; j += per)
        dup |= (L.pts[PS * j] == ui) & (L.pts[PS * j + 1] == vi) &
               (KIND != 3 || __float_as_int(L.pts[PS * j + 5]) == __float_as_int(L.pts[PS * i + 5]));
      if (dup) L.list[i] = 1;
    }
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < k; i += POSE_THREADS) mine += L.list[i] == 0;
    if (mine) atomicAdd(&L.n_distinct, mine);
  }
  __syncthreads();
  const bool enough = L.n_distinct >= prm.n_pts_align && L.n_distinct >= 4;
  if (tid == 0) {
    obj_valid[slot] = 0;
    obj_cluster[slot] = cluster;
    obj_model[slot] = cl_model[cluster];
    if (hyp_out) hyp_out[slot].n_best = 0;   // (no winner unless the task gets that far)
  }
  if (!enough) return;
  PP_T(0);

  // ---- hypotheses: one per lane ------------------------------------------------------
  // n_hypotheses < 0: that many, all of them (no adaptive stop after the first 256)
  const bool always_all = prm.n_hypotheses < 0;
  int H = prm.n_hypotheses > 0 ? prm.n_hypotheses : (always_all ? -prm.n_hypotheses : 1024);
  // A small cluster has fewer distinct samples than that: C(k, 3) triples x (k - 3) points to pick the root with -- 140
  // for the 7 points mean shift's MinPts lets through, 280, 504, 840 for 8, 9, 10.  Those are ENUMERATED, every one
  // once, instead of H random draws with repetitions: nothing a random draw could find is missed, and the clusters of
  // clutter -- which never settle and so ran all 1024 -- take one to four rounds of 256 instead of four.
  const int n_enum = (k >= 4 && k <= 10) ? k * (k - 1) * (k - 2) / 6 * (k - 3) : 0;
  const bool enumerate = n_enum > 0 && n_enum <= H;
  if (enumerate) H = n_enum;
  unsigned long long best_key = 0ull;  // (inliers << 32) | ~hypothesis id
  Pose34 best_pose;
  // The random stream of a task is keyed by (model id, number of the cluster among its model's
  // clusters, replica) and not by the cluster's row in this context's table: a rank that holds a
  // shard of the models numbers its rows differently, its objects must still be the ones the
  // single-context frame reports (model ids are global, a model's clusters keep their order).
  int ordinal = 0;
  {
    const int model = cl_model[cluster];
    while (ordinal < cluster && cl_model[cluster - ordinal - 1] == model) ++ordinal;
  }
  const uint64_t task_key = seed ^ ((uint64_t)(cl_model[cluster] + 1) << 48) ^
                            ((uint64_t)(ordinal + 1) << 36) ^ ((uint64_t)(replica + 1) << 32);
  auto hypothesis = [&](const int h) {
    uint64_t st = task_key ^ (uint64_t)h;
    splitmix64(st);
    // 4 correspondences with pairwise distinct image coordinates (:76-98)
    int i0 = -1, i1 = -1, i2 = -1, i3 = -1;
    // (several cameras: P3P needs its three points and the one that picks the root in ONE image -- the
    // image of the first draw; a candidate from another image counts like a repeated coordinate)
    auto same_uv = [&](int a, int b) {
      if (KIND == 3 && __float_as_int(L.pts[PS * a + 5]) != __float_as_int(L.pts[PS * b + 5])) return true;
      return L.pts[PS * a] == L.pts[PS * b] && L.pts[PS * a + 1] == L.pts[PS * b + 1];
    };
    // uniform in [0, k): the high word of (upper 32 random bits) x k -- one v_mul_hi_u32 (a 64-bit `% k` by a runtime k is
    // a ~100-instruction sequence, four to seven times per hypothesis)
    auto draw = [&](uint64_t& s_) { return DRAW_MULHI ? (int)(((splitmix64(s_) >> 32) * (uint64_t)(unsigned)k) >> 32) : (int)(splitmix64(s_) % (uint64_t)k); };
    if (enumerate) {
      // h = triple * (k - 3) + f: the triple by its rank among the C(k, 3) in lexicographic order, the fourth point the
      // f-th of the others
      int tr = h / (k - 3);
      const int f = h - tr * (k - 3);
      int a = 0;
      for (; a < k - 2; ++a) {
        const int c = (k - 1 - a) * (k - 2 - a) / 2;   // triples that start at a
        if (tr < c) break;
        tr -= c;
      }
      int b = a + 1;
      for (; b < k - 1; ++b) {
        const int c = k - 1 - b;                       // triples (a, b, *)
        if (tr < c) break;
        tr -= c;
      }
      const int c3 = b + 1 + tr;
      int d = f;                                       // the f-th index outside {a, b, c3}
      if (d >= a) ++d;
      if (d >= b) ++d;
      if (d >= c3) ++d;
      if (same_uv(a, b) || same_uv(a, c3) || same_uv(b, c3) || same_uv(d, a) || same_uv(d, b) || same_uv(d, c3)) return;
      i0 = a; i1 = b; i2 = c3; i3 = d;
    } else {
    i0 = draw(st);
    for (int tries = 0; tries < 16 && i1 < 0; ++tries) {
      const int c = draw(st);
      if (!same_uv(c, i0)) i1 = c;
    }
    if (i1 < 0) return;
    for (int tries = 0; tries < 16 && i2 < 0; ++tries) {
      const int c = draw(st);
      if (!same_uv(c, i0) && !same_uv(c, i1)) i2 = c;
    }
    if (i2 < 0) return;
    for (int tries = 0; tries < 16 && i3 < 0; ++tries) {
      const int c = draw(st);
      if (!same_uv(c, i0) && !same_uv(c, i1) && !same_uv(c, i2)) i3 = c;
    }
    if (i3 < 0) return;
    }
    const DevCam& cam = cam_of<KIND>(cams, L.pts + PS * i0);   // the sample's camera
    // disambiguate the P3P roots with the 4th point
    Pose34 cand;
    float best4 = __builtin_inff();
    bool have = false;
    const float* p4 = L.pts + PS * i3;
#if MH_P3P_FP64
    double X[3][3], y[3][3];
    auto load = [&](int s, int pi) {
      const float* p = L.pts + PS * pi;
      X[s][0] = p[2];
      X[s][1] = p[3];
      X[s][2] = p[4];
      y[s][0] = ((double)p[0] - cam.K[2]) / cam.K[0];
      y[s][1] = ((double)p[1] - cam.K[3]) / cam.K[1];
      y[s][2] = 1.0;
      normalize3(y[s]);
    };
    load(0, i0);
    load(1, i1);
    load(2, i2);
    p3p(X, y, [&](const double* Rc_, const double* tc_) {
      Pose34 w;
      to_world(cam, Rc_, tc_, w);
      const float e = reproj_err2(w.r, w.t, cam, p4[2], p4[3], p4[4], p4[0], p4[1]);
      if (e < best4) {
        best4 = e;
        cand = w;
        have = true;
      }
    });
#else
    float X[3][3], y[3][3];
    const float ifx = rcp_f(cam.K[0]), ify = rcp_f(cam.K[1]);
    auto load = [&](int s, int pi) {
      const float* p = L.pts + PS * pi;
      X[s][0] = p[2];
      X[s][1] = p[3];
      X[s][2] = p[4];
      const float bx = (p[0] - cam.K[2]) * ifx, by = (p[1] - cam.K[3]) * ify;
      const float inv = __builtin_amdgcn_rsqf(bx * bx + by * by + 1.0f);
      y[s][0] = bx * inv;
      y[s][1] = by * inv;
      y[s][2] = inv;
    };
    load(0, i0);
    load(1, i1);
    load(2, i2);
    p3p_f(X, y, [&](const float* Rc_, const float* tc_) {
      Pose34 w;
      to_world_f(cam, Rc_, tc_, w);
      const float e = reproj_err2(w.r, w.t, cam, p4[2], p4[3], p4[4], p4[0], p4[1]);
      if (e < best4) {
        best4 = e;
        cand = w;
        have = true;
      }
    });
#endif
    if (!have) return;
    int cnt = 0;
    if (KIND == 3 || MH_P3P_FP64) {   // (every point in its own camera: the two transforms of project() as they are)
      for (int i = 0; i < k; ++i) {
        const float* p = L.pts + PS * i;
        cnt += reproj_err2(cand.r, cand.t, cam_of<KIND>(cams, p), p[2], p[3], p[4], p[0], p[1]) < prm.error_threshold;
      }
    } else {
      // The count that RANKS the hypotheses: 256 lanes x k points were 45 k of a round's 63 k cycles in project()'s own
      // arithmetic (two unfused 3 x 3 transforms, two IEEE divisions, a double compare: ~75 instructions per point).  Here
      // the pose is composed with the camera once per hypothesis (M = Rc^T R, m = Rc^T (t - tc)) and a point costs nine
      // multiply-adds, one v_rcp_f32 and the pixel arithmetic: ~20.  The values differ from project()'s in the last
      // bits, so a point ON the threshold can count differently -- which only moves which of two near-equal hypotheses
      // wins; every decision the reference takes by project() (the inlier set the refine works on, ":204"'s more than
      // MinNPtsObject, FILTER's scores) is taken by project()'s exact arithmetic in pose_refine.
      const float* Rc = cam.Rc;
      float M[9], m[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) M[r * 3 + c] = Rc[0 + r] * cand.r[0 + c] + Rc[3 + r] * cand.r[3 + c] + Rc[6 + r] * cand.r[6 + c];
        m[r] = Rc[0 + r] * (cand.t[0] - cam.tc[0]) + Rc[3 + r] * (cand.t[1] - cam.tc[1]) + Rc[6 + r] * (cand.t[2] - cam.tc[2]);
      }
      const float fx = cam.K[0], fy = cam.K[1], cx0 = cam.K[2], cy0 = cam.K[3], thr = prm.error_threshold;
      for (int i = 0; i < k; ++i) {
        const float* p = L.pts + PS * i;
        const float X_ = p[2], Y_ = p[3], Z_ = p[4];
        const float cz = fmaf(M[6], X_, fmaf(M[7], Y_, fmaf(M[8], Z_, m[2])));
        const float cx = fmaf(M[0], X_, fmaf(M[1], Y_, fmaf(M[2], Z_, m[0])));
        const float cy = fmaf(M[3], X_, fmaf(M[4], Y_, fmaf(M[5], Z_, m[1])));
        const float iz = rcp_f(cz);
        const float du = fmaf(cx * iz, fx, cx0) - p[0], dv = fmaf(cy * iz, fy, cy0) - p[1];
        cnt += (cz >= 0.001f) & (fmaf(du, du, dv * dv) < thr);
      }
    }
    const unsigned long long key = ((unsigned long long)cnt << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)h);
    if (key > best_key) {
      best_key = key;
      best_pose = cand;
    }
  
  };
  // Two stages: the first 256 hypotheses run on four wavefronts (one per SIMD); the other
  // n_hypotheses - 256 only if, with the inlier ratio w the best of them reached, another
  // all-inlier sample is still likely to exist: (1 - w^4)^256 >= 1e-3 (the usual adaptive
  // RANSAC bound; 4 = the P3P sample plus the point that picks its root).
  auto wg_argmax = [&]() -> unsigned long long {
    unsigned long long wkey = best_key;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(wkey, off);
      wkey = o > wkey ? o : wkey;
    }
    if (lane == 0) L.wave_best[wave] = wkey;
    __syncthreads();
    unsigned long long g = 0ull;
    for (int w = 0; w < POSE_THREADS / 64; ++w) g = L.wave_best[w] > g ? L.wave_best[w] : g;
    return g;
  };
  const int HA = H < 256 ? H : 256;
  for (int h = tid; h < HA; h += POSE_THREADS) hypothesis(h);
  PP_T(1);
  unsigned long long gkey = wg_argmax();
  if (H <= HA && tid == 0) {
    atomicAdd(&counts->n_hyp, HA);
    atomicAdd(&counts->n_pose_tasks, 1);
  }
  if (H > HA) {
    const int cnt_a = (int)(gkey >> 32);
    const float w = (float)cnt_a / (float)k;
    const float w4 = w * w * w * w;
    const bool settled = !always_all && cnt_a > prm.min_n_pts_object && powf(1.f - w4, (float)HA) < 1e-3f;
    if (tid == 0) {   // what was evaluated (bench.py reports it)
      atomicAdd(&counts->n_hyp, settled ? HA : H);
      atomicAdd(&counts->n_pose_tasks, 1);
    }
    if (!settled) {
      for (int h = HA + tid; h < H; h += POSE_THREADS) hypothesis(h);
      __syncthreads();   // everybody has read wave_best of the first stage
      gkey = wg_argmax();
    }
  }
  const int best_cnt = (int)(gkey >> 32);
  // needs MORE than MinNPtsObject inliers (:204).  A winner that falls short by one or two is not given up yet: the
  // reference's hypothesis is a least-squares fit of n_pts_align (5) points, a P3P pose of three noisy points is a
  // cruder judge of a small cluster -- with 7 inliers among 9 points, all 7 required, no triple's pose may reach every
  // one of them (tests/tools/frame_stress.py scene 241: the oracle finds the object with every seed, the P3P stage with
  // none of 6 x 4 x 1024 hypotheses).  Such a near miss gets the local optimisation the reference's hypothesis has built
  // in: a plain-residual refine on the inliers it has (at least n_pts_align of them), then the count again.
  const bool near_miss = best_cnt <= prm.min_n_pts_object && best_cnt >= prm.n_pts_align && best_cnt >= 5 &&
                         best_cnt + 2 > prm.min_n_pts_object;
  if (best_cnt <= prm.min_n_pts_object && !near_miss) return;
  if (best_key == gkey) {                        // unique: the key embeds the hypothesis id
    for (int i = 0; i < 9; ++i) L.best_pose[i] = best_pose.r[i];
    for (int i = 0; i < 3; ++i) L.best_pose[9 + i] = best_pose.t[i];
  }
  __syncthreads();

  PP_T(2);
  if (SPLIT) {
    // split launch (PoseSplit): the winner goes to memory, the refine is pose_refine_kernel's -- one wavefront per task
    // there, several tasks per compute unit, instead of this workgroup's half compute unit held by one wavefront of four
    if (tid < 12) hyp_out[slot].pose[tid] = L.best_pose[tid];
    if (tid == 0) {
      hyp_out[slot].n_best = best_cnt;
      hyp_out[slot].flags = (near_miss ? 1 : 0) | ((replica & 1) << 1);   // (bit 1: the replica fits the inliers well inside the threshold, pose_refine)
    }
    return;
  }
  if (SPLIT) return;   // (not reached)
  // ---- refine on the winner's inliers, wavefront 0 ----------------------------------------
  if (wave != 0) return;
  float R[9], t[3];
  for (int i = 0; i < 9; ++i) R[i] = L.best_pose[i];
  for (int i = 0; i < 3; ++i) t[i] = L.best_pose[9 + i];
#ifdef POSE_PROF
  pose_refine<KIND>(L.pts, L.list, k, R, t, (near_miss ? 1 : 0) | ((replica & 1) << 1), cams, prm, alpha, lane, slot, obj_pose, obj_ninl, obj_err, obj_valid,
                    fuse, fa, pp_loc, &t_prof);
  if (threadIdx.x == 0 && obj_valid[slot]) {
    for (int i = 0; i < 6; ++i) atomicAdd(&g_pose_prof[i], pp_loc[i]);
    atomicAdd(&g_pose_prof[7], 1ull);
    pp_end.done = true;
  }
#else
  pose_refine<KIND>(L.pts, L.list, k, R, t, (near_miss ? 1 : 0) | ((replica & 1) << 1), cams, prm, alpha, lane, slot, obj_pose, obj_ninl, obj_err, obj_valid,
                    fuse, fa);
#endif
}

// The launch: a 1-D grid of workgroups shares ALL (cluster, replica) tasks of the launch -- of one frame, or of the B
// frames of a batch (FrameBatch): the tasks are numbered through the frames and dealt round-robin, task t to workgroup
// t mod gridDim.x.  The cluster counts live on the device, so the grid is a guess (PoseTail::grid; the library feeds it
// from what earlier launches found, PoseTail::feedback); every workgroup of this kernel needs half a compute unit to
// start, idle or not -- with one grid row per frame (round 2) a batch of eight frames launched 256 workgroups for its
// ~64 tasks, and a rank that owns a tenth of the models launched as many for a handful.
// In a frame the workgroup that finishes the frame's LAST task advances the object-slot count past this launch's slots,
// counts the valid objects and -- fused FILTER -- runs the step that follows (a frame without tasks: workgroup
// frame mod gridDim.x does).
constexpr int POSE_GRID = 160;

// What the workgroup that finishes a frame's LAST task does (all POSE_THREADS threads): the object-slot count past this
// launch's slots, the count of valid objects, the feedback word -- and, fused FILTER, the step that follows.
// The frame's buffers as the fused FILTER step sees them (arena of frame f applied)
__device__ __forceinline__ FilterBuffers fused_filter_buffers(const FilterFuseArgs* __restrict__ fuse_args, const unsigned long long a) {
  FilterBuffers ffb = fuse_args->fb;
  ffb.corr = frame_ptr(ffb.corr, a); ffb.m_rep = frame_ptr(ffb.m_rep, a); ffb.model_off = frame_ptr(ffb.model_off, a);
  ffb.obj_model = frame_ptr(ffb.obj_model, a); ffb.obj_pose = frame_ptr(ffb.obj_pose, a);
  ffb.obj_score = frame_ptr(ffb.obj_score, a); ffb.obj_score_raw = frame_ptr(ffb.obj_score_raw, a);
  ffb.obj_valid = frame_ptr(ffb.obj_valid, a); ffb.obj_npts = frame_ptr(ffb.obj_npts, a);
  ffb.best = frame_ptr(ffb.best, a); ffb.obj_clsize = frame_ptr(ffb.obj_clsize, a);
  ffb.new_members = frame_ptr(ffb.new_members, a); ffb.cl_model = frame_ptr(ffb.cl_model, a);
  ffb.cl_begin = frame_ptr(ffb.cl_begin, a); ffb.cl_count = frame_ptr(ffb.cl_count, a);
  return ffb;
}

// F1 -- score and keypoint claims -- of the objects a frame held BEFORE the launch (FILTER2: the POSE objects FILTER kept,
// slots [0, obj_base)): they compete with their re-estimates like the reference's list does (POSE2 appends to
// frameData.objects, ...REPROJECTION_CPU.hpp:299; FILTER_PROJECTION_CPU.hpp:96 scores every object).  Dealt over the
// launch's workgroups at their START, four slots per workgroup (one wavefront each): slot o belongs to workgroup
// (o / 4) mod G.  Returns how many slots this workgroup took.  (Round 5, first form: the closing workgroup did them all
// at the end -- ten kept objects were three rounds of ~10 us behind everything else, 0.74 -> 0.77 ms for a frame alone.)
__device__ __forceinline__ int pose_kept_f1(const FilterFuseArgs* __restrict__ fuse_args, const unsigned long long a, const int obj_base,
                                            const int max_objects, const DevCam& cam) {
  const int n_old = obj_base < max_objects ? obj_base : max_objects;
  const int G = (int)gridDim.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NW = POSE_THREADS / 64;
  int mine = 0;
  FilterBuffers ffb;
  bool have = false;
  for (int o0 = (int)blockIdx.x * NW; o0 < n_old; o0 += G * NW) {
    if (!have) {
      ffb = fused_filter_buffers(fuse_args, a);
      have = true;
    }
    mine += min(NW, n_old - o0);
    const int o = o0 + wave;
    if (o < n_old && ffb.obj_valid[o])   // (wave-uniform)
      filter_score_wave(ffb, cam, fuse_args->feature_distance, o, ffb.obj_model[o], ffb.obj_pose + 7 * (size_t)o,
                        ffb.obj_pose + 7 * (size_t)o + 4, lane);
  }
  return mine;
}

__device__ void pose_close_frame(const int f, const unsigned long long a, const int n_tasks, const int obj_base,
                                 const int max_objects, int32_t* obj_valid0, FrameCounts* counts0, const PoseTail& tail0,
                                 const FilterFuseArgs* __restrict__ fuse_args, const DevCam& cam, const FrameBatch& fbx) {
  int n_slots = obj_base + n_tasks;
  if (n_slots > max_objects) n_slots = max_objects;
  if (tail0.feedback && threadIdx.x == 0) tail0.feedback[f] = n_tasks;   // what the next launches size their grids by
  if (tail0.snap_valid) {
    const int32_t* obj_valid = frame_ptr(obj_valid0, a);
    int c = 0;
    for (int i = threadIdx.x; i < n_slots; i += POSE_THREADS) c += obj_valid[i] != 0;
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    __shared__ int wave_c[POSE_THREADS / 64];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wave_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < POSE_THREADS / 64; ++w) tot += wave_c[w];
      tail0.snap_valid[4 * f] = tot;
    }
  }
  int32_t* n_slots_dev = frame_ptr(tail0.n_slots, a);
  if (threadIdx.x == 0) *n_slots_dev = n_slots;
  if (fuse_args) {
    // the frame's FILTER step, here instead of in a launch of its own: this workgroup scores the objects the frame
    // held before the launch, then F2..F4 and (FILTER2) the result block
    FilterBuffers ffb = fused_filter_buffers(fuse_args, a);
    FilterTail ftail = fuse_args->tail;
    ftail.ticket = frame_ptr(ftail.ticket, a);
    if (ftail.snap_kept) ftail.snap_kept += 4 * f;
    if (ftail.snap_all) ftail.snap_all += 4 * f;
    ftail.result = frame_ptr(ftail.result, (unsigned long long)f * fbx.result_bytes);
    __shared__ FilterLds FS;
    __syncthreads();
    // F1 -- an object's score and its claims -- of THIS launch's objects was run by the wavefront that refined each of
    // them (pose_refine), that of the objects the frame already held by the launch's workgroups at their start
    // (pose_kept_f1): every one of them is counted in the frame's ticket, or ran in the launch before this one.
    filter_finish(FS, ffb, fuse_args->min_points, fuse_args->min_score, n_slots, n_slots_dev,
                  frame_ptr(fuse_args->n_clusters_dev, a), frame_ptr(counts0, a), ftail);
    __syncthreads();   // (FS and the task's LDS are reused by this workgroup's next frame)
  }
}

template <int KIND, bool SPLIT>
__global__ __launch_bounds__(POSE_THREADS, SPLIT ? MH_POSE_SPLIT_WAVES : MH_POSE_MIN_WAVES) void pose_kernel(
    const mh_corr* __restrict__ corr0, const float4* __restrict__ depth0, float alpha,
    const int32_t* __restrict__ members0,
    const int32_t* __restrict__ cl_model0, const int32_t* __restrict__ cl_begin0,
    const int32_t* __restrict__ cl_count0, const int32_t* __restrict__ n_clusters_dev0, DevCam cam,
    const DevCam* __restrict__ cam_table, const int32_t* __restrict__ img_of0, int n_images,
    mh_pose_params prm, uint64_t seed0,
    const int32_t* obj_base_dev0, int max_objects,
    int32_t* __restrict__ obj_model0, float* __restrict__ obj_pose0, int32_t* __restrict__ obj_ninl0,
    float* __restrict__ obj_err0, int32_t* __restrict__ obj_cluster0, int32_t* obj_valid0,
    FrameCounts* counts0, PoseTail tail0, const FilterFuseArgs* __restrict__ fuse_args, FrameBatch fbx,
    PoseHyp* hyp0 /* split launch: winners go here, pose_refine_kernel refines and closes the frames */) {
  static_assert(POSE_THREADS == FT, "the fused FILTER runs on the POSE workgroup's threads");
  MH_TRACE_SCOPE(mh::TK_POSE);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  PoseLds<KIND>& L = *reinterpret_cast<PoseLds<KIND>*>(smem);
  const int R_ = prm.max_objects_per_cluster;
  const int G = (int)gridDim.x;
  const int n_frames = fbx.n > 1 ? fbx.n : 1;
  int rank_base = 0;   // (mod G) tasks of the frames before the current one
  // Every frame's task count and first object slot, fetched by one thread per frame in ONE round trip: both are final
  // when the launch starts (the cluster count of POSE2 is FILTER's, written by the launch before this one).  Walking the
  // frames with two dependent global loads each kept every workgroup of a 32-frame batch resident for ~50 us it spent
  // on nothing else -- on half a compute unit that no MATCH workgroup could use meanwhile.
  __shared__ int fr_tasks[MH_MAX_BATCH], fr_obj_base[MH_MAX_BATCH];
  if ((int)threadIdx.x < n_frames) {
    const unsigned long long a = (unsigned long long)threadIdx.x * fbx.arena;
    fr_tasks[threadIdx.x] = *frame_ptr(n_clusters_dev0, a) * R_;
    fr_obj_base[threadIdx.x] = obj_base_dev0 ? *frame_ptr(obj_base_dev0, a) : 0;
  }
  __syncthreads();
  for (int f = 0; f < n_frames; ++f) {
    // frame f of a batch: its copy of the working arrays, its counts snapshot and result block
    const unsigned long long a = (unsigned long long)f * fbx.arena;
    unsigned int* ticket = SPLIT ? nullptr : frame_ptr(tail0.ticket, a);
    const uint64_t seed = fbx.n > 1 ? fbx.seed[f] : seed0;
    const int n_tasks = fr_tasks[f];
    const int obj_base = fr_obj_base[f];
    bool last = false;
    // fused FILTER2: the kept objects' F1s, dealt over the workgroups (a split launch: the refine kernel closes the
    // frames after this kernel has ended; one launch: they count in the frame's ticket like tasks)
    const int n_kept = (fuse_args && n_tasks > 0) ? (obj_base < max_objects ? obj_base : max_objects) : 0;
    if (n_kept > 0) {
      const int took = pose_kept_f1(fuse_args, a, obj_base, max_objects, cam);
      if (ticket && took > 0 && frame_work_done(ticket, (unsigned)took, (unsigned)(n_tasks + n_kept))) last = true;
    }
    int first = ((int)blockIdx.x - rank_base) % G;
    if (first < 0) first += G;
    for (int task = first; task < n_tasks; task += G) {
      pose_task<KIND, SPLIT>(L, task / R_, task % R_, a, corr0, depth0, alpha, members0, cl_model0, cl_begin0, cl_count0, cam,
                      cam_table, img_of0, n_images, prm, seed, obj_base, max_objects, obj_model0, obj_pose0, obj_ninl0,
                      obj_err0, obj_cluster0, obj_valid0, counts0, hyp0, SPLIT ? nullptr : fuse_args);
      __syncthreads();  // LDS is reused by the next task
      if (ticket && frame_work_done(ticket, 1u, (unsigned)(n_tasks + n_kept))) last = true;
    }
    rank_base = (rank_base + n_tasks) % G;
    if (ticket && n_tasks == 0) last = (int)blockIdx.x == f % G;   // nobody has a task here: one workgroup still closes the frame
    if (!last) continue;   // (uniform over the workgroup)
    pose_close_frame(f, a, n_tasks, obj_base, max_objects, obj_valid0, counts0, tail0, fuse_args, cam, fbx);
  }
}

// ---- the refine of a split launch -----------------------------------------------------------------------------------
// One wavefront per task, four tasks per workgroup at a time, RCAP cluster points of each cached in LDS (larger clusters
// work out of the frame's global scratch): a refine needs a sixteenth of a compute unit's wavefront slots where the fused
// kernel holds half a unit -- 255 registers x 4 wavefronts, three of them gone, 52 KB of LDS -- for the 55% of a task's
// time that one wavefront refines.  The tasks of the batch are numbered through the frames like pose_kernel's; workgroup b
// takes tasks 4 (round G + b) + wave in round `round`.  After every round the workgroup counts its finished tasks into
// the frames' tickets; a frame whose last task that was is closed by this workgroup (pose_close_frame, all threads).
// (plain frames: 512 points -- 12 KB per wavefront, three workgroups per compute unit; the reference's own test frame has a
// cluster of 330 matches, and a cluster that does not fit refines out of global scratch at a tenth of the speed: 1.4 ms)
template <int KIND> struct RefineCap { static constexpr int value = KIND == 0 ? 512 : 320; };
template <int KIND>
struct RefineLds {
  float pts[POSE_THREADS / 64][RefineCap<KIND>::value * PointStride<KIND>::value];
  int list[POSE_THREADS / 64][RefineCap<KIND>::value];
  int fr_tasks[MH_MAX_BATCH], fr_obj_base[MH_MAX_BATCH], fr_first[MH_MAX_BATCH + 1];
  int done_frame[POSE_THREADS / 64];
  int closed[POSE_THREADS / 64], n_closed;
  DevCam cams[MH_MAX_IMAGES];
};
#ifndef MH_REFINE_MIN_WAVES
#define MH_REFINE_MIN_WAVES 3
#endif
template <int KIND>
__global__ __launch_bounds__(POSE_THREADS, MH_REFINE_MIN_WAVES) void pose_refine_kernel(
    const mh_corr* __restrict__ corr0, const float4* __restrict__ depth0, float alpha,
    const int32_t* __restrict__ members0, const int32_t* __restrict__ cl_begin0, const int32_t* __restrict__ cl_count0,
    const int32_t* __restrict__ n_clusters_dev0, DevCam cam, const DevCam* __restrict__ cam_table,
    const int32_t* __restrict__ img_of0, int n_images, mh_pose_params prm, const int32_t* obj_base_dev0, int max_objects,
    float* __restrict__ obj_pose0, int32_t* __restrict__ obj_ninl0, float* __restrict__ obj_err0, int32_t* obj_valid0,
    FrameCounts* counts0, PoseTail tail0, const FilterFuseArgs* __restrict__ fuse_args, FrameBatch fbx, PoseSplit sp) {
  MH_TRACE_SCOPE(mh::TK_POSE);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  RefineLds<KIND>& L = *reinterpret_cast<RefineLds<KIND>*>(smem);
  constexpr int PS = PointStride<KIND>::value;
  constexpr int NW = POSE_THREADS / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R_ = prm.max_objects_per_cluster;
  const int G = (int)gridDim.x;
  const int n_frames = fbx.n > 1 ? fbx.n : 1;
  if (tid < n_frames) {
    const unsigned long long a = (unsigned long long)tid * fbx.arena;
    L.fr_tasks[tid] = *frame_ptr(n_clusters_dev0, a) * R_;
    L.fr_obj_base[tid] = obj_base_dev0 ? *frame_ptr(obj_base_dev0, a) : 0;
  }
  const DevCam* cams = &cam;
  if (KIND == 3) {
    for (int i = tid; i < n_images * (int)(sizeof(DevCam) / 4); i += POSE_THREADS)
      reinterpret_cast<float*>(L.cams)[i] = reinterpret_cast<const float*>(cam_table)[i];
    cams = L.cams;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int f = 0; f < n_frames; ++f) {
      L.fr_first[f] = run;
      run += L.fr_tasks[f];
    }
    L.fr_first[n_frames] = run;
  }
  __syncthreads();
  const int total = L.fr_first[n_frames];
  // frames without a task: nobody will count them down -- workgroup f mod G closes them now
  for (int f = 0; f < n_frames; ++f) {
    if (L.fr_tasks[f] == 0 && (int)blockIdx.x == f % G)   // (uniform over the workgroup)
      pose_close_frame(f, (unsigned long long)f * fbx.arena, 0, L.fr_obj_base[f], max_objects, obj_valid0, counts0, tail0,
                       fuse_args, cam, fbx);
  }
  const int n_rounds = (total + NW * G - 1) / (NW * G);
  for (int round = 0; round < n_rounds; ++round) {
    const int T = NW * (round * G + (int)blockIdx.x) + wave;
    int f_mine = -1;
    if (T < total) {
      int f = 0;
      while (T >= L.fr_first[f + 1]) ++f;   // (wave-uniform)
      f_mine = f;
      const int task = T - L.fr_first[f];
      const int cluster = task / R_, replica = task - cluster * R_;
      const unsigned long long fa = (unsigned long long)f * fbx.arena;
      const int slot = L.fr_obj_base[f] + cluster * R_ + replica;
      const PoseHyp* hyp = frame_ptr(sp.hyp, fa);
      if (slot < max_objects && hyp[slot].n_best > 0) {
        const mh_corr* __restrict__ corr = frame_ptr(corr0, fa);
        const float4* __restrict__ depth = frame_ptr(depth0, fa);
        const int32_t* __restrict__ members = frame_ptr(members0, fa);
        const int32_t* __restrict__ img_of = frame_ptr(img_of0, fa);
        int k = frame_ptr(cl_count0, fa)[cluster];
        const int begin = frame_ptr(cl_begin0, fa)[cluster];
        if (k > POSE_MAX_PTS) k = POSE_MAX_PTS;   // (pose_task has raised ERR_POSE_CAP)
        // the cluster's points: this wavefront's LDS cache, or the frame's scratch in global memory (every replica of a
        // cluster writes the same values there; the inlier lists are per replica)
        float* pts = L.pts[wave];
        int* list = L.list[wave];
        if (k > RefineCap<KIND>::value) {
          pts = frame_ptr(sp.pts, fa) + (size_t)begin * PS;
          list = frame_ptr(sp.list, fa) + (size_t)(replica & 3) * sp.max_m + begin;
        }
        for (int i = lane; i < k; i += 64) {
          const int mi = members[begin + i];
          const mh_corr c = corr[mi];
          float* p = pts + PS * i;
          p[0] = c.u;
          p[1] = c.v;
          p[2] = c.x;
          p[3] = c.y;
          p[4] = c.z;
          if (KIND == 3) p[5] = __int_as_float(img_of[mi]);
          if (KIND == 1 || KIND == 2) {
            const float4 d = depth[mi];
            p[5] = d.x;
            p[6] = d.y;
            p[7] = d.z;
            p[8] = d.w;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (k > RefineCap<KIND>::value) __threadfence();   // (global scratch written by other lanes of this wavefront)
        float R[9], t[3];
        for (int i = 0; i < 9; ++i) R[i] = hyp[slot].pose[i];
        for (int i = 0; i < 3; ++i) t[i] = hyp[slot].pose[9 + i];
#ifdef POSE_PROF
        pose_refine<KIND>(pts, list, k, R, t, hyp[slot].flags & 3, cams, prm, alpha, lane, slot, frame_ptr(obj_pose0, fa),
                          frame_ptr(obj_ninl0, fa), frame_ptr(obj_err0, fa), frame_ptr(obj_valid0, fa), fuse_args, fa, nullptr, nullptr);
#else
        pose_refine<KIND>(pts, list, k, R, t, hyp[slot].flags & 3, cams, prm, alpha, lane, slot, frame_ptr(obj_pose0, fa),
                          frame_ptr(obj_ninl0, fa), frame_ptr(obj_err0, fa), frame_ptr(obj_valid0, fa), fuse_args, fa);
#endif
      }
    }
    if (lane == 0) L.done_frame[wave] = f_mine;
    __threadfence();   // release: this round's results
    __syncthreads();
    if (tid == 0) {
      int nc = 0;
      for (int w = 0; w < NW; ++w) {
        const int f = L.done_frame[w];
        if (f < 0) continue;
        unsigned int* ticket = frame_ptr(tail0.ticket, (unsigned long long)f * fbx.arena);
        const unsigned int n = (unsigned)L.fr_tasks[f];
        const unsigned int was = atomicAdd(ticket, 1u);
        if (was + 1u == n) {
          atomicExch(ticket, 0u);
          L.closed[nc++] = f;
        }
      }
      L.n_closed = nc;
    }
    __syncthreads();
    const int nc = L.n_closed;
    if (nc) __threadfence();   // acquire: everybody else's results
    for (int c = 0; c < nc; ++c) {
      const int f = L.closed[c];
      pose_close_frame(f, (unsigned long long)f * fbx.arena, L.fr_tasks[f], L.fr_obj_base[f], max_objects, obj_valid0,
                       counts0, tail0, fuse_args, cam, fbx);
    }
    __syncthreads();   // (done_frame / closed are rewritten by the next round)
  }
}

__global__ void project_test_kernel(const float* __restrict__ pose7, const mh_corr* __restrict__ corr,
                                    int n, DevCam cam, float thr, uint8_t* __restrict__ inlier,
                                    float* __restrict__ err2, int32_t* __restrict__ n_inliers) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  TM T;
  tm_from_pose(T, pose7, pose7 + 4);
  bool in = false;
  if (i < n) {
    const mh_corr c = corr[i];
    const float e = reproj_err2(T.r, T.t, cam, c.x, c.y, c.z, c.u, c.v);
    in = e < thr;
    if (inlier) inlier[i] = in;
    if (err2) err2[i] = e;
  }
  const unsigned long long m = __ballot(in);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_inliers, __popcll(m));
}

}  // namespace

#ifdef POSE_PROF
extern "C" int mh_debug_pose_prof(unsigned long long out[32], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pose_prof), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_pose_prof), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// the fused FILTER step's arguments, from the launch's kernel arguments (safe to reuse at once) to where pose_kernel reads them
__global__ void store_fuse_args_kernel(FilterFuseArgs* dst, FilterFuseArgs v) {
  const int n = (int)(sizeof(FilterFuseArgs) / 4);
  for (int i = threadIdx.x; i < n; i += 64) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(&v)[i];
}

template <int KIND>
static void launch_pose_kind(const mh_corr* corr, const float4* depth, float alpha, const int32_t* members,
                             const int32_t* cl_model, const int32_t* cl_begin, const int32_t* cl_count,
                             const int32_t* n_clusters_dev, int max_clusters, const DevCam& cam,
                             const DevCam* cam_table, const int32_t* img_of, int n_images,
                             const mh_pose_params& p, uint64_t seed,
                             const int32_t* obj_base_dev,
                             int max_objects, int32_t* obj_model, float* obj_pose, int32_t* obj_ninl,
                             float* obj_err, int32_t* obj_cluster, int32_t* obj_valid, FrameCounts* counts,
                             const PoseTail& tail, hipStream_t s, const FilterFuse* fuse, const FrameBatch* batch,
                             const PoseSplit* split) {
  static DynLds attr;   // one per KIND (this function is a template)
  static DynLds attr_s;
  attr.ensure(pose_kernel<KIND, false>, sizeof(PoseLds<KIND>));
  attr_s.ensure(pose_kernel<KIND, true>, sizeof(PoseLds<KIND>));
  // the fused FILTER step's arguments: on the device already unless they have changed since this context's last launch
  const FilterFuseArgs* fuse_dev = nullptr;
  if (fuse && fuse->fb && fuse->tail && tail.ticket && fuse->dev && fuse->shadow && fuse->shadow_valid) {
    FilterFuseArgs now;
    std::memset(&now, 0, sizeof now);   // (padding bytes compare equal)
    now.fb = *fuse->fb;
    now.tail = *fuse->tail;
    now.tail.grid = 0;   // (the stand-alone FILTER launch's grid: follows the launches' feedback from frame to frame, unused here --
                         //  left in, it made every frame's arguments "new" and cost two one-wavefront launches per frame)
    now.feature_distance = fuse->feature_distance;
    now.min_score = fuse->min_score;
    now.min_points = fuse->min_points;
    now.n_clusters_dev = fuse->n_clusters_dev;
    if (!*fuse->shadow_valid || std::memcmp(&now, fuse->shadow, sizeof now) != 0) {
      hipLaunchKernelGGL(store_fuse_args_kernel, dim3(1), dim3(64), 0, s, fuse->dev, now);
      std::memcpy(fuse->shadow, &now, sizeof now);
      *fuse->shadow_valid = true;
    }
    fuse_dev = fuse->dev;
  }
  // one row of workgroups for all frames of the launch (tail.grid = the caller's guess of the launch's task count)
  const int n_frames = batch && batch->n > 1 ? batch->n : 1;
  static const int pose_grid = std::max(8, exp_int("MH_POSE_GRID", POSE_GRID));
  const int grid_cap = tail.grid > 0 ? std::min(tail.grid, pose_grid) : std::min(pose_grid, 96 * n_frames);
  const long all_slots = (long)max_clusters * p.max_objects_per_cluster * n_frames;
  // Two launches (PoseSplit: the frame paths hand in the scratch; MH_POSE_SPLIT=0 in experiment builds: one): the
  // hypotheses of every task, then the refines one wavefront each
  static const bool split_on = exp_int("MH_POSE_SPLIT", 1) != 0;
  // (one frame alone with a handful of tasks: the launch the split adds costs more than the compute units it frees --
  //  0.688 against 0.675 ms per isolated frame; the objects are the same bits either way)
  const bool two = split_on && split && split->hyp && tail.ticket && p.max_objects_per_cluster <= 4 &&
                   (n_frames > 1 || (tail.grid > 0 ? tail.grid : 96) > 48);
#define MH_POSE_LAUNCH(SPLIT_)                                                                                             \
  hipLaunchKernelGGL((pose_kernel<KIND, SPLIT_>), dim3((unsigned)std::max(1L, std::min((long)grid_cap, all_slots))),       \
                     dim3(POSE_THREADS), sizeof(PoseLds<KIND>), s, corr, depth, alpha, members, cl_model, cl_begin,        \
                     cl_count, n_clusters_dev, cam, cam_table, img_of, n_images, p, seed, obj_base_dev, max_objects,      \
                     obj_model, obj_pose, obj_ninl, obj_err, obj_cluster, obj_valid, counts, tail, fuse_dev,              \
                     batch ? *batch : FrameBatch(), two ? split->hyp : (PoseHyp*)nullptr)
  if (two) MH_POSE_LAUNCH(true);
  else MH_POSE_LAUNCH(false);
#undef MH_POSE_LAUNCH
  if (two) {
    static DynLds attr_r;
    attr_r.ensure(pose_refine_kernel<KIND>, sizeof(RefineLds<KIND>));
    // four tasks per workgroup and round; the guess of the task count sizes the grid, more tasks take more rounds
    const long guess = tail.grid > 0 ? tail.grid : 96L * n_frames;
    static const long g2_cap = std::max(1, exp_int("MH_POSE_RGRID", 256));
    static const int g2_div = std::max(1, exp_int("MH_POSE_RDIV", 4));   // tasks per workgroup the grid is sized for
    const unsigned g2 = (unsigned)std::max(1L, std::min(std::min(g2_cap, (guess + g2_div - 1) / g2_div), (all_slots + 3) / 4));
    hipLaunchKernelGGL(pose_refine_kernel<KIND>, dim3(g2), dim3(POSE_THREADS), sizeof(RefineLds<KIND>), s, corr, depth, alpha,
                       members, cl_begin, cl_count, n_clusters_dev, cam, cam_table, img_of, n_images, p, obj_base_dev,
                       max_objects, obj_pose, obj_ninl, obj_err, obj_valid, counts, tail, fuse_dev,
                       batch ? *batch : FrameBatch(), *split);
  }
}

void launch_pose(const mh_corr* corr, const float* depth4, int depth_kind, float alpha,
                 const int32_t* members, const int32_t* cl_model,
                 const int32_t* cl_begin, const int32_t* cl_count, const int32_t* n_clusters_dev,
                 int max_clusters, const DevCam& cam, const mh_pose_params& prm, uint64_t seed,
                 const int32_t* obj_base_dev, int max_objects, int32_t* obj_model,
                 float* obj_pose, int32_t* obj_ninl, float* obj_err, int32_t* obj_cluster,
                 int32_t* obj_valid, FrameCounts* counts, const PoseTail& tail, hipStream_t s,
                 const PoseImages& images, const FilterFuse* fuse, const FrameBatch* batch, const PoseSplit* split) {
  if (max_clusters <= 0) return;
  mh_pose_params p = prm;
  p.max_objects_per_cluster = prm.max_objects_per_cluster > 0 ? prm.max_objects_per_cluster : 1;
  static const bool repass = exp_int("MH_POSE_REPASS", 1) != 0;
  if (p.lm_iters_l2 < 0) p.lm_iters_l2 = 0;
  if (!repass) p.lm_iters_l2 = p.lm_iters_l2 > 0 ? -p.lm_iters_l2 : -1;
  const float4* d4 = reinterpret_cast<const float4*>(depth4);
  const int kind = depth4 ? depth_kind : 0;
#define POSE_ARGS corr, d4, alpha, members, cl_model, cl_begin, cl_count, n_clusters_dev, max_clusters, cam, \
                  images.cams, images.img_of, images.n_images, p,                                       \
                  seed, obj_base_dev, max_objects, obj_model, obj_pose, obj_ninl, obj_err,         \
                  obj_cluster,                                                                             \
                  obj_valid, counts, tail, s, fuse, batch, split
  if (images.img_of && images.cams && kind == 0)
    launch_pose_kind<3>(POSE_ARGS);
  else if (kind == 1)
    launch_pose_kind<1>(POSE_ARGS);
  else if (kind == 2)
    launch_pose_kind<2>(POSE_ARGS);
  else
    launch_pose_kind<0>(POSE_ARGS);
#undef POSE_ARGS
}

// Resource use of pose_kernel<KIND> as the runtime reports it (occupancy of the RANSAC kernel beside
// every GPU figure): VGPRs per lane, LDS per workgroup, threads, resident workgroups per compute unit, spill bytes.
template <int KIND>
static int pose_info_kind(int32_t out[8]) {
  static DynLds attr;
  attr.ensure(pose_kernel<KIND, true>, sizeof(PoseLds<KIND>));   // (the hypothesis kernel of the frame paths' two launches)
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(pose_kernel<KIND, true>)) != hipSuccess) return MH_ERR_HIP;
  int blocks = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, pose_kernel<KIND, true>, POSE_THREADS, sizeof(PoseLds<KIND>)) != hipSuccess)
    blocks = 0;
  out[0] = fa.numRegs;
  out[1] = (int32_t)(fa.sharedSizeBytes + sizeof(PoseLds<KIND>));
  out[2] = POSE_THREADS;
  out[3] = blocks;
  out[4] = (int32_t)fa.localSizeBytes;
  out[5] = POSE_THREADS / 64;
  out[6] = out[7] = 0;
  return MH_OK;
}
int pose_kernel_info(int kind, int32_t out[8]) {
  switch (kind) {
    case 1: return pose_info_kind<1>(out);
    case 2: return pose_info_kind<2>(out);
    case 3: return pose_info_kind<3>(out);
    default: return pose_info_kind<0>(out);
  }
}

void launch_project_test(const float* pose7, const mh_corr* corr, int n, const DevCam& cam,
                         float thr, uint8_t* inlier, float* err2, int32_t* n_inliers,
                         hipStream_t s) {
  hipMemsetAsync(n_inliers, 0, sizeof(int32_t), s);
  if (n <= 0) return;
  hipLaunchKernelGGL(project_test_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pose7, corr, n,
                     cam, thr, inlier, err2, n_inliers);
}

}  // namespace mh
