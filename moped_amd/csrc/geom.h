// Device geometry shared by POSE and FILTER.  The projection path reproduces the
// reference's fp32 operation order exactly (no fused multiply-add), so inlier
// decisions agree bit-for-bit with project()/testAllPoints
// (moped2/libmoped/include/moped.hpp:175-200,330-354;
//  src/pose/POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:166-180).
#pragma once
#include "steps.h"

namespace mh {

struct TM {
  float r[9];  // row-major rotation
  float t[3];
};

// TransformMatrix::init (moped.hpp:175-182): q = (x,y,z,w), must be normalised.
__host__ __device__ inline void tm_from_pose(TM& T, const float* q, const float* t) {
#ifdef __HIP_DEVICE_COMPILE__
#define MUL(a, b) __fmul_rn((a), (b))
#define SUB(a, b) __fsub_rn((a), (b))
#define ADD(a, b) __fadd_rn((a), (b))
#else
#define MUL(a, b) ((a) * (b))
#define SUB(a, b) ((a) - (b))
#define ADD(a, b) ((a) + (b))
#endif
  const float q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  const float two = 2.f, one = 1.f;
  T.r[0] = SUB(SUB(one, MUL(MUL(two, q1), q1)), MUL(MUL(two, q2), q2));
  T.r[1] = SUB(MUL(MUL(two, q0), q1), MUL(MUL(two, q3), q2));
  T.r[2] = ADD(MUL(MUL(two, q0), q2), MUL(MUL(two, q3), q1));
  T.r[3] = ADD(MUL(MUL(two, q0), q1), MUL(MUL(two, q3), q2));
  T.r[4] = SUB(SUB(one, MUL(MUL(two, q0), q0)), MUL(MUL(two, q2), q2));
  T.r[5] = SUB(MUL(MUL(two, q1), q2), MUL(MUL(two, q3), q0));
  T.r[6] = SUB(MUL(MUL(two, q0), q2), MUL(MUL(two, q3), q1));
  T.r[7] = ADD(MUL(MUL(two, q1), q2), MUL(MUL(two, q3), q0));
  T.r[8] = SUB(SUB(one, MUL(MUL(two, q0), q0)), MUL(MUL(two, q1), q1));
  T.t[0] = t[0];
  T.t[1] = t[1];
  T.t[2] = t[2];
#undef MUL
#undef SUB
#undef ADD
}

#ifdef __HIPCC__
// TransformMatrix::transform (moped.hpp:183-188)
__device__ __forceinline__ void tm_apply(const float* r, const float* t, float x, float y, float z,
                                         float& ox, float& oy, float& oz) {
  ox = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, r[0]), __fmul_rn(y, r[1])), __fmul_rn(z, r[2])), t[0]);
  oy = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, r[3]), __fmul_rn(y, r[4])), __fmul_rn(z, r[5])), t[1]);
  oz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, r[6]), __fmul_rn(y, r[7])), __fmul_rn(z, r[8])), t[2]);
}

// TransformMatrix::inverseTransform (moped.hpp:190-200)
__device__ __forceinline__ void tm_apply_inv(const float* r, const float* t, float x, float y,
                                             float z, float& ox, float& oy, float& oz) {
  const float d0 = __fsub_rn(x, t[0]), d1 = __fsub_rn(y, t[1]), d2 = __fsub_rn(z, t[2]);
  ox = __fadd_rn(__fadd_rn(__fmul_rn(d0, r[0]), __fmul_rn(d1, r[3])), __fmul_rn(d2, r[6]));
  oy = __fadd_rn(__fadd_rn(__fmul_rn(d0, r[1]), __fmul_rn(d1, r[4])), __fmul_rn(d2, r[7]));
  oz = __fadd_rn(__fadd_rn(__fmul_rn(d0, r[2]), __fmul_rn(d1, r[5])), __fmul_rn(d2, r[8]));
}

// project() then the squared pixel error of testAllPoints / FILTER:
// z < 0.001 -> the reference returns (FLT_MAX, FLT_MAX) and the error overflows.
__device__ __forceinline__ float reproj_err2(const float* pr, const float* pt, const DevCam& cam,
                                             float X, float Y, float Z, float u, float v) {
  float wx, wy, wz, cx, cy, cz;
  tm_apply(pr, pt, X, Y, Z, wx, wy, wz);
  tm_apply_inv(cam.Rc, cam.tc, wx, wy, wz, cx, cy, cz);
  float pu = 3.402823466e+38f, pv = 3.402823466e+38f;
  if (!((double)cz < 0.001)) {
    pu = __fadd_rn(__fmul_rn(__fdiv_rn(cx, cz), cam.K[0]), cam.K[2]);
    pv = __fadd_rn(__fmul_rn(__fdiv_rn(cy, cz), cam.K[1]), cam.K[3]);
  }
  const float du = __fsub_rn(pu, u), dv = __fsub_rn(pv, v);
  return __fadd_rn(__fmul_rn(du, du), __fmul_rn(dv, dv));
}
#endif

}  // namespace mh
