// moped3d's depth front end on the device, the parts that decide which
// features and matches reach CLUSTER:
//   DEPTHFILTER_CPU   (moped3d/libmoped/src/depthfilter/DEPTHFILTER_CPU.hpp:117-254): features
//                     (ToFilter = 1) or a model's matches (ToFilter = 2) survive only where their
//                     density per square metre of scene surface, taken over image patches and
//                     dilated 3x3, exceeds Density;
//   MATCH_ADAPTIVE_FLANN_CPU's ratio (moped3d/.../match/MATCH_ADAPTIVE_FLANN_CPU.hpp:193-215,
//                     361-376): the ratio-test threshold of a query depends on the depth under it
//                     and on the model of its nearest neighbour.
// Arithmetic follows the reference's expressions literally (float / double mix included); the
// build has -ffp-contract=off.  The per-match parts live in group_kernel (group.hip).
#include <algorithm>

#include "steps.h"

namespace mh {

namespace {

constexpr int DP_THREADS = 256;

// projectPoint (:38-43): Float x = (u - K[2]) / K[0]*depth
__device__ __forceinline__ void project_point(float u, float v, float depth, const float* K, float* o) {
  o[0] = __fmul_rn(__fdiv_rn(__fsub_rn(u, K[2]), K[0]), depth);
  o[1] = __fmul_rn(__fdiv_rn(__fsub_rn(v, K[3]), K[1]), depth);
  o[2] = depth;
}
// Pt<3>::euclDist (include/moped.hpp:125-126): d = pt[x] - this[x]; r += d*d; sqrt
__device__ __forceinline__ float eucl_dist(const float* a, const float* b) {
  float r = 0.f;
  for (int x = 0; x < 3; ++x) {
    const float d = __fsub_rn(b[x], a[x]);
    r = __fadd_rn(r, __fmul_rn(d, d));
  }
  return sqrtf(r);
}

// One workgroup per patch: minimum depth (:157-166, std::min semantics: NaN never wins), then
// the patch's area at that depth (getArea :50-61, with the reference's `y1 = min(.., width)`)
// and 1.0 / area as the double every feature of the patch adds to its density.
// blockIdx.y = frame of a batch: its own depth map (DepthMaps), its own patch map behind the frames' before it.
__global__ __launch_bounds__(DP_THREADS) void depth_patch_kernel(const float4* __restrict__ img, int w, int h,
                                                                 float k0, float k1, float k2, float k3, int patch,
                                                                 int pw, double* __restrict__ inv_size, DepthMaps maps) {
  __shared__ float red[DP_THREADS];
  if (blockIdx.y) {
    img = maps.img[blockIdx.y];
    inv_size += (size_t)blockIdx.y * gridDim.x;
  }
  const int p = blockIdx.x, px = p % pw, py = p / pw;
  const int x0 = px * patch, y0 = py * patch;
  const int x1 = min((px + 1) * patch, w), y1 = min((py + 1) * patch, h);
  float m = 1e10f;
  for (int i = threadIdx.x; i < (x1 - x0) * (y1 - y0); i += DP_THREADS) {
    const int x = x0 + i % (x1 - x0), y = y0 + i / (x1 - x0);
    const float d = img[(size_t)y * w + x].z;   // Image::getDepth (include/moped.hpp:279-284)
    m = d < m ? d : m;
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = DP_THREADS / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float o = red[threadIdx.x + s];
      red[threadIdx.x] = o < red[threadIdx.x] ? o : red[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float K[4] = {k0, k1, k2, k3};
    const int y1ref = min((py + 1) * patch, w);   // :170, as written
    float c0[3], c1[3], c2[3];
    project_point((float)x0, (float)y0, red[0], K, c0);
    project_point((float)x0, (float)y1ref, red[0], K, c1);
    project_point((float)x1, (float)y0, red[0], K, c2);
    const float area = __fmul_rn(eucl_dist(c0, c2), eucl_dist(c0, c1));
    inv_size[p] = 1.0 / (double)area;   // `1.0 / sizeMap[..]` (:189)
  }
}

}  // namespace

namespace {

constexpr int DF_MAX_PATCHES = 4096;

// DEPTHFILTER with ToFilter = 1 (:181-211): keep[q] = 1 where the dilated density of the
// feature's patch exceeds `filter`.  Single workgroup.
__global__ __launch_bounds__(1024) void feature_density_kernel(const float* __restrict__ q_uv, int Q,
                                                               const int32_t* __restrict__ q_count, int patch, int pw,
                                                               int ph, const double* __restrict__ inv_size,
                                                               float filter, uint8_t* __restrict__ keep) {
  __shared__ int cnt[DF_MAX_PATCHES];
  __shared__ float val[DF_MAX_PATCHES];
  const int P = pw * ph;
  if (blockIdx.y) {   // frame of a batch: keypoints and flags frame after frame, the patch maps too
    q_uv += 2 * (size_t)blockIdx.y * Q;
    keep += (size_t)blockIdx.y * Q;
    inv_size += (size_t)blockIdx.y * P;
  }
  if (q_count) Q = min(Q, *q_count);
  for (int p = threadIdx.x; p < P; p += blockDim.x) cnt[p] = 0;
  __syncthreads();
  for (int q = threadIdx.x; q < Q; q += blockDim.x) atomicAdd(&cnt[patch_of(q_uv[2 * q], q_uv[2 * q + 1], patch, pw, ph)], 1);
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += blockDim.x) val[p] = density_replay(cnt[p], inv_size[p]);
  __syncthreads();
  for (int q = threadIdx.x; q < Q; q += blockDim.x) {
    const int p = patch_of(q_uv[2 * q], q_uv[2 * q + 1], patch, pw, ph);
    keep[q] = dilated(val, p, pw, ph) > filter;
  }
}

}  // namespace

void launch_depth_patches(const DepthImage& dimg, const float K[4], int patch, double* inv_size, hipStream_t s,
                          const DepthMaps* maps, int n_frames) {
  const int pw = (dimg.w + patch - 1) / patch, ph = (dimg.h + patch - 1) / patch;
  const bool batch = maps && n_frames > 1;
  hipLaunchKernelGGL(depth_patch_kernel, dim3(pw * ph, batch ? n_frames : 1), dim3(DP_THREADS), 0, s, dimg.img, dimg.w,
                     dimg.h, K[0], K[1], K[2], K[3], patch, pw, inv_size, batch ? *maps : DepthMaps());
}

void launch_feature_density(const float* q_uv, int Q, const int32_t* q_count, int patch, int pw, int ph,
                            const double* inv_size, float filter, uint8_t* keep, hipStream_t s, int n_frames) {
  if (Q <= 0) return;
  hipLaunchKernelGGL(feature_density_kernel, dim3(1, std::max(1, n_frames)), dim3(1024), 0, s, q_uv, Q, q_count, patch, pw,
                     ph, inv_size, filter, keep);
}

}  // namespace mh
