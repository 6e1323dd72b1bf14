// moped3d's DEPTHFILL step on the device: DEPTH_FILL_EXACT_CPU::fillInScaled
// (moped3d/libmoped/src/depthfill/DEPTH_FILL_EXACT_CPU.hpp:268-349; config.hpp:39 ships (8, false)).
// The depth map [h][w][4] stays in HBM: its holes (z < 0) get the depth of a nearby valid pixel, x / y / norm are
// recomputed for them (:249-273), and the step's second output, the distance map that DEPTHMAP_PROP and
// CLUSTER_LINKAGE read, is written beside it -- no trip through the host for a frame whose depth map is already on
// the device.
//
// The fill itself (:176-243) is a FIFO wavefront over the DOWNSCALED map (80 x 60 for a 640 x 480 map): a source is
// pushed on from a pixel whether or not the pixel still belongs to it, improvements are strict, ties go to the first
// arrival -- the result depends on the queue order, so the queue is replayed as it is: ONE wavefront pops the
// elements one at a time, nine lanes look at the popped pixel's 3 x 3 neighbourhood (nine different pixels: the
// updates of one element do not interact), the pushes of an element go to the tail in (dy, dx) order by ballot +
// prefix count.  Everything the loop touches lives in LDS (best distance, source, validity, a circular queue), one
// pop is ~3 dependent LDS round trips.  The rest -- downscale, seeds in raster order, upsampling (nearest neighbour
// with the reference's late row advance, :80-82, or bilinear, :93-168), normalisation -- is parallel.
#include "context.h"

namespace mh {

namespace {

constexpr int DF_THREADS = 1024;
constexpr int DF_MAX_PIX = 8192;      // pixels of the downscaled map (LDS resident)
constexpr int DF_QUEUE = 16384;       // live queue entries (circular)

struct DfLds {
  float best[DF_MAX_PIX];
  unsigned short src[DF_MAX_PIX];
  unsigned char valid[DF_MAX_PIX];
  unsigned int queue[DF_QUEUE];   // (source << 16) | pixel
  int head, tail, overflow;
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One workgroup.  zfill / fdist: [dh][dw] the filled downscaled depths and the fill's distance map (times `scale`).
__global__ __launch_bounds__(DF_THREADS) void depth_fill_kernel(const float4* __restrict__ depth, int w, int h, int scale,
                                                                int dw, int dh, float* __restrict__ zfill,
                                                                float* __restrict__ fdist, int32_t* __restrict__ err) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  DfLds& L = *reinterpret_cast<DfLds*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = dw * dh;
  // nearest-neighbour downsample (:317-323), validity (:190), best distance 0 / 1e30 (:205)
  for (int p = tid; p < n; p += DF_THREADS) {
    const int y = p / dw, x = p - y * dw;
    const float z = depth[(size_t)(y * scale) * w + x * scale].z;
    zfill[p] = z;
    const bool v = z >= 0.f;
    L.valid[p] = v ? 1 : 0;
    L.best[p] = v ? 0.f : 1e30f;
    L.src[p] = (unsigned short)p;
  }
  if (tid == 0) {
    L.head = 0;
    L.tail = 0;
    L.overflow = 0;
  }
  __syncthreads();
  if (tid < 64) {
    // seeds in raster order (:200-204): valid pixels with a hole among their in-image neighbours
    int tail = 0;
    for (int p0 = 0; p0 < n; p0 += 64) {
      const int p = p0 + lane;
      bool seed = false;
      if (p < n && L.valid[p]) {
        const int y = p / dw, x = p - y * dw;
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx, yy = y + dy;
            if (xx >= 0 && xx < dw && yy >= 0 && yy < dh && !L.valid[yy * dw + xx]) seed = true;
          }
      }
      const unsigned long long m = __ballot(seed);
      if (seed) L.queue[(tail + __popcll(m & ((1ull << lane) - 1ull))) & (DF_QUEUE - 1)] = ((unsigned)p << 16) | (unsigned)p;
      tail += __popcll(m);
    }
    wave_sync();   // (tail <= n <= DF_MAX_PIX < DF_QUEUE)
    // the queue, one element at a time (:215-241)
    const int dy = lane / 3 - 1, dx = lane - (lane / 3) * 3 - 1;   // lanes 0..8: dy outer, dx inner
    const float dil = (float)scale;
    int head = 0;
    while (head < tail) {
      const unsigned e = L.queue[head & (DF_QUEUE - 1)];
      ++head;
      const int s = (int)(e >> 16), q = (int)(e & 0xffffu);
      const int sy = s / dw, sx = s - sy * dw;
      const int qy = q / dw, qx = q - qy * dw;
      const int xp = qx + dx, yp = qy + dy;
      const int pp = yp * dw + xp;
      bool upd = false;
      float d = 0.f;
      if (lane < 9 && xp >= 0 && xp < dw && yp >= 0 && yp < dh && !L.valid[pp]) {
        d = __fmul_rn(sqrtf((float)((xp - sx) * (xp - sx) + (yp - sy) * (yp - sy))), dil);   // :227
        upd = d < L.best[pp];
      }
      const unsigned long long m = __ballot(upd);
      if (tail + __popcll(m) - head > DF_QUEUE) {   // the live part would no longer fit the ring: stop, flagged
        if (lane == 0) L.overflow = 1;
        break;
      }
      if (upd) {
        L.best[pp] = d;
        L.src[pp] = (unsigned short)s;
        L.queue[(tail + __popcll(m & ((1ull << lane) - 1ull))) & (DF_QUEUE - 1)] = ((unsigned)s << 16) | (unsigned)pp;
      }
      tail += __popcll(m);
      wave_sync();
    }
  }
  __syncthreads();
  if (tid == 0 && L.overflow) atomicOr(err, 1);
  for (int p = tid; p < n; p += DF_THREADS) {
    fdist[p] = L.best[p];
    if (!L.valid[p] && L.src[p] != p) zfill[p] = zfill[L.src[p]];   // setDepth(xp, yp, getDepth(x0, y0)): sources are valid pixels, never rewritten
  }
}

// Upsampling + normalisation, one thread per pixel of the full map.
__global__ void depth_fill_upscale_kernel(float4* __restrict__ depth, int w, int h, int scale, int dw, int dh, int bilinear,
                                          const float* __restrict__ zfill, const float* __restrict__ fdist, float k0, float k1,
                                          float k2, float k3, float* __restrict__ dist_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const int uy = i / w, ux = i - uy * w;
  float4 px = depth[i];
  if (px.z >= 0.f) {   // valid: left alone, distance 0 (:306-308)
    dist_out[i] = 0.f;
    return;
  }
  float z, fd;
  if (!bilinear) {
    // NNInterp (:69-85): lx advances before the pixel, ly after the row
    const int lx = min(ux / scale, dw - 1);
    const int ly = min(uy == 0 ? 0 : (uy - 1) / scale, dh - 1);
    z = zfill[ly * dw + lx];
    fd = fdist[ly * dw + lx];
  } else {
    // bilinearInterp (:93-168): the influences are doubles that lose `delta` per pixel since the last multiple
    const double delta = 1.0 / scale;
    double up = 1, left = 1;
    for (int k = uy % scale; k > 0; --k) up -= delta;
    for (int k = ux % scale; k > 0; --k) left -= delta;
    int x0 = ux / scale, y0 = uy / scale, x1 = x0 + 1, y1 = y0 + 1;
    float w00 = (float)(left * up), w01 = (float)(left * (1 - up)), w10 = (float)((1 - left) * up),
          w11 = (float)((1 - left) * (1 - up));
    if (x1 == dw) {
      w00 = __fadd_rn(w00, w01); w01 = 0.f;
      w01 = __fadd_rn(w01, w11); w11 = 0.f;
      x1 = x0;
    }
    if (y1 == dh) {
      w00 = __fadd_rn(w00, w10); w10 = 0.f;
      w10 = __fadd_rn(w10, w11); w11 = 0.f;
      y1 = y0;
    }
    x0 = min(x0, dw - 1); x1 = min(x1, dw - 1); y0 = min(y0, dh - 1); y1 = min(y1, dh - 1);
    auto mix = [&](const float* m) {
      float r = __fmul_rn(w00, m[y0 * dw + x0]);
      r = __fadd_rn(r, __fmul_rn(w01, m[y1 * dw + x0]));
      r = __fadd_rn(r, __fmul_rn(w10, m[y0 * dw + x1]));
      return __fadd_rn(r, __fmul_rn(w11, m[y1 * dw + x1]));
    };
    z = mix(zfill);
    fd = mix(fdist);
  }
  // normalizeDepthmap (:249-273)
  const float x = __fmul_rn(__fdiv_rn(__fsub_rn((float)ux, k2), k0), z);
  const float y = __fmul_rn(__fdiv_rn(__fsub_rn((float)uy, k3), k1), z);
  px.x = x;
  px.y = y;
  px.z = z;
  px.w = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
  depth[i] = px;
  dist_out[i] = fd;
}

__global__ void depth_fill_scale1_kernel(const float* __restrict__ fdist, int n, float* __restrict__ dist_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dist_out[i] = fdist[i];
}

__global__ void depth_count_valid_kernel(const float4* __restrict__ depth, int n, int32_t* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool v = i < n && depth[i].z >= 0.f;
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __popcll(m));
}

}  // namespace

}  // namespace mh

using namespace mh;

extern "C" int mh_depth_fill(mh_ctx* ctx, float* depth_xyzn_dev, int width, int height, int scale_factor, int bilinear,
                             const float K[4], float* fill_distance_dev, int* scale_used) {
  if (!ctx) return MH_ERR_ARG;
  if (!depth_xyzn_dev || !fill_distance_dev || !K || width <= 0 || height <= 0 || (scale_factor < 1 && scale_factor != -1)) {
    ctx->err = "mh_depth_fill: bad arguments";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc = use_stream(ctx)) return rc;
  hipStream_t s = ctx->stream;
  const int n_full = width * height;
  // [overflow word, valid count] + the downscaled maps: the context's own (the status outlives the call)
  if (!ctx->df_buf) {
    MH_HIP(ctx, hipMalloc(&ctx->df_buf, 64 + 2 * sizeof(float) * (size_t)DF_MAX_PIX));
    MH_HIP(ctx, hipMemsetAsync(ctx->df_buf, 0, 64, s));
  }
  int32_t* words = reinterpret_cast<int32_t*>(ctx->df_buf);
  float* zfill = reinterpret_cast<float*>(ctx->df_buf + 64);
  float* fdist = zfill + DF_MAX_PIX;
  int scale = scale_factor;   // (the overflow word words[0] is sticky until mh_depth_fill_status reads it)
  if (scale == -1) {   // :283-296: the factor follows the share of holes (one small reduction + a 4-byte read)
    MH_HIP(ctx, hipMemsetAsync(words + 1, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(depth_count_valid_kernel, dim3((n_full + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(depth_xyzn_dev), n_full, words + 1);
    int32_t valid_count = 0;
    MH_HIP(ctx, hipMemcpyAsync(&valid_count, words + 1, sizeof valid_count, hipMemcpyDeviceToHost, s));
    MH_HIP(ctx, hipStreamSynchronize(s));
    const float invalid_ratio = ((float)width * height - valid_count) / (width * height);
    scale = invalid_ratio < 0.1 ? 1 : invalid_ratio < 0.2 ? 2 : invalid_ratio < 0.4 ? 4 : invalid_ratio < 0.6 ? 8 : 16;
  }
  const int dw = width / scale, dh = height / scale;
  if (dw < 1 || dh < 1) {
    ctx->err = "mh_depth_fill: scale factor larger than the map";
    return MH_ERR_ARG;
  }
  if ((long)dw * dh > DF_MAX_PIX) {
    ctx->err = "mh_depth_fill: the downscaled map has more than 8192 pixels (the fill is LDS resident)";
    return MH_ERR_CAPACITY;
  }
  if (scale_used) *scale_used = scale;
  static DynLds attr;
  attr.ensure(depth_fill_kernel, sizeof(DfLds));
  hipLaunchKernelGGL(depth_fill_kernel, dim3(1), dim3(DF_THREADS), sizeof(DfLds), s,
                     reinterpret_cast<const float4*>(depth_xyzn_dev), width, height, scale, dw, dh, zfill, fdist, words);
  if (scale == 1)   // :336-338: the filled map never reaches the frame, the distance map does
    hipLaunchKernelGGL(depth_fill_scale1_kernel, dim3((n_full + 255) / 256), dim3(256), 0, s, fdist, n_full, fill_distance_dev);
  else
    hipLaunchKernelGGL(depth_fill_upscale_kernel, dim3((n_full + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<float4*>(depth_xyzn_dev), width, height, scale, dw, dh, bilinear ? 1 : 0, zfill, fdist,
                       K[0], K[1], K[2], K[3], fill_distance_dev);
  MH_HIP(ctx, hipGetLastError());
  return MH_OK;
}

extern "C" int mh_depth_fill_status(mh_ctx* ctx) {
  if (!ctx) return MH_ERR_ARG;
  if (!ctx->df_buf) return MH_OK;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc = use_stream(ctx)) return rc;
  int32_t word = 0;
  MH_HIP(ctx, hipMemcpyAsync(&word, ctx->df_buf, sizeof word, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (word) {
    MH_HIP(ctx, hipMemsetAsync(ctx->df_buf, 0, sizeof(int32_t), ctx->stream));
    ctx->err = "mh_depth_fill: the fill's queue outgrew its ring (the maps of that call are incomplete)";
    return MH_ERR_CAPACITY;
  }
  return MH_OK;
}

extern "C" int mh_depth_fill_host(mh_ctx* ctx, float* depth_xyzn_host, int width, int height, int scale_factor, int bilinear,
                                  const float K[4], float* fill_distance_host, int* scale_used) {
  if (!ctx) return MH_ERR_ARG;
  if (!depth_xyzn_host || !fill_distance_host || width <= 0 || height <= 0) {
    ctx->err = "mh_depth_fill_host: bad arguments";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc = use_stream(ctx)) return rc;
  const size_t px = (size_t)width * height;
  if (px > ctx->own_depth_px) {
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_depth) MH_HIP(ctx, hipFree(ctx->own_depth));
    if (ctx->own_fill) MH_HIP(ctx, hipFree(ctx->own_fill));
    ctx->own_depth = ctx->own_fill = nullptr;
    ctx->own_depth_px = 0;
    MH_HIP(ctx, hipMalloc(&ctx->own_depth, px * 4 * sizeof(float)));
    MH_HIP(ctx, hipMalloc(&ctx->own_fill, px * sizeof(float)));
    ctx->own_depth_px = px;
  }
  MH_HIP(ctx, hipMemcpyAsync(ctx->own_depth, depth_xyzn_host, px * 4 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  if (int rc = mh_depth_fill(ctx, ctx->own_depth, width, height, scale_factor, bilinear, K, ctx->own_fill, scale_used)) return rc;
  MH_HIP(ctx, hipMemcpyAsync(depth_xyzn_host, ctx->own_depth, px * 4 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipMemcpyAsync(fill_distance_host, ctx->own_fill, px * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  return mh_depth_fill_status(ctx);   // (synchronises the stream)
}
