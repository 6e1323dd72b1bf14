// Experiment (not product code): sustained dense f16 rate of the two MFMA shapes on RANDOM operands held in registers
// (eight different A and B fragments per wavefront, cycled), two wavefronts per SIMD on every CU: v_mfma_f32_32x32x16_f16
// (pass B's instruction) against v_mfma_f32_16x16x32_f16 (MI355X_MICROARCH.md: ~1.15x in bare loops).
//   hipcc --offload-arch=gfx950 -O2 scripts/experiments/mfma_shapes_rate.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
// make -C moped_amd/host builds it as moped_amd/host/mfma_rate; `mfma_rate --json` is what bench.py runs for
// roofline.sustained_mfma_only (the rate THIS box's matrix pipes sustain: boxes of the pool differ by 8%).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ inline unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ inline half8 rnd8(unsigned seed) {
  half8 h;
  for (int i = 0; i < 8; ++i) h[i] = (_Float16)((float)(hash(seed * 8 + i) & 0xFFFF) / 65536.f * 0.25f);   // SIFT-like: non-negative, < 0.25
  return h;
}

template <int SHAPE>   // 0: 32x32x16, four 32x32 accumulators; 1: 16x16x32, sixteen 16x16 accumulators (the same 64 VGPRs)
__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, int iters) {
  half8 a[8], b[8];
  const unsigned t = blockIdx.x * 512 + threadIdx.x;
  for (int i = 0; i < 8; ++i) {
    a[i] = rnd8(t * 16 + i);
    b[i] = rnd8(t * 16 + 8 + i);
  }
  float s = 0.f;
  if (SHAPE == 0) {
    v16f acc[4];
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[(u + c) & 7], acc[c], 0, 0, 0);
    }
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 16; ++r) s += acc[c][r];
  } else {
    v4f acc[16];
    for (int c = 0; c < 16; ++c)
      for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)      // the same flops per iteration: 4 x 16 MFMAs of half the size... x 2
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(u + c) & 7], b[(u + 2 * c) & 7], acc[c], 0, 0, 0);
    }
    for (int c = 0; c < 16; ++c)
      for (int r = 0; r < 4; ++r) s += acc[c][r];
  }
  if (s == 12345.678f) out[0] = s;   // keep the loop
}

static bool g_quiet = false;
template <int SHAPE>
static double run(int blocks, int iters, const char* label) {
  double best = 0.0;
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(blocks), dim3(512), 0, 0, out, iters / 10);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // per iteration and wavefront: SHAPE 0: 32 MFMAs x 2*32*32*16 flops; SHAPE 1: 64 MFMAs x 2*16*16*32 flops
    const double per_iter = SHAPE == 0 ? 32.0 * 2 * 32 * 32 * 16 : 64.0 * 2 * 16 * 16 * 32;
    const double flops = per_iter * iters * 8.0 * blocks;
    if (!g_quiet) std::printf("%s: %.1f ms: %.0f TFLOP/s\n", label, ms, flops / ms * 1e-9);
    if (rep > 0 && flops / ms * 1e-9 > best) best = flops / ms * 1e-9;   // (the first repetition still ramps)
  }
  hipFree(out);
  return best;
}

int main(int argc, char** argv) {
  int cus = 256;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 3;
  cus = p.multiProcessorCount;
  if (argc > 1 && !std::strcmp(argv[1], "--json")) {
    g_quiet = true;
    const double r32 = run<0>(cus, 8000, ""), r16 = run<1>(cus, 8000, "");
    std::printf("{\"f16_32x32x16_tflops\": %.1f, \"f16_16x16x32_tflops\": %.1f, \"cus\": %d}\n", r32, r16, cus);
    return 0;
  }
  std::printf("%d CUs\n", cus);
  run<0>(cus, 20000, "32x32x16 f16, random operands, two wavefronts per SIMD");
  run<1>(cus, 20000, "16x16x32 f16, random operands, two wavefronts per SIMD");
  run<0>(cus, 20000, "32x32x16 f16 again                                   ");
  return 0;
}
