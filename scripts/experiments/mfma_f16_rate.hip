// Experiment (not product code): what dense f16 MFMA rate does the chip SUSTAIN?  Pass B of the two-stage MATCH sits at
// ~1.3 PFLOP/s = 51-54% of the 2.5 PFLOP/s the data sheet quotes at 2.4 GHz whatever its wavefront-level schedule
// looks like; this loop has nothing but v_mfma_f32_32x32x16_f16 in it -- registers only, four independent
// accumulators per wavefront, W wavefronts per SIMD on every CU -- and says what the matrix pipes deliver when they
// are never kept waiting, for seconds, i.e. under the clock the power budget allows.
//   hipcc --offload-arch=gfx950 -O2 scripts/experiments/mfma_f16_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, float seed) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(seed + 0.001f * (threadIdx.x + i));
    b[i] = (_Float16)(seed - 0.002f * (threadIdx.x - i));
  }
  v16f acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 12345.678f) out[0] = s;   // keep the loop
}

template <int CHAINS>
static void run(int threads, int blocks, int iters, const char* label) {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, iters / 10, 0.5f);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 16 * 8.0 * CHAINS * iters * (threads / 64) * (double)blocks;
    std::printf("%s: %d wavefronts per CU x %d independent chains, %.1f ms: %.0f TFLOP/s\n", label, threads / 64, CHAINS, ms,
                flops / ms * 1e-9);
  }
  hipFree(out);
}

int main() {
  int cus = 256;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) == hipSuccess) cus = p.multiProcessorCount;
  std::printf("%d CUs, clock %d MHz\n", cus, p.clockRate / 1000);
  run<1>(256, cus, 20000, "one wavefront per SIMD, dependent chain ");
  run<4>(256, cus, 6000, "one wavefront per SIMD, 4 chains         ");
  run<4>(512, cus, 4000, "two wavefronts per SIMD, 4 chains        ");
  run<2>(512, cus, 40000, "two wavefronts per SIMD, 2 chains, 2.5 s ");
  return 0;
}
