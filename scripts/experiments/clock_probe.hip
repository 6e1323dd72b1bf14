// What the shader clock is while a kernel runs: s_memtime (shader clock cycles, clock64) against the constant 100 MHz
// counter (wall_clock64) over a dependent FMA chain -- (a) one short kernel at a time with the chip idle in between
// (the calling convention of one synchronous frame), (b) the same kernel back to back behind a load that keeps every
// compute unit busy (the pipelined benchmark).   hipcc --offload-arch=gfx950 -O2 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void probe_kernel(int iters, unsigned long long* out, float* sink) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) a = fmaf(a, 1.0000001f, 1e-7f);
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = w1 - w0;
  }
  if (a == 12345.f) *sink = a;
}
__global__ void load_kernel(int iters, float* sink) {
  float a = threadIdx.x * 1e-3f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
  for (int i = 0; i < iters; ++i) {
    a = fmaf(a, 1.0000001f, 1e-7f);
    b = fmaf(b, 1.0000001f, 1e-7f);
    c = fmaf(c, 1.0000001f, 1e-7f);
    d = fmaf(d, 1.0000001f, 1e-7f);
  }
  if (a + b + c + d == 12345.f) *sink = a;
}

int main() {
  unsigned long long* out;
  float* sink;
  hipMalloc(&out, 2 * 64 * sizeof(unsigned long long));
  hipMalloc(&sink, 4);
  hipStream_t s, s2;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  unsigned long long h[2];
  auto run = [&](const char* what, int reps, int idle_us, bool loaded) {
    double mhz_sum = 0, us_sum = 0;
    for (int r = 0; r < reps; ++r) {
      if (loaded) hipLaunchKernelGGL(load_kernel, dim3(2048), dim3(256), 0, s2, 400000, sink);
      hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, s, 40000, out, sink);   // ~160 k cycles: the length of a POSE task
      hipStreamSynchronize(s);
      hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
      const double us = h[1] / 100.0;   // 100 MHz
      mhz_sum += h[0] / us;
      us_sum += us;
      if (loaded) hipStreamSynchronize(s2);
      if (idle_us) std::this_thread::sleep_for(std::chrono::microseconds(idle_us));
    }
    printf("%-78s shader clock %6.0f MHz, the chain took %6.1f us\n", what, mhz_sum / reps, us_sum / reps);
  };
  run("warm-up", 20, 0, true);
  run("one short kernel at a time, 2 ms idle between them", 50, 2000, false);
  run("one short kernel at a time, 200 us idle between them", 50, 200, false);
  run("one short kernel at a time, back to back", 200, 0, false);
  run("the same kernel beside a load on every compute unit", 20, 0, true);
  return 0;
}
