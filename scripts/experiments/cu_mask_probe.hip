// Which compute units does a CU-masked stream (hipExtStreamCreateWithCUMask) reach on this part, and what does a
// cross-stream event hand-off cost?  build: hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.hip
// Prints, for a few masks, the set of (XCC, SE, CU) the workgroups of a 4096-block kernel ran on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include <chrono>

__global__ void where_kernel(unsigned* out, int spin) {
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    out[blockIdx.x] = ((xcc & 0xF) << 16) | ((hw >> 8) & 0xFF) | (((hw >> 13) & 7) << 8 & 0) ;
    out[blockIdx.x] = ((xcc & 0xF) << 8) | ((hw >> 8) & 0xFF);   // XCC | SE(3) SH(1) CU(4)
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) { }
}
__global__ void tiny_kernel(int* p) { if (threadIdx.x == 0 && p) atomicAdd(p, 1); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int run_mask(const char* name, const std::vector<uint32_t>& mask, unsigned* d_out, int blocks) {
  hipStream_t s;
  CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  CK(hipMemsetAsync(d_out, 0xFF, blocks * 4, s));
  hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(1024), 0, s, d_out, 2000);   // 20 us per block: all CUs get work
  CK(hipStreamSynchronize(s));
  std::vector<unsigned> h(blocks);
  CK(hipMemcpy(h.data(), d_out, blocks * 4, hipMemcpyDeviceToHost));
  std::set<unsigned> cus(h.begin(), h.end());
  int per_xcc[16] = {0};
  for (unsigned c : cus) per_xcc[(c >> 8) & 0xF]++;
  std::printf("%-28s -> %3zu CUs;", name, cus.size());
  for (int x = 0; x < 8; ++x) std::printf(" xcc%d:%d", x, per_xcc[x]);
  std::printf("\n");
  CK(hipStreamDestroy(s));
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  std::printf("%s: %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
  const int blocks = 4096;
  unsigned* d_out;
  CK(hipMalloc(&d_out, blocks * 4));
  const int words = 8;   // 256 bits
  std::vector<uint32_t> all(words, 0xFFFFFFFFu);
  run_mask("all 256 bits", all, d_out, blocks);
  std::vector<uint32_t> m(words, 0xFFFFFFFFu);
  m[7] = 0;                                   // bits 224..255 off
  run_mask("bits 0..223", m, d_out, blocks);
  m.assign(words, 0u); m[7] = 0xFFFFFFFFu;    // only bits 224..255
  run_mask("bits 224..255", m, d_out, blocks);
  m.assign(words, 0u); m[0] = 0xFFFFFFFFu;    // only bits 0..31
  run_mask("bits 0..31", m, d_out, blocks);
  m.assign(words, 0xFFFFFFFFu); m[0] &= ~0xFFu;   // bits 0..7 off
  run_mask("all but bits 0..7", m, d_out, blocks);
  m.assign(words, 0u); m[0] = 0xFFu;
  run_mask("bits 0..7", m, d_out, blocks);
  m.assign(words, 0u); for (int i = 0; i < 256; i += 8) m[i / 32] |= 1u << (i % 32);
  run_mask("every 8th bit", m, d_out, blocks);
  // cost of an event hand-off between two streams: A -> B -> A, 200 round trips
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t e1, e2;
  CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  int* d_cnt;
  CK(hipMalloc(&d_cnt, 4));
  for (int rep = 0; rep < 2; ++rep) {
    const int n = 200;
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
      hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, a, d_cnt);
      CK(hipEventRecord(e1, a));
      CK(hipStreamWaitEvent(b, e1, 0));
      hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, b, d_cnt);
      CK(hipEventRecord(e2, b));
      CK(hipStreamWaitEvent(a, e2, 0));
    }
    CK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2 * n; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, a, d_cnt);
    CK(hipDeviceSynchronize());
    const double us1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
    std::printf("two kernels per round trip: across two streams with events %.1f us, in one stream %.1f us\n", us / n, us1 / n);
  }
  return 0;
}
