// Experiment (not product code): is v_mfma_f32_32x32x2_f32 bit-identical to an fmaf chain over k?
// The match kernel's canonical arithmetic is dot = fmaf chain k = 0..127 (shared bit for bit with the
// oracle); an MFMA formulation could only keep the bit-exact parity if the matrix pipe accumulates
// C + a0*b0 + a1*b1 as two fused multiply-adds in k order.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off scripts/experiments/mfma_f32_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void mfma_kernel(const float* __restrict__ A /*[32][128]*/, const float* __restrict__ B /*[32][128]*/,
                            float* __restrict__ out /*[32][32]*/) {
  const int lane = threadIdx.x;
  v16f acc = {0};
  for (int k0 = 0; k0 < 128; k0 += 2) {
    // 32x32x2: lane l supplies A[i = l % 32][k0 + l / 32] and B[k0 + l / 32][j = l % 32]
    const float a = A[(lane % 32) * 128 + k0 + lane / 32];
    const float b = B[(lane % 32) * 128 + k0 + lane / 32];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  // D layout: lane l, register r: j = l % 32, i = 8 * (r / 4) + 4 * (l / 32)... the documented 32x32 layout:
  // i = (r / 4) * 8 + (l / 32) * 4 + (r % 4), j = l % 32
  for (int r = 0; r < 16; ++r) {
    const int i = (r / 4) * 8 + (lane / 32) * 4 + (r % 4), j = lane % 32;
    out[i * 32 + j] = acc[r];
  }
}

__global__ void chain_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int mode) {
  const int i = threadIdx.x / 32, j = threadIdx.x % 32;
  float s = 0.f;
  if (mode == 0) {            // fmaf chain, k ascending
    for (int k = 0; k < 128; ++k) s = fmaf(A[i * 128 + k], B[j * 128 + k], s);
  } else if (mode == 1) {     // per pair: fma(a1, b1, fma(a0, b0, s)) is mode 0; here: s + (a0*b0 + a1*b1) with one rounding each
    for (int k = 0; k < 128; k += 2) s = s + fmaf(A[i * 128 + k + 1], B[j * 128 + k + 1], A[i * 128 + k] * B[j * 128 + k]);
  } else {                    // pair order swapped
    for (int k = 0; k < 128; k += 2) s = fmaf(A[i * 128 + k], B[j * 128 + k], fmaf(A[i * 128 + k + 1], B[j * 128 + k + 1], s));
  }
  out[i * 32 + j] = s;
}

int main() {
  std::vector<float> A(32 * 128), B(32 * 128);
  srand(7);
  for (auto& x : A) x = (float)rand() / RAND_MAX * 0.4f;
  for (auto& x : B) x = (float)rand() / RAND_MAX * 0.4f;
  float *dA, *dB, *dO;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> m(1024), c(1024);
  hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dO);
  hipMemcpy(m.data(), dO, 4096, hipMemcpyDeviceToHost);
  const char* names[3] = {"fmaf chain k ascending", "s + fma(a1,b1,a0*b0)", "fmaf chain, pairs swapped"};
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(1024), 0, 0, dA, dB, dO, mode);
    hipMemcpy(c.data(), dO, 4096, hipMemcpyDeviceToHost);
    int same = 0;
    double maxrel = 0;
    for (int e = 0; e < 1024; ++e) {
      same += m[e] == c[e];
      maxrel = fmax(maxrel, fabs((double)m[e] - c[e]) / fabs((double)c[e]));
    }
    printf("mfma_f32_32x32x2f32 vs %-28s: %4d / 1024 bit-identical, max rel diff %.3e\n", names[mode], same, maxrel);
  }
  return 0;
}
