// Micro-benchmark: cycles per v_mfma_f32_16x16x4_f32 as a function of the number of independent accumulator
// chains (1..4) the instructions alternate between - i.e. the real dependent-accumulate latency.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define M(acc) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0)
template <int C>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      M(a0);
      if (C > 1) M(a1);
      if (C > 2) M(a2);
      if (C > 3) M(a3);
      if (C > 4) M(a4);
      if (C > 5) M(a5);
    }
  }
  long long t1 = clock64();
  f32x4 s = a0 + a1 + a2 + a3 + a4 + a5;
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int C>
void run(float* out, long long* cyc) {
  int iters = 5000;
  hipLaunchKernelGGL(k<C>, dim3(128), dim3(256), 0, 0, out, cyc, 100);
  hipLaunchKernelGGL(k<C>, dim3(128), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%d chains: %.1f cycles per MFMA (%.1f per round of %d)\n", C, h / (double)(iters * 4 * C), h / (double)(iters * 4), C);
}
int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 64);
  run<1>(out, cyc); run<2>(out, cyc); run<3>(out, cyc); run<4>(out, cyc); run<5>(out, cyc); run<6>(out, cyc);
  return 0;
}
