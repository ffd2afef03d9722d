// Micro-benchmark: sustained issue rate of v_mfma_f32_16x16x4_f32 and the shader clock it runs at.
//   hipcc --offload-arch=gfx950 -O3 tools/mb/mfma_clock.hip -o tools/mb/mfma_clock && tools/mb/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, int chains) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    if (chains > 1) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    if (chains > 2) a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
    if (chains > 2) a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  f32x4 s = a0 + a1 + a2 + a3;
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = w1 - w0; }
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 16);
  long long h[2];
  for (int chains : {1, 2, 4}) for (int wgs : {64, 128, 256, 512}) {
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, cyc, 1000, chains); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, cyc, iters, chains); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    double n = (double)iters * (chains == 4 ? 4 : chains);
    printf("chains %d wgs %3d: %.3f ms  clock64 %lld (%.1f per mfma)  wall(100MHz) %lld -> %.2f us; event-derived %.1f ns/mfma; clock64 rate %.0f MHz\n",
           chains, wgs, ms, h[0], h[0] / n, h[1], h[1] / 100.0, ms * 1e6 / n, h[0] / (h[1] / 100.0));
  }
  return 0;
}
