// Micro-benchmark (round 5): do the 128 workgroups of a serial kernel slow each other down when they all pull the SAME
// weight image (44 x 16 bytes per lane = 180 KB per workgroup, 1 KB of consecutive memory per wave instruction) out of L2
// at the same moment - the same lines, the same channels - and would replicas of the image (each workgroup reading copy
// blockIdx % R, the copies shifted against each other) help?  Prints the time from kernel start to "all loads landed" of
// workgroup 0 and the kernel time, for R = 1, 2, 4, 8, 16.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mb/prologue_hotspot.hip -o tools/mb/prologue_hotspot
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NLD 44
__global__ __launch_bounds__(256) void k(const float* __restrict__ img, size_t rep_stride, int R, float* out, long long* cyc) {
  const float* base = img + (size_t)(blockIdx.x % R) * rep_stride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long t0 = clock64();
  f32x4 v[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) v[j] = *reinterpret_cast<const f32x4*>(base + (((size_t)wave * NLD + j) * 64 + lane) * 4);
  f32x4 s = v[0];
#pragma unroll
  for (int j = 1; j < NLD; ++j) s += v[j];
  long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const size_t img_floats = 4 * NLD * 64 * 4;                 // 180 KB
  const size_t stride = img_floats + 64 * 37;                  // copies shifted by an odd number of 256-byte pieces
  float *img, *out; long long* cyc;
  (void)hipMalloc(&img, 16 * stride * 4); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
  (void)hipMemset(img, 0, 16 * stride * 4);
  for (int R : {1, 2, 4, 8, 16, 1}) {
    for (int warm = 0; warm < 3; ++warm) hipLaunchKernelGGL(k, dim3(128), dim3(256), 0, 0, img, stride, R, out, cyc);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k, dim3(128), dim3(256), 0, 0, img, stride, R, out, cyc);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[128]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0, mx = 0; for (int i = 0; i < 128; ++i) { mean += h[i]; if (h[i] > mx) mx = h[i]; }
    printf("replicas %2d: loads landed after %.0f cycles (mean over workgroups; max %.0f) = %.2f us at 2.35 GHz; kernel %.2f us\n", R, mean / 128, mx,
           mean / 128 / 2350.0, ms * 1e3 / 20);
  }
  return 0;
}
