// Micro-benchmark (round 6, r5 verdict item 8): one K = 32 block of a weight-gradient product D[16][16] += A^T B per wave,
//   mode 0  exact fp32: 8 x v_mfma_f32_16x16x4_f32
//   mode 1  three-way bf16 split of BOTH streamed operands (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)) and the
//           six leading cross products (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid) on 6 x v_mfma_f32_16x16x32_bf16
//   mode 2  the six bf16 matrix instructions alone (operands already split: what a PRE-split operand would cost in the pipe)
// wgrad_partial streams both operands from memory (saved activations x deltas), so the split runs per element on the VALU -
// which shares the SIMD's issue with the matrix pipe (tools/mb/mfma_valu_overlap.hip).  Cycles per K-block, one wave per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mb/bf16_split.hip -o tools/mb/bf16_split
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)x[i];
    const float r1 = x[i] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    hi[i] = h;
    mid[i] = m;
    lo[i] = (__bf16)r2;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = 1e-3f * threadIdx.x + i; b[i] = 0.5f + 1e-4f * threadIdx.x * i; }
  bf16x8 ah, am, al, bh, bm, bl;
  split3(a, ah, am, al);
  split3(b, bh, bm, bl);
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));      // fresh operands every block (streamed)
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i + 1], b[i + 1], acc2, 0, 0, 0);
      }
    } else {
      if (MODE == 1) {
        split3(a, ah, am, al);
        split3(b, bh, bm, bl);
      } else {
        asm volatile("" : "+v"(ah), "+v"(am), "+v"(al), "+v"(bh), "+v"(bm), "+v"(bl));
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc2, 0, 0, 0);
    }
  }
  long long t1 = clock64();
  acc = acc + acc2;
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * sizeof(float));
  hipMalloc(&cyc, 8 * sizeof(long long));
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    k<0><<<256, 256>>>(out, cyc, iters);
    k<1><<<256, 256>>>(out, cyc, iters);
    k<2><<<256, 256>>>(out, cyc, iters);
  }
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("cycles per K = 32 block and wave (one wave per SIMD, 256 workgroups):\n");
  printf("  exact fp32, 8 x v_mfma_f32_16x16x4_f32                     %7.1f\n", (double)h[0] / iters);
  printf("  3-way bf16 split of both operands + 6 x v_mfma_16x16x32_bf16 %7.1f   (%.2f x)\n", (double)h[1] / iters, (double)h[0] / h[1]);
  printf("  the 6 bf16 matrix instructions alone (operands pre-split)  %7.1f   (%.2f x)\n", (double)h[2] / iters, (double)h[0] / h[2]);
  return 0;
}
