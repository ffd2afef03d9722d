// Micro-benchmark (round 5): does VALU work overlap with v_mfma_f32_16x16x4_f32 on gfx950 -
//   (a) inside ONE wave (VALU instructions issued between its own MFMAs), and
//   (b) between the TWO waves a 512-thread workgroup places on each SIMD?
// The round-3 note "fp32 MFMA and VALU time add within a wave" came from a re-ordered real kernel; this isolates it.
// Per iteration: 16 MFMAs (4 accumulator chains) and 16*NV VALU ops (8 independent chains; EXP: v_exp_f32 instead of v_fma_f32).
//   mode 0  MFMA only                         mode 1  VALU only
//   mode 2  one wave, interleaved (MFMA, NV VALU, MFMA, ...)      mode 3  one wave, blocks (16 MFMA then 16 NV VALU)
//   mode 4  512 threads: waves 0-3 MFMA only, waves 4-7 VALU only (partners on the same SIMDs)
//   mode 5  512 threads: every wave runs blocks, waves 4-7 start with the VALU block (complementary phases)
//   mode 6  512 threads: every wave runs blocks in the same phase, HALF the work per wave (the split of one tile over 8 waves)
// Build: hipcc --offload-arch=gfx950 -O3 tools/mb/mfma_valu_overlap.hip -o tools/mb/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int NV, bool EXP>
__device__ __forceinline__ void valu_group(float (&v)[8], float m) {
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[q & 7]));
    else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q & 7]) : "v"(m));
  }
}
template <int NM>
__device__ __forceinline__ void mfma_group(f32x4 (&a)[4], float x, float y) {
#pragma unroll
  for (int q = 0; q < NM; ++q) { a[q & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[q & 3], 0, 0, 0); SB(); }
}

template <int MODE, int NV, bool EXP, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, long long* cyc, int iters) {
  f32x4 a[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = 0.5f + 1e-3f * threadIdx.x + q;
  const float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f, m = 0.999f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  long long t0 = clock64();
  if (MODE == 0 || (MODE == 4 && wave < 4)) {
    for (int i = 0; i < iters; ++i) mfma_group<16>(a, x, y);
  } else if (MODE == 1 || (MODE == 4 && wave >= 4)) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) valu_group<NV, EXP>(v, m);
    }
  } else if (MODE == 2) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { mfma_group<1>(a, x, y); valu_group<NV, EXP>(v, m); SB(); }
    }
  } else if (MODE == 3 || (MODE == 5 && wave < 4)) {
    for (int i = 0; i < iters; ++i) {
      mfma_group<16>(a, x, y);
#pragma unroll
      for (int u = 0; u < 16; ++u) valu_group<NV, EXP>(v, m);
      SB();
    }
  } else if (MODE == 5) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) valu_group<NV, EXP>(v, m);
      SB();
      mfma_group<16>(a, x, y);
    }
  } else if (MODE == 6) {
    for (int i = 0; i < iters; ++i) {
      mfma_group<8>(a, x, y);
#pragma unroll
      for (int u = 0; u < 8; ++u) valu_group<NV, EXP>(v, m);
      SB();
    }
  }
  long long t1 = clock64();
  f32x4 s = a[0] + a[1] + a[2] + a[3];
  float r = s[0] + s[1] + s[2] + s[3];
#pragma unroll
  for (int q = 0; q < 8; ++q) r += v[q];
  out[blockIdx.x * THREADS + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int MODE, int NV, bool EXP, int THREADS>
void run(float* out, long long* cyc, const char* what) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE, NV, EXP, THREADS>), dim3(128), dim3(THREADS), 0, 0, out, cyc, 50);
  hipLaunchKernelGGL((k<MODE, NV, EXP, THREADS>), dim3(128), dim3(THREADS), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("mode %d NV %d %s %-58s cycles/iter by wave:", MODE, NV, EXP ? "exp" : "fma", what);
  for (int w = 0; w < THREADS / 64; ++w) printf(" %6.0f", h[w] / (double)iters);
  printf("\n");
}
template <int NV, bool EXP>
void suite(float* out, long long* cyc) {
  run<0, NV, EXP, 256>(out, cyc, "MFMA only (16)");
  run<1, NV, EXP, 256>(out, cyc, "VALU only (16 NV)");
  run<2, NV, EXP, 256>(out, cyc, "one wave, interleaved");
  run<3, NV, EXP, 256>(out, cyc, "one wave, blocks");
  run<4, NV, EXP, 512>(out, cyc, "partners: waves 0-3 MFMA, 4-7 VALU");
  run<5, NV, EXP, 512>(out, cyc, "partners: blocks, complementary phases (2x work per SIMD)");
  run<6, NV, EXP, 512>(out, cyc, "partners: blocks, same phase, half the work per wave");
}
int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 1024 * 512 * 4); (void)hipMalloc(&cyc, 64);
  suite<2, false>(out, cyc);
  suite<4, false>(out, cyc);
  suite<8, false>(out, cyc);
  suite<2, true>(out, cyc);
  suite<4, true>(out, cyc);
  return 0;
}
