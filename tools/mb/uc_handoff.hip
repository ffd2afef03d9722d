// Micro-benchmark: can workgroups of ONE launch hand rows to other workgroups (other CUs / XCDs) through UNCACHED
// device memory with plain stores - producer: stores, s_waitcnt vmcnt(0), one relaxed atomic increment per wave;
// consumer: poll the counter, plain loads - and what does the producer pay?
//   hipcc --offload-arch=gfx950 -O3 tools/mb/uc_handoff.hip -o tools/mb/uc_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NPROD 128
#define STEPS 12
#define ROWF4 4   // float4 per thread per step: 256 threads x 4 x 16 B = 16 KB per producer and step

__global__ __launch_bounds__(256) void k(float* __restrict__ data, unsigned* __restrict__ cnt, unsigned* __restrict__ bad,
                                         long long* __restrict__ cyc, int publish, int work) {
  const int tid = threadIdx.x, wg = blockIdx.x;
  if (wg < NPROD) {
    long long t0 = clock64(), tw = 0;
    float acc = tid;
    for (int s = 0; s < STEPS; ++s) {
      for (int i = 0; i < work; ++i) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);   // stands for the step's compute
      f32x4* dst = reinterpret_cast<f32x4*>(data) + ((size_t)(s * NPROD + wg) * 256 + tid) * ROWF4;
      for (int j = 0; j < ROWF4; ++j) dst[j] = f32x4{(float)s, (float)wg, (float)tid, acc * 0.f + j};
      for (int i = 0; i < work; ++i) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);   // more compute behind the stores
      if (publish) {
        long long a = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tw += clock64() - a;
        if ((tid & 63) == 0) __hip_atomic_fetch_add(cnt + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (tid == 0 && wg == 0) { cyc[0] = clock64() - t0; cyc[1] = tw; }
    if (acc == 12345.f) bad[1] = 1;
  } else if (publish) {
    const int c = wg - NPROD;
    unsigned nbad = 0;
    long long t0 = clock64();
    for (int s = 0; s < STEPS; ++s) {
      int spins = 0;
      while (__hip_atomic_load(cnt + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NPROD * 4) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > 2000000) { nbad += 1000000; break; }
      }
      // every consumer checks the rows of producer (c + q * 37) % NPROD, q = 0..7
      for (int q = 0; q < 8; ++q) {
        const int p = (c + q * 37) % NPROD;
        const f32x4* src = reinterpret_cast<const f32x4*>(data) + ((size_t)(s * NPROD + p) * 256 + tid) * ROWF4;
        for (int j = 0; j < ROWF4; ++j) {
          f32x4 v = src[j];
          if (v[0] != (float)s || v[1] != (float)p || v[2] != (float)tid || v[3] != (float)j) ++nbad;
        }
      }
    }
    if (nbad) atomicAdd(bad, nbad);
    if (tid == 0 && c == 0) cyc[2] = clock64() - t0;
  }
}

int main() {
  const size_t bytes = (size_t)STEPS * NPROD * 256 * ROWF4 * 16;
  for (int mode = 0; mode < 2; ++mode) {
    float* data = nullptr; unsigned *cnt, *bad; long long* cyc;
    hipError_t e = mode ? hipExtMallocWithFlags((void**)&data, bytes, hipDeviceMallocUncached) : hipMalloc((void**)&data, bytes);
    if (e != hipSuccess) { printf("alloc mode %d failed: %s\n", mode, hipGetErrorString(e)); continue; }
    // the counters live in uncached memory in both modes (atomics go to L2 / memory anyway)
    (void)hipExtMallocWithFlags((void**)&cnt, 4096, hipDeviceMallocUncached);
    (void)hipMalloc(&bad, 64); (void)hipMalloc(&cyc, 64);
    for (int work : {2000, 500}) {
      for (int publish : {0, 1}) {
        unsigned tot_bad = 0; float ms_sum = 0; long long c[3] = {0, 0, 0};
        for (int rep = 0; rep < 20; ++rep) {
          (void)hipMemset(data, 0xff, bytes); (void)hipMemset(cnt, 0, 4096); (void)hipMemset(bad, 0, 64);
          hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
          (void)hipEventRecord(e0);
          hipLaunchKernelGGL(k, dim3(publish ? 2 * NPROD : NPROD), dim3(256), 0, 0, data, cnt, bad, cyc, publish, work);
          (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
          float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms_sum += ms;
          unsigned hb[2]; (void)hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost); tot_bad += hb[0];
          (void)hipMemcpy(c, cyc, 24, hipMemcpyDeviceToHost);
        }
        printf("%s memory, work %4d, publish %d: %.1f us per launch, producer %lld cycles (%lld in vmcnt(0) waits), consumer %lld cycles, mismatches %u\n",
               mode ? "UNCACHED" : "default ", work, publish, ms_sum / 20 * 1e3, c[0], c[1], c[2], tot_bad);
      }
    }
    (void)hipFree(data);
  }
  return 0;
}
