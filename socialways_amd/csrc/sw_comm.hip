// sw_comm.hip - a latency-optimal gradient all-reduce for the data-parallel training step (SURVEY 8e, section 5 last row):
// the three flat gradient buckets of a step are 112 KB / 112 KB / 344 KB - a ring all-reduce over 8 ranks is 14 latency-bound
// hops for them.  Here every rank holds an EXCHANGE BUFFER that all its peers have mapped (hipIpc), and one kernel per bucket
// runs a two-hop exchange over the direct xGMI links:
//
//   hop 1 (reduce-scatter)  rank r STORES slice p of its gradient into peer p's buffer (slot r), for every p, then raises
//                           flag[hop 1][r] at p;  p waits for all W flags and sums the W slots of its slice IN RANK ORDER
//                           (every element is reduced by exactly one rank in a fixed order: replicas get identical bits);
//   hop 2 (all-gather)      p stores its reduced slice into every peer's `out` region (slice p) and raises flag[hop 2][p];
//                           every rank waits for all W flags and copies `out` over its gradient buffer.
//
// Remote traffic is stores only (a load over the fabric costs a round trip).  A launch is NBLK workgroups; workgroup b owns
// chunk b of every slice on every rank and synchronises only with the workgroups b of the peers (flags per workgroup), so
// there is no grid-wide barrier.  Flags carry a monotonically rising epoch (kept in the buffer, advanced by the kernel:
// a launch recorded in a hipGraph needs no changing argument).  Payload stores are released and flags raised / polled at
// SYSTEM scope (peers are other devices, or other processes on this device).  A wait gives up after ~4 s of the constant
// 100 MHz clock and leaves an error code in the buffer (sw_comm_status) instead of hanging the GPU.
#include "../../include/socialways_hip.h"
#include "sw_common.h"
#include <cstring>
#include <cstdlib>
#include <hip/hip_runtime.h>

#define SW_COMM_MAXW 16          // ranks
#define SW_COMM_MAXBLK 32        // workgroups per launch
#define SW_COMM_FLAG_STRIDE 16   // uint32 per flag (one 64-byte line each)
#define SW_COMM_TIMEOUT_TICKS 400000000ULL   // 4 s of wall_clock64() (100 MHz)

namespace {
// layout of an exchange buffer (bytes): header | flags | recv [W][Ls] | out [W][Ls], Ls = slice capacity in floats
struct CommLayout {
  size_t flags, recv, out, total;
  long long ls_cap;
};
__host__ __device__ inline long long comm_slice_floats(long long n, int W, int nblk) {   // multiple of 4 * nblk
  const long long q = 4LL * nblk;
  return ((n + (long long)W * q - 1) / ((long long)W * q)) * q;
}
__host__ __device__ inline CommLayout comm_layout(int W, long long max_floats) {
  CommLayout L;
  L.ls_cap = comm_slice_floats(max_floats, W, SW_COMM_MAXBLK);
  L.flags = 256;      // header: [0] status, [16 + b] epoch of workgroup b
  L.recv = L.flags + (size_t)2 * SW_COMM_MAXW * SW_COMM_MAXBLK * SW_COMM_FLAG_STRIDE * 4;
  L.out = L.recv + (size_t)W * L.ls_cap * 4;
  L.total = L.out + (size_t)W * L.ls_cap * 4;
  return L;
}
struct CommArgs {
  char* peer[SW_COMM_MAXW];   // every rank's exchange buffer as mapped HERE (own buffer at [rank])
  int rank, W, nblk;
  long long n, ls, ls_cap;
  size_t flags, recv, out;
};
__device__ __forceinline__ unsigned* comm_flag(char* buf, size_t flags, int hop, int src, int blk) {
  return reinterpret_cast<unsigned*>(buf + flags) + ((size_t)(hop * SW_COMM_MAXW + src) * SW_COMM_MAXBLK + blk) * SW_COMM_FLAG_STRIDE;
}
// all threads: their payload stores are complete and visible system-wide before thread 0 raises the flags
__device__ __forceinline__ void comm_publish(const CommArgs& A, int hop, int blk, unsigned e) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's stores have been acknowledged (memory / the peer)
  __syncthreads();
  if (threadIdx.x < 64) {                               // ONE wave releases (a cached buffer: the write-back of this CU's L2)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if ((int)threadIdx.x < A.W)
      __hip_atomic_store(comm_flag(A.peer[threadIdx.x], A.flags, hop, A.rank, blk), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// thread s < W polls the flag rank s raises here; returns false (for every thread) on a time-out
__device__ __forceinline__ bool comm_wait(const CommArgs& A, int hop, int blk, unsigned e, int* ok_lds) {
  if (threadIdx.x == 0) *ok_lds = 1;
  __syncthreads();
  if ((int)threadIdx.x < A.W) {
    unsigned* f = comm_flag(A.peer[A.rank], A.flags, hop, threadIdx.x, blk);
    const unsigned long long t0 = wall_clock64();
    // (epochs are compared as a signed distance: the counter may wrap)
    while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > SW_COMM_TIMEOUT_TICKS) {
        *ok_lds = 0;
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  return *ok_lds != 0;
}
}  // namespace

__global__ __launch_bounds__(256) void allreduce_direct_kernel(CommArgs A, float* __restrict__ grad) {
  __shared__ int ok_lds;
  const int b = blockIdx.x, W = A.W, r = A.rank;
  char* mine = A.peer[r];
  unsigned* hdr = reinterpret_cast<unsigned*>(mine);
  const unsigned e = hdr[16 + b] + 1u;
  const long long chunk = A.ls / A.nblk, c0 = (long long)b * chunk;      // floats; multiple of 4
  const int nq = (int)(chunk >> 2);
  // ---- hop 1: slice p of this rank's gradient -> slot r of rank p's recv region -------------------------------------
  for (int pp = 0; pp < W; ++pp) {
    const int p = (r + 1 + pp) % W;                                        // own slice last; peers start on different links
    const long long g0 = (long long)p * A.ls + c0;
    float* dst = reinterpret_cast<float*>(A.peer[p] + A.recv) + (size_t)r * A.ls_cap + c0;
    for (int q = threadIdx.x; q < nq; q += 256) {
      const long long g = g0 + 4LL * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (g + 3 < A.n) v = ld4(grad + g);
      else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (g + k < A.n) v[k] = grad[g + k];
      }
      st4(dst + 4 * q, v);
    }
  }
  comm_publish(A, 0, b, e);
  bool ok = comm_wait(A, 0, b, e, &ok_lds);
  // ---- reduce this rank's slice in rank order, hop 2: the sum -> slice r of every rank's out region ------------------
  {
    const float* rv = reinterpret_cast<const float*>(mine + A.recv) + c0;
    for (int q = threadIdx.x; q < nq; q += 256) {
      f32x4 s = ld4(rv + 4 * q);
      for (int src = 1; src < W; ++src) s += ld4(rv + (size_t)src * A.ls_cap + 4 * q);
      for (int pp = 0; pp < W; ++pp) {
        const int p = (r + 1 + pp) % W;
        st4(reinterpret_cast<float*>(A.peer[p] + A.out) + (size_t)r * A.ls_cap + c0 + 4 * q, s);
      }
    }
  }
  comm_publish(A, 1, b, e);
  ok = comm_wait(A, 1, b, e, &ok_lds) && ok;
  // ---- the all-reduced gradient back over the rank's buffer ----------------------------------------------------------
  {
    const float* ov = reinterpret_cast<const float*>(mine + A.out) + c0;
    for (int p = 0; p < W; ++p) {
      const long long g0 = (long long)p * A.ls + c0;
      for (int q = threadIdx.x; q < nq; q += 256) {
        const long long g = g0 + 4LL * q;
        const f32x4 v = ld4(ov + (size_t)p * A.ls_cap + 4 * q);
        if (g + 3 < A.n) st4(grad + g, v);
        else {
#pragma unroll
          for (int k = 0; k < 4; ++k) if (g + k < A.n) grad[g + k] = v[k];
        }
      }
    }
  }
  if (threadIdx.x == 0) {
    hdr[16 + b] = e;
    if (!ok) hdr[0] = 1u;      // a peer never arrived: the result is garbage, say so (sw_comm_status)
  }
}

extern "C" long long sw_comm_bytes(int world, long long max_floats) {
  if (world < 1 || world > SW_COMM_MAXW || max_floats < 1) return SW_EARG;
  return (long long)comm_layout(world, max_floats).total;
}
// The exchange buffer is device memory of its own (not the caller's allocator: it must be exportable as ONE hipIpc
// allocation and should be uncached - every access to it is a hand-off).  Zero-filled: flags and epochs start at 0.
extern "C" int sw_comm_alloc(long long bytes, void** ptr) {
  if (bytes < 1 || !ptr) return SW_EARG;
  void* p = nullptr;
  // SW_COMM_CACHED=1 (tests): ordinary cached device memory - the exchange then relies on the release / acquire fences alone
  static const bool cached = getenv("SW_COMM_CACHED") && atoi(getenv("SW_COMM_CACHED")) != 0;
  hipError_t e = cached ? hipErrorNotSupported : hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc(&p, (size_t)bytes);
  }
  if (e != hipSuccess) { sw_set_error("sw_comm_alloc", e); return SW_EHIP; }
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { sw_set_error("sw_comm_alloc(memset)", e); (void)hipFree(p); return SW_EHIP; }
  *ptr = p;
  return SW_OK;
}
extern "C" int sw_comm_free(void* ptr) {
  if (!ptr) return SW_OK;
  hipError_t e = hipFree(ptr);
  if (e != hipSuccess) { sw_set_error("sw_comm_free", e); return SW_EHIP; }
  return SW_OK;
}
extern "C" int sw_comm_ipc_export(void* ptr, void* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  if (!ptr || !handle64) return SW_EARG;
  hipError_t e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), ptr);
  if (e != hipSuccess) { sw_set_error("hipIpcGetMemHandle", e); return SW_EHIP; }
  return SW_OK;
}
extern "C" int sw_comm_ipc_import(const void* handle64, void** ptr) {
  if (!handle64 || !ptr) return SW_EARG;
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle64, 64);
  hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) { sw_set_error("hipIpcOpenMemHandle", e); return SW_EHIP; }
  return SW_OK;
}
extern "C" int sw_comm_ipc_close(void* ptr) {
  if (!ptr) return SW_OK;
  hipError_t e = hipIpcCloseMemHandle(ptr);
  if (e != hipSuccess) { sw_set_error("hipIpcCloseMemHandle", e); return SW_EHIP; }
  return SW_OK;
}
extern "C" int sw_comm_status(const void* own_buf, int* status) {
  if (!own_buf || !status) return SW_EARG;
  unsigned v = 0;
  hipError_t e = hipMemcpy(&v, own_buf, 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { sw_set_error("sw_comm_status", e); return SW_EHIP; }
  *status = (int)v;
  return SW_OK;
}
extern "C" int sw_allreduce_direct(void* const* peer_bufs, int rank, int world, long long max_floats, float* grad, long long n,
                                   void* stream) {
  if (!peer_bufs || !grad || world < 1 || world > SW_COMM_MAXW || rank < 0 || rank >= world || n < 0 || n > max_floats)
    return SW_EARG;
  if (n == 0) return SW_OK;
  const CommLayout L = comm_layout(world, max_floats);
  CommArgs A;
  for (int p = 0; p < world; ++p) {
    if (!peer_bufs[p]) return SW_EARG;
    A.peer[p] = static_cast<char*>(peer_bufs[p]);
  }
  A.rank = rank; A.W = world;
  // ~4 KB of every slice per workgroup, 4 .. SW_COMM_MAXBLK workgroups: the same on every rank (a function of n and W)
  long long nb = (n * 4 / world + 4095) / 4096;
  A.nblk = (int)(nb < 4 ? 4 : nb > SW_COMM_MAXBLK ? SW_COMM_MAXBLK : nb);
  A.n = n;
  A.ls = comm_slice_floats(n, world, A.nblk);
  A.ls_cap = L.ls_cap;
  if (A.ls > A.ls_cap) return SW_ESHAPE;
  A.flags = L.flags; A.recv = L.recv; A.out = L.out;
  SW_LAUNCH(allreduce_direct_kernel, dim3(A.nblk), dim3(256), 0, (hipStream_t)stream, A, grad);
  SW_CHECK_LAUNCH("allreduce_direct_kernel");
  return SW_OK;
}
