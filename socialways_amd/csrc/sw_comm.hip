// sw_comm.hip - a latency-optimal gradient all-reduce for the data-parallel training step (SURVEY 8e, section 5 last row):
// the three flat gradient buckets of a step are 112 KB / 112 KB / 344 KB - a ring all-reduce over 8 ranks is 14 latency-bound
// hops for them.  Here every rank holds an EXCHANGE BUFFER that all its peers have mapped (hipIpc), and one kernel per bucket
// runs a two-hop exchange over the direct xGMI links:
//
//   hop 1 (reduce-scatter)  rank r STORES slice p of its gradient into peer p's buffer (slot r), for every p;  p sums the W
//                           slots of its slice IN RANK ORDER (every element is reduced by exactly one rank in a fixed order:
//                           replicas get identical bits);
//   hop 2 (all-gather)      p stores its reduced slice into every peer's `out` region (slice p); every rank copies `out` over
//                           its gradient buffer - and, in the _adam form, applies its optimizer step from it.
//
// Remote traffic is stores only (a load over the fabric costs a round trip).  The hand-off is DATA-TAGGED: the payload
// travels as 8-byte granules {value, epoch}, two per 16-byte system-scope (sc0 sc1, write-through) store; the consumer polls
// the granules themselves with L2-bypassing system-scope loads until both tags carry this call's epoch.  No flag, no
// s_waitcnt drain in front of a flag, no fence (a system-scope release / acquire writes back / invalidates the XCD's whole L2,
// full of the backward pass's saved rows: 9 us per call measured), no workgroup barrier: a hop costs one store becoming
// visible plus one load (MI355X_MICROARCH.md, hand-off price list: granules for latency; 8-byte granules observed untorn).
// The epoch is kept in the buffer and advanced by the kernel - a launch recorded in a hipGraph needs no changing argument;
// buffers start zeroed and epochs at 1.  A launch is NBLK workgroups, every thread owning the same pairs of floats of every
// slice on every rank.  A poll gives up after SW_COMM_TIMEOUT_S seconds (default 30) of the constant 100 MHz clock instead of
// hanging the GPU; a rank whose wait timed out publishes NOTHING from that wait (no hop-2 stores with a valid tag, no
// gradient write, no optimizer step), leaves an error code in its buffer (sw_comm_status) and every later call on that
// buffer returns at once - its peers then time out in turn.  The host makes the status collective before it trusts an
// epoch (trainer.train_epoch).
#include "../../include/socialways_hip.h"
#include "sw_common.h"
#include "sw_wgrad.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <hip/hip_runtime.h>

#define SW_COMM_MAXW 16          // ranks
#define SW_COMM_MAXBLK 128       // workgroups per launch
#define SW_COMM_TICKS_PER_S 100000000ULL       // wall_clock64(): constant 100 MHz

namespace {
// layout of an exchange buffer (bytes): header | recv [W][Ls] granules | out [W][Ls] granules, Ls = slice capacity in floats
struct CommLayout {
  size_t recv, out, total;
  long long ls_cap;
};
__host__ __device__ inline long long comm_slice_floats(long long n, int W, int nblk) {   // multiple of 4 * nblk
  const long long q = 4LL * nblk;
  return ((n + (long long)W * q - 1) / ((long long)W * q)) * q;
}
__host__ __device__ inline CommLayout comm_layout(int W, long long max_floats) {
  CommLayout L;
  L.ls_cap = comm_slice_floats(max_floats, W, SW_COMM_MAXBLK);
  L.recv = 1024;      // header: [0] status, [16] epoch of the last call, [17] workgroups of the running call that have finished
  L.out = L.recv + (size_t)W * L.ls_cap * 8;
  L.total = L.out + (size_t)W * L.ls_cap * 8;
  return L;
}
struct CommArgs {
  char* peer[SW_COMM_MAXW];   // every rank's exchange buffer as mapped HERE (own buffer at [rank])
  int rank, W, nblk;
  long long n, ls, ls_cap;
  size_t recv, out;
  unsigned long long timeout_ticks;
};
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// system-scope 16-byte accesses to exchange buffers (own or a peer's): two granules {value bits, epoch}
__device__ __forceinline__ void st_gran(char* p, float a, float b, unsigned e) {
  const u32x4 v = {__float_as_uint(a), e, __float_as_uint(b), e};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// Two granules {value bits, epoch} = one 16-byte system-scope load (buffer_load_dwordx4 sc0 sc1: served from memory, not
// from this XCD's L2 or the CU's L1) through the compiler's buffer-load builtin: the COMPILER places the waits (round 5 used
// inline-asm loads with a hand-placed s_waitcnt - correct with that codegen, fragile across compilers; two 8-byte atomic
// loads per pair cost ~2 us per launch).  The 8-byte halves of such a load have been observed untorn (MI355X_MICROARCH.md).
typedef __amdgpu_buffer_rsrc_t comm_rsrc;
__device__ __forceinline__ comm_rsrc comm_make_rsrc(const char* base) {     // wave-uniform base, byte offsets per lane
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
}
struct Gran2 { u32x4 q; };      // the two granules of a 16-byte pair: {value a, tag a, value b, tag b}
__device__ __forceinline__ Gran2 ld_gran(comm_rsrc r, unsigned off) {
  Gran2 g;
  g.q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 1 | 16);      // aux: sc0 | sc1
  return g;
}
// the W granule pairs at byte offsets off + k * stride (k < W) of the own buffer, polled until every tag is e; false on a time-out
__device__ __forceinline__ bool poll_grans(Gran2 (&v)[SW_COMM_MAXW], comm_rsrc r, unsigned off, unsigned stride, int W, unsigned e,
                                           unsigned long long timeout_ticks) {
  unsigned long long t0 = 0;
  for (;;) {
#pragma unroll
    for (int k = 0; k < SW_COMM_MAXW; ++k)
      if (k < W) v[k] = ld_gran(r, off + (unsigned)k * stride);         // W loads in flight
    bool all = true;
#pragma unroll
    for (int k = 0; k < SW_COMM_MAXW; ++k)
      if (k < W) all = all && v[k].q[1] == e && v[k].q[3] == e;
    if (all) return true;
    if (t0 == 0) t0 = wall_clock64();
    else if (wall_clock64() - t0 > timeout_ticks) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}
}  // namespace

// ADAM: the rank also applies the optimizer step to its replica while it copies the reduced gradient out (the arithmetic of
// sw_adam_packed / the in-reduction updates, sw_wgrad.h; a Discriminator's registered weight images follow the update): one
// launch per bucket instead of exchange + update.  The bias corrections (two double-precision pow) are computed by one lane
// under hop 1.
template <bool ADAM>
__global__ __launch_bounds__(256) void allreduce_direct_kernel(CommArgs A, float* __restrict__ grad, WgAdam ad) {
  __shared__ float bcs[2];
  if (ADAM && threadIdx.x == 64) wg_adam_bc_compute(ad.step, ad.beta1, ad.beta2, bcs[0], bcs[1]);
  const int b = blockIdx.x, W = A.W, r = A.rank;   // (b: this workgroup's chunk of every slice)
  char* mine = A.peer[r];
  const comm_rsrc rs = comm_make_rsrc(mine);      // the own buffer: everything this rank polls
  unsigned* hdr = reinterpret_cast<unsigned*>(mine);
  // a wait of an EARLIER call on this buffer timed out: the exchange is dead (sw_comm_status says so to the host); nothing is
  // sent, nothing is written, nobody is waited for - one time-out costs one time-out, not one per remaining call of the epoch
  // (read next to the epoch, one round trip; lanes that leave early are simply gone - a barrier does not wait for them)
  const unsigned dead = __hip_atomic_load(&hdr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ONE epoch per call for the whole buffer (a per-workgroup epoch could collide: the chunking depends on n, so a granule is
  // written by different workgroup indices in different calls).  It advances when the LAST workgroup of the launch finishes -
  // by then every workgroup has read it.
  const unsigned e = __hip_atomic_load(&hdr[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  if (dead != 0u) return;
  const long long chunk = A.ls / A.nblk, c0 = (long long)b * chunk;      // floats; multiple of 4
  const int np = (int)(chunk >> 1);                                       // pairs of floats in this workgroup's chunk
  bool ok = true;
  // ---- hop 1: slice p of this rank's gradient -> slot r of rank p's recv region -------------------------------------
  for (int pp = 0; pp < W; ++pp) {
    const int p = (r + 1 + pp) % W;                                        // own slice last; peers start on different links
    const long long g0 = (long long)p * A.ls + c0;
    char* dst = A.peer[p] + A.recv + ((size_t)r * A.ls_cap + c0) * 8;
    for (int i = threadIdx.x; i < np; i += 256) {
      const long long g = g0 + 2LL * i;
      const float a = g < A.n ? grad[g] : 0.f, bb = g + 1 < A.n ? grad[g + 1] : 0.f;
      st_gran(dst + (size_t)i * 16, a, bb, e);
    }
  }
  // ---- this rank's slice summed in rank order as the slots arrive, hop 2: the sum -> slice r of every rank's out region
  {
    for (int i = threadIdx.x; i < np; i += 256) {
      Gran2 v[SW_COMM_MAXW];
      if (!poll_grans(v, rs, (unsigned)(A.recv + (size_t)c0 * 8 + (size_t)i * 16), (unsigned)(A.ls_cap * 8), W, e, A.timeout_ticks)) {
        ok = false;          // a peer never arrived: this pair is NOT published (its readers time out in turn)
        continue;
      }
      float s0 = __uint_as_float(v[0].q[0]), s1 = __uint_as_float(v[0].q[2]);
#pragma unroll
      for (int src = 1; src < SW_COMM_MAXW; ++src)
        if (src < W) {
          s0 += __uint_as_float(v[src].q[0]);
          s1 += __uint_as_float(v[src].q[2]);
        }
      for (int pp = 0; pp < W; ++pp) {
        const int p = (r + 1 + pp) % W;
        st_gran(A.peer[p] + A.out + ((size_t)r * A.ls_cap + c0) * 8 + (size_t)i * 16, s0, s1, e);
      }
    }
  }
  if constexpr (ADAM) __syncthreads();      // bcs
  // ---- the all-reduced gradient back over the rank's buffer (and the optimizer step) ----------------------------------
  {
    for (int i = threadIdx.x; i < np; i += 256) {
      Gran2 vs[SW_COMM_MAXW];
      // (a thread that has already given up on a peer does not wait a second time: the call has failed)
      if (!ok || !poll_grans(vs, rs, (unsigned)(A.out + (size_t)c0 * 8 + (size_t)i * 16), (unsigned)(A.ls_cap * 8), W, e, A.timeout_ticks)) {
        ok = false;          // no valid sum for these elements: neither the gradient nor the weights are touched
        continue;
      }
#pragma unroll
      for (int p = 0; p < SW_COMM_MAXW; ++p) {
        if (p >= W) continue;
        const long long g = (long long)p * A.ls + c0 + 2LL * i;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (g + k >= A.n) continue;
          const float val = __uint_as_float(vs[p].q[2 * k]);
          grad[g + k] = val;
          if constexpr (ADAM) wg_adam_fin(ad, wg_adam_pre(ad, grad + g + k), bcs[0], bcs[1], val);
        }
      }
    }
  }
  if (!ok) __hip_atomic_store(&hdr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // a peer never arrived (sw_comm_status)
  __syncthreads();
  if (threadIdx.x == 0) {
    if (__hip_atomic_fetch_add(&hdr[17], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)A.nblk - 1u) {
      __hip_atomic_store(&hdr[17], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&hdr[16], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

extern "C" long long sw_comm_bytes(int world, long long max_floats) {
  if (world < 1 || world > SW_COMM_MAXW || max_floats < 1) return SW_EARG;
  const CommLayout L = comm_layout(world, max_floats);
  if (L.total >= (size_t)0x7fffffff) return SW_ESHAPE;      // the kernel addresses its own buffer with 32-bit byte offsets
  return (long long)L.total;
}
// The exchange buffer is device memory of its own (not the caller's allocator: it must be exportable as ONE hipIpc
// allocation and should be uncached - every access to it is a hand-off).  Zero-filled: flags and epochs start at 0.
extern "C" int sw_comm_alloc(long long bytes, void** ptr) {
  if (bytes < 1 || !ptr) return SW_EARG;
  void* p = nullptr;
  // SW_COMM_CACHED=1 (tests): ordinary cached device memory.  The kernel has no fences: visibility then rests on its
  // system-scope (sc0 sc1) write-through stores and L2-bypassing loads alone, which is what every access to the buffer is.
  // Uncached memory is the intended form; if the runtime refuses it the fallback is SAID, never silent.
  static const bool cached = getenv("SW_COMM_CACHED") && atoi(getenv("SW_COMM_CACHED")) != 0;
  hipError_t e = cached ? hipErrorNotSupported : hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (!cached)
      fprintf(stderr, "socialways_hip: sw_comm_alloc: uncached device memory unavailable (%s); the exchange buffer is ordinary "
                      "cached memory accessed with system-scope stores / loads only\n", hipGetErrorString(e));
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
  // SW_COMM_FAULT_INJECT=1 (tests of the harness around this path - bench.py's exchange report runs it in a child job):
  // the process dies here, the way a GPU fault inside the never-hardware-tested peer mapping would take it down
  if (getenv("SW_COMM_FAULT_INJECT") && atoi(getenv("SW_COMM_FAULT_INJECT")) == 1) abort();
  if (getenv("SW_COMM_FAULT_INJECT") && atoi(getenv("SW_COMM_FAULT_INJECT")) == 2) {      // ... or the mapping is refused
    sw_set_error("hipIpcOpenMemHandle (SW_COMM_FAULT_INJECT)", hipErrorInvalidValue);
    return SW_EHIP;
  }
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
  // the whole device, not the null stream: exchange kernels run on the caller's (non-blocking) streams
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(&v, own_buf, 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { sw_set_error("sw_comm_status", e); return SW_EHIP; }
  *status = (int)v;
  return SW_OK;
}
static int allreduce_direct_launch(void* const* peer_bufs, int rank, int world, long long max_floats, float* grad, long long n,
                                   const WgAdam* adam, void* stream) {
  if (!peer_bufs || !grad || world < 1 || world > SW_COMM_MAXW || rank < 0 || rank >= world || n < 0 || n > max_floats)
    return SW_EARG;
  if (n == 0) return SW_OK;
  const CommLayout L = comm_layout(world, max_floats);
  if (L.total >= (size_t)0x7fffffff) return SW_ESHAPE;      // (sw_comm_bytes refuses such a buffer too)
  CommArgs A;
  for (int p = 0; p < world; ++p) {
    if (!peer_bufs[p]) return SW_EARG;
    A.peer[p] = static_cast<char*>(peer_bufs[p]);
  }
  A.rank = rank; A.W = world;
  // ~2 KB of every slice per workgroup (one pair of floats = one 16-byte granule pair per thread and slice: every poll of
  // a phase is ONE round trip), 4 .. SW_COMM_MAXBLK workgroups: the same on every rank (a function of n and W)
  long long nb = (n * 4 / world + 2047) / 2048;
  A.nblk = (int)(nb < 4 ? 4 : nb > SW_COMM_MAXBLK ? SW_COMM_MAXBLK : nb);
  A.n = n;
  A.ls = comm_slice_floats(n, world, A.nblk);
  A.ls_cap = L.ls_cap;
  if (A.ls > A.ls_cap) return SW_ESHAPE;
  A.recv = L.recv; A.out = L.out;
  // how long a poll waits for a peer (rank skew: a rank that evaluates / saves / captures while the others step)
  static const double timeout_s = getenv("SW_COMM_TIMEOUT_S") && atof(getenv("SW_COMM_TIMEOUT_S")) > 0.0 ? atof(getenv("SW_COMM_TIMEOUT_S")) : 30.0;
  A.timeout_ticks = (unsigned long long)(timeout_s * (double)SW_COMM_TICKS_PER_S);
  if (adam) SW_LAUNCH(allreduce_direct_kernel<true>, dim3(A.nblk), dim3(256), 0, (hipStream_t)stream, A, grad, *adam);
  else SW_LAUNCH(allreduce_direct_kernel<false>, dim3(A.nblk), dim3(256), 0, (hipStream_t)stream, A, grad, WgAdam());
  SW_CHECK_LAUNCH("allreduce_direct_kernel");
  return SW_OK;
}
extern "C" int sw_allreduce_direct(void* const* peer_bufs, int rank, int world, long long max_floats, float* grad, long long n,
                                   void* stream) {
  return allreduce_direct_launch(peer_bufs, rank, world, max_floats, grad, n, nullptr, stream);
}
extern "C" int sw_allreduce_direct_adam(void* const* peer_bufs, int rank, int world, long long max_floats, float* grad, long long n,
                                        float* w, float* m, float* v, const float* step, double lr, double beta1, double beta2,
                                        double eps, int disc_Tp, void* stream) {
  if (!w || !m || !v || !step || n < 1) return SW_EARG;
  WgAdam ad;
  ad.w = w; ad.m = m; ad.v = v; ad.g0 = grad; ad.step = step; ad.n = (size_t)n;
  ad.lr = lr; ad.beta1 = beta1; ad.beta2 = beta2; ad.eps = eps;
  if (disc_Tp > 0) {
    const DiscImages di = sw_disc_images_for(w, disc_Tp);
    ad.img = const_cast<float*>(di.img);
    ad.tab = di.tab;
  }
  return allreduce_direct_launch(peer_bufs, rank, world, max_floats, grad, n, &ad, stream);
}
