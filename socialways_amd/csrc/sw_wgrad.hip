// sw_wgrad.hip - deferred weight gradients: a grouped, split-K "TN" GEMM
//
//        dW[N][K] (+)= sum_r delta[r][n] * act[r][k]        db[n] (+)= sum_r delta[r][n]
//
// over the time-major rows the serial BPTT kernels leave behind (what autograd does for the
// reference's nn.Linear / nn.LSTM weight grads, train.py:495,538).  One launch handles a batch of
// problems; on the host every problem is cut into column blocks of <= 64 act columns (the bias
// gradient rides along as a "ones" column).  A workgroup owns one row slice of one column block:
//   * it streams the slice through LDS in chunks of Rc rows; the float4 loads of chunk i+1 are
//     issued into registers before the MFMAs of chunk i and committed to LDS after them, so every
//     delta / act element is read once, coalesced, with its latency under the matrix work;
//   * wave w owns the 16-row output tiles nt = w, w+4, .. (<= 4) x all <= 4 column tiles:
//     16 accumulators, 8 LDS operand reads per 16 MFMAs (MFMA K dimension = rows, 4 per instruction);
//   * one partial per slice goes to the workspace; a second kernel reduces the slices in a fixed
//     order (deterministic: no float atomics).
#include "../../include/socialways_hip.h"
#include "sw_common.h"
#include "sw_wgrad.h"
#include "sw_wgrad_dev.h"
#include <stdlib.h>

// One wave = one job: a 64 x 64 output block (4 x 4 MFMA tiles, 16 accumulators) of one column block
// of one problem over one row slice.  Operands come straight from global memory in MFMA layout
// (A: delta[r0+lg][n0+16i+ln], B: act[r0+lg][k0+16kt+ln], 64 B contiguous per 16 lanes), software
// pipelined two 4-row groups ahead; no LDS, no barriers - every wave streams independently and the
// SIMDs hold 4 such waves.  Each delta element is read once per column block, each act element once
// per 64-row output block: L2 traffic, not HBM (the rows were written microseconds earlier).
// Branch-free streaming body, specialised on the tile counts (NI output-row tiles x KT column tiles of
// this wave's 64x64 block): loads are unconditional from clamped addresses and masked by a 0/1
// factor, so the hot loop is NI+KT loads, a few multiplies and NI*KT MFMAs - no exec-mask branches,
// no accumulator shuffling through control flow.
// workgroups that carry jobs: the job count padded to whole rounds of 8 XCDs x chunks of 4 (see the kernel)
__host__ __device__ inline int wg_grid_jobs(int total_jobs) { return (total_jobs + 31) & ~31; }
#ifdef SW_WG_STAMP      // timing experiment (tools/build_variant.sh): start / end of every workgroup of the last launch
__device__ unsigned long long g_wg_stamps[4 * 4096];
extern "C" int sw_debug_wg_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif
__global__ __launch_bounds__(SW_THREADS, 2) void wgrad_partial_kernel(WgBatch batch, float* __restrict__ ws,
                                                                      const float* __restrict__ adam_step, double beta1,
                                                                      double beta2, float* __restrict__ bc_out) {
  __shared__ __attribute__((aligned(16))) float red[SW_WG_RED_FLOATS];   // per-wave 64 x <=69 blocks, summed before the store
  // Adam's bias corrections for the reduction that follows (it applies the update): one extra workgroup, no job delayed
  if ((int)blockIdx.x >= wg_grid_jobs(batch.total_jobs)) {
    if (threadIdx.x == 0) {
      float bc1, bc2s;
      wg_adam_bc_compute(adam_step, beta1, beta2, bc1, bc2s);
      bc_out[0] = bc1;
      bc_out[1] = bc2s;
    }
    return;
  }
  // Workgroup v runs on XCD v % 8 (round-robin dispatch).  Jobs are numbered so that the output blocks of one row slice
  // are consecutive (and wg_finalize puts the problems with 4 blocks first, then 2, then 1): chunks of 4 consecutive jobs
  // go to ONE XCD, the chunks round-robin over the XCDs - the 4 gate blocks of the LSTM problem fetch their h rows into
  // one L2 instead of four, and every XCD gets the same mix of problems.
  const int v = blockIdx.x, l = v >> 3;
  const int job = ((((l >> 2) << 3) + (v & 7)) << 2) + (l & 3);
  if (job >= batch.total_jobs) return;
#ifdef SW_WG_STAMP
  const unsigned long long t0 = wall_clock64();
#endif
  wg_job(batch, ws, job, red);
#ifdef SW_WG_STAMP
  if (threadIdx.x == 0 && blockIdx.x < 2048) {      // launches below 400 jobs (discriminator passes at m1): second half
    int p = 0;
    while (p + 1 < batch.np && job >= batch.job0s[p + 1]) ++p;
    unsigned long long* o = g_wg_stamps + (batch.total_jobs < 400 ? 4 * 2048 : 0) + 4 * blockIdx.x;
    o[0] = t0;
    o[1] = wall_clock64();
    o[2] = p;
    o[3] = batch.total_jobs;
  }
#endif
}
// scratch for those two floats: a ring of slots per device (launches of one stream are ordered; 64 slots cover
// concurrent streams)
__device__ float g_wg_bc[2 * 64];
static float* wg_bc_slot() {
  static float* base[32] = {};
  static unsigned next = 0;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return nullptr;
  if (!base[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wg_bc)) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    base[dev] = (float*)p;
  }
  return base[dev] + 2 * (next++ % 64);
}

// out element e of problem p = sum over slices.  A wave owns SW_WG_REL consecutive elements (lanes el = lane %
// REL: one coalesced segment per slice) and splits the slices SW_WG_RSUB ways (sub = lane / REL handles slices
// q = sub, sub + RSUB, ..), 4 independent loads in flight per lane; the sub-sums meet in a fixed shuffle tree.
#ifndef SW_WG_RSUB
#define SW_WG_RSUB 4
#endif
#define SW_WG_REL (64 / SW_WG_RSUB)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgBatch batch, const float* __restrict__ ws, WgAdam ad) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int lane = gid & 63, el = lane % SW_WG_REL, sub = lane / SW_WG_REL;
  const int i = (gid >> 6) * SW_WG_REL + el;
  bool live = i < batch.total_out;
  int ii = live ? i : batch.total_out - 1;
  int p = 0;
#pragma unroll 1
  while (p + 1 < batch.np && ii >= batch.out0s[p + 1]) ++p;
  const WgProblem& P = batch.p[p];
  const int e = ii - P.out0;
  const int Kc = P.K + P.K2 + P.ones;
  const size_t stride = (size_t)P.N * Kc;
  const float* src = ws + P.ws_off + e;
  // destination(s) of this element; with the fused Adam update their optimizer state is fetched now, under the sums
  const int n = e / Kc, k = e - n * Kc;
  float* dst = k < P.K ? P.dW + (size_t)n * P.ldw + k : k < P.K + P.K2 ? P.dW2 + (size_t)n * P.ldw2 + (k - P.K) : P.db + n;
  float* dst2 = (k >= P.K + P.K2 && P.db2) ? P.db2 + n : nullptr;   // LSTM b_ih / b_hh share their gradient
  WgAdamPre a1 = {}, a2 = {};
  float old1 = 0.f, old2 = 0.f;
  if (ad.w) {
    a1 = wg_adam_pre(ad, dst);
    if (dst2) a2 = wg_adam_pre(ad, dst2);
  }
  if (P.accumulate) {
    old1 = *dst;
    if (dst2) old2 = *dst2;
  }
  float s = 0.f;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int q = sub;
  for (; q + 3 * SW_WG_RSUB < P.nsplit; q += 4 * SW_WG_RSUB) {  // fixed combination order
    s += src[(size_t)q * stride];
    s1 += src[(size_t)(q + SW_WG_RSUB) * stride];
    s2 += src[(size_t)(q + 2 * SW_WG_RSUB) * stride];
    s3 += src[(size_t)(q + 3 * SW_WG_RSUB) * stride];
  }
  for (; q < P.nsplit; q += SW_WG_RSUB) s += src[(size_t)q * stride];
  s = (s + s1) + (s2 + s3);
#pragma unroll
  for (int o = SW_WG_REL; o < 64; o <<= 1) s += __shfl_xor(s, o);
  float bc1 = 1.f, bc2s = 1.f;
  if (ad.w) wg_adam_bc(ad, bc1, bc2s);   // wave-uniform; computed here so that graph and eager steps share the code
  if (!live || sub != 0) return;
  const float g = P.accumulate ? old1 + s : s;
  *dst = g;
  if (ad.w) wg_adam_fin(ad, a1, bc1, bc2s, g);
  if (dst2) {
    const float g2 = P.accumulate ? old2 + s : s;
    *dst2 = g2;
    if (ad.w) wg_adam_fin(ad, a2, bc1, bc2s, g2);
  }
}

// host side -------------------------------------------------------------------------------------
static bool wg_shape_ok(int N, int K) {   // every 64-column block of delta / act must split into whole lane vectors
  for (int n0 = 0; n0 < N; n0 += 64)
    if (!wg_tiles(N - n0 < 64 ? N - n0 : 64, false)) return false;
  return K == 0 || wg_tiles(K, true) != 0;
}

int wg_add(WgBatch& b, const float* delta, int ldd, const float* act, int lda, int R, int N, int K, float* dW,
           int ldw, float* db, float* db2, int accumulate) {
  if (N > 256 || (ldd & 3) || (lda & 3)) return SW_ESHAPE;
  if ((long long)R * ldd >= (1LL << 31) || (long long)R * lda >= (1LL << 31)) return SW_ESHAPE;   // 32-bit element offsets in the kernel
  // column blocks of <= 64 REAL act columns; the ones column (bias gradient) rides with the last block on the VALU
  // (it used to open a block of its own whenever K was a multiple of 64: every delta row read again for a row sum)
  if (K < 1) return SW_ESHAPE;
  // A last output block narrower than 64 delta columns (fc2: 80 = 64 + 16) is a problem of its own: every output block
  // of a problem gets the same number of row slices, and the narrow block's jobs - a quarter of the matrix work over the
  // same rows - finished at two thirds of the launch (and, the blocks alternating with the workgroup index, all on the
  // even XCDs).  With its own split it gets fewer, longer slices.
  const int Nfull = N > 64 && (N & 63) && (N & 63) <= 32 && b.np + 2 * ((K + 63) / 64) <= SW_WG_MAXP ? (N & ~63) : N;
  for (int n0 = 0; n0 < N; n0 = (n0 == 0 ? Nfull : N)) {
    const int Nseg = n0 == 0 ? Nfull : N - Nfull;
    for (int c0 = 0; c0 < K; c0 += 64) {
      if (b.np >= SW_WG_MAXP) return SW_ESHAPE;
      const int c1 = c0 + 64 < K ? c0 + 64 : K;
      const bool has_ones = db && c1 == K;
      WgProblem& P = b.p[b.np++];
      P.delta = delta + n0; P.ldd = ldd; P.act = act + c0; P.lda = lda;
      P.R = R; P.N = Nseg; P.K = c1 - c0; P.ones = has_ones ? 1 : 0;
      P.dW = dW + (size_t)n0 * ldw + c0; P.ldw = ldw; P.db = has_ones ? db + n0 : nullptr;
      P.db2 = has_ones && db2 ? db2 + n0 : nullptr;
      P.accumulate = accumulate;
      P.pre = 0;
      P.act2 = nullptr; P.dW2 = nullptr; P.lda2 = P.ldw2 = P.K2 = P.row0 = 0;
      if (!wg_shape_ok(Nseg, P.K)) return SW_ESHAPE;
      P.nbn = (Nseg + 15) / 16;
      P.nbk = P.K > 0 ? wg_tiles(P.K, true) : 1;     // the ones column costs no matrix tile (VALU)
    }
  }
  return SW_OK;
}

int wg_add_tail(WgBatch& b, const float* delta, int ldd, const float* act, int lda, int R, int N, int K, float* dW,
                int ldw, const float* act2, int lda2, int K2, float* dW2, int ldw2, int row0, float* db, float* db2,
                int accumulate) {
  if (N > 256 || (ldd & 3) || (lda & 3) || (lda2 & 3) || K != 64 || K2 != 4 || b.np >= SW_WG_MAXP)   // the tail runs on the
    return SW_ESHAPE;                                                                                  // VALU: 4 columns, float4 rows
  if ((long long)R * ldd >= (1LL << 31) || (long long)R * lda >= (1LL << 31) || (long long)R * lda2 >= (1LL << 31)) return SW_ESHAPE;
  WgProblem& P = b.p[b.np++];
  P.delta = delta; P.ldd = ldd; P.act = act; P.lda = lda; P.R = R; P.N = N; P.K = K; P.ones = db ? 1 : 0;
  P.dW = dW; P.ldw = ldw; P.db = db; P.db2 = db ? db2 : nullptr; P.accumulate = accumulate; P.pre = 0;
  P.act2 = act2; P.lda2 = lda2; P.K2 = K2; P.dW2 = dW2; P.ldw2 = ldw2; P.row0 = row0;
  if (!wg_shape_ok(N, K)) return SW_ESHAPE;
  P.nbn = (N + 15) / 16;
  P.nbk = 4;
  return SW_OK;
}

int wg_add_pre(WgBatch& b, int N, int K, float* dW, int ldw, float* db, int nslices) {
  if (b.np >= SW_WG_MAXP || N < 1 || K < 1 || !db || nslices < 1) return SW_ESHAPE;
  WgProblem& P = b.p[b.np++];
  P.delta = nullptr; P.act = nullptr; P.ldd = P.lda = 0; P.R = 0;
  P.N = N; P.K = K; P.ones = 1; P.dW = dW; P.ldw = ldw; P.db = db; P.db2 = nullptr; P.accumulate = 0;
  P.nbn = (N + 15) / 16; P.nbk = (K + 1 + 15) / 16;
  P.act2 = nullptr; P.dW2 = nullptr; P.lda2 = P.ldw2 = P.K2 = P.row0 = 0;
  P.pre = nslices;
  b.top_reserved += (size_t)nslices * N * (K + 1);
  if (b.top_reserved > SW_WG_WS_FLOATS) return SW_ESHAPE;
  P.ws_off = SW_WG_WS_FLOATS - b.top_reserved;
  return SW_OK;
}

// Cost of a problem in wave-cycles: per 4-row group a wave issues NI x KT MFMAs (32 cycles each), the tail / ones columns
// on the VALU (4 cycles per instruction) and SW_WG_GROUP_C0 cycles of loads, masks and address arithmetic; a problem has
// ceil(N/64) output blocks.  C0 and the VALU term are fitted on per-workgroup start / end stamps of the launches at the
// metric shape (-DSW_WG_STAMP): the LSTM problem measures 848 cycles per group and wave = 512 +
// 80 + 256.  With C0 = 0 the narrow problems (K = 32 blocks, the 2048-row S / z blocks of fc1.0) were undersplit and their
// jobs ended the generator's launch 8 us after the average job.
#ifndef SW_WG_GROUP_C0
#define SW_WG_GROUP_C0 256.0
#endif
#ifndef SW_WG_ROUND_UP
#define SW_WG_ROUND_UP 0.75
#endif
static double wg_cost(const WgProblem& P) {
  double c = 0;
  for (int n0 = 0; n0 < P.N; n0 += 64) {
    int ni = (P.N - n0 + 15) / 16;
    if (ni > 4) ni = 4;
    double per_group = ni * P.nbk * 32.0 + ni * (P.K2 + P.ones) * 4.0 + SW_WG_GROUP_C0;   // MFMAs, tail / ones columns (VALU),   // + loads / masks / address arithmetic of a group (fitted on
    if (per_group < 192.0) per_group = 192.0;                // per-job stamps of the generator pass)
    c += per_group;
  }
  return c * (P.R / 4.0 + 8.0);
}
double wg_total_work(const WgBatch& b) {
  double w = 0;
  for (int i = 0; i < b.np; ++i)
    if (!b.p[i].pre) w += wg_cost(b.p[i]);
  return w;
}

size_t wg_finalize(WgBatch& b) {
  // problems in the order 4 / 3 / 2 / 1 output blocks (stable), precomputed-partial problems last: the kernel's job chunks
  // of 4 then hold whole row slices (every problem's job count is a multiple of its block count)
  {
    auto key = [](const WgProblem& P) { return P.pre ? 0 : (P.N + 63) / 64; };
    for (int i = 1; i < b.np; ++i) {          // insertion sort, <= 24 entries
      const WgProblem t = b.p[i];
      int j = i;
      for (; j > 0 && key(b.p[j - 1]) < key(t); --j) b.p[j] = b.p[j - 1];
      b.p[j] = t;
    }
  }
  // ~1024 workgroups = 4096 wave-jobs per launch (4 per SIMD), equal cost each
  const double total = wg_total_work(b) + 1.0;
  // workgroups per launch: every wave should carry >= ~16K cycles of work (fixed per-workgroup costs -
  // pipeline fill, LDS reduction, partial store - are ~8 us), at most 1024 (two rounds of residency)
  static const double grain = 8192.0 * (1.0 + SW_WG_GROUP_C0 / 512.0);   // cycles of work per wave (swept in round 3)
  double target = total / grain / 4.0;
  if (target < 64.0) target = 64.0;
  // ... at most ONE round of residency (2 workgroups x 256 CUs) - a sharp optimum once every workgroup streams with a
  // full pipeline (swept: 512 -> 63 us, 448 -> 72, 576 -> 78, 1024 -> 69 for the generator pass at m1) - unless the
  // batch is so large that a second round still leaves each wave several grains of work (dense crowds: better balance)
  const double maxwg = target >= 4096.0 ? 1024.0 : 512.0;
  if (target > maxwg) target = maxwg;
  // A batch whose natural size lies between one workgroup per CU and one resident round (a discriminator pass at the
  // metric shape: ~300) runs best with AT MOST one workgroup on every CU: beyond 256 some CUs get two and the launch
  // lasts as long as those (swept in round 3: 224 / 240 / 256 / 264 / 288 workgroups -> 0.4056 / 0.4057 / 0.4025 /
  // 0.4095 / 0.4090 ms per training step).
  const double small_env = 256.0;
  const bool one_per_cu = target > 256.0 && target < 448.0;
  if (one_per_cu) target = small_env;
  auto assign = [&](double tgt) {
    size_t ws = 0;
    int job = 0, out = 0;
    for (int i = 0; i < b.np; ++i) {
      WgProblem& P = b.p[i];
      const int NB = (P.N + 63) / 64;
      int ns;
      if (P.pre) {
        ns = P.pre;
      } else {
        // workgroups (4 row slices each) per output block.  All workgroups of a launch start together and the launch
        // lasts as long as its longest job: a share of 1.3 rounded DOWN leaves jobs 30 % above the average (the decoder's
        // fc1.0 z block at m1 ended the launch 5 us after everything else), rounded up they merely finish early
        const double share = tgt * wg_cost(P) / total / NB;
        ns = (int)(share + (share < 4.0 && maxwg <= 512.0 ? SW_WG_ROUND_UP : 0.5));   // (two rounds: the second evens out)
        int cap = (P.R + 127) / 128;  // at least 32 rows per wave
        if (ns > cap) ns = cap;
        if (ns > SW_WG_MAXSPLIT) ns = SW_WG_MAXSPLIT;
        if (ns < 1) ns = 1;
      }
      P.nsplit = ns;
      P.job0 = job;
      b.job0s[i] = job;
      b.out0s[i] = out;
      if (!P.pre) job += ns * NB;
      P.out0 = out;
      const int Kc = P.K + P.K2 + P.ones;
      out += P.N * Kc;
      if (!P.pre) {
        P.ws_off = ws;
        ws += (size_t)ns * P.N * Kc;
      }
    }
    b.total_jobs = job;
    b.total_out = out;
    return ws;
  };
  size_t ws = assign(target);
  // the per-problem rounding may overshoot the count: step down until it fits (one workgroup per CU / one resident round)
  const int limit = one_per_cu ? (int)small_env : (target >= maxwg ? (int)maxwg : 1 << 30);
  for (double tgt = target - 4.0; b.total_jobs > limit && tgt > 64.0; tgt -= 4.0) ws = assign(tgt);
  return ws + b.top_reserved;
}

// Host-side handle that carries the problems of one kernel sequence to a later launch (one wgrad launch per
// backward pass instead of one per module: the launches are occupancy-bound, not work-bound).
struct sw_wgrad_batch {
  WgBatch b;
};
extern "C" sw_wgrad_batch* sw_wgrad_batch_new(void) { return new sw_wgrad_batch(); }
extern "C" void sw_wgrad_batch_free(sw_wgrad_batch* h) { delete h; }
WgBatch* wg_pending(sw_wgrad_batch* h) { return h ? &h->b : nullptr; }

int wg_launch(WgBatch& b, float* ws, hipStream_t stream) {
  if (b.np == 0) return SW_OK;
  size_t need = wg_finalize(b);
  if (need > SW_WG_WS_FLOATS) return SW_ESHAPE;
  return wg_launch_finalized(b, ws, stream);
}

int wg_reduce_launch_adam(WgBatch& b, float* ws, const WgAdam& ad, hipStream_t stream) {
  if (b.total_out == 0) return SW_OK;
  SW_LAUNCH(wgrad_reduce_kernel, dim3((b.total_out * SW_WG_RSUB + 255) / 256), dim3(256), 0, stream, b, ws, ad);
  SW_CHECK_LAUNCH("wgrad_reduce_kernel");
  return SW_OK;
}
int wg_reduce_launch(WgBatch& b, float* ws, hipStream_t stream) { return wg_reduce_launch_adam(b, ws, WgAdam(), stream); }
int wg_launch_adam(WgBatch& b, float* ws, WgAdam& ad, hipStream_t stream) {
  if (b.np == 0) return SW_OK;
  size_t need = wg_finalize(b);
  if (need > SW_WG_WS_FLOATS) return SW_ESHAPE;
  if (b.total_out == 0) return SW_OK;
  ad.bc = nullptr;
  if (b.total_jobs > 0) {
    float* bc = ad.w ? wg_bc_slot() : nullptr;
    SW_LAUNCH(wgrad_partial_kernel, dim3(wg_grid_jobs(b.total_jobs) + (bc ? 1 : 0)), dim3(SW_THREADS), 0, stream, b, ws, ad.step,
                       ad.beta1, ad.beta2, bc);
    SW_CHECK_LAUNCH("wgrad_partial_kernel");
    ad.bc = bc;
  }
  return wg_reduce_launch_adam(b, ws, ad, stream);
}

int wg_launch_finalized(WgBatch& b, float* ws, hipStream_t stream) {
  if (b.total_out == 0) return SW_OK;
  if (b.total_jobs > 0) {
    SW_LAUNCH(wgrad_partial_kernel, dim3(wg_grid_jobs(b.total_jobs)), dim3(SW_THREADS), 0, stream, b, ws, (const float*)nullptr,
                       0.0, 0.0, (float*)nullptr);
    SW_CHECK_LAUNCH("wgrad_partial_kernel");
  }
  return wg_reduce_launch(b, ws, stream);
}
