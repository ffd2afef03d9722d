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
#ifndef SW_WG_DEPTH
#define SW_WG_DEPTH 4
#endif
#ifndef SW_WG_DEPTH5
#define SW_WG_DEPTH5 2
#endif
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
#define SW_WG_RLD 69   // LDS row stride of a wave's 64 x (<= 69) block: 64 act columns + tail segment + ones
template <int NI, int KT>
__device__ __forceinline__ void wg_run(const float* __restrict__ dbase, const float* __restrict__ abase, int ldd, int lda,
                                       int rbeg, int rend, int rmax, const int* acol, const float* amask,
                                       const int* bcol, const float* bmask, const float* bone, int lg, int ln,
                                       float* __restrict__ mine, const float* __restrict__ abase2, int lda2, int row0) {
  f32x4 acc[NI][KT];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[i][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr int DEPTH = KT == 5 ? SW_WG_DEPTH5 : SW_WG_DEPTH;  // 4-row groups in flight (swept on the GPU: 4 / 2)
  // The pipeline registers hold the RAW loaded values; row masks / the ones column are applied when a group is
  // consumed.  (Arithmetic attached to the load sits in front of the loop's back edge, so every load of a body
  // iteration had to complete inside it: the compiler drained the pipeline - s_waitcnt vmcnt(0) - once per DEPTH
  // groups.)
  float a[DEPTH][NI], b[DEPTH][KT];
  auto load = [&](int r0, float* av, float* bv) {
    const int rc = min(r0 + lg, rmax);
    const float* dr = dbase + (size_t)rc * ldd;
    const float* ar = abase + (size_t)max(rc, row0) * lda;     // rows below row0 have no `act` operand
#pragma unroll
    for (int i = 0; i < NI; ++i) av[i] = dr[acol[i]];
#pragma unroll
    for (int kt = 0; kt < (KT < 5 ? KT : 4); ++kt) bv[kt] = ar[bcol[kt]];
    if (KT == 5) bv[4] = (abase2 + (size_t)rc * lda2)[bcol[4]];   // tail segment | ones
  };
#pragma unroll
  for (int q = 0; q < DEPTH - 1; ++q) load(rbeg + 4 * q, a[q], b[q]);
  for (int r = rbeg; r < rend; r += 4 * DEPTH) {
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) {
      load(r + 4 * (q + DEPTH - 1), a[(q + DEPTH - 1) % DEPTH], b[(q + DEPTH - 1) % DEPTH]);
      asm volatile("" ::: "memory");   // the loads are issued HERE (DEPTH - 1 groups ahead), not sunk to their uses
#pragma unroll
      for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(a[q][i]));   // ... and group q is first touched here
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) asm volatile("" : "+v"(b[q][kt]));
      const int rr = r + 4 * q + lg;
      const float rs = rr < rend ? 1.0f : 0.0f;
      const float rs0 = rr >= row0 ? rs : 0.0f;
      float av[NI], bv[KT];
#pragma unroll
      for (int i = 0; i < NI; ++i) av[i] = a[q][i] * (amask[i] * rs);
#pragma unroll
      for (int kt = 0; kt < (KT < 5 ? KT : 4); ++kt) bv[kt] = fmaf(b[q][kt], bmask[kt], bone[kt]) * (bone[kt] > 0.f ? rs : rs0);
      if (KT == 5) bv[4] = fmaf(b[q][4], bmask[4], bone[4]) * rs;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acc[i][kt] = SW_MFMA(av[i], bv[kt], acc[i][kt]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kt < 4 || ln < SW_WG_RLD - 64) mine[(i * 16 + 4 * lg + r) * SW_WG_RLD + kt * 16 + ln] = acc[i][kt][r];
    }
  }
}

// One wave = one job: a 64 x 64 output block (NI x KT MFMA tiles) of one column block of one problem
// over one row slice.  Operands come straight from global memory in MFMA layout (A: delta[r0+lg][n0+16i+ln],
// B: act[r0+lg][16kt+ln], 64 B contiguous per 16 lanes), software pipelined 5 four-row groups ahead; no
// LDS in the loop, no barriers - every wave streams independently, 3 waves per SIMD.  The 4 waves of a
// workgroup take 4 consecutive row slices of the same block and sum them through LDS at the end.
__global__ __launch_bounds__(SW_THREADS, 2) void wgrad_partial_kernel(WgBatch batch, float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) float red[4][64 * SW_WG_RLD];   // per-wave 64 x <=69 block, summed before the store
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int job = blockIdx.x;            // one workgroup = 4 consecutive row slices of one output block
  int p = 0;
#pragma unroll 1
  while (p + 1 < batch.np && job >= batch.job0s[p + 1]) ++p;
  const WgProblem& P = batch.p[p];
  const int j = job - P.job0;
  const int NB = (P.N + 63) >> 6;       // 64-row output blocks
  const int sg = j / NB, nb = j - sg * NB;
  const int s = sg * 4 + wave;          // this wave's row slice (may be empty)
  const int N = P.N, K = P.K, Kc = P.K + P.K2 + P.ones;
  const int n0 = nb * 64;
  const int NI = min(4, (N - n0 + 15) >> 4), KT = (Kc + 15) >> 4;
  const int nsub = P.nsplit * 4;
  const int rows_per = (((P.R + nsub - 1) / nsub) + 3) & ~3;
  const int rbeg = min(P.R, s * rows_per);
  const int rend = min(P.R, rbeg + rows_per);
  int acol[4], bcol[5];
  float amask[4], bmask[5], bone[5];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int n = n0 + 16 * i + ln, k = 16 * i + ln;
    acol[i] = min(n, N - 1);
    amask[i] = n < N ? 1.0f : 0.0f;
    bcol[i] = min(k, max(K - 1, 0));
    bmask[i] = k < K ? 1.0f : 0.0f;
    bone[i] = (P.ones && k == K + P.K2) ? 1.0f : 0.0f;
  }
  {  // k tile 4 exists only with a tail segment (K == 64): columns of act2, then the ones column
    const int c = ln;
    bcol[4] = min(c, max(P.K2 - 1, 0));
    bmask[4] = c < P.K2 ? 1.0f : 0.0f;
    bone[4] = (P.ones && c == P.K2) ? 1.0f : 0.0f;
  }
  float* mine = red[wave];
#define WG_CASE(ni, kt)                                                                                          \
  case ni * 8 + kt:                                                                                              \
    wg_run<ni, kt>(P.delta, P.act, P.ldd, P.lda, rbeg, rend, P.R - 1, acol, amask, bcol, bmask, bone, lg, ln, mine, \
                   P.act2 ? P.act2 : P.delta, P.act2 ? P.lda2 : P.ldd, P.row0);                                  \
    break;
  switch (NI * 8 + KT) {
    WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 3) WG_CASE(1, 4) WG_CASE(1, 5)
    WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 3) WG_CASE(2, 4) WG_CASE(2, 5)
    WG_CASE(3, 1) WG_CASE(3, 2) WG_CASE(3, 3) WG_CASE(3, 4) WG_CASE(3, 5)
    WG_CASE(4, 1) WG_CASE(4, 2) WG_CASE(4, 3) WG_CASE(4, 4) WG_CASE(4, 5)
  }
#undef WG_CASE
  sw_barrier();
  // one partial per workgroup (4 row slices summed): ws[ws_off + (sg*N + n)*Kc + k]
  float* out = ws + P.ws_off + (size_t)sg * N * Kc;
  const int rows = min(64, N - n0), cols = min(SW_WG_RLD, Kc);
  // element e = rr * cols + cc walks the block row-major; (rr, cc) advance incrementally (one division per thread)
  const int dq = SW_THREADS / cols, dr = SW_THREADS - dq * cols;
  int rr = threadIdx.x / cols, cc = threadIdx.x - rr * cols;
  for (int e = threadIdx.x; e < rows * cols; e += SW_THREADS) {
    const int o = rr * SW_WG_RLD + cc;
    const float v = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
    out[(size_t)(n0 + rr) * Kc + cc] = v;
    cc += dr;
    rr += dq;
    if (cc >= cols) {
      cc -= cols;
      ++rr;
    }
  }
}

// out element e of problem p = sum over slices.  A wave owns SW_WG_REL consecutive elements (lanes el = lane %
// REL: one coalesced segment per slice) and splits the slices SW_WG_RSUB ways (sub = lane / REL handles slices
// q = sub, sub + RSUB, ..), 4 independent loads in flight per lane; the sub-sums meet in a fixed shuffle tree.
#ifndef SW_WG_RSUB
#define SW_WG_RSUB 4
#endif
#define SW_WG_REL (64 / SW_WG_RSUB)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgBatch batch, const float* __restrict__ ws) {
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
  if (!live || sub != 0) return;
  int n = e / Kc, k = e - n * Kc;
  if (k < P.K) {
    float* dst = P.dW + (size_t)n * P.ldw + k;
    *dst = P.accumulate ? *dst + s : s;
  } else if (k < P.K + P.K2) {
    float* dst = P.dW2 + (size_t)n * P.ldw2 + (k - P.K);
    *dst = P.accumulate ? *dst + s : s;
  } else {
    P.db[n] = P.accumulate ? P.db[n] + s : s;
    if (P.db2) P.db2[n] = P.accumulate ? P.db2[n] + s : s;  // LSTM b_ih / b_hh share their gradient
  }
}

// host side -------------------------------------------------------------------------------------
int wg_add(WgBatch& b, const float* delta, int ldd, const float* act, int lda, int R, int N, int K, float* dW,
           int ldw, float* db, float* db2, int accumulate) {
  if (N > 256 || (ldd & 3) || (lda & 3)) return SW_ESHAPE;
  const int total = K + (db ? 1 : 0);  // act columns incl. the ones column
  for (int c0 = 0; c0 < total; c0 += 64) {
    if (b.np >= SW_WG_MAXP) return SW_ESHAPE;
    const int c1 = c0 + 64 < total ? c0 + 64 : total;
    const bool has_ones = db && c1 == total;
    WgProblem& P = b.p[b.np++];
    // a block that holds only the ones column multiplies no act column: keep its (masked) loads in bounds
    P.delta = delta; P.ldd = ldd; P.act = (has_ones && c1 - c0 == 1) ? act : act + c0; P.lda = lda;
    P.R = R; P.N = N; P.K = (c1 - c0) - (has_ones ? 1 : 0); P.ones = has_ones ? 1 : 0;
    P.dW = dW + c0; P.ldw = ldw; P.db = has_ones ? db : nullptr; P.db2 = has_ones ? db2 : nullptr;
    P.accumulate = accumulate;
    P.pre = 0;
    P.act2 = nullptr; P.dW2 = nullptr; P.lda2 = P.ldw2 = P.K2 = P.row0 = 0;
    P.nbn = (N + 15) / 16;
    P.nbk = (c1 - c0 + 15) / 16;
  }
  return SW_OK;
}

int wg_add_tail(WgBatch& b, const float* delta, int ldd, const float* act, int lda, int R, int N, int K, float* dW,
                int ldw, const float* act2, int lda2, int K2, float* dW2, int ldw2, int row0, float* db, float* db2,
                int accumulate) {
  if (N > 256 || (ldd & 3) || (lda & 3) || K != 64 || K2 < 1 || K2 + (db ? 1 : 0) > SW_WG_RLD - 64 || b.np >= SW_WG_MAXP)
    return SW_ESHAPE;
  WgProblem& P = b.p[b.np++];
  P.delta = delta; P.ldd = ldd; P.act = act; P.lda = lda; P.R = R; P.N = N; P.K = K; P.ones = db ? 1 : 0;
  P.dW = dW; P.ldw = ldw; P.db = db; P.db2 = db ? db2 : nullptr; P.accumulate = accumulate; P.pre = 0;
  P.act2 = act2; P.lda2 = lda2; P.K2 = K2; P.dW2 = dW2; P.ldw2 = ldw2; P.row0 = row0;
  P.nbn = (N + 15) / 16;
  P.nbk = 5;
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

// Cost of a problem in wave-cycles: per 4-row group a wave issues NI x KT MFMAs (32 cycles each) but
// never less than the issue time of its ~8 operand loads; a problem has ceil(N/64) output blocks.
static double wg_cost(const WgProblem& P) {
  double c = 0;
  for (int n0 = 0; n0 < P.N; n0 += 64) {
    int ni = (P.N - n0 + 15) / 16;
    if (ni > 4) ni = 4;
    double per_group = ni * P.nbk * 32.0;
    if (per_group < 192.0) per_group = 192.0;
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
  // ~1024 workgroups = 4096 wave-jobs per launch (4 per SIMD), equal cost each
  const double total = wg_total_work(b) + 1.0;
  // workgroups per launch: every wave should carry >= ~16K cycles of work (fixed per-workgroup costs -
  // pipeline fill, LDS reduction, partial store - are ~8 us), at most 1024 (two rounds of residency)
  static const double grain = getenv("SW_WG_GRAIN") ? atof(getenv("SW_WG_GRAIN")) : 8192.0;   // tuning knob (cycles of work per wave)
  double target = total / grain / 4.0;
  if (target < 64.0) target = 64.0;
  // ... at most ONE round of residency (2 workgroups x 256 CUs) - a sharp optimum once every workgroup streams with a
  // full pipeline (swept: 512 -> 63 us, 448 -> 72, 576 -> 78, 1024 -> 69 for the generator pass at m1) - unless the
  // batch is so large that a second round still leaves each wave several grains of work (dense crowds: better balance)
  static const double maxwg_env = getenv("SW_WG_MAXWG") ? atof(getenv("SW_WG_MAXWG")) : 0.0;
  const double maxwg = maxwg_env > 0.0 ? maxwg_env : (target >= 4096.0 ? 1024.0 : 512.0);
  if (target > maxwg) target = maxwg;
  size_t ws = 0;
  int job = 0, out = 0;
  for (int i = 0; i < b.np; ++i) {
    WgProblem& P = b.p[i];
    const int NB = (P.N + 63) / 64;
    int ns;
    if (P.pre) {
      ns = P.pre;
    } else {
      ns = (int)(target * wg_cost(P) / total / NB + 0.5);   // workgroups (4 row slices each) per output block
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

int wg_launch_finalized(WgBatch& b, float* ws, hipStream_t stream) {
  if (b.total_out == 0) return SW_OK;
  if (b.total_jobs > 0) {
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3(b.total_jobs), dim3(SW_THREADS), 0, stream, b, ws);
    SW_CHECK_LAUNCH("wgrad_partial_kernel");
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((b.total_out * SW_WG_RSUB + 255) / 256), dim3(256), 0, stream, b, ws);
  SW_CHECK_LAUNCH("wgrad_reduce_kernel");
  return SW_OK;
}
