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

#define WG_LDS_FLOATS 12288  // 48 KB staging per workgroup -> 3 workgroups per CU

__host__ __device__ inline int wg_ld(int tiles) {  // row stride = 16 (mod 32): the two rows a half-wave
  int ld = tiles * 16;                             // reads never share an LDS bank
  return (ld & 31) == 0 ? ld + 16 : ld;
}
__host__ __device__ inline int wg_pow2(int x) {
  int p = 4;
  while (p < x) p <<= 1;
  return p;
}

__global__ __launch_bounds__(SW_THREADS) void wgrad_partial_kernel(WgBatch batch, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int job = blockIdx.x;
  int p = 0;
#pragma unroll 1
  while (p + 1 < batch.np && job >= batch.p[p + 1].job0) ++p;
  const WgProblem& P = batch.p[p];
  const int s = job - P.job0;
  const int N = P.N, K = P.K, Kc = P.K + P.ones;
  const int NT = (N + 15) >> 4, KT = (Kc + 15) >> 4;
  const int ldn = wg_ld(NT), ldk = wg_ld(KT);
  const int CW = wg_pow2(NT * 4), CWk = wg_pow2(KT * 4);  // float4 lanes per staged row
  const int rpd = SW_THREADS / CW, rpa = SW_THREADS / CWk;  // rows per staging iteration
  int Rc = WG_LDS_FLOATS / (ldn + ldk);
  Rc = min(min(Rc, 64), min(8 * rpd, 4 * rpa)) & ~3;
  float* dst = smem;
  float* ast = smem + Rc * ldn;
  const int rows_per = (((P.R + P.nsplit - 1) / P.nsplit) + 3) & ~3;
  const int rbeg = s * rows_per;
  const int rend = min(P.R, rbeg + rows_per);
  const float* dptr = P.delta;
  const float* aptr = P.act;
  const int ldd = P.ldd, lda = P.lda;
  const int ones = P.ones;
  // this thread's staging coordinates
  const int drow = threadIdx.x / CW, dcol = (threadIdx.x & (CW - 1)) * 4;
  const int arow = threadIdx.x / CWk, acol = (threadIdx.x & (CWk - 1)) * 4;
  const bool dlive = dcol < NT * 16, alive = acol < KT * 16;

  f32x4 pd[8], pa[4];
  auto issue = [&](int r0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      int rr = drow + u * rpd, r = r0 + rr;
      if (dlive && rr < Rc && r < rend) {
        const float* q = dptr + (size_t)r * ldd + dcol;
        if (dcol + 3 < N) v = ld4(q);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (dcol + e < N) v[e] = q[e];
        }
      }
      pd[u] = v;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      int rr = arow + u * rpa, r = r0 + rr;
      if (alive && rr < Rc && r < rend) {
        const float* q = aptr + (size_t)r * lda + acol;
        if (acol + 3 < K) v = ld4(q);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acol + e < K ? q[e] : ((ones && acol + e == K) ? 1.0f : 0.f);
        }
      }
      pa[u] = v;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int rr = drow + u * rpd;
      if (dlive && rr < Rc) st4(dst + rr * ldn + dcol, pd[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int rr = arow + u * rpa;
      if (alive && rr < Rc) st4(ast + rr * ldk + acol, pa[u]);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) acc[i][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* abase = dst + lg * ldn + wave * 16 + ln;
  const float* bbase = ast + lg * ldk + ln;
  if (rbeg < rend) issue(rbeg);
  for (int r0 = rbeg; r0 < rend; r0 += Rc) {
    commit();
    sw_barrier();
    if (r0 + Rc < rend) issue(r0 + Rc);
    const int gmax = min(Rc, (rend - r0 + 3) & ~3);
    for (int g = 0; g < gmax; g += 4) {
      float b[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) b[kt] = kt < KT ? bbase[g * ldk + kt * 16] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (wave + 4 * i < NT) {
          float a = abase[g * ldn + i * 64];
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            if (kt < KT) acc[i][kt] = SW_MFMA(a, b[kt], acc[i][kt]);
          }
        }
      }
    }
    sw_barrier();
  }
  // partial of this slice: ws[ws_off + (s*N + n)*Kc + k]
  float* out = ws + P.ws_off + (size_t)s * N * Kc;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int nt = wave + 4 * i;
    if (nt < NT) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        int k = kt * 16 + ln;
        if (kt < KT && k < Kc) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int n = nt * 16 + 4 * lg + r;
            if (n < N) out[(size_t)n * Kc + k] = acc[i][kt][r];
          }
        }
      }
    }
  }
}

// out element e of problem p = sum over slices, 16 lanes per element (slices q = lane&15, +16, ...),
// combined by a fixed shuffle tree.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgBatch batch, const float* __restrict__ ws) {
  int gid = blockIdx.x * 256 + threadIdx.x;
  int i = gid >> 4, sub = gid & 15;
  bool live = i < batch.total_out;
  int ii = live ? i : batch.total_out - 1;
  int p = 0;
#pragma unroll 1
  while (p + 1 < batch.np && ii >= batch.p[p + 1].out0) ++p;
  const WgProblem& P = batch.p[p];
  const int e = ii - P.out0;
  const int Kc = P.K + P.ones;
  const size_t stride = (size_t)P.N * Kc;
  const float* src = ws + P.ws_off + e;
  float s = 0.f;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int q = sub;
  for (; q + 48 < P.nsplit; q += 64) {  // 4 independent loads in flight per lane, fixed combination order
    s += src[(size_t)q * stride];
    s1 += src[(size_t)(q + 16) * stride];
    s2 += src[(size_t)(q + 32) * stride];
    s3 += src[(size_t)(q + 48) * stride];
  }
  for (; q < P.nsplit; q += 16) s += src[(size_t)q * stride];
  s = (s + s1) + (s2 + s3);
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 4);
  s += __shfl_xor(s, 8);
  if (!live || sub != 0) return;
  int n = e / Kc, k = e - n * Kc;
  if (k < P.K) {
    float* dst = P.dW + (size_t)n * P.ldw + k;
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
    P.delta = delta; P.ldd = ldd; P.act = act + c0; P.lda = lda;
    P.R = R; P.N = N; P.K = (c1 - c0) - (has_ones ? 1 : 0); P.ones = has_ones ? 1 : 0;
    P.dW = dW + c0; P.ldw = ldw; P.db = has_ones ? db : nullptr; P.db2 = has_ones ? db2 : nullptr;
    P.accumulate = accumulate;
    P.nbn = (N + 15) / 16;
    P.nbk = (c1 - c0 + 15) / 16;
  }
  return SW_OK;
}

// Cost model of one row slice: per 4-row group a wave issues NI x KT MFMAs (NI = its n-tiles) and the
// workgroup pays a roughly constant staging / barrier price worth ~16 MFMAs.  Slices are sized so
// that every workgroup of the launch carries the same cost: a narrow problem with many rows (bias-
// like shapes, K = 4) is staging-bound and must be cut as finely as a wide one.
static double wg_cost(const WgProblem& P) {
  int ni = (P.nbn + 3) / 4;
  return (double)P.R * (16.0 + ni * P.nbk);
}
double wg_total_work(const WgBatch& b) {
  double w = 0;
  for (int i = 0; i < b.np; ++i) w += wg_cost(b.p[i]);
  return w;
}

size_t wg_finalize(WgBatch& b) {
  // ~768 workgroups per launch (3 per CU by LDS and registers), equal cost each
  const double total = wg_total_work(b) + 1.0;
  size_t ws = 0;
  int job = 0, out = 0;
  for (int i = 0; i < b.np; ++i) {
    WgProblem& P = b.p[i];
    int ns = (int)(768.0 * wg_cost(P) / total + 0.5);
    int cap = (P.R + 31) / 32;  // at least 32 rows per slice
    if (ns > cap) ns = cap;
    if (ns > SW_WG_MAXSPLIT) ns = SW_WG_MAXSPLIT;
    if (ns < 1) ns = 1;
    P.nsplit = ns;
    P.job0 = job;
    job += ns;
    P.out0 = out;
    const int Kc = P.K + P.ones;
    out += P.N * Kc;
    P.ws_off = ws;
    ws += (size_t)ns * P.N * Kc;
  }
  b.total_jobs = job;
  b.total_out = out;
  return ws;
}

int wg_launch(WgBatch& b, float* ws, hipStream_t stream) {
  if (b.np == 0) return SW_OK;
  size_t need = wg_finalize(b);
  if (need > SW_WG_WS_FLOATS) return SW_ESHAPE;
  if (b.total_jobs == 0 || b.total_out == 0) return SW_OK;
  hipLaunchKernelGGL(wgrad_partial_kernel, dim3(b.total_jobs), dim3(SW_THREADS), WG_LDS_FLOATS * 4, stream, b, ws);
  SW_CHECK_LAUNCH("wgrad_partial_kernel");
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((b.total_out * 16 + 255) / 256), dim3(256), 0, stream, b, ws);
  SW_CHECK_LAUNCH("wgrad_reduce_kernel");
  return SW_OK;
}
