// sw_wgrad.hip - deferred weight gradients: a grouped, split-K "TN" GEMM
//
//        dW[N][K] (+)= sum_r delta[r][n] * act[r][k]        db[n] (+)= sum_r delta[r][n]
//
// over the time-major rows the serial BPTT kernels leave behind (what autograd does for the
// reference's nn.Linear / nn.LSTM weight grads, train.py:495,538).  One launch handles up to
// SW_WG_MAXP problems; each wave owns one 32x32 output block of one problem for one slice of the
// rows (2x2 MFMA tiles, K dimension = rows, 4 per instruction), writes its partial to the
// workspace, and a second kernel reduces the slices in a fixed order (deterministic - no float
// atomics).
#include "../../include/socialways_hip.h"
#include "sw_common.h"
#include "sw_wgrad.h"

__global__ __launch_bounds__(SW_THREADS) void wgrad_partial_kernel(WgBatch batch, float* __restrict__ ws) {
  const int lane = sw_lane(), ln = lane & 15, lg = lane >> 4;
  int job = blockIdx.x * 4 + sw_wave();
  if (job >= batch.total_jobs) return;
  int p = 0;
#pragma unroll 1
  while (p + 1 < batch.np && job >= batch.p[p + 1].job0) ++p;
  const WgProblem& P = batch.p[p];
  int j = job - P.job0;
  // job -> (split, block); the 4 waves of a workgroup take 4 consecutive blocks of one split so
  // they re-read the same rows through L1
  const int nblk = P.nbn * P.nbk;
  const int s = j / nblk;
  const int blk = j - s * nblk;
  const int bn = blk / P.nbk, bk = blk - bn * P.nbk;
  const int n0 = bn * 32, k0 = bk * 32;
  const int rows_per = (P.R + P.nsplit - 1) / P.nsplit;
  const int rbeg = s * ((rows_per + 3) & ~3);
  const int rend = min(P.R, rbeg + ((rows_per + 3) & ~3));
  const bool bias = P.db != nullptr && bk == 0;

  f32x4 acc[2][2], accb[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    accb[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool nok0 = n0 + ln < P.N, nok1 = n0 + 16 + ln < P.N;
  const bool kok0 = k0 + ln < P.K, kok1 = k0 + 16 + ln < P.K;
  const float* dp = P.delta + n0 + ln;
  const float* ap = P.act + k0 + ln;
  for (int r = rbeg; r < rend; r += 4) {
    const int rr = r + lg;
    const bool rok = rr < rend;
    const float* drow = dp + (size_t)rr * P.ldd;
    const float* arow = ap + (size_t)rr * P.lda;
    float a0 = (rok && nok0) ? drow[0] : 0.f;
    float a1 = (rok && nok1) ? drow[16] : 0.f;
    float b0 = (rok && kok0) ? arow[0] : 0.f;
    float b1 = (rok && kok1) ? arow[16] : 0.f;
    acc[0][0] = SW_MFMA(a0, b0, acc[0][0]);
    acc[0][1] = SW_MFMA(a0, b1, acc[0][1]);
    acc[1][0] = SW_MFMA(a1, b0, acc[1][0]);
    acc[1][1] = SW_MFMA(a1, b1, acc[1][1]);
    if (bias) {
      accb[0] = SW_MFMA(a0, 1.0f, accb[0]);
      accb[1] = SW_MFMA(a1, 1.0f, accb[1]);
    }
  }
  // partial layout: ws[P.ws_off + s*(N*K) + n*K + k]; bias: ws[P.wsb_off + s*N + n]
  float* out = ws + P.ws_off + (size_t)s * P.N * P.K;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int k = k0 + 16 * c + ln;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int n = n0 + 16 * a + 4 * lg + r;
        if (n < P.N && k < P.K) out[(size_t)n * P.K + k] = acc[a][c][r];
      }
    }
    if (bias && ln == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int n = n0 + 16 * a + 4 * lg + r;
        if (n < P.N) ws[P.wsb_off + (size_t)s * P.N + n] = accb[a][r];
      }
    }
  }
}

__global__ void wgrad_reduce_kernel(WgBatch batch, const float* __restrict__ ws) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch.total_out) return;
  int p = 0;
#pragma unroll 1
  while (p + 1 < batch.np && i >= batch.p[p + 1].out0) ++p;
  const WgProblem& P = batch.p[p];
  int e = i - P.out0;
  const int nk = P.N * P.K;
  if (e < nk) {
    const float* src = ws + P.ws_off + e;
    float s = 0.f;
    for (int q = 0; q < P.nsplit; ++q) s += src[(size_t)q * nk];
    int n = e / P.K, k = e - n * P.K;
    float* dst = P.dW + (size_t)n * P.ldw + k;
    *dst = P.accumulate ? *dst + s : s;
  } else {
    int n = e - nk;
    const float* src = ws + P.wsb_off + n;
    float s = 0.f;
    for (int q = 0; q < P.nsplit; ++q) s += src[(size_t)q * P.N];
    // a second bias vector (LSTM b_ih / b_hh share their gradient) is written too
    P.db[n] = P.accumulate ? P.db[n] + s : s;
    if (P.db2) P.db2[n] = P.accumulate ? P.db2[n] + s : s;
  }
}

// host side -------------------------------------------------------------------------------------
void wg_add(WgBatch& b, const float* delta, int ldd, const float* act, int lda, int R, int N, int K, float* dW,
            int ldw, float* db, float* db2, int accumulate) {
  WgProblem& P = b.p[b.np++];
  P.delta = delta; P.ldd = ldd; P.act = act; P.lda = lda;
  P.R = R; P.N = N; P.K = K; P.dW = dW; P.ldw = ldw; P.db = db; P.db2 = db2; P.accumulate = accumulate;
  P.nbn = (N + 31) / 32;
  P.nbk = (K + 31) / 32;
}

size_t wg_finalize(WgBatch& b) {
  // slices: aim at ~2048 wave-jobs per launch overall, at least 32 rows per slice
  size_t ws = 0;
  int job = 0, out = 0;
  for (int i = 0; i < b.np; ++i) {
    WgProblem& P = b.p[i];
    int nblk = P.nbn * P.nbk;
    // weight the split count by the row count so that long problems get more slices
    int want = (int)((2048.0 * ((double)P.R * nblk)) / (wg_total_work(b) + 1.0) / nblk + 0.5);
    int cap = (P.R + 31) / 32;
    int ns = want < 1 ? 1 : want;
    if (ns > cap) ns = cap;
    if (ns > SW_WG_MAXSPLIT) ns = SW_WG_MAXSPLIT;
    if (ns < 1) ns = 1;
    P.nsplit = ns;
    P.job0 = job;
    job += ns * nblk;
    P.out0 = out;
    out += P.N * P.K + (P.db ? P.N : 0);
    P.ws_off = ws;
    ws += (size_t)ns * P.N * P.K;
    P.wsb_off = ws;
    ws += P.db ? (size_t)ns * P.N : 0;
  }
  b.total_jobs = job;
  b.total_out = out;
  return ws;
}

double wg_total_work(const WgBatch& b) {
  double w = 0;
  for (int i = 0; i < b.np; ++i) w += (double)b.p[i].R * b.p[i].nbn * b.p[i].nbk;
  return w;
}

int wg_launch(WgBatch& b, float* ws, hipStream_t stream) {
  if (b.np == 0) return SW_OK;
  wg_finalize(b);
  if (b.total_jobs == 0 || b.total_out == 0) return SW_OK;
  hipLaunchKernelGGL(wgrad_partial_kernel, dim3((b.total_jobs + 3) / 4), dim3(SW_THREADS), 0, stream, b, ws);
  SW_CHECK_LAUNCH("wgrad_partial_kernel");
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((b.total_out + 255) / 256), dim3(256), 0, stream, b, ws);
  SW_CHECK_LAUNCH("wgrad_reduce_kernel");
  return SW_OK;
}
