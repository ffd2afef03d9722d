// sw_wide.hip - the WIDE path: the reference's model at hidden sizes above the fused kernels' 64 units (`--hidden-size`,
// train.py:42-44, 76-81: encoder / social / discriminator width H, noise H/2, decoder 2.5H -> 2.5H -> 1.25H -> 0.625H -> 2)
// as TIME-STEP-level kernels: one launch per LSTM step (gate products + cell in one kernel) and per decoder layer
// (product + bias + activation, or product + activation derivative on the way back), over all agents of the packed
// batch at once, driven by socialways_amd/wide.py with an explicit backward pass (time-major saved rows, deferred
// weight gradients through the grouped split-K GEMM of sw_wgrad.hip) and captured into one hipGraph per step.
//
// Why not the fused design: at 128 units W_hh alone is 256 KB - it fills the 512 registers per lane of a 4-wave
// workgroup - and the decoder's 320 x 320 + 160 x 320 weights (600 KB) fit neither registers nor the 160 KB LDS next to
// it, so a sequence kernel would stream 600 KB per tile and step from L2.  Here every product runs over the whole batch
// (2048 agents = 32 row blocks x N / 64 column blocks of workgroups), weights are read once per workgroup through L1 /
// L2, and the recurrence is carried by kernel boundaries.  Same tiling convention as the rest of the library
// (sw_common.h): D[16 units][16 agents] += W[16 units][K] X[K][16 agents] on v_mfma_f32_16x16x4_f32, a wave owns 16
// agents x 64 units (4 accumulators), lanes fetch float4s of their weight / activation rows (k = 16 j + 4 lg + r).
#include "../../include/socialways_hip.h"
#include "sw_common.h"
#include "sw_wgrad.h"
#include <cstddef>
#include <cstring>

namespace {
enum { EPI_NONE = 0, EPI_RELU = 1, EPI_LRELU = 2, EPI_DRELU = 3, EPI_DLRELU = 4 };

__device__ __forceinline__ float epi_apply(float v, int epi, float aux) {
  switch (epi) {
    case EPI_RELU: return fmaxf(v, 0.f);
    case EPI_LRELU: return sw_lrelu(v);
    case EPI_DRELU: return aux > 0.f ? v : 0.f;
    case EPI_DLRELU: return aux > 0.f ? v : 0.2f * v;
  }
  return v;
}

// y[r][n] = epi( sum_k x[r][k] w[n][k] + bias[n] + cin[r][n] ; aux[r][n] )
//   x element (r, k) at x + r x_rs + k x_cs, w element (n, k) at w + n w_rs + k w_cs.  XV / WV: the operand has unit k
//   stride, a row stride that is a multiple of 4 and K % 4 == 0 -> float4 loads (the model's layers); otherwise scalar
//   loads with free strides (the 3-wide pair features, transposed operands of the small composition products).
//   OV: N % 4 == 0 and every row stride of y / cin / aux a multiple of 4 -> float4 epilogue.
template <bool XV, bool WV, bool OV>
__global__ __launch_bounds__(256) void wide_gemm_kernel(const float* __restrict__ x, long long x_rs, int x_cs,
                                                        const float* __restrict__ w, long long w_rs, int w_cs,
                                                        const float* __restrict__ bias, const float* __restrict__ cin,
                                                        int cin_ld, const float* __restrict__ aux, int aux_ld, long long R,
                                                        int K, int N, float* __restrict__ y, int y_ld, int epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 15, lg = lane >> 4;
  const int ncb = (N + 63) >> 6;
  const long long rb = blockIdx.x / ncb;
  const int n0 = (int)(blockIdx.x - rb * ncb) * 64;
  const long long r0 = rb * 64 + 16 * wave;
  if (r0 >= R) return;
  const long long row = r0 + ln;
  const bool rv = row < R;
  const float* xr = x + (rv ? row : R - 1) * x_rs;
  const int nt = min(4, (N - n0 + 15) >> 4);        // live 16-unit tiles of this column block (wave-uniform)
  const float* wr[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wr[t] = w + (long long)min(n0 + 16 * t + ln, N - 1) * w_rs;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 16 * t + 4 * lg + q;
      acc[t][q] = (bias && n < N) ? bias[n] : 0.f;
    }
  }
  // Operands of k-step j (16 columns): one float4 of the lane's activation row and of each of its 4 weight rows.  Three
  // k-steps are in flight: the loads of step j + 2 are issued before the 16 matrix instructions of step j (a wave alone on
  // its SIMD - these grids are a few hundred workgroups - has nothing else to hide the L2 round trip behind).  Loads are
  // unconditional from clamped addresses; a k-step beyond K contributes zeros (its activation operand is masked).
  auto load = [&](int k0, f32x4& xv, f32x4 (&wv)[4]) {
    const int kk = k0 + 4 * lg;
    if constexpr (XV) {
      xv = ld4(xr + min(kk, K - 4));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[q] = xr[(long long)min(kk + q, K - 1) * x_cs];
    }
    if constexpr (WV) {
#pragma unroll
      for (int t = 0; t < 4; ++t) wv[t] = ld4(wr[t] + min(kk, K - 4));
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[t][q] = wr[t][(long long)min(kk + q, K - 1) * w_cs];
    }
  };
  auto mma = [&](int k0, f32x4 xv, const f32x4 (&wv)[4]) {
    if (k0 >= K) return;                             // wave-uniform; no memory operation inside
    const int kk = k0 + 4 * lg;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float b = kk + q < K ? xv[q] : 0.f;
      acc[0] = SW_MFMA(wv[0][q], b, acc[0]);
      if (nt > 1) acc[1] = SW_MFMA(wv[1][q], b, acc[1]);
      if (nt > 2) acc[2] = SW_MFMA(wv[2][q], b, acc[2]);
      if (nt > 3) acc[3] = SW_MFMA(wv[3][q], b, acc[3]);
    }
  };
  f32x4 xa, xb, xc, wa[4], wb[4], wc[4];
  load(0, xa, wa);
  load(16, xb, wb);
  for (int k0 = 0; k0 < K; k0 += 48) {
    load(k0 + 32, xc, wc);
    mma(k0, xa, wa);
    load(k0 + 48, xa, wa);
    mma(k0 + 16, xb, wb);
    load(k0 + 64, xb, wb);
    mma(k0 + 32, xc, wc);
  }
  if (!rv) return;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = n0 + 16 * t + 4 * lg;
    if (n >= N) continue;
    if constexpr (OV) {
      f32x4 v = acc[t];
      if (cin) v = v + ld4(cin + row * cin_ld + n);
      if (epi != EPI_NONE) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (epi >= EPI_DRELU) a = ld4(aux + row * aux_ld + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = epi_apply(v[q], epi, a[q]);
      }
      st4(y + row * y_ld + n, v);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (n + q < N) {
          float v = acc[t][q];
          if (cin) v += cin[row * cin_ld + n + q];
          v = epi_apply(v, epi, epi >= EPI_DRELU ? aux[row * aux_ld + n + q] : 0.f);
          y[row * y_ld + n + q] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// LDS-staged product core.  Lanes of an MFMA operand hold one ROW each (16 rows x 64-byte pieces per load instruction): read
// like that from global memory a wave instruction is 64 separate 16-byte accesses for the address unit, and the first
// version of these kernels - operands straight from global memory, three k-steps in flight - ran at 12 us for a
// 2048 x 320 x 320 product that holds 4.4 us of matrix instructions (rocprofv3: bound by address processing, not by
// latency: the prefetch depth changed nothing).  So a workgroup stages 64-column chunks of its activation rows (16 AT
// agents) and of its 64 weight rows into LDS with COALESCED loads (a wave instruction = 4 rows x 256 contiguous bytes),
// double-buffered (the global loads of chunk c + 1 are in flight under the products of chunk c, one barrier per chunk),
// and the waves read their row-per-lane operands from LDS (row stride 68 floats).
//   AT agent tiles per workgroup, wave -> agent tile wave % AT, unit group wave / AT of UT unit tiles each (4 / AT groups x
//   UT tiles = the 4 unit tiles of the 64 staged weight rows).  acc[UT] in/out.
//   xrow(i): pointer to activation row i of the tile (0 .. 16 AT - 1, clamped by the caller), wrow(i): weight row i (0..63).
// Requires K % 4 == 0, 16-byte aligned rows.
// ---------------------------------------------------------------------------------------------------------------------------
#define WIDE_LDS 68
template <int AT>
struct WideSmem {
  float xs[2][16 * AT][WIDE_LDS];
  float ws[2][64][WIDE_LDS];
};
template <int AT, int UT, class XRow, class WRow>
__device__ __forceinline__ void wide_core(WideSmem<AT>& sm, XRow xrow, WRow wrow, int K, f32x4 (&acc)[UT]) {
  static_assert((4 / AT) * UT == 4, "4 unit tiles per workgroup");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, ln = lane & 15, lg = lane >> 4;
  const int c4 = t & 15, rr = t >> 4;
  const int at = wave % AT, ug = wave / AT;
  constexpr int NX = AT;          // float4s of the activation tile per thread (16 AT rows x 16 float4 / 256 threads)
  const float* xp[NX];
  const float* wp[4];
#pragma unroll
  for (int i = 0; i < NX; ++i) xp[i] = xrow(rr + 16 * i) + 4 * c4;
#pragma unroll
  for (int i = 0; i < 4; ++i) wp[i] = wrow(rr + 16 * i) + 4 * c4;
  const int nch = (K + 63) >> 6;
  // TWO chunks of global loads in flight (register stages A, B): at 32 matrix instructions per wave and chunk (0.45 us) one
  // chunk ahead did not cover the L2 round trip - the backward LSTM step (K = 4H, 8 chunks) ran latency-bound at 1.2 us per
  // chunk.  Loads are unconditional (the last chunk is re-fetched by the prefetches beyond the end).
  f32x4 xa[NX], wa[4], xb[NX], wb[4];
  auto gload = [&](int c, f32x4 (&xr)[NX], f32x4 (&wr)[4]) {
    const int kc = min(64 * min(c, nch - 1) + 4 * c4, K - 4);
#pragma unroll
    for (int i = 0; i < NX; ++i) xr[i] = ld4(xp[i] + kc - 4 * c4);
#pragma unroll
    for (int i = 0; i < 4; ++i) wr[i] = ld4(wp[i] + kc - 4 * c4);
  };
  auto lstore = [&](int c, int buf, const f32x4 (&xr)[NX], const f32x4 (&wr)[4]) {
    const bool kv = 64 * c + 4 * c4 < K;       // columns beyond K: zeros (arithmetic on the loaded value, not a conditional load)
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NX; ++i) st4(&sm.xs[buf][rr + 16 * i][4 * c4], kv ? xr[i] : z);
#pragma unroll
    for (int i = 0; i < 4; ++i) st4(&sm.ws[buf][rr + 16 * i][4 * c4], kv ? wr[i] : z);
  };
  auto compute = [&](int c) {
    const int buf = c & 1;
    const int nj = min(4, (K - 64 * c + 15) >> 4);
    for (int j = 0; j < nj; ++j) {
      const f32x4 b = ld4(&sm.xs[buf][16 * at + ln][16 * j + 4 * lg]);
      f32x4 a[UT];
#pragma unroll
      for (int u = 0; u < UT; ++u) a[u] = ld4(&sm.ws[buf][16 * (ug * UT + u) + ln][16 * j + 4 * lg]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < UT; ++u) acc[u] = SW_MFMA(a[u][q], b[q], acc[u]);
    }
  };
  gload(0, xa, wa);
  gload(1, xb, wb);
  lstore(0, 0, xa, wa);
  gload(2, xa, wa);
  __syncthreads();
  // invariant at the top of an iteration for chunk c (even): LDS[c & 1] holds chunk c, stage B chunk c + 1, stage A chunk c + 2
  for (int c = 0; c < nch; c += 2) {
    compute(c);
    lstore(c + 1, (c + 1) & 1, xb, wb);
    gload(c + 3, xb, wb);
    __syncthreads();
    if (c + 1 < nch) compute(c + 1);
    lstore(c + 2, c & 1, xa, wa);
    gload(c + 4, xa, wa);
    __syncthreads();
  }
}

// y = epi(x W^T + bias + cin; aux) on the staged core: workgroup = 16 AT rows x 64 columns (AT = 2: wave = 16 rows x 32
// columns; AT = 1: 16 rows x 16 columns - twice the workgroups, for products that would leave half of the chip idle).
template <int AT, bool OV>
__global__ __launch_bounds__(256) void wide_gemm_lds_kernel(const float* __restrict__ x, long long x_rs, const float* __restrict__ w,
                                                            long long w_rs, const float* __restrict__ bias,
                                                            const float* __restrict__ cin, int cin_ld, const float* __restrict__ aux,
                                                            int aux_ld, long long R, int K, int N, float* __restrict__ y, int y_ld,
                                                            int epi) {
  constexpr int UT = AT;
  __shared__ WideSmem<AT> sm;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 15, lg = lane >> 4;
  const int ncb = (N + 63) >> 6;
  const long long rb = blockIdx.x / ncb;
  const int n0 = (int)(blockIdx.x - rb * ncb) * 64;
  const long long r0 = rb * 16 * AT;
  const int at = wave % AT, ug = wave / AT;
  f32x4 acc[UT];
#pragma unroll
  for (int u = 0; u < UT; ++u)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 16 * (UT * ug + u) + 4 * lg + q;
      acc[u][q] = (bias && n < N) ? bias[n] : 0.f;
    }
  wide_core<AT, UT>(sm, [&](int i) { return x + min(r0 + i, R - 1) * x_rs; },
                    [&](int i) { return w + (long long)min(n0 + i, N - 1) * w_rs; }, K, acc);
  const long long row = r0 + 16 * at + ln;
  if (row >= R) return;
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    const int n = n0 + 16 * (UT * ug + u) + 4 * lg;
    if (n >= N) continue;
    if constexpr (OV) {
      f32x4 v = acc[u];
      if (cin) v = v + ld4(cin + row * cin_ld + n);
      if (epi != EPI_NONE) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (epi >= EPI_DRELU) a = ld4(aux + row * aux_ld + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = epi_apply(v[q], epi, a[q]);
      }
      st4(y + row * y_ld + n, v);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (n + q < N) {
          float v = acc[u][q];
          if (cin) v += cin[row * cin_ld + n + q];
          v = epi_apply(v, epi, epi >= EPI_DRELU ? aux[row * aux_ld + n + q] : 0.f);
          y[row * y_ld + n + q] = v;
        }
      }
    }
  }
}

// Products with a contracted width of at most 8 (the heads' K = 1 / 2 / n_latent back-products, the 3-wide pair features,
// the rank-1 bias terms of the composition): no matrix instruction pays here - one thread per output element, K fused
// multiply-adds, same operand addressing and epilogue as wide_gemm_kernel.  (The general kernel spent 10 us on each.)
__global__ __launch_bounds__(256) void wide_smallk_kernel(const float* __restrict__ x, long long x_rs, int x_cs,
                                                          const float* __restrict__ w, long long w_rs, int w_cs,
                                                          const float* __restrict__ bias, const float* __restrict__ cin, int cin_ld,
                                                          const float* __restrict__ aux, int aux_ld, long long R, int K, int N,
                                                          float* __restrict__ y, int y_ld, int epi) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= R * N) return;
  const long long r = e / N;
  const int n = (int)(e - r * N);
  float v = bias ? bias[n] : 0.f;
  for (int k = 0; k < K; ++k) v = fmaf(x[r * x_rs + (long long)k * x_cs], w[n * w_rs + (long long)k * w_cs], v);
  if (cin) v += cin[r * cin_ld + n];
  y[r * y_ld + n] = epi_apply(v, epi, epi >= EPI_DRELU ? aux[r * aux_ld + n] : 0.f);
}

// One LSTM step for all agents (nn.LSTM gate order i f g o, train.py:254, 278):
//   pre = Wx x4 + b1 (+ b2) + Whh h_prev ; c' = f c + i g ; h' = o tanh(c')
// Wx [4H][4] is the 4-d input's matrix (the encoder's composed W_ih W_embed, the discriminator's W_ih), Whh [4H][H].
// A wave owns 16 agents x 16 hidden units of all four gates (the cell update is lane-local); a workgroup = 4 waves = 64
// agents of one unit block (its weight rows are shared through L1).
__global__ __launch_bounds__(256) void wide_lstm_fwd_kernel(const float* __restrict__ x4, int x_ld, const float* __restrict__ h_prev,
                                                            int hp_ld, const float* __restrict__ c_prev,
                                                            const float* __restrict__ Wx, const float* __restrict__ b1,
                                                            const float* __restrict__ b2, const float* __restrict__ Whh,
                                                            int B, int H, float* __restrict__ gates, float* __restrict__ c_out,
                                                            float* __restrict__ h_out, int h_ld, float* __restrict__ h_out2,
                                                            int h2_ld) {
  __shared__ WideSmem<4> sm;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 15, lg = lane >> 4;
  const int nub = H >> 4;
  const int rb = blockIdx.x / nub, u0 = (blockIdx.x - rb * nub) * 16;
  const int r0 = rb * 64;
  const int row = r0 + 16 * wave + ln;
  const bool rv = row < B;
  const int rc = rv ? row : B - 1;
  f32x4 acc[4];
  const float xb = x4[(size_t)rc * x_ld + lg];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 b = ld4(b1 + g * H + u0 + 4 * lg);
    if (b2) b = b + ld4(b2 + g * H + u0 + 4 * lg);
    acc[g] = SW_MFMA(Wx[(size_t)(g * H + u0 + ln) * 4 + lg], xb, b);
  }
  if (h_prev)      // staged weight row i = gate i / 16, unit u0 + i % 16; all waves of the workgroup take part (barriers inside)
    wide_core<4, 4>(sm, [&](int i) { return h_prev + (size_t)min(r0 + i, B - 1) * hp_ld; },
                    [&](int i) { return Whh + ((size_t)(i >> 4) * H + u0 + (i & 15)) * H; }, H, acc);
  if (!rv) return;
  f32x4 cp = {0.f, 0.f, 0.f, 0.f};
  if (c_prev) cp = ld4(c_prev + (size_t)row * H + u0 + 4 * lg);
  f32x4 gi, gf, gg, go, cn, hn;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    gi[q] = sw_sigmoid(acc[0][q]);
    gf[q] = sw_sigmoid(acc[1][q]);
    gg[q] = sw_tanh(acc[2][q]);
    go[q] = sw_sigmoid(acc[3][q]);
    cn[q] = fmaf(gf[q], cp[q], gi[q] * gg[q]);
    hn[q] = go[q] * sw_tanh(cn[q]);
  }
  float* gr = gates + (size_t)row * 4 * H + u0 + 4 * lg;
  st4(gr, gi);
  st4(gr + H, gf);
  st4(gr + 2 * H, gg);
  st4(gr + 3 * H, go);
  st4(c_out + (size_t)row * H + u0 + 4 * lg, cn);
  st4(h_out + (size_t)row * h_ld + u0 + 4 * lg, hn);
  if (h_out2) st4(h_out2 + (size_t)row * h2_ld + u0 + 4 * lg, hn);
}

// Backward of one LSTM step for all agents: dh = dh_ext + dgates_{t+1} Whh (the recurrent path; WhhT [H][4H] is the
// transposed matrix), then the element-wise cell backward -> dgates_t (pre-activation gradients, the rows the weight
// gradients contract over) and dc_{t-1}.  A wave owns 16 agents x 16 units; a workgroup = the 4 unit tiles of one
// 64-unit block for the same 16 agents (the dgates rows are shared through L1).
template <int AT>       // agent tiles per workgroup: 2 (32 agents x 64 units) or 1 (16 x 64: twice the workgroups for small batches)
__global__ __launch_bounds__(256) void wide_lstm_bwd_kernel(const float* __restrict__ dh_ext, int dhe_ld,
                                                            const float* __restrict__ dh_ext2, int dhe2_ld,
                                                            const float* __restrict__ dg_next, const float* __restrict__ WhhT,
                                                            const float* __restrict__ gates, const float* __restrict__ c,
                                                            const float* __restrict__ c_prev, const float* __restrict__ dc_in,
                                                            int B, int H, float* __restrict__ dgates, float* __restrict__ dc_out) {
  constexpr int UT = AT;          // unit tiles per wave (4 / AT unit groups x UT = 4)
  __shared__ WideSmem<AT> sm;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 15, lg = lane >> 4;
  const int nub = (H + 63) >> 6;
  const int rb = blockIdx.x / nub, ub = (blockIdx.x - rb * nub) * 64;
  const int r0 = rb * 16 * AT;
  const int at = wave % AT, ug = wave / AT;
  f32x4 acc[UT];
#pragma unroll
  for (int u = 0; u < UT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (dg_next)
    wide_core<AT, UT>(sm, [&](int i) { return dg_next + (size_t)min(r0 + i, B - 1) * 4 * H; },
                      [&](int i) { return WhhT + (size_t)min(ub + i, H - 1) * 4 * H; }, 4 * H, acc);
  const int row = r0 + 16 * at + ln;
  if (row >= B) return;
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    const int u0 = ub + 16 * (UT * ug + u);
    if (u0 >= H) continue;
    f32x4 dh = acc[u];
    const size_t e = (size_t)row * H + u0 + 4 * lg;
    if (dh_ext) dh = dh + ld4(dh_ext + (size_t)row * dhe_ld + u0 + 4 * lg);
    if (dh_ext2) dh = dh + ld4(dh_ext2 + (size_t)row * dhe2_ld + u0 + 4 * lg);
    const float* gr = gates + (size_t)row * 4 * H + u0 + 4 * lg;
    const f32x4 gi = ld4(gr), gf = ld4(gr + H), gg = ld4(gr + 2 * H), go = ld4(gr + 3 * H), ct = ld4(c + e);
    f32x4 cp = {0.f, 0.f, 0.f, 0.f}, dc = cp;
    if (c_prev) cp = ld4(c_prev + e);
    if (dc_in) dc = ld4(dc_in + e);
    f32x4 di, df, dg, dO, dcp;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float tc = sw_tanh(ct[q]);
      const float dct = fmaf(dh[q] * go[q], 1.0f - tc * tc, dc[q]);
      di[q] = dct * gg[q] * gi[q] * (1.0f - gi[q]);
      df[q] = dct * cp[q] * gf[q] * (1.0f - gf[q]);
      dg[q] = dct * gi[q] * (1.0f - gg[q] * gg[q]);
      dO[q] = dh[q] * tc * go[q] * (1.0f - go[q]);
      dcp[q] = dct * gf[q];
    }
    float* dq = dgates + (size_t)row * 4 * H + u0 + 4 * lg;
    st4(dq, di);
    st4(dq + H, df);
    st4(dq + 2 * H, dg);
    st4(dq + 3 * H, dO);
    st4(dc_out + e, dcp);
  }
}

// Last decoder layer + position integration of a decode step (train.py:330, 422-424): v = a3 W4^T + b4 (2 outputs, K = D3:
// a dot product per agent - 16 lanes per agent, shuffle tree), p_i = p_{i-1} + v_i; the prediction row (p, v) agent-major
// and the 4-d input of the re-fed encoder step time-major.  One launch instead of a product, an integration and two copies.
__global__ __launch_bounds__(256) void wide_out_fwd_kernel(const float* __restrict__ a3, int D3, const float* __restrict__ W4,
                                                           const float* __restrict__ b4, float* __restrict__ p, int B,
                                                           float* __restrict__ pred4_i, int pred_ld, float* __restrict__ x4_tm) {
  const int b = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const int bc = min(b, B - 1);
  float vx = 0.f, vy = 0.f;
  for (int k = l; k < D3; k += 16) {
    const float a = a3[(size_t)bc * D3 + k];
    vx = fmaf(a, W4[k], vx);
    vy = fmaf(a, W4[D3 + k], vy);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    vx += __shfl_xor(vx, o);
    vy += __shfl_xor(vy, o);
  }
  if (b >= B || l != 0) return;
  vx += b4[0];
  vy += b4[1];
  const float px = p[2 * b] + vx, py = p[2 * b + 1] + vy;
  p[2 * b] = px;
  p[2 * b + 1] = py;
  const f32x4 r = {px, py, vx, vy};
  st4(pred4_i + (size_t)b * pred_ld, r);
  if (x4_tm) st4(x4_tm + (size_t)b * 4, r);
}
// ... and the backward of that step: dx4 = dgates_t Wx (the re-fed encoder step's input gradient, K = 4H: a dot product per
// agent and component against WxT [4][4H]), d p_i = dpred.p + dx4.p + d p_{i+1}, d v_i = dpred.v + dx4.v + d p_i, and
// dz3 = dv W4 (rank 2).  16 lanes per agent.
__global__ __launch_bounds__(256) void wide_out_bwd_kernel(const float* __restrict__ dpred4_i, int pred_ld,
                                                           const float* __restrict__ dg, const float* __restrict__ WxT, int H4,
                                                           float* __restrict__ dp_run, int B, float* __restrict__ dv,
                                                           const float* __restrict__ W4, int D3, float* __restrict__ dz3) {
  const int b = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const int bc = min(b, B - 1);
  f32x4 g = ld4(dpred4_i + (size_t)bc * pred_ld);
  if (dg) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const float* dr = dg + (size_t)bc * H4;
    for (int k = 4 * l; k < H4; k += 64) {
      const f32x4 d = ld4(dr + k), w0 = ld4(WxT + k), w1 = ld4(WxT + H4 + k), w2 = ld4(WxT + 2 * H4 + k), w3 = ld4(WxT + 3 * H4 + k);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s0 = fmaf(d[q], w0[q], s0);
        s1 = fmaf(d[q], w1[q], s1);
        s2 = fmaf(d[q], w2[q], s2);
        s3 = fmaf(d[q], w3[q], s3);
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      s0 += __shfl_xor(s0, o);
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
      s3 += __shfl_xor(s3, o);
    }
    g = g + f32x4{s0, s1, s2, s3};
  }
  const float dpx = g[0] + dp_run[2 * bc], dpy = g[1] + dp_run[2 * bc + 1];
  const float dvx = g[2] + dpx, dvy = g[3] + dpy;
  if (b >= B) return;
  for (int k = l; k < D3; k += 16) dz3[(size_t)b * D3 + k] = fmaf(dvx, W4[k], dvy * W4[D3 + k]);
  if (l == 0) {
    dp_run[2 * b] = dpx;
    dp_run[2 * b + 1] = dpy;
    st4(dv + (size_t)b * 4, f32x4{dvx, dvy, 0.f, 0.f});
  }
}

// out[r][c] = sum_t in[t][r][c] (fixed order), 2-d blocks with row strides
__global__ __launch_bounds__(256) void wide_sum_steps_kernel(const float* __restrict__ in, long long t_stride, int in_ld, int T,
                                                             long long R, int C, float* __restrict__ out, int out_ld) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= R * C) return;
  const long long r = e / C;
  const int cc = (int)(e - r * C);
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += in[t * t_stride + r * in_ld + cc];
  out[r * out_ld + cc] = s;
}
// dst[tab.dst + c rows + r] = src[tab.src + r cols + c] for every matrix of the table (int4: src offset, rows, cols, dst
// offset; all in floats): the transposed copies the backward products read (dx = dy W wants W^T rows), one launch for all
// matrices of a module.  One workgroup per 32 x 32 tile, through LDS (coalesced on both sides).
__global__ __launch_bounds__(256) void wide_transpose_kernel(const float* __restrict__ src, const int4* __restrict__ tab, int ntab,
                                                             float* __restrict__ dst) {
  __shared__ float tile[32][33];
  int b = blockIdx.x, m = 0;
  for (; m < ntab; ++m) {
    const int tiles = ((tab[m].y + 31) >> 5) * ((tab[m].z + 31) >> 5);
    if (b < tiles) break;
    b -= tiles;
  }
  if (m >= ntab) return;
  const int4 T = tab[m];
  const int tc = (T.z + 31) >> 5;
  const int r0 = (b / tc) * 32, c0 = (b % tc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < T.y && c0 + tx < T.z) tile[i][tx] = src[T.x + (size_t)(r0 + i) * T.z + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < T.z && r0 + tx < T.y) dst[T.w + (size_t)(c0 + i) * T.y + r0 + tx] = tile[tx][i];
}

// MFMA A-operand IMAGES of weight matrices (as swimg::OP_* of the fused path, sw_common.h): for a matrix Mx [R][K] - M
// itself or its transpose - the float4 of (row tile t, k-step j, lane l) = Mx[16 t + (l & 15)][16 j + 4 (l >> 4) .. + 3] at
// dst + ((t K/16 + j) 64 + l) 4, so that a wave loads a tile's operand as 1 KB of consecutive memory.  The LSTM sequence
// kernels below hold W_hh (forward) / W_hh^T (BPTT) in registers for a whole sequence and load them from these images.
// table entry (6 ints): src offset, R, K, dst offset, transposed (Mx = M^T, M is [K][R]), source row stride (0: dense).  One
// thread per float4.
__global__ __launch_bounds__(256) void wide_opimage_kernel(const float* __restrict__ src, const int* __restrict__ tab, int ntab,
                                                           float* __restrict__ dst) {
  long long f = (long long)blockIdx.x * 256 + threadIdx.x;
  int m = 0;
  for (; m < ntab; ++m) {
    const long long n4 = (long long)tab[6 * m + 1] * tab[6 * m + 2] / 4;
    if (f < n4) break;
    f -= n4;
  }
  if (m >= ntab) return;
  const int so = tab[6 * m], K = tab[6 * m + 2], dofs = tab[6 * m + 3], tr = tab[6 * m + 4], R = tab[6 * m + 1];
  const int ld = tab[6 * m + 5] ? tab[6 * m + 5] : (tr ? R : K);      // row stride of the source (a column block of a wider matrix)
  const int KJ = K >> 4;
  const int l = (int)(f & 63);
  const long long tj = f >> 6;
  const int t = (int)(tj / KJ), j = (int)(tj - (long long)t * KJ);
  const int r = 16 * t + (l & 15), k = 16 * j + 4 * (l >> 4);
  f32x4 v;
  if (!tr) v = ld4(src + so + (size_t)r * ld + k);
  else {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = src[so + (size_t)(k + q) * ld + r];
  }
  st4(dst + dofs + 4 * f, v);
}

// ---------------------------------------------------------------------------------------------------------------------------
// LSTM SEQUENCE kernels for H = 64 NU (NU = 1, 2): the T steps of one observation sequence in ONE launch per 16-agent tile,
// W_hh (forward) / W_hh^T (BPTT) register-resident for the whole sequence - at 128 units that is 256 of the 512 registers a
// lane owns at one wave per SIMD -, h / dgates exchanged through LDS tiles, one barrier per step: the design of
// enc_lstm_fwd/bwd_kernel (sw_lstm.hip) at twice the width.  Replaces T launches of wide_lstm_fwd / _bwd (9.5 / 13.7 us each)
// by 4.1 us per step.  Wave w owns the unit tiles NU w .. NU w + NU - 1 of all four gates.
// ---------------------------------------------------------------------------------------------------------------------------
template <int NU>
__global__ __launch_bounds__(256) void wide_lstm_seq_fwd_kernel(const float* __restrict__ x4 /*[T][B][4]*/,
                                                                const float* __restrict__ Wx, const float* __restrict__ b1,
                                                                const float* __restrict__ b2, const float* __restrict__ whh_img,
                                                                int B, int T, float* __restrict__ gates /*[T][B][4H]*/,
                                                                float* __restrict__ cs /*[T][B][H]*/,
                                                                float* __restrict__ hs /*[T+1][B][H], slab 0 = h_0*/,
                                                                float* __restrict__ h_last2, int h2_ld) {
  constexpr int H = 64 * NU, KJ = H / 16, HLD = H + 4, UT = 4 * NU;     // UT unit tiles per gate
  __shared__ __attribute__((aligned(16))) float hbuf[2][16 * HLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 15, lg = lane >> 4;
  const int a0 = blockIdx.x * 16;
  const int b = min(a0 + ln, B - 1);          // padding lanes of the last tile: replicas of agent B - 1 (same values, same rows)
  f32x4 whh[4][NU][KJ];
  float wx[4][NU];
  f32x4 bias[4][NU];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      const int tile = g * UT + NU * wave + k;
#pragma unroll
      for (int j = 0; j < KJ; ++j) whh[g][k][j] = ld4(whh_img + (((size_t)tile * KJ + j) * 64 + lane) * 4);
      const int u0 = 16 * (NU * wave + k);
      wx[g][k] = Wx[(size_t)(g * H + u0 + ln) * 4 + lg];
      bias[g][k] = ld4(b1 + g * H + u0 + 4 * lg);
      if (b2) bias[g][k] = bias[g][k] + ld4(b2 + g * H + u0 + 4 * lg);
    }
  f32x4 c[NU], h[NU];
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    c[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    h[k] = ld4(hs + (size_t)b * H + 16 * (NU * wave + k) + 4 * lg);      // h_0 (zeros for the model's sequences)
    st4(&hbuf[0][ln * HLD + 16 * (NU * wave + k) + 4 * lg], h[k]);
  }
  float xa = x4[(size_t)b * 4 + lg];
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const float xb = xa;
    xa = x4[((size_t)min(t + 1, T - 1) * B + b) * 4 + lg];             // the next step's input, in flight under this step
    const float* hrow = &hbuf[t & 1][ln * HLD + 4 * lg];
    f32x4 acc[4][NU];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < NU; ++k) acc[g][k] = SW_MFMA(wx[g][k], xb, bias[g][k]);
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const f32x4 hv = ld4(hrow + 16 * j);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int k = 0; k < NU; ++k) acc[g][k] = SW_MFMA(whh[g][k][j][q], hv[q], acc[g][k]);
    }
    float* grow = gates + ((size_t)t * B + b) * 4 * H + 4 * lg;
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      const int u0 = 16 * (NU * wave + k);
      f32x4 gi, gf, gg, go;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gi[q] = sw_sigmoid(acc[0][k][q]);
        gf[q] = sw_sigmoid(acc[1][k][q]);
        gg[q] = sw_tanh(acc[2][k][q]);
        go[q] = sw_sigmoid(acc[3][k][q]);
        c[k][q] = fmaf(gf[q], c[k][q], gi[q] * gg[q]);
        h[k][q] = go[q] * sw_tanh(c[k][q]);
      }
      st4(&hbuf[(t + 1) & 1][ln * HLD + u0 + 4 * lg], h[k]);
      st4(grow + u0, gi);
      st4(grow + H + u0, gf);
      st4(grow + 2 * H + u0, gg);
      st4(grow + 3 * H + u0, go);
      st4(cs + ((size_t)t * B + b) * H + u0 + 4 * lg, c[k]);
      st4(hs + ((size_t)(t + 1) * B + b) * H + u0 + 4 * lg, h[k]);
    }
    sw_barrier();
  }
  if (h_last2) {
#pragma unroll
    for (int k = 0; k < NU; ++k) st4(h_last2 + (size_t)b * h2_ld + 16 * (NU * wave + k) + 4 * lg, h[k]);
  }
}

// BPTT of a sequence: for t = T-1 .. 0: dh_t = [t == T-1: dh_ext + dh_ext2] + W_hh^T dgates_{t+1} (dgates_T = dg_init or none),
// cell backward with the saved rows -> dgates_t (to memory through an LDS tile, one agent row per store instruction, and
// kept in LDS as the next step's operand), dc carried in registers (dc_T = dc_init or 0).
template <int NU>
__global__ __launch_bounds__(256) void wide_lstm_seq_bwd_kernel(const float* __restrict__ dh_ext, int dhe_ld,
                                                                const float* __restrict__ dh_ext2, int dhe2_ld,
                                                                const float* __restrict__ dg_init, const float* __restrict__ dc_init,
                                                                const float* __restrict__ whhT_img, const float* __restrict__ gates,
                                                                const float* __restrict__ cs, int B, int T,
                                                                float* __restrict__ dgates) {
  constexpr int H = 64 * NU, K = 4 * H, KJ = K / 16, GLD = K + 4;
  __shared__ __attribute__((aligned(16))) float dgbuf[2][16 * GLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 15, lg = lane >> 4;
  const int a0 = blockIdx.x * 16;
  const int b = min(a0 + ln, B - 1);
  f32x4 wT[NU][KJ];
#pragma unroll
  for (int k = 0; k < NU; ++k)
#pragma unroll
    for (int j = 0; j < KJ; ++j) wT[k][j] = ld4(whhT_img + (((size_t)(NU * wave + k) * KJ + j) * 64 + lane) * 4);
  f32x4 dc[NU], dh[NU];
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int u0 = 16 * (NU * wave + k);
    dc[k] = dc_init ? ld4(dc_init + (size_t)b * H + u0 + 4 * lg) : f32x4{0.f, 0.f, 0.f, 0.f};
    dh[k] = dh_ext ? ld4(dh_ext + (size_t)b * dhe_ld + u0 + 4 * lg) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (dh_ext2) dh[k] = dh[k] + ld4(dh_ext2 + (size_t)b * dhe2_ld + u0 + 4 * lg);
  }
  // dgates_T (the step behind the sequence, if any) into the tile the first iteration reads: a wave copies 4 agents' rows
  {
    float* tile = dgbuf[T & 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int a = 4 * wave + q, bb = min(a0 + a, B - 1);
      for (int c4 = lane; c4 < K / 4; c4 += 64)
        st4(tile + a * GLD + 4 * c4, dg_init ? ld4(dg_init + (size_t)bb * K + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f});
    }
  }
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    // the saved rows of step t are requested FIRST: their round trip runs under the 256 matrix instructions below
    f32x4 gi[NU], gf[NU], gg[NU], go[NU], ct[NU], cp[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      const int u0 = 16 * (NU * wave + k);
      const float* gr = gates + ((size_t)t * B + b) * K + u0 + 4 * lg;
      gi[k] = ld4(gr); gf[k] = ld4(gr + H); gg[k] = ld4(gr + 2 * H); go[k] = ld4(gr + 3 * H);
      ct[k] = ld4(cs + ((size_t)t * B + b) * H + u0 + 4 * lg);
      cp[k] = ld4(cs + ((size_t)max(t - 1, 0) * B + b) * H + u0 + 4 * lg);      // unconditional; zeroed below for t = 0
    }
    // recurrent part: dh += W_hh^T dgates_{t+1} from the LDS tile (zeros when nothing follows)
    {
      const float* drow = &dgbuf[(t + 1) & 1][ln * GLD + 4 * lg];
      f32x4 acc[NU], acc1[NU];
#pragma unroll
      for (int k = 0; k < NU; ++k) acc[k] = acc1[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j0 = 0; j0 < KJ; j0 += 8) {
        f32x4 d[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = ld4(drow + 16 * (j0 + j));
#pragma unroll
        for (int j = 0; j < 8; j += 2)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < NU; ++k) {
              acc[k] = SW_MFMA(wT[k][j0 + j][q], d[j][q], acc[k]);
              acc1[k] = SW_MFMA(wT[k][j0 + j + 1][q], d[j + 1][q], acc1[k]);
            }
      }
#pragma unroll
      for (int k = 0; k < NU; ++k) dh[k] = dh[k] + (acc[k] + acc1[k]);
    }
    float* tile = dgbuf[t & 1];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      const int u0 = 16 * (NU * wave + k);
      if (t == 0) cp[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 di, df, dg, dO;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float tc = sw_tanh(ct[k][q]);
        const float dct = fmaf(dh[k][q] * go[k][q], 1.0f - tc * tc, dc[k][q]);
        di[q] = dct * gg[k][q] * gi[k][q] * (1.0f - gi[k][q]);
        df[q] = dct * cp[k][q] * gf[k][q] * (1.0f - gf[k][q]);
        dg[q] = dct * gi[k][q] * (1.0f - gg[k][q] * gg[k][q]);
        dO[q] = dh[k][q] * tc * go[k][q] * (1.0f - go[k][q]);
        dc[k][q] = dct * gf[k][q];
      }
      float* tr = tile + ln * GLD + u0 + 4 * lg;
      st4(tr, di);
      st4(tr + H, df);
      st4(tr + 2 * H, dg);
      st4(tr + 3 * H, dO);
      dh[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    sw_barrier();      // LDS only: a full __syncthreads() would also drain the 32 KB of dgates rows stored in the previous step
    // dgates_t rows to memory from the tile: a wave writes 4 agents' rows, consecutive lanes consecutive float4s
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int a = 4 * wave + q, bb = min(a0 + a, B - 1);
      for (int c4 = lane; c4 < K / 4; c4 += 64) st4(dgates + ((size_t)t * B + bb) * K + 4 * c4, ld4(tile + a * GLD + 4 * c4));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The DECODE LOOP of predict() at 128 hidden units (train.py:415-432) as ONE persistent launch per 16-agent tile - the wide
// counterpart of dec_rollout_fwd_kernel.  Nothing is weight-resident: per decode step a workgroup STREAMS the operand
// images (sw_wide_opimage) of W1[:, :H] (320 x 128), W2 (160 x 320), W3 (80 x 160) and W_hh (512 x 128) - 0.66 MB - from L2
// straight into MFMA A operands, 8 image loads (1 KB each) in flight per wave, while the activations of the tile move
// between the layers through LDS tiles (one barrier per layer).  u = W1[:, H:] [S; z] + b1 is constant over the steps
// (train.py:411, 421) and comes in as the initial accumulators of layer 1.  Per step and wave: 744 matrix instructions
// (11 us) instead of five launches (36 us).  Leaves exactly the rows the step-level kernels leave (a1 / a2 / a3, the
// re-fed encoder steps' gates / c / h, x4, the prediction).
// ---------------------------------------------------------------------------------------------------------------------------
struct WideDecFwd {
  const float *w1h_img, *w2_img, *w3_img, *whh_img;     // operand images
  const float *u /*[B][D1]*/, *b2, *b3, *W4 /*[2][D3]*/, *b4, *Wx /*[4H][4]*/, *bx1, *bx2 /*[4H]*/;
  const float* p0;                                       // [B][p0_ld]: last observed position
  int p0_ld;
  float *a1, *a2, *a3;                                   // [Tp][B][D1 / D2 / D3]
  float* pred4;                                          // [B][Tp][4]
  float* x4;                                             // [To + Tp][B][4] time-major: rows To .. To + Tp - 1 written
  float *gates, *cs, *hs;                                // [Ta][B][4H], [Ta][B][H], [Ta + 1][B][H] (hs[t + 1] = h_t)
  float* cat;                                            // [Tp][B][D1]: columns 0 .. H-1 of slab i + 1 receive h of step To + i
  int B, To, Tp;
};
// acc[t] += sum_j img(tile[t], j) x b_j for the tiles of one layer: the loads of the linear sequence (j, t) run PD ahead
template <int NT, int KJ, int PD>
__device__ __forceinline__ void wide_stream_mm(const float* __restrict__ img, const int (&tile)[NT], const float* brow,
                                               f32x4 (&acc)[NT], int lane) {
  constexpr int N = NT * KJ;
  f32x4 ring[PD];
  auto addr = [&](int n) {
    const int nn = n < N ? n : N - 1;
    return img + (((size_t)tile[nn % NT] * KJ + nn / NT) * 64 + lane) * 4;
  };
#pragma unroll
  for (int n = 0; n < PD; ++n) ring[n] = ld4(addr(n));
  f32x4 bnext = ld4(brow);
#pragma unroll
  for (int j = 0; j < KJ; ++j) {
    const f32x4 b = bnext;
    bnext = ld4(brow + 16 * (j + 1 < KJ ? j + 1 : j));      // the next k-step's activation operand (LDS), one step ahead
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = j * NT + t;
      const f32x4 a = ring[n % PD];
      ring[n % PD] = ld4(addr(n + PD));
      asm volatile("" ::: "memory");      // the refill is issued HERE: neither hoisted (the unrolled loop would otherwise
#pragma unroll                            // request the whole layer at once: 400 registers) nor sunk to its use
      for (int q = 0; q < 4; ++q) acc[t] = SW_MFMA(a[q], b[q], acc[t]);
    }
  }
}
__global__ __launch_bounds__(256) void wide_dec_loop_fwd_kernel(WideDecFwd A) {
  constexpr int H = 128, D1 = 320, D2 = 160, D3 = 80, PD = 16;
  constexpr int HL = H + 4, L1 = D1 + 4, L2 = D2 + 4, L3 = D3 + 4;
  __shared__ __attribute__((aligned(16))) float hb[2][16 * HL];
  __shared__ __attribute__((aligned(16))) float a1b[16 * L1];
  __shared__ __attribute__((aligned(16))) float a2b[16 * L2];
  __shared__ __attribute__((aligned(16))) float a3b[16 * L3];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ln = lane & 15, lg = lane >> 4;
  const int B = A.B, To = A.To, Tp = A.Tp;
  const int a0 = blockIdx.x * 16;
  const int b = min(a0 + ln, B - 1);            // padding lanes of the last tile: replicas of agent B - 1
  // tiles of this wave per layer (duplicates compute and store the same values)
  const int t1[5] = {5 * wave, 5 * wave + 1, 5 * wave + 2, 5 * wave + 3, 5 * wave + 4};
  const int t2[3] = {wave, wave + 4, 8 + (wave & 1)};
  const int t3[2] = {wave, 4};
  int tl[8];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    tl[2 * g] = g * 8 + 2 * wave;
    tl[2 * g + 1] = g * 8 + 2 * wave + 1;
  }
  // constants of the tile / wave
  f32x4 u[5], bias2[3], bias3[2], biasl[8], w4[2][5];
  float wx[8];
#pragma unroll
  for (int t = 0; t < 5; ++t) u[t] = ld4(A.u + (size_t)b * D1 + 16 * t1[t] + 4 * lg);
#pragma unroll
  for (int t = 0; t < 3; ++t) bias2[t] = ld4(A.b2 + 16 * t2[t] + 4 * lg);
#pragma unroll
  for (int t = 0; t < 2; ++t) bias3[t] = ld4(A.b3 + 16 * t3[t] + 4 * lg);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int row0 = (t >> 1) * H + 16 * (2 * wave + (t & 1));       // gate t / 2, unit tile 2 wave + (t & 1)
    biasl[t] = ld4(A.bx1 + row0 + 4 * lg) + ld4(A.bx2 + row0 + 4 * lg);
    wx[t] = A.Wx[(size_t)(row0 + ln) * 4 + lg];
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    w4[0][j] = ld4(A.W4 + 20 * lg + 4 * j);
    w4[1][j] = ld4(A.W4 + D3 + 20 * lg + 4 * j);
  }
  const float b4x = A.b4[0], b4y = A.b4[1];
  float px = A.p0[(size_t)b * A.p0_ld], py = A.p0[(size_t)b * A.p0_ld + 1];
  f32x4 c[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int u0 = 16 * (2 * wave + k);
    c[k] = ld4(A.cs + ((size_t)(To - 1) * B + b) * H + u0 + 4 * lg);
    st4(&hb[0][ln * HL + u0 + 4 * lg], ld4(A.hs + ((size_t)To * B + b) * H + u0 + 4 * lg));
  }
  __syncthreads();
  for (int i = 0; i < Tp; ++i) {
    const float* hrow = &hb[i & 1][ln * HL + 4 * lg];
    // the image bases are made opaque once per step: the ~250 image addresses of a step are the same in every step, and
    // hoisted out of the loop as loop invariants they alone would take 500 registers
    const float *w1i = A.w1h_img, *w2i = A.w2_img, *w3i = A.w3_img, *whi = A.whh_img;
    asm volatile("" : "+s"(w1i), "+s"(w2i), "+s"(w3i), "+s"(whi));
    // ---- layer 1: a1 = lrelu(W1[:, :H] h + u) ----
    {
      f32x4 acc[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) acc[t] = u[t];
      wide_stream_mm<5, H / 16, PD>(w1i, t1, hrow, acc, lane);
      float* g1 = A.a1 + ((size_t)i * B + b) * D1 + 4 * lg;
#pragma unroll
      for (int t = 0; t < 5; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = sw_lrelu(acc[t][q]);
        st4(&a1b[ln * L1 + 16 * t1[t] + 4 * lg], acc[t]);
        st4(g1 + 16 * t1[t], acc[t]);
      }
    }
    sw_barrier();
    // ---- layer 2: a2 = lrelu(W2 a1 + b2) ----
    {
      f32x4 acc[3] = {bias2[0], bias2[1], bias2[2]};
      wide_stream_mm<3, D1 / 16, PD>(w2i, t2, &a1b[ln * L1 + 4 * lg], acc, lane);
      float* g2 = A.a2 + ((size_t)i * B + b) * D2 + 4 * lg;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = sw_lrelu(acc[t][q]);
        st4(&a2b[ln * L2 + 16 * t2[t] + 4 * lg], acc[t]);
        st4(g2 + 16 * t2[t], acc[t]);
      }
    }
    sw_barrier();
    // ---- layer 3: a3 = W3 a2 + b3 (no activation, train.py:327-330) ----
    {
      f32x4 acc[2] = {bias3[0], bias3[1]};
      wide_stream_mm<2, D2 / 16, PD>(w3i, t3, &a2b[ln * L2 + 4 * lg], acc, lane);
      float* g3 = A.a3 + ((size_t)i * B + b) * D3 + 4 * lg;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        st4(&a3b[ln * L3 + 16 * t3[t] + 4 * lg], acc[t]);
        st4(g3 + 16 * t3[t], acc[t]);
      }
    }
    sw_barrier();
    // ---- output layer + integration: every wave for itself (each keeps its own copy of the running position) ----
    float vx = 0.f, vy = 0.f;
    {
      const float* ar = &a3b[ln * L3 + 20 * lg];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const f32x4 av = ld4(ar + 4 * j);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          vx = fmaf(av[q], w4[0][j][q], vx);
          vy = fmaf(av[q], w4[1][j][q], vy);
        }
      }
      vx += __shfl_xor(vx, 16);
      vy += __shfl_xor(vy, 16);
      vx += __shfl_xor(vx, 32);
      vy += __shfl_xor(vy, 32);
      vx += b4x;
      vy += b4y;
      px += vx;
      py += vy;
      const f32x4 r = {px, py, vx, vy};
      st4(A.pred4 + ((size_t)b * Tp + i) * 4, r);                   // every lane of agent ln holds the same row: all store it
      st4(A.x4 + ((size_t)(To + i) * B + b) * 4, r);
    }
    if (i + 1 == Tp) break;                                          // the step after the last decode is dead compute (train.py:430)
    // ---- re-fed encoder step To + i on (p, v) ----
    {
      const float xb = lg == 0 ? px : (lg == 1 ? py : (lg == 2 ? vx : vy));
      f32x4 acc[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = SW_MFMA(wx[t], xb, biasl[t]);
      wide_stream_mm<8, H / 16, PD>(whi, tl, hrow, acc, lane);
      const size_t trow = (size_t)(To + i) * B + b;
      float* gr = A.gates + trow * 4 * H + 4 * lg;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int u0 = 16 * (2 * wave + k);
        f32x4 gi, gf, gg, go, hn;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          gi[q] = sw_sigmoid(acc[k][q]);
          gf[q] = sw_sigmoid(acc[2 + k][q]);
          gg[q] = sw_tanh(acc[4 + k][q]);
          go[q] = sw_sigmoid(acc[6 + k][q]);
          c[k][q] = fmaf(gf[q], c[k][q], gi[q] * gg[q]);
          hn[q] = go[q] * sw_tanh(c[k][q]);
        }
        st4(&hb[(i + 1) & 1][ln * HL + u0 + 4 * lg], hn);
        st4(gr + u0, gi);
        st4(gr + H + u0, gf);
        st4(gr + 2 * H + u0, gg);
        st4(gr + 3 * H + u0, go);
        st4(A.cs + trow * H + u0 + 4 * lg, c[k]);
        st4(A.hs + ((size_t)(To + i + 1) * B + b) * H + u0 + 4 * lg, hn);
        st4(A.cat + ((size_t)(i + 1) * B + b) * D1 + u0 + 4 * lg, hn);
      }
    }
    sw_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Backward of the decode loop at 128 hidden units in ONE persistent launch per 16-agent tile (the wide counterpart of
// dec_rollout_bwd_kernel): per decode step i = Tp-1 .. 0
//   [re-fed encoder step To + i, if it exists]  dh = dhcat_{i+1} + W_hh^T dgates_{To+i+1}, cell backward -> dgates_{To+i};
//                                               dx4 = dgates Wx (K split over the waves)
//   d p / d v chain (train.py:422-424), dz3 = dv W4, dz2 = (dz3 W3) lrelu'(a2), dz1 = (dz2 W2) lrelu'(a1), dhcat_i = dz1 W1[:, :H]
// with the operand images of W_hh^T, W3^T, W2^T, W1[:, :H]^T streamed from L2 (wide_stream_mm), delta tiles in LDS, dhcat / dc
// carried in registers (the wave that produces dhcat's unit tiles is the one whose LSTM cells consume them).  Data gradients
// only: the rows it leaves (dgates, dv, dz3, dz2, dz1) are what the deferred weight-gradient products contract over.
// ---------------------------------------------------------------------------------------------------------------------------
struct WideDecBwd {
  const float *whhT_img, *w3T_img, *w2T_img, *w1hT_img, *wxT_img;    // [H][4H], [D2][D3], [D1][D2], [H][D1], [16 (4 live)][4H]
  const float* W4;                                                    // [2][D3]
  const float* dpred4;                                                // [B][Tp][4]
  const float *a1, *a2, *gates, *cs;                                  // saved rows (as the forward kernel left them)
  float *dgates, *dv, *dz3, *dz2, *dz1;                               // [Ta][B][4H], [Tp][B][4], [Tp][B][D3 / D2 / D1]
  float *dhcat_out, *dc_out;                                          // [B][H]: d h_{To-1} from decode step 0, d c_{To-1}
  int B, To, Tp;
};
__global__ __launch_bounds__(256) void wide_dec_loop_bwd_kernel(WideDecBwd A) {
  constexpr int H = 128, D1 = 320, D2 = 160, D3 = 80, K4 = 4 * H, PD = 16;
  constexpr int GL = K4 + 4, L1 = D1 + 4, L2 = D2 + 4, L3 = D3 + 4;
  __shared__ __attribute__((aligned(16))) float dgt[2][16 * GL];
  __shared__ __attribute__((aligned(16))) float dz1b[16 * L1];
  __shared__ __attribute__((aligned(16))) float dz2b[16 * L2];
  __shared__ __attribute__((aligned(16))) float dz3b[16 * L3];
  __shared__ __attribute__((aligned(16))) float dxp[4][16 * 4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ln = lane & 15, lg = lane >> 4;
  const int B = A.B, To = A.To, Tp = A.Tp, Ta = To + Tp - 1;
  const int a0 = blockIdx.x * 16;
  const int b = min(a0 + ln, B - 1);
  const int tu[2] = {2 * wave, 2 * wave + 1};                        // unit tiles of this wave (LSTM cells, dhcat)
  const int t2[3] = {wave, wave + 4, 8 + (wave & 1)};                 // dz2 tiles (duplicates store the same values)
  const int t1[5] = {5 * wave, 5 * wave + 1, 5 * wave + 2, 5 * wave + 3, 5 * wave + 4};
  f32x4 w4[2][5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    w4[0][j] = ld4(A.W4 + 20 * lg + 4 * j);
    w4[1][j] = ld4(A.W4 + D3 + 20 * lg + 4 * j);
  }
  f32x4 dh[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float dpx = 0.f, dpy = 0.f;
  for (int i = Tp - 1; i >= 0; --i) {
    const float *whi = A.whhT_img, *w3i = A.w3T_img, *w2i = A.w2T_img, *w1i = A.w1hT_img, *wxi = A.wxT_img;
    asm volatile("" : "+s"(whi), "+s"(w3i), "+s"(w2i), "+s"(w1i), "+s"(wxi));    // image addresses: per step, not hoisted
    // the saved activations of this step (for the LeakyReLU derivatives) are requested now
    f32x4 s2[3], s1[5];
#pragma unroll
    for (int t = 0; t < 3; ++t) s2[t] = ld4(A.a2 + ((size_t)i * B + b) * D2 + 16 * t2[t] + 4 * lg);
#pragma unroll
    for (int t = 0; t < 5; ++t) s1[t] = ld4(A.a1 + ((size_t)i * B + b) * D1 + 16 * t1[t] + 4 * lg);
    f32x4 dx4 = {0.f, 0.f, 0.f, 0.f};
    if (i + 1 < Tp) {
      // ---- re-fed encoder step t = To + i: its h fed decode step i + 1 (dh holds dhcat_{i+1}) and LSTM step t + 1 ----
      const int t = To + i;
      float* cur = dgt[t & 1];
      if (t + 1 < Ta) {
        f32x4 acc[2] = {dh[0], dh[1]};
        wide_stream_mm<2, K4 / 16, PD>(whi, tu, &dgt[(t + 1) & 1][ln * GL + 4 * lg], acc, lane);
        dh[0] = acc[0];
        dh[1] = acc[1];
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int u0 = 16 * tu[k];
        const float* gr = A.gates + ((size_t)t * B + b) * K4 + u0 + 4 * lg;
        const f32x4 gi = ld4(gr), gf = ld4(gr + H), gg = ld4(gr + 2 * H), go = ld4(gr + 3 * H);
        const f32x4 ct = ld4(A.cs + ((size_t)t * B + b) * H + u0 + 4 * lg), cp = ld4(A.cs + ((size_t)(t - 1) * B + b) * H + u0 + 4 * lg);
        f32x4 di, df, dg, dO;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float tc = sw_tanh(ct[q]);
          const float dct = fmaf(dh[k][q] * go[q], 1.0f - tc * tc, dc[k][q]);
          di[q] = dct * gg[q] * gi[q] * (1.0f - gi[q]);
          df[q] = dct * cp[q] * gf[q] * (1.0f - gf[q]);
          dg[q] = dct * gi[q] * (1.0f - gg[q] * gg[q]);
          dO[q] = dh[k][q] * tc * go[q] * (1.0f - go[q]);
          dc[k][q] = dct * gf[q];
        }
        float* tr = cur + ln * GL + u0 + 4 * lg;
        st4(tr, di);
        st4(tr + H, df);
        st4(tr + 2 * H, dg);
        st4(tr + 3 * H, dO);
      }
      sw_barrier();
      // dgates_t rows to memory from the tile (a wave writes 4 agents' rows, consecutive lanes consecutive float4s) ...
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int a = 4 * wave + q, bb = min(a0 + a, B - 1);
#pragma unroll
        for (int c4 = 0; c4 < K4 / 4 / 64; ++c4)
          st4(A.dgates + ((size_t)t * B + bb) * K4 + 4 * (lane + 64 * c4), ld4(cur + a * GL + 4 * (lane + 64 * c4)));
      }
      // ... and dx4 = dgates Wx: this wave's quarter of the 4H columns (8 k-steps of the one WxT row tile)
      {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* drow = cur + ln * GL + 4 * lg + 128 * wave;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4 a = ld4(wxi + (((size_t)(8 * wave + j)) * 64 + lane) * 4), d = ld4(drow + 16 * j);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = SW_MFMA(a[q], d[q], acc);
        }
        if (lg == 0) st4(&dxp[wave][ln * 4], acc);                    // rows 0..3 of the tile = the 4 input components
      }
      sw_barrier();
      dx4 = (ld4(&dxp[0][ln * 4]) + ld4(&dxp[1][ln * 4])) + (ld4(&dxp[2][ln * 4]) + ld4(&dxp[3][ln * 4]));
    }
    // ---- position / velocity chain, dz3 = dv W4 (every wave for itself: each keeps its own copy of d p) ----
    {
      const f32x4 g = ld4(A.dpred4 + ((size_t)b * Tp + i) * 4) + dx4;
      dpx += g[0];
      dpy += g[1];
      const float dvx = g[2] + dpx, dvy = g[3] + dpy;
      st4(A.dv + ((size_t)i * B + b) * 4, f32x4{dvx, dvy, 0.f, 0.f});
      float* zr = A.dz3 + ((size_t)i * B + b) * D3 + 20 * lg;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaf(dvx, w4[0][j][q], dvy * w4[1][j][q]);
        st4(&dz3b[ln * L3 + 20 * lg + 4 * j], v);
        st4(zr + 4 * j, v);
      }
    }
    sw_barrier();
    // ---- dz2 = (dz3 W3) lrelu'(a2) ----
    {
      f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      wide_stream_mm<3, D3 / 16, PD>(w3i, t2, &dz3b[ln * L3 + 4 * lg], acc, lane);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = sw_lrelu_grad(s2[t][q], acc[t][q]);
        st4(&dz2b[ln * L2 + 16 * t2[t] + 4 * lg], acc[t]);
        st4(A.dz2 + ((size_t)i * B + b) * D2 + 16 * t2[t] + 4 * lg, acc[t]);
      }
    }
    sw_barrier();
    // ---- dz1 = (dz2 W2) lrelu'(a1) ----
    {
      f32x4 acc[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      wide_stream_mm<5, D2 / 16, PD>(w2i, t1, &dz2b[ln * L2 + 4 * lg], acc, lane);
#pragma unroll
      for (int t = 0; t < 5; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = sw_lrelu_grad(s1[t][q], acc[t][q]);
        st4(&dz1b[ln * L1 + 16 * t1[t] + 4 * lg], acc[t]);
        st4(A.dz1 + ((size_t)i * B + b) * D1 + 16 * t1[t] + 4 * lg, acc[t]);
      }
    }
    sw_barrier();
    // ---- dhcat_i = dz1 W1[:, :H]: the gradient w.r.t. h of LSTM step To - 1 + i, kept in registers for the next iteration ----
    {
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      wide_stream_mm<2, D1 / 16, PD>(w1i, tu, &dz1b[ln * L1 + 4 * lg], acc, lane);
      dh[0] = acc[0];
      dh[1] = acc[1];
    }
    // (the next iteration's first write to dz3b / dgt sits behind its own barriers; dz1b is next written three barriers on)
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    st4(A.dhcat_out + (size_t)b * H + 16 * tu[k] + 4 * lg, dh[k]);
    st4(A.dc_out + (size_t)b * H + 16 * tu[k] + 4 * lg, dc[k]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The HEADS of the wide discriminator (train.py:280-292, 300-309) for a 16-agent tile and both future branches in ONE launch
// (forward) / ONE launch (backward) instead of nine / seven product launches of ~5 us each: the observation fc, the
// prediction encoder, the classifier and the latent-code head are 64-wide layers at 128 hidden units - a handful of matrix
// instructions per wave between barriers, activations in LDS tiles, weights from their operand images (sw_wide_opimage).
// Generic in H (H2 = H / 2 a multiple of 16), 4 Tp a multiple of 16, nb = 1 or 2 branches, any latent-code count <= 16.
// ---------------------------------------------------------------------------------------------------------------------------
#define WH_MAXH 256
struct WideHeads {
  // operand images: forward (of0 [H2][H], of1 [H2][H2], pe0 [H2][K4], pe1 [H2][H2], cl0 [H2][H], la0 [H2][H]) or, for the
  // backward kernel, of their transposes (of0T [H][H2], of1T, pe0T [K4][H2], pe1T, cl0T [H][H2], la0T)
  const float *of0, *of1, *pe0, *pe1, *cl0, *la0;
  const float *b_of0, *b_of1, *b_pe0, *b_pe1, *b_cl0, *b_la0;     // biases (forward)
  const float *cl1w, *cl1b, *la1w, *la1b;                          // [1][H2], [1], [nl][H2], [nl]
  const float* hT;                                                 // [B][H]
  const float* px;                                                 // [nb B][K4]
  float *o1, *q1, *both, *c1, *l1, *label, *code;                  // [B][H2], [nb B][H2], [nb B][H], .., [nb B], [nb B][nl]
  // backward
  const float *dlab, *dcod;                                        // [nb B][4] (col 0), [nb B][nlp]
  float *dc1, *dl1, *dboth, *dq1, *docode, *do1, *dhT, *dpx;       // deltas (docode / do1 / dhT only with need_obs, dpx with want_dpred)
  int B, H, K4, nb, nl, nlp, need_obs, want_dpred;
  // forward, loss != 0: the LSGAN / info-loss terms of the pass (train.py:484-494, 512-523) formed where label / code_hat appear:
  // dlab = gl (label - targets[t_br]), dcod = gc (code_hat - z[:, :nl]) on branch 0 (zero on branch 1), and the tile's sums of
  // squares {branch-0 label, branch-0 code, branch-1 label} to part[tile][3]
  int loss, t0, t1, zld;
  float gl, gc;
  const float *targets, *z;
  float* part;
  float *dlab_w, *dcod_w;      // [nb B][4], [nb B][nlp] (written by the forward kernel with loss != 0)
};
// acc += W[16 t ..][.] x over K = 16 KJ columns: A operand from the image (tile t), B operand = row `brow` of an LDS tile
__device__ __forceinline__ f32x4 wh_tile(const float* __restrict__ img, int t, int KJ, const float* brow, f32x4 acc, int lane) {
  f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
  const float* ip = img + ((size_t)t * KJ * 64 + lane) * 4;
  for (int j0 = 0; j0 < KJ; j0 += 8) {            // eight image loads in flight, then their matrix instructions
    f32x4 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = min(j0 + u, KJ - 1);
      a[u] = ld4(ip + (size_t)j * 256);
      b[u] = ld4(brow + 16 * j);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (j0 + u < KJ) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (u & 1) acc1 = SW_MFMA(a[u][q], b[u][q], acc1);
          else acc = SW_MFMA(a[u][q], b[u][q], acc);
        }
      }
  }
  return acc + acc1;
}
__global__ __launch_bounds__(256) void wide_disc_heads_fwd_kernel(WideHeads A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = A.H, H2 = H >> 1, K4 = A.K4, nb = A.nb, nl = A.nl, B = A.B;
  const int LH = H + 4, L2 = H2 + 4, LK = K4 + 4;
  float* hTb = smem;                       // [16][LH]
  float* o1b = hTb + 16 * LH;              // [16][L2]
  float* pxb = o1b + 16 * L2;              // [2][16][LK]
  float* q1b = pxb + 2 * 16 * LK;          // [2][16][L2]
  float* bob = q1b + 2 * 16 * L2;          // [2][16][LH]   both = obsv_code | pred_code
  float* c1b = bob + 2 * 16 * LH;          // [2][16][L2]
  float* l1b = c1b + 2 * 16 * L2;          // [2][16][L2]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ln = lane & 15, lg = lane >> 4;
  const int a0 = blockIdx.x * 16;
  const int b = min(a0 + ln, B - 1);
  const int NT2 = H2 >> 4;
  // stage h_T and the prediction rows of the tile (coalesced: 16 threads per row)
  for (int i = threadIdx.x; i < 16 * (H >> 2); i += 256) {
    const int a = i / (H >> 2), c4 = i - a * (H >> 2);
    st4(hTb + a * LH + 4 * c4, ld4(A.hT + (size_t)min(a0 + a, B - 1) * H + 4 * c4));
  }
  for (int i = threadIdx.x; i < nb * 16 * (K4 >> 2); i += 256) {
    const int br = i / (16 * (K4 >> 2)), r = i - br * 16 * (K4 >> 2), a = r / (K4 >> 2), c4 = r - a * (K4 >> 2);
    st4(pxb + (br * 16 + a) * LK + 4 * c4, ld4(A.px + ((size_t)br * B + min(a0 + a, B - 1)) * K4 + 4 * c4));
  }
  __syncthreads();
  // ---- o1 = lrelu(of0 h_T + b); q1[br] = lrelu(pe0 px[br] + b) (independent of each other: same phase) ----
  for (int t = wave; t < NT2; t += 4) {
    f32x4 acc = wh_tile(A.of0, t, H >> 4, hTb + ln * LH + 4 * lg, ld4(A.b_of0 + 16 * t + 4 * lg), lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = sw_lrelu(acc[q]);
    st4(o1b + ln * L2 + 16 * t + 4 * lg, acc);
    st4(A.o1 + (size_t)b * H2 + 16 * t + 4 * lg, acc);
    for (int br = 0; br < nb; ++br) {
      f32x4 aq = wh_tile(A.pe0, t, K4 >> 4, pxb + (br * 16 + ln) * LK + 4 * lg, ld4(A.b_pe0 + 16 * t + 4 * lg), lane);
#pragma unroll
      for (int q = 0; q < 4; ++q) aq[q] = sw_lrelu(aq[q]);
      st4(q1b + (br * 16 + ln) * L2 + 16 * t + 4 * lg, aq);
      st4(A.q1 + ((size_t)br * B + b) * H2 + 16 * t + 4 * lg, aq);
    }
  }
  __syncthreads();
  // ---- obsv_code = of1 o1 + b -> both[br][:, :H2]; pred_code[br] = pe1 q1[br] + b -> both[br][:, H2:] ----
  for (int t = wave; t < NT2; t += 4) {
    const f32x4 oc = wh_tile(A.of1, t, H2 >> 4, o1b + ln * L2 + 4 * lg, ld4(A.b_of1 + 16 * t + 4 * lg), lane);
    for (int br = 0; br < nb; ++br) {
      const f32x4 pc = wh_tile(A.pe1, t, H2 >> 4, q1b + (br * 16 + ln) * L2 + 4 * lg, ld4(A.b_pe1 + 16 * t + 4 * lg), lane);
      st4(bob + (br * 16 + ln) * LH + 16 * t + 4 * lg, oc);
      st4(bob + (br * 16 + ln) * LH + H2 + 16 * t + 4 * lg, pc);
      float* g = A.both + ((size_t)br * B + b) * H + 16 * t + 4 * lg;
      st4(g, oc);
      st4(g + H2, pc);
    }
  }
  __syncthreads();
  // ---- c1 = lrelu(cl0 both + b), l1 = lrelu(la0 both + b) ----
  for (int t = wave; t < NT2; t += 4)
    for (int br = 0; br < nb; ++br) {
      const float* brow = bob + (br * 16 + ln) * LH + 4 * lg;
      f32x4 ac = wh_tile(A.cl0, t, H >> 4, brow, ld4(A.b_cl0 + 16 * t + 4 * lg), lane);
      f32x4 al = wh_tile(A.la0, t, H >> 4, brow, ld4(A.b_la0 + 16 * t + 4 * lg), lane);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ac[q] = sw_lrelu(ac[q]);
        al[q] = sw_lrelu(al[q]);
      }
      st4(c1b + (br * 16 + ln) * L2 + 16 * t + 4 * lg, ac);
      st4(l1b + (br * 16 + ln) * L2 + 16 * t + 4 * lg, al);
      st4(A.c1 + ((size_t)br * B + b) * H2 + 16 * t + 4 * lg, ac);
      st4(A.l1 + ((size_t)br * B + b) * H2 + 16 * t + 4 * lg, al);
    }
  __syncthreads();
  // ---- label = cl1 c1 + b (1 output), code = la1 l1 + b (nl outputs): dot products, 8 threads per (branch, agent) ----
  float* lred = hTb;             // (dead since the first layer) [2 branches][16 agents][2]: squared label / code differences
  {
    const int br = threadIdx.x >> 7 & 1, a = (threadIdx.x >> 3) & 15, l8 = threadIdx.x & 7;   // 2 branches x 16 agents x 8 lanes
    float ql = 0.f, qc = 0.f;
    if (br < nb) {
      const float* cr = c1b + (br * 16 + a) * L2;
      const float* lr = l1b + (br * 16 + a) * L2;
      const bool live = a0 + a < B;
      const size_t row = (size_t)br * B + min(a0 + a, B - 1);
      const float tgt = A.loss ? A.targets[br ? A.t1 : A.t0] : 0.f;
      for (int n = -1; n < nl; ++n) {          // n = -1: the label
        const float* wr = n < 0 ? A.cl1w : A.la1w + (size_t)n * H2;
        const float* xr = n < 0 ? cr : lr;
        float s = 0.f;
        for (int k = l8; k < H2; k += 8) s = fmaf(xr[k], wr[k], s);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 1);
        if (l8 == 0 && live) {
          if (n < 0) {
            const float v = s + A.cl1b[0];
            A.label[row] = v;
            if (A.loss) {
              const float d = v - tgt;
              ql = d * d;
              A.dlab_w[row * 4] = A.gl * d;
            }
          } else {
            const float v = s + A.la1b[n];
            A.code[row * nl + n] = v;
            if (A.loss) {
              const float d = br == 0 ? v - A.z[(size_t)(a0 + a) * A.zld + n] : 0.f;     // the info term: fake branch only
              qc = fmaf(d, d, qc);
              A.dcod_w[row * A.nlp + n] = A.gc * d;
            }
          }
        }
      }
    }
    if (A.loss && l8 == 0) {
      lred[((threadIdx.x >> 7 & 1) * 16 + a) * 2] = ql;
      lred[((threadIdx.x >> 7 & 1) * 16 + a) * 2 + 1] = qc;
    }
  }
  if (A.loss) {
    __syncthreads();
    if (threadIdx.x < 3) {       // fixed order: the tile's 16 agents one after the other
      const int br = threadIdx.x == 2 ? 1 : 0, w = threadIdx.x == 1 ? 1 : 0;
      float v = 0.f;
      if (br < nb)
        for (int a = 0; a < 16; ++a) v += lred[(br * 16 + a) * 2 + w];
      A.part[(size_t)blockIdx.x * 3 + threadIdx.x] = v;
    }
  }
}

__global__ __launch_bounds__(256) void wide_disc_heads_bwd_kernel(WideHeads A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = A.H, H2 = H >> 1, K4 = A.K4, nb = A.nb, nl = A.nl, nlp = A.nlp, B = A.B;
  const int LH = H + 4, L2 = H2 + 4;
  float* dc1b = smem;                       // [2][16][L2]
  float* dl1b = dc1b + 2 * 16 * L2;         // [2][16][L2]
  float* dbob = dl1b + 2 * 16 * L2;         // [2][16][LH]
  float* dq1b = dbob + 2 * 16 * LH;         // [2][16][L2]  (docode in slot 0 rows after dq1 is consumed? no: own tile below)
  float* docb = dq1b + 2 * 16 * L2;         // [16][L2]
  float* do1b = docb + 16 * L2;             // [16][L2]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ln = lane & 15, lg = lane >> 4;
  const int a0 = blockIdx.x * 16;
  const int b = min(a0 + ln, B - 1);
  const int NT2 = H2 >> 4, NTH = H >> 4;
  // ---- dc1 = (dlabel cl1) lrelu'(c1), dl1 = (dcode la1) lrelu'(l1): element-wise (rank 1 / rank nl) ----
  for (int i = threadIdx.x; i < nb * 16 * (H2 >> 2); i += 256) {
    const int br = i / (16 * (H2 >> 2)), r = i - br * 16 * (H2 >> 2), a = r / (H2 >> 2), c4 = r - a * (H2 >> 2);
    const size_t row = (size_t)br * B + min(a0 + a, B - 1);
    const float dl = A.dlab[row * 4];
    const f32x4 cw = ld4(A.cl1w + 4 * c4), c1v = ld4(A.c1 + row * H2 + 4 * c4), l1v = ld4(A.l1 + row * H2 + 4 * c4);
    f32x4 vc, vl = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < nl; ++n) {
      const float dcn = A.dcod[row * nlp + n];
      const f32x4 lw = ld4(A.la1w + (size_t)n * H2 + 4 * c4);
#pragma unroll
      for (int q = 0; q < 4; ++q) vl[q] = fmaf(dcn, lw[q], vl[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vc[q] = sw_lrelu_grad(c1v[q], dl * cw[q]);
      vl[q] = sw_lrelu_grad(l1v[q], vl[q]);
    }
    st4(dc1b + (br * 16 + a) * L2 + 4 * c4, vc);
    st4(dl1b + (br * 16 + a) * L2 + 4 * c4, vl);
    if (a0 + a < B) {
      st4(A.dc1 + row * H2 + 4 * c4, vc);
      st4(A.dl1 + row * H2 + 4 * c4, vl);
    }
  }
  __syncthreads();
  // ---- dboth = cl0^T dc1 + la0^T dl1 (H rows) ----
  for (int t = wave; t < NTH; t += 4)
    for (int br = 0; br < nb; ++br) {
      f32x4 acc = wh_tile(A.cl0, t, H2 >> 4, dc1b + (br * 16 + ln) * L2 + 4 * lg, f32x4{0.f, 0.f, 0.f, 0.f}, lane);
      acc = wh_tile(A.la0, t, H2 >> 4, dl1b + (br * 16 + ln) * L2 + 4 * lg, acc, lane);
      st4(dbob + (br * 16 + ln) * LH + 16 * t + 4 * lg, acc);
      st4(A.dboth + ((size_t)br * B + b) * H + 16 * t + 4 * lg, acc);
    }
  __syncthreads();
  // ---- dq1 = (pe1^T dpcode) lrelu'(q1); docode = sum over the branches of dboth[:, :H2] ----
  for (int t = wave; t < NT2; t += 4) {
    for (int br = 0; br < nb; ++br) {
      f32x4 acc = wh_tile(A.pe1, t, H2 >> 4, dbob + (br * 16 + ln) * LH + H2 + 4 * lg, f32x4{0.f, 0.f, 0.f, 0.f}, lane);
      const f32x4 qv = ld4(A.q1 + ((size_t)br * B + b) * H2 + 16 * t + 4 * lg);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = sw_lrelu_grad(qv[q], acc[q]);
      st4(dq1b + (br * 16 + ln) * L2 + 16 * t + 4 * lg, acc);
      st4(A.dq1 + ((size_t)br * B + b) * H2 + 16 * t + 4 * lg, acc);
    }
    if (A.need_obs) {
      f32x4 v = ld4(dbob + ln * LH + 16 * t + 4 * lg);
      if (nb > 1) v = v + ld4(dbob + (16 + ln) * LH + 16 * t + 4 * lg);
      st4(docb + ln * L2 + 16 * t + 4 * lg, v);
      st4(A.docode + (size_t)b * H2 + 16 * t + 4 * lg, v);
    }
  }
  __syncthreads();
  // ---- do1 = (of1^T docode) lrelu'(o1);  d/d(pred) of branch 0 = pe0^T dq1 (4 Tp rows) ----
  if (A.need_obs)
    for (int t = wave; t < NT2; t += 4) {
      f32x4 acc = wh_tile(A.of1, t, H2 >> 4, docb + ln * L2 + 4 * lg, f32x4{0.f, 0.f, 0.f, 0.f}, lane);
      const f32x4 ov = ld4(A.o1 + (size_t)b * H2 + 16 * t + 4 * lg);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = sw_lrelu_grad(ov[q], acc[q]);
      st4(do1b + ln * L2 + 16 * t + 4 * lg, acc);
      st4(A.do1 + (size_t)b * H2 + 16 * t + 4 * lg, acc);
    }
  if (A.want_dpred)
    for (int t = wave; t < (K4 >> 4); t += 4) {
      const f32x4 acc = wh_tile(A.pe0, t, H2 >> 4, dq1b + ln * L2 + 4 * lg, f32x4{0.f, 0.f, 0.f, 0.f}, lane);
      st4(A.dpx + (size_t)b * K4 + 16 * t + 4 * lg, acc);
    }
  if (!A.need_obs) return;
  __syncthreads();
  // ---- dh_T = of0^T do1 ----
  for (int t = wave; t < NTH; t += 4) {
    const f32x4 acc = wh_tile(A.of0, t, H2 >> 4, do1b + ln * L2 + 4 * lg, f32x4{0.f, 0.f, 0.f, 0.f}, lane);
    st4(A.dhT + (size_t)b * H + 16 * t + 4 * lg, acc);
  }
}
}  // namespace

extern "C" int sw_wide_transpose(const float* src, const int* tab /*device, ntab x 4*/, int ntab, int total_tiles, float* dst,
                                 void* stream) {
  if (!src || !tab || !dst || ntab < 1 || total_tiles < 1) return SW_EARG;
  SW_LAUNCH(wide_transpose_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, src, (const int4*)tab, ntab, dst);
  SW_CHECK_LAUNCH("wide_transpose_kernel");
  return SW_OK;
}

extern "C" int sw_wide_dec_loop_supported(int H) { return H == 128 ? 1 : 0; }
// The decode loop at 128 hidden units in one launch (see wide_dec_loop_fwd_kernel).  Images: w1h = W1[:, :H] [320][128], w2
// [160][320], w3 [80][160], whh [512][128] (sw_wide_opimage); u [B][320] = W1[:, H:] [S; z] + b1.
extern "C" int sw_wide_dec_loop_fwd(const float* w1h_img, const float* w2_img, const float* w3_img, const float* whh_img,
                                    const float* u, const float* b2, const float* b3, const float* W4, const float* b4,
                                    const float* Wx, const float* bx1, const float* bx2, const float* p0, int p0_ld, float* a1,
                                    float* a2, float* a3, float* pred4, float* x4, float* gates, float* cs, float* hs, float* cat,
                                    int B, int H, int To, int Tp, void* stream) {
  if (!w1h_img || !w2_img || !w3_img || !whh_img || !u || !b2 || !b3 || !W4 || !b4 || !Wx || !bx1 || !bx2 || !p0 || !a1 || !a2 || !a3 ||
      !pred4 || !x4 || !gates || !cs || !hs || !cat || B < 1 || To < 1 || Tp < 1 || p0_ld < 2)
    return SW_EARG;
  if (!sw_wide_dec_loop_supported(H)) return SW_ESHAPE;
  WideDecFwd A{w1h_img, w2_img, w3_img, whh_img, u, b2, b3, W4, b4, Wx, bx1, bx2, p0, p0_ld, a1, a2, a3, pred4, x4, gates, cs, hs, cat,
               B, To, Tp};
  SW_LAUNCH(wide_dec_loop_fwd_kernel, dim3((B + 15) / 16), dim3(256), 0, (hipStream_t)stream, A);
  SW_CHECK_LAUNCH("wide_dec_loop_fwd_kernel");
  return SW_OK;
}

// Backward of the decode loop at 128 units in one launch (wide_dec_loop_bwd_kernel).  Images: whhT = W_hh^T [128][512], w3T = W3^T
// [160][80], w2T = W2^T [320][160], w1hT = W1[:, :H]^T [128][320], wxT = Wx^T zero-padded to 16 rows [16][512].
extern "C" int sw_wide_dec_loop_bwd(const float* whhT_img, const float* w3T_img, const float* w2T_img, const float* w1hT_img,
                                    const float* wxT_img, const float* W4, const float* dpred4, const float* a1, const float* a2,
                                    const float* gates, const float* cs, float* dgates, float* dv, float* dz3, float* dz2, float* dz1,
                                    float* dhcat_out, float* dc_out, int B, int H, int To, int Tp, void* stream) {
  if (!whhT_img || !w3T_img || !w2T_img || !w1hT_img || !wxT_img || !W4 || !dpred4 || !a1 || !a2 || !gates || !cs || !dgates || !dv ||
      !dz3 || !dz2 || !dz1 || !dhcat_out || !dc_out || B < 1 || To < 1 || Tp < 1)
    return SW_EARG;
  if (!sw_wide_dec_loop_supported(H)) return SW_ESHAPE;
  WideDecBwd A{whhT_img, w3T_img, w2T_img, w1hT_img, wxT_img, W4, dpred4, a1, a2, gates, cs, dgates, dv, dz3, dz2, dz1, dhcat_out,
               dc_out, B, To, Tp};
  SW_LAUNCH(wide_dec_loop_bwd_kernel, dim3((B + 15) / 16), dim3(256), 0, (hipStream_t)stream, A);
  SW_CHECK_LAUNCH("wide_dec_loop_bwd_kernel");
  return SW_OK;
}

// The heads of the wide discriminator in one launch per direction (wide_disc_heads_fwd/bwd_kernel).  `p` = 52 host pointers
// / integers in the order of struct WideHeads (see socialways_amd/wide.py: _heads_args).
extern "C" int sw_wide_disc_heads_supported(int H, int K4, int nl) {
  return (H >= 32 && H <= WH_MAXH && (H & 31) == 0 && K4 >= 16 && (K4 & 15) == 0 && nl >= 1 && nl <= 16) ? 1 : 0;
}
static int wide_heads_launch(const long long* p, int backward, void* stream) {
  if (!p) return SW_EARG;
  WideHeads A;
  const float** cf = reinterpret_cast<const float**>(&A);
  // the struct starts with 18 const float* (6 images, 6 biases, 4 small weights, hT, px) + 7 float* + 2 const float* + 8 float*
  static_assert(offsetof(WideHeads, B) == 35 * sizeof(void*), "WideHeads layout");
  for (int i = 0; i < 35; ++i) cf[i] = (const float*)(uintptr_t)p[i];
  A.B = (int)p[35]; A.H = (int)p[36]; A.K4 = (int)p[37]; A.nb = (int)p[38]; A.nl = (int)p[39]; A.nlp = (int)p[40];
  A.need_obs = (int)p[41]; A.want_dpred = (int)p[42];
  A.loss = (int)p[43]; A.t0 = (int)p[44]; A.t1 = (int)p[45]; A.zld = (int)p[46];
  double gl, gc;
  memcpy(&gl, &p[47], 8);
  memcpy(&gc, &p[48], 8);
  A.gl = (float)gl; A.gc = (float)gc;
  A.targets = (const float*)(uintptr_t)p[49]; A.z = (const float*)(uintptr_t)p[50]; A.part = (float*)(uintptr_t)p[51];
  A.dlab_w = (float*)(uintptr_t)p[25];      // the backward kernel's dlab / dcod inputs (struct slots 25, 26): same buffers
  A.dcod_w = (float*)(uintptr_t)p[26];
  if (A.loss && !backward && (!A.targets || !A.z || !A.part || !A.dlab_w || !A.dcod_w || A.t0 < 0 || A.t1 < 0 || A.zld < A.nl))
    return SW_EARG;
  if (A.B < 1 || A.nb < 1 || A.nb > 2 || !sw_wide_disc_heads_supported(A.H, A.K4, A.nl)) return SW_ESHAPE;
  const int H = A.H, H2 = H / 2, LH = H + 4, L2 = H2 + 4, LK = A.K4 + 4;
  const int lds_f = (16 * LH + 16 * L2 + 2 * 16 * LK + 2 * 16 * L2 + 2 * 16 * LH + 4 * 16 * L2) * 4;
  const int lds_b = (4 * 16 * L2 + 2 * 16 * LH + 2 * 16 * L2 + 2 * 16 * L2) * 4;
  const int lds = backward ? lds_b : lds_f;
  static int have_f = 0, have_b = 0;
  int& have = backward ? have_b : have_f;
  if (have < lds) {
    hipError_t e = hipFuncSetAttribute(backward ? (const void*)wide_disc_heads_bwd_kernel : (const void*)wide_disc_heads_fwd_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { sw_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e); return SW_EHIP; }
    have = lds;
  }
  const dim3 grid((A.B + 15) / 16), block(256);
  if (backward) SW_LAUNCH(wide_disc_heads_bwd_kernel, grid, block, lds, (hipStream_t)stream, A);
  else SW_LAUNCH(wide_disc_heads_fwd_kernel, grid, block, lds, (hipStream_t)stream, A);
  SW_CHECK_LAUNCH("wide_disc_heads_kernel");
  return SW_OK;
}
extern "C" int sw_wide_disc_heads_fwd(const long long* p, void* stream) { return wide_heads_launch(p, 0, stream); }
extern "C" int sw_wide_disc_heads_bwd(const long long* p, void* stream) { return wide_heads_launch(p, 1, stream); }

extern "C" int sw_wide_opimage(const float* src, const int* tab /*device, ntab x 6*/, int ntab, long long total_float4, float* dst,
                               void* stream) {
  if (!src || !tab || !dst || ntab < 1 || total_float4 < 1) return SW_EARG;
  SW_LAUNCH(wide_opimage_kernel, dim3((unsigned)((total_float4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, tab, ntab, dst);
  SW_CHECK_LAUNCH("wide_opimage_kernel");
  return SW_OK;
}

extern "C" int sw_wide_lstm_seq_supported(int H) { return H == 64 || H == 128 ? 1 : 0; }

extern "C" int sw_wide_lstm_seq_fwd(const float* x4, const float* Wx, const float* b1, const float* b2, const float* whh_img, int B,
                                    int H, int T, float* gates, float* cs, float* hs, float* h_last2, int h2_ld, void* stream) {
  if (!x4 || !Wx || !b1 || !whh_img || !gates || !cs || !hs || B < 1 || T < 1 || (h_last2 && (h2_ld < H || (h2_ld & 3)))) return SW_EARG;
  if (!sw_wide_lstm_seq_supported(H)) return SW_ESHAPE;
  const dim3 grid((B + 15) / 16), block(256);
  if (H == 64)
    SW_LAUNCH((wide_lstm_seq_fwd_kernel<1>), grid, block, 0, (hipStream_t)stream, x4, Wx, b1, b2, whh_img, B, T, gates, cs, hs, h_last2,
              h2_ld);
  else
    SW_LAUNCH((wide_lstm_seq_fwd_kernel<2>), grid, block, 0, (hipStream_t)stream, x4, Wx, b1, b2, whh_img, B, T, gates, cs, hs, h_last2,
              h2_ld);
  SW_CHECK_LAUNCH("wide_lstm_seq_fwd_kernel");
  return SW_OK;
}

extern "C" int sw_wide_lstm_seq_bwd(const float* dh_ext, int dhe_ld, const float* dh_ext2, int dhe2_ld, const float* dg_init,
                                    const float* dc_init, const float* whhT_img, const float* gates, const float* cs, int B, int H,
                                    int T, float* dgates, void* stream) {
  if (!whhT_img || !gates || !cs || !dgates || B < 1 || T < 1 || (dh_ext && (dhe_ld < H || (dhe_ld & 3))) ||
      (dh_ext2 && (dhe2_ld < H || (dhe2_ld & 3))))
    return SW_EARG;
  if (!sw_wide_lstm_seq_supported(H)) return SW_ESHAPE;
  const dim3 grid((B + 15) / 16), block(256);
  if (H == 64)
    SW_LAUNCH((wide_lstm_seq_bwd_kernel<1>), grid, block, 0, (hipStream_t)stream, dh_ext, dhe_ld, dh_ext2, dhe2_ld, dg_init, dc_init,
              whhT_img, gates, cs, B, T, dgates);
  else
    SW_LAUNCH((wide_lstm_seq_bwd_kernel<2>), grid, block, 0, (hipStream_t)stream, dh_ext, dhe_ld, dh_ext2, dhe2_ld, dg_init, dc_init,
              whhT_img, gates, cs, B, T, dgates);
  SW_CHECK_LAUNCH("wide_lstm_seq_bwd_kernel");
  return SW_OK;
}

extern "C" int sw_wide_gemm(const float* x, long long x_rs, int x_cs, const float* w, long long w_rs, int w_cs, const float* bias,
                            const float* cin, int cin_ld, const float* aux, int aux_ld, long long R, int K, int N, float* y,
                            int y_ld, int epi, void* stream) {
  if (!x || !w || !y || R < 0 || K < 1 || N < 1 || y_ld < N || epi < 0 || epi > 4 || (epi >= 3 && !aux) || (cin && cin_ld < N) ||
      (aux && aux_ld < N))
    return SW_EARG;
  if (R == 0) return SW_OK;
  const long long blocks = ((R + 63) / 64) * ((N + 63) / 64);
  if (blocks > 0x7fffffffLL) return SW_ESHAPE;
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  const bool xv = x_cs == 1 && (x_rs & 3) == 0 && (K & 3) == 0 && al(x);
  const bool wv = w_cs == 1 && (w_rs & 3) == 0 && (K & 3) == 0 && al(w);
  const bool ov = (N & 3) == 0 && (y_ld & 3) == 0 && al(y) && (!cin || ((cin_ld & 3) == 0 && al(cin))) &&
                  (!aux || ((aux_ld & 3) == 0 && al(aux)));
  hipStream_t st = (hipStream_t)stream;
  if (K <= 8 && R * N <= 0x7fffffffLL * 256) {
    SW_LAUNCH(wide_smallk_kernel, dim3((unsigned)((R * N + 255) / 256)), dim3(256), 0, st, x, x_rs, x_cs, w, w_rs, w_cs, bias, cin, cin_ld,
              aux, aux_ld, R, K, N, y, y_ld, epi);
    SW_CHECK_LAUNCH("wide_smallk_kernel");
    return SW_OK;
  }
  if (xv && wv && R >= 16) {      // the model's layers: LDS-staged core, 32 x 64 tiles (16 x 64 while those leave CUs idle)
    const long long b2 = ((R + 31) / 32) * ((N + 63) / 64), b1 = ((R + 15) / 16) * ((N + 63) / 64);
    if (b1 > 0x7fffffffLL) return SW_ESHAPE;
#define WIDE_LDS_GEMM(AT, OV, NB)                                                                                              \
  SW_LAUNCH((wide_gemm_lds_kernel<AT, OV>), dim3((unsigned)(NB)), dim3(256), 0, st, x, x_rs, w, w_rs, bias, cin, cin_ld, aux, aux_ld, \
            R, K, N, y, y_ld, epi)
    if (b2 >= 384) {
      if (ov) WIDE_LDS_GEMM(2, true, b2);
      else WIDE_LDS_GEMM(2, false, b2);
    } else {
      if (ov) WIDE_LDS_GEMM(1, true, b1);
      else WIDE_LDS_GEMM(1, false, b1);
    }
#undef WIDE_LDS_GEMM
    SW_CHECK_LAUNCH("wide_gemm_lds_kernel");
    return SW_OK;
  }
  // everything else (3-wide pair features, the K = 1 / 2 products of the heads, transposed operands of the small
  // composition products, a handful of rows): operands straight from global memory with free strides
  const dim3 grid((unsigned)blocks), block(256);
#define WIDE_GEMM(XV, WV, OV)                                                                                              \
  SW_LAUNCH((wide_gemm_kernel<XV, WV, OV>), grid, block, 0, st, x, x_rs, x_cs, w, w_rs, w_cs, bias, cin, cin_ld, aux, aux_ld, \
            R, K, N, y, y_ld, epi)
  if (xv && wv && ov) WIDE_GEMM(true, true, true);
  else if (xv && wv) WIDE_GEMM(true, true, false);
  else if (xv && ov) WIDE_GEMM(true, false, true);
  else if (wv && ov) WIDE_GEMM(false, true, true);
  else WIDE_GEMM(false, false, false);
#undef WIDE_GEMM
  SW_CHECK_LAUNCH("wide_gemm_kernel");
  return SW_OK;
}

extern "C" int sw_wide_lstm_fwd(const float* x4, int x_ld, const float* h_prev, int hp_ld, const float* c_prev, const float* Wx,
                                const float* b1, const float* b2, const float* Whh, int B, int H, float* gates, float* c_out,
                                float* h_out, int h_ld, float* h_out2, int h2_ld, void* stream) {
  if (!x4 || !Wx || !b1 || !Whh || !gates || !c_out || !h_out || B < 1 || H < 16 || (H & 15) || x_ld < 4 || h_ld < H ||
      (h_ld & 3) || (h_prev && (hp_ld < H || (hp_ld & 3))) || (h_out2 && (h2_ld < H || (h2_ld & 3))))
    return SW_EARG;
  SW_LAUNCH(wide_lstm_fwd_kernel, dim3((unsigned)(((B + 63) / 64) * (H / 16))), dim3(256), 0, (hipStream_t)stream, x4, x_ld,
            h_prev, hp_ld, c_prev, Wx, b1, b2, Whh, B, H, gates, c_out, h_out, h_ld, h_out2, h2_ld);
  SW_CHECK_LAUNCH("wide_lstm_fwd_kernel");
  return SW_OK;
}

extern "C" int sw_wide_lstm_bwd(const float* dh_ext, int dhe_ld, const float* dh_ext2, int dhe2_ld, const float* dg_next,
                                const float* WhhT, const float* gates, const float* c, const float* c_prev, const float* dc_in,
                                int B, int H, float* dgates, float* dc_out, void* stream) {
  if (!gates || !c || !dgates || !dc_out || B < 1 || H < 16 || (H & 15) || (dg_next && !WhhT) ||
      (dh_ext && (dhe_ld < H || (dhe_ld & 3))) || (dh_ext2 && (dhe2_ld < H || (dhe2_ld & 3))))
    return SW_EARG;
  const int nub = (H + 63) / 64;
  if (((B + 31) / 32) * nub >= 512)      // enough workgroups for the chip at 32 agents each
    SW_LAUNCH((wide_lstm_bwd_kernel<2>), dim3((unsigned)(((B + 31) / 32) * nub)), dim3(256), 0, (hipStream_t)stream, dh_ext, dhe_ld,
              dh_ext2, dhe2_ld, dg_next, WhhT, gates, c, c_prev, dc_in, B, H, dgates, dc_out);
  else
    SW_LAUNCH((wide_lstm_bwd_kernel<1>), dim3((unsigned)(((B + 15) / 16) * nub)), dim3(256), 0, (hipStream_t)stream, dh_ext, dhe_ld,
              dh_ext2, dhe2_ld, dg_next, WhhT, gates, c, c_prev, dc_in, B, H, dgates, dc_out);
  SW_CHECK_LAUNCH("wide_lstm_bwd_kernel");
  return SW_OK;
}

extern "C" int sw_wide_out_fwd(const float* a3, int D3, const float* W4, const float* b4, float* p, int B, float* pred4_i,
                               int pred_ld, float* x4_tm, void* stream) {
  if (!a3 || !W4 || !b4 || !p || !pred4_i || B < 1 || D3 < 1 || (pred_ld & 3)) return SW_EARG;
  SW_LAUNCH(wide_out_fwd_kernel, dim3((B + 15) / 16), dim3(256), 0, (hipStream_t)stream, a3, D3, W4, b4, p, B, pred4_i, pred_ld,
            x4_tm);
  SW_CHECK_LAUNCH("wide_out_fwd_kernel");
  return SW_OK;
}
extern "C" int sw_wide_out_bwd(const float* dpred4_i, int pred_ld, const float* dg, const float* WxT, int H4, float* dp_run, int B,
                               float* dv, const float* W4, int D3, float* dz3, void* stream) {
  if (!dpred4_i || !dp_run || !dv || !W4 || !dz3 || B < 1 || D3 < 1 || (pred_ld & 3) || (dg && (!WxT || H4 < 64 || (H4 & 63))))
    return SW_EARG;
  SW_LAUNCH(wide_out_bwd_kernel, dim3((B + 15) / 16), dim3(256), 0, (hipStream_t)stream, dpred4_i, pred_ld, dg, WxT, H4, dp_run, B,
            dv, W4, D3, dz3);
  SW_CHECK_LAUNCH("wide_out_bwd_kernel");
  return SW_OK;
}
extern "C" int sw_wide_sum_steps(const float* in, long long t_stride, int in_ld, int T, long long R, int C, float* out,
                                 int out_ld, void* stream) {
  if (!in || !out || T < 1 || R < 1 || C < 1 || in_ld < C || out_ld < C) return SW_EARG;
  SW_LAUNCH(wide_sum_steps_kernel, dim3((unsigned)((R * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, t_stride, in_ld,
            T, R, C, out, out_ld);
  SW_CHECK_LAUNCH("wide_sum_steps_kernel");
  return SW_OK;
}

// Weight gradients of a whole backward pass: n problems dW[N][K] = delta^T act, db[N] = column sums of delta, through the
// grouped split-K GEMM (sw_wgrad.hip) - as few launches as its 24-problem batches allow.  desc = n x 10 host values
// (pointers as integers): delta, ldd, act, lda, R, N, K, dW, ldw, db (0: no bias).
extern "C" int sw_wide_wgrad(const long long* desc, int n, float* wgrad_ws, void* stream) {
  if (!desc || n < 1 || !wgrad_ws) return SW_EARG;
  WgBatch b;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n; ++i) {
    const long long* d = desc + 10 * (size_t)i;
    const float* delta = (const float*)(uintptr_t)d[0];
    const float* act = (const float*)(uintptr_t)d[2];
    float* dW = (float*)(uintptr_t)d[7];
    float* db = (float*)(uintptr_t)d[9];
    const int ldd = (int)d[1], lda = (int)d[3], R = (int)d[4], N = (int)d[5], K = (int)d[6], ldw = (int)d[8];
    if (!delta || !act || !dW || R < 1 || N < 1 || K < 1) return SW_EARG;
    for (int n0 = 0; n0 < N; n0 += 256) {
      const int nn = N - n0 < 256 ? N - n0 : 256;
      WgBatch trial = b;
      int rc = wg_add(trial, delta + n0, ldd, act, lda, R, nn, K, dW + (size_t)n0 * ldw, ldw, db ? db + n0 : nullptr, nullptr, 0);
      if (rc == SW_ESHAPE && b.np > 0) {        // the batch is full: launch it and start the next one with this problem
        if (int r2 = wg_launch(b, wgrad_ws, st)) return r2;
        b = WgBatch();
        trial = WgBatch();
        rc = wg_add(trial, delta + n0, ldd, act, lda, R, nn, K, dW + (size_t)n0 * ldw, ldw, db ? db + n0 : nullptr, nullptr, 0);
      }
      if (rc) return rc;
      b = trial;
    }
  }
  return wg_launch(b, wgrad_ws, st);
}
