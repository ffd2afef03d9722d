// sw_modules.hip - backward passes of the reference's STAND-ALONE sub-modules (AttentionPooling, train.py:153-175, on a
// dense (B,B,F) embedding tensor; helpers for EncoderLstm / DecoderFC / EmbedSocialFeatures) so that a user who
// composes them differently from predict() still gets gradients (model.py: _AttFn, _EncFn, _DecFn, _EmbFn).
// None of this is on the training step's path: the kernels are plain VALU code written for any scene size, the
// matrix work goes through the grouped weight-gradient GEMM (sw_wgrad.hip).
#include "../../include/socialways_hip.h"
#include "sw_common.h"
#include "sw_wgrad.h"

// ---- y[r][n] (+)= bias[n] + sum_k x[r][k] w[k w_rs + n w_cs]: a row-times-matrix product with free strides on w
//      (W or W^T).  One wave = 16 rows x (up to) 64 output columns on the matrix cores, operands straight from global
//      memory (clamped addresses, masked values: no load under a branch); the generic-width path (generic.py) spends most
//      of its time here - the first version (one thread per output element) ran 33 ms of a 40 ms step at 128 hidden units.
__global__ __launch_bounds__(256) void rows_gemm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        int w_rs, int w_cs, const float* __restrict__ bias, long long R,
                                                        int K, int N, float* __restrict__ y, int ldy, int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 15, lg = lane >> 4;
  const int ncb = (N + 63) >> 6;
  const long long g = (long long)blockIdx.x * 4 + wave;
  const long long rt = g / ncb;
  const int n0 = (int)(g - rt * ncb) * 64;
  if (rt * 16 >= R) return;
  const long long row = rt * 16 + ln;
  const bool rv = row < R;
  const float* xr = x + (rv ? row : R - 1) * ldx;
  // C layout: acc[t][q] = y[row ln][n0 + 16t + 4lg + q]
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 16 * t + 4 * lg + q;
      acc[t][q] = (bias && n < N) ? bias[n] : 0.f;
    }
  }
  size_t wn[4];      // column offsets of the lane's A operands: output column n0 + 16t + ln (clamped)
  bool nv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = n0 + 16 * t + ln;
    nv[t] = n < N;
    wn[t] = (size_t)(nv[t] ? n : N - 1) * w_cs;
  }
  for (int k0 = 0; k0 < K; k0 += 16) {
    float bv[4], av[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = k0 + 4 * lg + q, kc = kk < K ? kk : K - 1;
      bv[q] = xr[kc];
#pragma unroll
      for (int t = 0; t < 4; ++t) av[t][q] = w[(size_t)kc * w_rs + wn[t]];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool kv = k0 + 4 * lg + q < K;
      const float b = (kv && rv) ? bv[q] : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = SW_MFMA((kv && nv[t]) ? av[t][q] : 0.f, b, acc[t]);
    }
  }
  if (!rv) return;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 16 * t + 4 * lg + q;
      if (n < N) {
        float* dst = y + row * ldy + n;
        *dst = accumulate ? *dst + acc[t][q] : acc[t][q];
      }
    }
  }
}
extern "C" int sw_rows_gemm(const float* x, int ldx, const float* w, int w_rs, int w_cs, const float* bias, long long R, int K,
                            int N, float* y, int ldy, int accumulate, void* stream) {
  if (!x || !w || !y || R < 0 || K < 1 || N < 1 || ldx < K || ldy < N) return SW_EARG;
  if (R == 0) return SW_OK;
  const long long waves = ((R + 15) / 16) * ((N + 63) / 64);
  SW_LAUNCH(rows_gemm_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, w, w_rs,
                     w_cs, bias, R, K, N, y, ldy, accumulate);
  SW_CHECK_LAUNCH("rows_gemm_kernel");
  return SW_OK;
}

// ---- dW[N][K] (+)= delta^T act, db[N] (+)= column sums of delta: one problem of the grouped weight-gradient GEMM
extern "C" int sw_linear_wgrad(const float* delta, int ldd, const float* act, int lda, int R, int N, int K, float* dW, int ldw,
                               float* db, float* wgrad_ws, int accumulate, void* stream) {
  if (!delta || !act || !dW || !wgrad_ws || R < 1 || N < 1 || K < 1) return SW_EARG;
  WgBatch b;
  if (wg_add(b, delta, ldd, act, lda, R, N, K, dW, ldw, db, nullptr, accumulate ? 1 : 0)) return SW_ESHAPE;
  return wg_launch(b, wgrad_ws, (hipStream_t)stream);
}

// ---- AttentionPooling on a dense embedding tensor, one workgroup per agent, any scene size -------------------------
// f [B][B][64] (only in-scene blocks are read), h [B][64], wh = W h + b [B][64] (sw_rows_gemm), scene_off [S+1].
namespace {
__device__ __forceinline__ void find_scene(const int* __restrict__ scene_off, int S, int i, int& s0, int& s1) {
  int lo = 0, hi = S - 1;
  while (lo < hi) {       // the scene with scene_off[s] <= i < scene_off[s+1]
    const int mid = (lo + hi + 1) >> 1;
    if (scene_off[mid] <= i) lo = mid;
    else hi = mid - 1;
  }
  s0 = scene_off[lo];
  s1 = scene_off[lo + 1];
}
__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return v;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return v;
}
__device__ __forceinline__ float dot64(const float* __restrict__ a, const float* __restrict__ b) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int u = 0; u < 64; u += 8) {
    const f32x4 x0 = ld4(a + u), y0 = ld4(b + u), x1 = ld4(a + u + 4), y1 = ld4(b + u + 4);
    s0 = fmaf(x0[0], y0[0], fmaf(x0[1], y0[1], fmaf(x0[2], y0[2], fmaf(x0[3], y0[3], s0))));
    s1 = fmaf(x1[0], y1[0], fmaf(x1[1], y1[1], fmaf(x1[2], y1[2], fmaf(x1[3], y1[3], s1))));
  }
  return s0 + s1;
}
}  // namespace

// sigma_ij = <f_ij, Wh_j>, sigma_ii := -1000, a_i = softmax_j(sigma_i), S_i = sum_j a_ij h_j (train.py:160-174);
// single-agent scenes keep S = 0 (:165).  attn [B][B] receives a_ij inside the scene block (scratch for sigma first).
__global__ __launch_bounds__(256) void attention_dense_row_fwd_kernel(const float* __restrict__ f, const float* __restrict__ h,
                                                                      const float* __restrict__ wh,
                                                                      const int* __restrict__ scene_off, int S, int B,
                                                                      float* __restrict__ attn, float* __restrict__ S_out) {
  __shared__ float red[4];
  __shared__ float part[4][64];
  const int i = blockIdx.x;
  int s0, s1;
  find_scene(scene_off, S, i, s0, s1);
  const int n = s1 - s0, t = threadIdx.x;
  float* arow = attn + (size_t)i * B + s0;
  if (n == 1) {
    if (t < 64) S_out[(size_t)i * 64 + t] = 0.f;
    if (t == 0) arow[0] = 0.f;
    return;
  }
  float m = -INFINITY;
  for (int j = t; j < n; j += 256) {
    const float sc = (s0 + j == i) ? -1000.0f : dot64(f + ((size_t)i * B + s0 + j) * 64, wh + (size_t)(s0 + j) * 64);
    arow[j] = sc;
    m = fmaxf(m, sc);
  }
  m = block_max(m, red);
  float sum = 0.f;
  for (int j = t; j < n; j += 256) {
    const float e = expf(arow[j] - m);
    arow[j] = e;
    sum += e;
  }
  sum = block_sum(sum, red);
  for (int j = t; j < n; j += 256) arow[j] = arow[j] / sum;
  __syncthreads();
  const int u = t & 63, g = t >> 6;
  float acc = 0.f;
  for (int j = g; j < n; j += 4) acc = fmaf(arow[j], h[(size_t)(s0 + j) * 64 + u], acc);
  part[g][u] = acc;
  __syncthreads();
  if (t < 64) S_out[(size_t)i * 64 + t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
}

// Row part of the backward: da_ij = <dS_i, h_j>, dsigma_ij = a_ij (da_ij - sum_j' a_ij' da_ij') (the masked diagonal is a
// constant: no gradient), df_ij = dsigma_ij Wh_j.  dsig [B][B]; df [B][B][64] must be zero outside the scene blocks.
__global__ __launch_bounds__(256) void attention_dense_row_bwd_kernel(const float* __restrict__ h, const float* __restrict__ wh,
                                                                      const float* __restrict__ attn,
                                                                      const float* __restrict__ dS,
                                                                      const int* __restrict__ scene_off, int S, int B,
                                                                      float* __restrict__ dsig, float* __restrict__ df) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  int s0, s1;
  find_scene(scene_off, S, i, s0, s1);
  const int n = s1 - s0, t = threadIdx.x;
  const float* arow = attn + (size_t)i * B + s0;
  float* drow = dsig + (size_t)i * B + s0;
  if (n == 1) {
    if (t == 0) drow[0] = 0.f;
    if (df && t < 16) st4(df + ((size_t)i * B + i) * 64 + 4 * t, f32x4{0.f, 0.f, 0.f, 0.f});
    return;
  }
  float tsum = 0.f;
  for (int j = t; j < n; j += 256) {
    const float da = dot64(dS + (size_t)i * 64, h + (size_t)(s0 + j) * 64);
    drow[j] = da;
    tsum = fmaf(arow[j], da, tsum);
  }
  tsum = block_sum(tsum, red);
  for (int j = t; j < n; j += 256) drow[j] = (s0 + j == i) ? 0.f : arow[j] * (drow[j] - tsum);
  __syncthreads();
  if (df) {
    for (int e = t; e < n * 16; e += 256) {
      const int j = e >> 4, q = e & 15;
      const float d = drow[j];
      const f32x4 w = ld4(wh + (size_t)(s0 + j) * 64 + 4 * q);
      st4(df + ((size_t)i * B + s0 + j) * 64 + 4 * q, f32x4{d * w[0], d * w[1], d * w[2], d * w[3]});
    }
  }
}

// Column part: dWh_j = sum_i dsigma_ij f_ij, dh_j = sum_i a_ij dS_i (the W^T dWh_j term is added by the caller).
__global__ __launch_bounds__(256) void attention_dense_col_bwd_kernel(const float* __restrict__ f, const float* __restrict__ attn,
                                                                      const float* __restrict__ dsig,
                                                                      const float* __restrict__ dS,
                                                                      const int* __restrict__ scene_off, int S, int B,
                                                                      float* __restrict__ dwh, float* __restrict__ dh) {
  __shared__ float p1[4][64], p2[4][64];
  const int j = blockIdx.x;
  int s0, s1;
  find_scene(scene_off, S, j, s0, s1);
  const int n = s1 - s0, t = threadIdx.x, u = t & 63, g = t >> 6;
  float a1 = 0.f, a2 = 0.f;
  if (n > 1) {
    for (int i = g; i < n; i += 4) {
      const size_t ij = (size_t)(s0 + i) * B + j;
      a1 = fmaf(dsig[ij], f[ij * 64 + u], a1);
      a2 = fmaf(attn[ij], dS[(size_t)(s0 + i) * 64 + u], a2);
    }
  }
  p1[g][u] = a1;
  p2[g][u] = a2;
  __syncthreads();
  if (t < 64) {
    dwh[(size_t)j * 64 + t] = (p1[0][t] + p1[1][t]) + (p1[2][t] + p1[3][t]);
    dh[(size_t)j * 64 + t] = (p2[0][t] + p2[1][t]) + (p2[2][t] + p2[3][t]);
  }
}

extern "C" int sw_attention_dense_fwd(const float* f, const float* h, const float* wh, const int* scene_off, int S, int B,
                                      float* attn, float* S_out, void* stream) {
  if (!f || !h || !wh || !scene_off || !attn || !S_out || S < 1 || B < 1) return SW_EARG;
  SW_LAUNCH(attention_dense_row_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, f, h, wh, scene_off, S, B, attn,
                     S_out);
  SW_CHECK_LAUNCH("attention_dense_row_fwd_kernel");
  return SW_OK;
}
extern "C" int sw_attention_dense_bwd(const float* f, const float* h, const float* wh, const float* attn, const float* dS,
                                      const int* scene_off, int S, int B, float* dsig, float* df, float* dwh, float* dh,
                                      void* stream) {
  if (!f || !h || !wh || !attn || !dS || !scene_off || !dsig || !dwh || !dh || S < 1 || B < 1) return SW_EARG;
  SW_LAUNCH(attention_dense_row_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, h, wh, attn, dS, scene_off, S,
                     B, dsig, df);
  SW_CHECK_LAUNCH("attention_dense_row_bwd_kernel");
  SW_LAUNCH(attention_dense_col_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, f, attn, dsig, dS, scene_off, S,
                     B, dwh, dh);
  SW_CHECK_LAUNCH("attention_dense_col_bwd_kernel");
  return SW_OK;
}

// dz [B][32] = du W1[:, 128:160] of a Tp = 1 DecoderFC rollout (gdelta of sw_dec_rollout_bwd called with this To, Tp = 1):
// the gradient w.r.t. the noise input, which predict() never needs (train.py:473: z is data)
extern "C" int sw_dec_fc_dz(const float* dec_w, const float* gdelta, int B, int To, float* dz, void* stream) {
  if (!dec_w || !gdelta || !dz || B < 1 || To < 2) return SW_EARG;
  const GDelta gd = gdelta_layout(B, To, 1);
  return sw_rows_gemm(gdelta + gd.du, 160, dec_w + swp::DEC_W1 + 128, 160, 1, nullptr, B, 160, 32, dz, 32, 0, stream);
}
