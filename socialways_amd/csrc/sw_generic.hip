// sw_generic.hip - the GENERIC-WIDTH path: the same model for any `--hidden-size` / latent-code count the reference accepts
// (train.py:42-44, 65, 76-81) as plain per-layer kernels.  The fused kernels of the other files keep one 64-unit layer per
// workgroup in registers and are the product for the BASELINE configs (all of them use 64 units); widths above 64 and
// latent-code counts other than 2 run here, layer by layer - correct to the same tolerances, launch-bound, an order of
// magnitude slower.  The matrix products are sw_rows_gemm / sw_linear_wgrad (sw_modules.hip); this file holds the
// element-wise and block-diagonal attention pieces:
//   LSTM cell (nn.LSTM gate order i f g o, train.py:254,278), ReLU / LeakyReLU(0.2), mean-squared-error terms
//   (train.py:484-494, 512-523), SocialFeatures on the in-scene pairs (train.py:208-241), AttentionPooling on pair
//   rows (train.py:153-175) for any feature / hidden width.
#include "../../include/socialways_hip.h"
#include "sw_common.h"

namespace {
__device__ __forceinline__ void scene_of(const int* __restrict__ scene_off, int S, int i, int& s, int& s0, int& s1) {
  int lo = 0, hi = S - 1;
  while (lo < hi) {       // the scene with scene_off[s] <= i < scene_off[s+1]
    const int mid = (lo + hi + 1) >> 1;
    if (scene_off[mid] <= i) lo = mid;
    else hi = mid - 1;
  }
  s = lo;
  s0 = scene_off[lo];
  s1 = scene_off[lo + 1];
}
__device__ __forceinline__ float gblock_sum(float v, float* red) {   // 256 threads
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return v;
}
__device__ __forceinline__ float gblock_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return v;
}
}  // namespace

// ---- LSTM cell, element-wise part: pre [B][4H] = W_ih x + b_ih + W_hh h + b_hh (gate blocks i | f | g | o) ------------
__global__ __launch_bounds__(256) void lstm_point_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ c_prev,
                                                             int B, int H, float* __restrict__ gates, float* __restrict__ c,
                                                             float* __restrict__ h) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)B * H) return;
  const int b = (int)(e / H), u = (int)(e - (long long)b * H);
  const float* p = pre + (size_t)b * 4 * H + u;
  const float i = sw_sigmoid(p[0]), f = sw_sigmoid(p[H]), g = sw_tanh(p[2 * H]), o = sw_sigmoid(p[3 * H]);
  const float cn = fmaf(f, c_prev ? c_prev[e] : 0.f, i * g);
  float* q = gates + (size_t)b * 4 * H + u;
  q[0] = i; q[H] = f; q[2 * H] = g; q[3 * H] = o;
  c[e] = cn;
  h[e] = o * sw_tanh(cn);
}
__global__ __launch_bounds__(256) void lstm_point_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c,
                                                             const float* __restrict__ c_prev, const float* __restrict__ dh,
                                                             const float* __restrict__ dc, int B, int H,
                                                             float* __restrict__ dpre, float* __restrict__ dc_prev) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)B * H) return;
  const int b = (int)(e / H), u = (int)(e - (long long)b * H);
  const float* q = gates + (size_t)b * 4 * H + u;
  const float i = q[0], f = q[H], g = q[2 * H], o = q[3 * H];
  const float tc = sw_tanh(c[e]);
  const float dhv = dh ? dh[e] : 0.f;
  const float dct = fmaf(dhv * o, 1.0f - tc * tc, dc ? dc[e] : 0.f);
  float* d = dpre + (size_t)b * 4 * H + u;
  d[0] = dct * g * i * (1.0f - i);
  d[H] = dct * (c_prev ? c_prev[e] : 0.f) * f * (1.0f - f);
  d[2 * H] = dct * i * (1.0f - g * g);
  d[3 * H] = dhv * tc * o * (1.0f - o);
  dc_prev[e] = dct * f;
}
extern "C" int sw_lstm_point_fwd(const float* pre, const float* c_prev, int B, int H, float* gates, float* c, float* h,
                                 void* stream) {
  if (!pre || !gates || !c || !h || B < 1 || H < 1) return SW_EARG;
  SW_LAUNCH(lstm_point_fwd_kernel, dim3((unsigned)(((long long)B * H + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre,
            c_prev, B, H, gates, c, h);
  SW_CHECK_LAUNCH("lstm_point_fwd_kernel");
  return SW_OK;
}
extern "C" int sw_lstm_point_bwd(const float* gates, const float* c, const float* c_prev, const float* dh, const float* dc,
                                 int B, int H, float* dpre, float* dc_prev, void* stream) {
  if (!gates || !c || !dpre || !dc_prev || B < 1 || H < 1) return SW_EARG;
  SW_LAUNCH(lstm_point_bwd_kernel, dim3((unsigned)(((long long)B * H + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
            gates, c, c_prev, dh, dc, B, H, dpre, dc_prev);
  SW_CHECK_LAUNCH("lstm_point_bwd_kernel");
  return SW_OK;
}

// ---- ReLU (kind 0) / LeakyReLU(0.2) (kind 1): y = act(x); backward dx = dy * act'(x), the sign taken from y ------------
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, long long n, int kind, float* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = x[i];
    y[i] = kind == 0 ? fmaxf(v, 0.f) : sw_lrelu(v);
  }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, long long n,
                                                      int kind, float* __restrict__ dx) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float g = dy[i];
    dx[i] = y[i] > 0.f ? g : (kind == 0 ? 0.f : 0.2f * g);
  }
}
extern "C" int sw_act_fwd(const float* x, long long n, int kind, float* y, void* stream) {
  if (!x || !y || n < 0 || (kind != 0 && kind != 1)) return SW_EARG;
  if (n == 0) return SW_OK;
  long long blocks = (n + 255) / 256;
  SW_LAUNCH(act_fwd_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, x, n, kind, y);
  SW_CHECK_LAUNCH("act_fwd_kernel");
  return SW_OK;
}
extern "C" int sw_act_bwd(const float* y, const float* dy, long long n, int kind, float* dx, void* stream) {
  if (!y || !dy || !dx || n < 0 || (kind != 0 && kind != 1)) return SW_EARG;
  if (n == 0) return SW_OK;
  long long blocks = (n + 255) / 256;
  SW_LAUNCH(act_bwd_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, y, dy, n, kind, dx);
  SW_CHECK_LAUNCH("act_bwd_kernel");
  return SW_OK;
}

// ---- nn.MSELoss on a [R][C] block against a block or a scalar target: out[0] = sum (a - b)^2 (the caller divides);
//      da = gscale * (a - b) when asked for.  One workgroup, fixed summation order. ---------------------------------------
__global__ __launch_bounds__(1024) void sqdiff_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                      const float* __restrict__ tb, int tb_idx, long long R, int C,
                                                      float gscale, float* __restrict__ out, float* __restrict__ da, int ldda) {
  __shared__ float red[16];
  const float t = tb ? tb[tb_idx] : 0.f;
  float s = 0.f;
  for (long long e = threadIdx.x; e < R * C; e += 1024) {
    const long long r = e / C;
    const int cidx = (int)(e - r * C);
    const float d = a[r * lda + cidx] - (b ? b[r * ldb + cidx] : t);
    s = fmaf(d, d, s);
    if (da) da[r * ldda + cidx] = gscale * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int w = 0; w < 16; ++w) v += red[w];
    if (out) out[0] = v;
  }
}
extern "C" int sw_sqdiff(const float* a, int lda, const float* b, int ldb, const float* target, int target_idx, long long R,
                         int C, float gscale, float* out_sum, float* da, int ldda, void* stream) {
  if (!a || R < 1 || C < 1 || lda < C || (b && ldb < C) || (!b && !target) || (da && ldda < C)) return SW_EARG;
  SW_LAUNCH(sqdiff_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, lda, b, ldb, target, target_idx, R, C, gscale,
            out_sum, da, ldda);
  SW_CHECK_LAUNCH("sqdiff_kernel");
  return SW_OK;
}

// ---- AttentionPooling on PAIR ROWS (train.py:153-175) for any widths: f [P][F] holds the embedded features of the ordered
//      in-scene pairs, row pair_off[s] + i_local n + j_local (single-agent scenes own no rows), wh = W h + b [B][F],
//      h [B][H].  sigma_ij = <f_ij, wh_j>, sigma_ii := -1000, a_i = softmax_j, S_i = sum_j a_ij h_j; n == 1: S = 0. --------
__global__ __launch_bounds__(256) void attn_pairs_fwd_kernel(const float* __restrict__ f, const float* __restrict__ wh,
                                                             const float* __restrict__ h, const int* __restrict__ scene_off,
                                                             const long long* __restrict__ pair_off, int S, int F, int H,
                                                             float* __restrict__ attn, float* __restrict__ S_out) {
  __shared__ float red[4];
  const int i = blockIdx.x, t = threadIdx.x;
  int s, s0, s1;
  scene_of(scene_off, S, i, s, s0, s1);
  const int n = s1 - s0;
  if (n == 1) {
    for (int u = t; u < H; u += 256) S_out[(size_t)i * H + u] = 0.f;
    return;
  }
  const long long p0 = pair_off[s] + (long long)(i - s0) * n;
  float* arow = attn + p0;
  float m = -INFINITY;
  for (int j = t; j < n; j += 256) {
    float sc = -1000.0f;
    if (s0 + j != i) {
      const float* fr = f + (size_t)(p0 + j) * F;
      const float* wr = wh + (size_t)(s0 + j) * F;
      float a0 = 0.f, a1 = 0.f;
      int k = 0;
      for (; k + 1 < F; k += 2) {
        a0 = fmaf(fr[k], wr[k], a0);
        a1 = fmaf(fr[k + 1], wr[k + 1], a1);
      }
      if (k < F) a0 = fmaf(fr[k], wr[k], a0);
      sc = a0 + a1;
    }
    arow[j] = sc;
    m = fmaxf(m, sc);
  }
  m = gblock_max(m, red);
  float sum = 0.f;
  for (int j = t; j < n; j += 256) {
    const float e = expf(arow[j] - m);
    arow[j] = e;
    sum += e;
  }
  sum = gblock_sum(sum, red);
  for (int j = t; j < n; j += 256) arow[j] = arow[j] / sum;
  __syncthreads();
  for (int u = t; u < H; u += 256) {
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc = fmaf(arow[j], h[(size_t)(s0 + j) * H + u], acc);
    S_out[(size_t)i * H + u] = acc;
  }
}
// row part of the backward (workgroup = query agent i): dsigma_ij = a_ij (<dS_i, h_j> - sum_j' a_ij' <dS_i, h_j'>), 0 on the
// masked diagonal; df_ij = dsigma_ij wh_j
__global__ __launch_bounds__(256) void attn_pairs_bwd_row_kernel(const float* __restrict__ wh, const float* __restrict__ h,
                                                                 const float* __restrict__ attn, const float* __restrict__ dS,
                                                                 const int* __restrict__ scene_off,
                                                                 const long long* __restrict__ pair_off, int S, int F, int H,
                                                                 float* __restrict__ dsig, float* __restrict__ df) {
  __shared__ float red[4];
  const int i = blockIdx.x, t = threadIdx.x;
  int s, s0, s1;
  scene_of(scene_off, S, i, s, s0, s1);
  const int n = s1 - s0;
  if (n == 1) return;
  const long long p0 = pair_off[s] + (long long)(i - s0) * n;
  const float* arow = attn + p0;
  float* drow = dsig + p0;
  float tsum = 0.f;
  for (int j = t; j < n; j += 256) {
    const float* hr = h + (size_t)(s0 + j) * H;
    const float* dr = dS + (size_t)i * H;
    float da = 0.f;
    for (int u = 0; u < H; ++u) da = fmaf(dr[u], hr[u], da);
    drow[j] = da;
    tsum = fmaf(arow[j], da, tsum);
  }
  tsum = gblock_sum(tsum, red);
  for (int j = t; j < n; j += 256) drow[j] = (s0 + j == i) ? 0.f : arow[j] * (drow[j] - tsum);
  __syncthreads();
  if (df) {
    for (long long e = t; e < (long long)n * F; e += 256) {
      const int j = (int)(e / F), k = (int)(e - (long long)j * F);
      df[(size_t)(p0 + j) * F + k] = drow[j] * wh[(size_t)(s0 + j) * F + k];
    }
  }
}
// column part (workgroup = key agent j): dwh_j = sum_i dsigma_ij f_ij, dh_j = sum_i a_ij dS_i
__global__ __launch_bounds__(256) void attn_pairs_bwd_col_kernel(const float* __restrict__ f, const float* __restrict__ attn,
                                                                 const float* __restrict__ dsig, const float* __restrict__ dS,
                                                                 const int* __restrict__ scene_off,
                                                                 const long long* __restrict__ pair_off, int S, int F, int H,
                                                                 float* __restrict__ dwh, float* __restrict__ dh) {
  const int j = blockIdx.x, t = threadIdx.x;
  int s, s0, s1;
  scene_of(scene_off, S, j, s, s0, s1);
  const int n = s1 - s0;
  if (n == 1) {
    for (int k = t; k < F; k += 256) dwh[(size_t)j * F + k] = 0.f;
    for (int u = t; u < H; u += 256) dh[(size_t)j * H + u] = 0.f;
    return;
  }
  const long long pj = pair_off[s] + (j - s0);     // pair (i, j) = pj + i_local n
  for (int k = t; k < F; k += 256) {
    float a = 0.f;
    for (int i = 0; i < n; ++i) a = fmaf(dsig[pj + (long long)i * n], f[(size_t)(pj + (long long)i * n) * F + k], a);
    dwh[(size_t)j * F + k] = a;
  }
  for (int u = t; u < H; u += 256) {
    float a = 0.f;
    for (int i = 0; i < n; ++i) a = fmaf(attn[pj + (long long)i * n], dS[(size_t)(s0 + i) * H + u], a);
    dh[(size_t)j * H + u] = a;
  }
}
extern "C" int sw_attn_pairs_fwd(const float* f, const float* wh, const float* h, const int* scene_off,
                                 const long long* pair_off, int S, int B, int F, int H, float* attn, float* S_out, void* stream) {
  if (!f || !wh || !h || !scene_off || !pair_off || !attn || !S_out || S < 1 || B < 1 || F < 1 || H < 1) return SW_EARG;
  SW_LAUNCH(attn_pairs_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, f, wh, h, scene_off, pair_off, S, F, H, attn, S_out);
  SW_CHECK_LAUNCH("attn_pairs_fwd_kernel");
  return SW_OK;
}
extern "C" int sw_attn_pairs_bwd(const float* f, const float* wh, const float* h, const float* attn, const float* dS,
                                 const int* scene_off, const long long* pair_off, int S, int B, int F, int H, float* dsig,
                                 float* df, float* dwh, float* dh, void* stream) {
  if (!f || !wh || !h || !attn || !dS || !scene_off || !pair_off || !dsig || !dwh || !dh || S < 1 || B < 1 || F < 1 || H < 1)
    return SW_EARG;
  SW_LAUNCH(attn_pairs_bwd_row_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, wh, h, attn, dS, scene_off, pair_off, S, F, H,
            dsig, df);
  SW_CHECK_LAUNCH("attn_pairs_bwd_row_kernel");
  SW_LAUNCH(attn_pairs_bwd_col_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, f, attn, dsig, dS, scene_off, pair_off, S, F, H,
            dwh, dh);
  SW_CHECK_LAUNCH("attn_pairs_bwd_col_kernel");
  return SW_OK;
}
