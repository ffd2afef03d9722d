// sw_lstm.hip - EncoderLstm over a whole sequence (reference train.py:245-269 as used by
// predict() train.py:397-404) and its BPTT.  One workgroup per 16-agent tile, 4 waves, W_hh in
// registers, h exchanged through a double-buffered 4 KB LDS tile: one barrier per time step.
#include "../../include/socialways_hip.h"
#include "sw_lstm_dev.h"

// x4[agent][t][comp] for the observation rule of get_traj_4d (train.py:131-133): v_0 := v_1.
__device__ __forceinline__ float obs_x4(const float* pos, int b, int t, int T, int comp) {
  const float* p = pos + (size_t)b * T * 2;
  if (comp < 2) return p[t * 2 + comp];
  int c = comp - 2;
  int tt = t == 0 ? 1 : t;
  return p[tt * 2 + c] - p[(tt - 1) * 2 + c];
}

__global__ __launch_bounds__(SW_THREADS) void enc_lstm_fwd_kernel(
    const float* __restrict__ x, int x_mode, const float* __restrict__ enc_w, const float* __restrict__ h0,
    const float* __restrict__ c0, int B, int T, float* __restrict__ hT, float* __restrict__ cT,
    float* __restrict__ y, float* __restrict__ act, float* __restrict__ x4s, int t0, const float* __restrict__ aux_src,
    float* __restrict__ aux_dst, long long aux_n) {
  // Workgroups beyond the agent tiles only copy aux_src -> aux_dst (the training step pulls z out of its pinned
  // host slot here: 128 tiles leave half of the CUs idle for the whole latency-bound kernel, the PCIe read is free)
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  if ((int)blockIdx.x >= tiles) {
    const long long n4 = aux_n >> 2, stride = (long long)(gridDim.x - tiles) * SW_THREADS;
    for (long long i = (long long)(blockIdx.x - tiles) * SW_THREADS + threadIdx.x; i < n4; i += stride)
      st4(aux_dst + 4 * i, ld4(aux_src + 4 * i));
    return;
  }
  __shared__ __attribute__((aligned(16))) float hbuf[2][SW_TILE * SW_HLD];
  __shared__ __attribute__((aligned(16))) float wx_lds[256 * 4];
  __shared__ __attribute__((aligned(16))) float bx_lds[256];
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16;
  const int a0 = blockIdx.x * SW_TILE;
  const int b = min(a0 + ln, B - 1);
  const bool live = (a0 + ln) < B;
  LstmW W;
  lstm_load_whh(W, enc_w + swp::ENC_WHH, u0, ln, lg);

  lstm_prep_rows(enc_w + swp::ENC_EMB_W, enc_w + swp::ENC_EMB_B, enc_w + swp::ENC_WIH, enc_w + swp::ENC_BIH,
                 enc_w + swp::ENC_BHH, true, wx_lds, bx_lds);
  // initial state
  f32x4 c = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};
  if (h0) h = ld4(h0 + (size_t)b * 64 + u0 + 4 * lg);
  if (c0) c = ld4(c0 + (size_t)b * 64 + u0 + 4 * lg);
  st4(&hbuf[0][ln * SW_HLD + u0 + 4 * lg], h);
  sw_barrier();
  lstm_load_wx(W, wx_lds, bx_lds, u0, ln, lg);

  // the input of step t+1 is fetched while step t computes: its L2/HBM latency would otherwise sit in front of
  // the first MFMA of every step
  auto load_x = [&](int t) { return x_mode == 0 ? obs_x4(x, b, t, T, lg) : x[((size_t)b * T + t) * 4 + lg]; };
  float xnext = load_x(0);
  for (int t = 0; t < T; ++t) {
    const float xb = xnext;
    if (t + 1 < T) xnext = load_x(t + 1);
    f32x4 gate[4];
    lstm_cell(W, xb, &hbuf[t & 1][ln * SW_HLD + 4 * lg], gate, c, h);
    st4(&hbuf[(t + 1) & 1][ln * SW_HLD + u0 + 4 * lg], h);
    if (live) {
      if (act) {
        float* row = act + ((size_t)(t0 + t) * B + b) * 384 + u0 + 4 * lg;
#pragma unroll
        for (int g = 0; g < 4; ++g) st4(row + g * 64, gate[g]);
        st4(row + 256, c);
        st4(row + 320, h);
      }
      if (y) st4(y + ((size_t)b * T + t) * 64 + u0 + 4 * lg, h);
      if (x4s && wave == 0) x4s[((size_t)(t0 + t) * B + b) * 4 + lg] = xb;
    }
    sw_barrier();
  }
  if (live) {
    st4(hT + (size_t)b * 64 + u0 + 4 * lg, h);
    st4(cT + (size_t)b * 64 + u0 + 4 * lg, c);
  }
}

// BPTT.  Per step: elementwise gate gradients (lane-local) -> dgates row to HBM (for the
// deferred weight-gradient GEMM) and to LDS -> dh_{t-1} = W_hh^T dgates on the matrix cores.
__global__ __launch_bounds__(SW_THREADS) void enc_lstm_bwd_kernel(
    const float* __restrict__ whh, const float* __restrict__ act, const float* __restrict__ c0,
    const float* __restrict__ dhT, const float* __restrict__ dcT, const float* __restrict__ dy, int B, int T,
    int t0, float* __restrict__ dgates, float* __restrict__ dh0, float* __restrict__ dc0) {
  __shared__ __attribute__((aligned(16))) float dgbuf[2][SW_TILE * SW_GLD];
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16;
  const int a0 = blockIdx.x * SW_TILE;
  const int b = min(a0 + ln, B - 1);
  const bool live = (a0 + ln) < B;
  LstmWT W;
  lstm_load_wT(W, whh, u0, ln, lg);
  f32x4 dh = {0.f, 0.f, 0.f, 0.f}, dc = {0.f, 0.f, 0.f, 0.f};
  if (dhT) dh = ld4(dhT + (size_t)b * 64 + u0 + 4 * lg);
  if (dcT) dc = ld4(dcT + (size_t)b * 64 + u0 + 4 * lg);
  // saved rows of step t are loaded one iteration ahead: their L2/HBM latency hides under the MFMAs
  auto load_row = [&](int t, f32x4 g[4], f32x4& ct_, f32x4& cp_) {
    const float* row = act + ((size_t)(t0 + t) * B + b) * 384 + u0 + 4 * lg;
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = ld4(row + q * 64);
    ct_ = ld4(row + 256);
    cp_ = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t0 + t > 0) cp_ = ld4(row - (size_t)B * 384 + 256);
    else if (c0) cp_ = ld4(c0 + (size_t)b * 64 + u0 + 4 * lg);
  };
  f32x4 gate[4], ct, cprev;
  load_row(T - 1, gate, ct, cprev);
  for (int t = T - 1; t >= 0; --t) {
    f32x4 ngate[4], nct, ncp, dgate[4];
    if (t > 0) load_row(t - 1, ngate, nct, ncp);
    if (dy) dh += ld4(dy + ((size_t)b * T + t) * 64 + u0 + 4 * lg);
    lstm_cell_bwd(gate, ct, cprev, dh, dc, dgate);
    float* dgl = &dgbuf[t & 1][ln * SW_GLD + u0 + 4 * lg];
    float* dgg = dgates + ((size_t)(t0 + t) * B + b) * 256 + u0 + 4 * lg;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      st4(dgl + g * 64, dgate[g]);
      if (live) st4(dgg + g * 64, dgate[g]);
    }
    sw_barrier();
    dh = lstm_dh_prev(W, &dgbuf[t & 1][ln * SW_GLD + 4 * lg]);
    if (t > 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) gate[g] = ngate[g];
      ct = nct;
      cprev = ncp;
    }
  }
  if (live) {
    if (dh0) st4(dh0 + (size_t)b * 64 + u0 + 4 * lg, dh);
    if (dc0) st4(dc0 + (size_t)b * 64 + u0 + 4 * lg, dc);
  }
}

// get_traj_4d (train.py:130-138) as a standalone op for the module-level API.
__global__ void traj4d_kernel(const float* __restrict__ obsv, const float* __restrict__ pred, int B, int To,
                              int Tp, float* __restrict__ o4, float* __restrict__ p4) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int n_o = B * To, n_p = pred ? B * Tp : 0;
  if (i < n_o) {
    int b = i / To, t = i - b * To;
    f32x4 v;
    v[0] = obs_x4(obsv, b, t, To, 0);
    v[1] = obs_x4(obsv, b, t, To, 1);
    v[2] = obs_x4(obsv, b, t, To, 2);
    v[3] = obs_x4(obsv, b, t, To, 3);
    st4(o4 + (size_t)i * 4, v);
  } else if (i < n_o + n_p) {
    int k = i - n_o, b = k / Tp, t = k - b * Tp;
    const float* p = pred + ((size_t)b * Tp + t) * 2;
    const float* q = t == 0 ? obsv + ((size_t)b * To + To - 1) * 2 : p - 2;
    f32x4 v = {p[0], p[1], p[0] - q[0], p[1] - q[1]};
    st4(p4 + (size_t)k * 4, v);
  }
}

extern "C" int sw_traj_4d(const float* obsv, const float* pred, int B, int To, int Tp, float* obsv4,
                          float* pred4, void* stream) {
  if (!obsv || !obsv4 || B < 0 || To < 2 || (pred && (!pred4 || Tp < 1))) return SW_EARG;
  if (B == 0) return SW_OK;
  int n = B * To + (pred ? B * Tp : 0);
  hipLaunchKernelGGL(traj4d_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, obsv, pred, B,
                     To, Tp, obsv4, pred4);
  SW_CHECK_LAUNCH("traj4d_kernel");
  return SW_OK;
}

extern "C" int sw_enc_lstm_fwd_aux(const float* x, int x_mode, const float* enc_w, const float* h0,
                                   const float* c0, int B, int T, float* hT, float* cT, float* y, float* act,
                                   float* x4s, int t0, const float* aux_src, float* aux_dst, long long aux_n,
                                   void* stream) {
  if (!x || !enc_w || !hT || !cT || B < 0 || T < 1 || t0 < 0 || (x_mode != 0 && x_mode != 1)) return SW_EARG;
  if (aux_n < 0 || (aux_n & 3) || (aux_n > 0 && (!aux_src || !aux_dst))) return SW_EARG;
  if (x_mode == 0 && T < 2) return SW_ESHAPE;  // the observation velocity rule needs 2 points
  if (B == 0) return SW_OK;
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  int extra = aux_n > 0 ? (int)((aux_n / 4 + SW_THREADS - 1) / SW_THREADS) : 0;
  if (extra > 64) extra = 64;
  hipLaunchKernelGGL(enc_lstm_fwd_kernel, dim3(tiles + extra), dim3(SW_THREADS), 0, (hipStream_t)stream, x, x_mode, enc_w,
                     h0, c0, B, T, hT, cT, y, act, x4s, t0, aux_src, aux_dst, aux_n);
  SW_CHECK_LAUNCH("enc_lstm_fwd_kernel");
  return SW_OK;
}
extern "C" int sw_enc_lstm_fwd(const float* x, int x_mode, const float* enc_w, const float* h0,
                               const float* c0, int B, int T, float* hT, float* cT, float* y, float* act,
                               float* x4s, int t0, void* stream) {
  return sw_enc_lstm_fwd_aux(x, x_mode, enc_w, h0, c0, B, T, hT, cT, y, act, x4s, t0, nullptr, nullptr, 0, stream);
}

extern "C" int sw_enc_lstm_bwd(const float* enc_w, const float* act, const float* c0, const float* dhT,
                               const float* dcT, const float* dy, int B, int T, int t0, float* dgates,
                               float* dh0, float* dc0, void* stream) {
  if (!enc_w || !act || !dgates || B < 0 || T < 1 || t0 < 0) return SW_EARG;
  if (B == 0) return SW_OK;
  hipLaunchKernelGGL(enc_lstm_bwd_kernel, dim3((B + SW_TILE - 1) / SW_TILE), dim3(SW_THREADS), 0,
                     (hipStream_t)stream, enc_w + swp::ENC_WHH, act, c0, dhT, dcT, dy, B, T, t0, dgates, dh0,
                     dc0);
  SW_CHECK_LAUNCH("enc_lstm_bwd_kernel");
  return SW_OK;
}
