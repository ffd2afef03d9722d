// sw_lstm.hip - EncoderLstm over a whole sequence (reference train.py:245-269 as used by
// predict() train.py:397-404) and its BPTT.  One workgroup per 16-agent tile, 4 waves, W_hh in
// registers, h exchanged through a double-buffered 4 KB LDS tile: one barrier per time step.
#include "../../include/socialways_hip.h"
#include "sw_lstm_dev.h"
#include <type_traits>
#include <cstdlib>

// XMODE 0: x = positions [B][T][2] (4-d state formed on the fly); 1: x = [B][T][4].  ACT / Y / X4S: which per-step
// rows are stored.  All of them are template parameters so that the step loop has NO conditional memory operation:
// the compiler then knows how many loads / stores are in flight and never waits for the stores of a step (with
// run-time conditions it put s_waitcnt vmcnt(0) in front of every step's barrier: one store round trip per step).
// Padding lanes of the last tile are exact replicas of agent B-1 (every load is clamped to it) and store the same
// values to the same rows.
template <int XMODE, bool ACT, bool Y, bool X4S>
__global__ __launch_bounds__(SW_THREADS) void enc_lstm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ enc_w, const float* __restrict__ h0,
    const float* __restrict__ c0, int B, int T, float* __restrict__ hT, float* __restrict__ cT,
    float* __restrict__ y, float* __restrict__ act, float* __restrict__ x4s, int t0, const float* __restrict__ aux_src,
    float* __restrict__ aux_dst, long long aux_n, const float* __restrict__ gimg) {
  // The FIRST workgroups of the grid only copy aux_src -> aux_dst (the training step pulls z out of its pinned host slot
  // here).  128 tiles leave half of the CUs idle for the whole latency-bound kernel: the PCIe read is free.  With more
  // tiles than CUs (dense crowds: 4 MB of z) the copy must START with the kernel - workgroups are dispatched in index
  // order; appended behind the tiles (round 3) they began when the last round of tiles did and the 4 MB then took
  // longer than that round: 113 -> 173 us - and keep several reads per thread in flight (PCIe round trips).
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  const int extra = (int)gridDim.x - tiles;
  if ((int)blockIdx.x < extra) {
    const long long n4 = aux_n >> 2, stride = (long long)extra * SW_THREADS;
    long long i = (long long)blockIdx.x * SW_THREADS + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      const f32x4 v0 = ld4(aux_src + 4 * i), v1 = ld4(aux_src + 4 * (i + stride));
      const f32x4 v2 = ld4(aux_src + 4 * (i + 2 * stride)), v3 = ld4(aux_src + 4 * (i + 3 * stride));
      st4(aux_dst + 4 * i, v0);
      st4(aux_dst + 4 * (i + stride), v1);
      st4(aux_dst + 4 * (i + 2 * stride), v2);
      st4(aux_dst + 4 * (i + 3 * stride), v3);
    }
    for (; i < n4; i += stride) st4(aux_dst + 4 * i, ld4(aux_src + 4 * i));
    return;
  }
  // h exchange between the waves: with ACT the whole saved row of a step is assembled in LDS (lstm_put_act_tile) and its
  // h columns are the next step's B operand; without, a plain [16][68] h tile
  constexpr int HS = ACT ? SW_ALD : SW_HLD, HO = ACT ? 320 : 0;
  __shared__ __attribute__((aligned(16))) float hbuf[2][SW_TILE * HS];
  __shared__ __attribute__((aligned(16))) float wx_lds[256 * 4];
  __shared__ __attribute__((aligned(16))) float bx_lds[256];
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16;
  const int a0 = ((int)blockIdx.x - extra) * SW_TILE;
  const int b = min(a0 + ln, B - 1);
  LstmW W;
  if (gimg) {   // weight images of this step, derived once by the staging launch (swimg, sw_common.h)
    lstm_load_img(W, gimg, wave, lane);
  } else {
    lstm_load_whh(W, enc_w + swp::ENC_WHH, u0, ln, lg);
    lstm_prep_rows(enc_w + swp::ENC_EMB_W, enc_w + swp::ENC_EMB_B, enc_w + swp::ENC_WIH, enc_w + swp::ENC_BIH,
                   enc_w + swp::ENC_BHH, true, wx_lds, bx_lds);
  }
  // initial state
  f32x4 c = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};
  if (h0) h = ld4(h0 + (size_t)b * 64 + u0 + 4 * lg);
  if (c0) c = ld4(c0 + (size_t)b * 64 + u0 + 4 * lg);
  st4(&hbuf[0][ln * HS + HO + u0 + 4 * lg], h);
  sw_barrier();
  if (!gimg) lstm_load_wx(W, wx_lds, bx_lds, u0, ln, lg);

  // the input of step t+1 is fetched while step t computes: its L2/HBM latency would otherwise sit in front of
  // the first MFMA of every step (the last step re-fetches its own input: no conditional load)
  float xa, xq = 0.f;
  auto load_x = [&](int t) {
    if constexpr (XMODE == 0) obs_x4_load(x, b, t, T, lg, xa, xq);
    else xa = x[((size_t)b * T + t) * 4 + lg];
  };
  load_x(0);
  asm volatile("" : "+v"(xa), "+v"(xq));   // waited for HERE: the loop header must see no pending load on any path in
  float* yrow = Y ? y + (size_t)b * T * 64 + u0 + 4 * lg : nullptr;
  float* xrow = X4S ? x4s + ((size_t)t0 * B + b) * 4 + lg : nullptr;
  for (int t = 0; t < T; ++t) {
    const float xb = XMODE == 0 ? xa - (lg >= 2 ? xq : 0.f) : xa;
    load_x(min(t + 1, T - 1));
    f32x4 gate[4];
    lstm_cell(W, xb, &hbuf[t & 1][ln * HS + HO + 4 * lg], gate, c, h);
    if constexpr (ACT) lstm_put_act_tile(hbuf[(t + 1) & 1], gate, c, h, ln, lg, u0);
    else st4(&hbuf[(t + 1) & 1][ln * HS + u0 + 4 * lg], h);
    if constexpr (Y) {
      st4(yrow, h);
      yrow += 64;
    }
    if constexpr (X4S) {   // all four waves hold the same x_t: they all store it (no per-wave branch)
      *xrow = xb;
      xrow += (size_t)B * 4;
    }
    sw_barrier();
    if constexpr (ACT) lstm_store_act_tile(hbuf[(t + 1) & 1], act + (size_t)(t0 + t) * B * 384, a0, B, wave, lane);
    asm volatile("" : "+v"(xa), "+v"(xq));   // the prefetched input is not touched before this point
  }
  st4(hT + (size_t)b * 64 + u0 + 4 * lg, h);
  st4(cT + (size_t)b * 64 + u0 + 4 * lg, c);
}

// ---- round-5 pilot: the same kernel on EIGHT waves (two per SIMD), weights split 8 ways -----------------------------
// The r4 verdict's lever for the register-resident serial kernels.  Wave w owns hidden units 8w .. 8w+7 of all four gates
// as two row tiles (swimg::OP_WHH8: a lane's result registers hold i, f | g, o of the same two units, so the cell update
// stays lane-local): 34 instead of 68 matrix instructions per wave and step, 32 instead of 64 weight registers, the saved
// row assembled in the same LDS tile.  Training-step instance only (positions in, saved rows + inputs out, weight images
// registered); selected by SW_ENC8=1.  Measured against the 4-wave kernel in DESIGN section 9 (round 5).
__global__ __launch_bounds__(512) void enc_lstm_fwd8_kernel(
    const float* __restrict__ x, const float* __restrict__ h0, const float* __restrict__ c0, int B, int T,
    float* __restrict__ hT, float* __restrict__ cT, float* __restrict__ act, float* __restrict__ x4s, int t0,
    const float* __restrict__ aux_src, float* __restrict__ aux_dst, long long aux_n, const float* __restrict__ gimg) {
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  const int extra = (int)gridDim.x - tiles;
  if ((int)blockIdx.x < extra) {      // z's pull out of the pinned slot, as in enc_lstm_fwd_kernel
    const long long n4 = aux_n >> 2, stride = (long long)extra * 512;
    long long i = (long long)blockIdx.x * 512 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
      const f32x4 v0 = ld4(aux_src + 4 * i), v1 = ld4(aux_src + 4 * (i + stride));
      st4(aux_dst + 4 * i, v0);
      st4(aux_dst + 4 * (i + stride), v1);
    }
    for (; i < n4; i += stride) st4(aux_dst + 4 * i, ld4(aux_src + 4 * i));
    return;
  }
  __shared__ __attribute__((aligned(16))) float hbuf[2][SW_TILE * SW_ALD];
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int ub = 8 * wave + 2 * lg;                 // this lane's two units (result rows 4 lg + r: gate r >> 1, unit ub + (r & 1))
  const int a0 = ((int)blockIdx.x - extra) * SW_TILE;
  const int b = min(a0 + ln, B - 1);
  f32x4 whh[2][4], bias[2];
  float wx[2];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl) {
#pragma unroll
    for (int j = 0; j < 4; ++j) whh[tl][j] = ld4(gimg + swimg::OP_WHH8 + ((((size_t)wave * 2 + tl) * 4 + j) * 64 + lane) * 4);
    const int rowA = (2 * tl + ((ln & 3) >> 1)) * 64 + 8 * wave + 2 * (ln >> 2) + (ln & 1);   // A row of lane ln
    wx[tl] = gimg[swimg::WX + rowA * 4 + lg];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[tl][r] = gimg[swimg::BX + (2 * tl + (r >> 1)) * 64 + ub + (r & 1)];
  }
  float2 c = {0.f, 0.f}, h = {0.f, 0.f};
  if (h0) h = *reinterpret_cast<const float2*>(h0 + (size_t)b * 64 + ub);
  if (c0) c = *reinterpret_cast<const float2*>(c0 + (size_t)b * 64 + ub);
  *reinterpret_cast<float2*>(&hbuf[0][ln * SW_ALD + 320 + ub]) = h;
  sw_barrier();
  float xa, xq = 0.f;
  obs_x4_load(x, b, 0, T, lg, xa, xq);
  asm volatile("" : "+v"(xa), "+v"(xq));
  float* xrow = x4s + ((size_t)t0 * B + b) * 4 + lg;
  for (int t = 0; t < T; ++t) {
    const float xb = xa - (lg >= 2 ? xq : 0.f);
    obs_x4_load(x, b, min(t + 1, T - 1), T, lg, xa, xq);
    const float* hrow = &hbuf[t & 1][ln * SW_ALD + 320 + 4 * lg];
    f32x4 bq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bq[j] = ld4(hrow + 16 * j);
    f32x4 acc0 = SW_MFMA(wx[0], xb, bias[0]), acc1 = SW_MFMA(wx[1], xb, bias[1]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0 = SW_MFMA(whh[0][j][r], bq[j][r], acc0);
        acc1 = SW_MFMA(whh[1][j][r], bq[j][r], acc1);
      }
    float2 gi, gf, gg, go;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float i_ = sw_sigmoid(acc0[e]), f_ = sw_sigmoid(acc0[2 + e]), g_ = sw_tanh(acc1[e]), o_ = sw_sigmoid(acc1[2 + e]);
      const float cp = e ? c.y : c.x;
      const float cn = fmaf(f_, cp, i_ * g_);
      const float hn = o_ * sw_tanh(cn);
      if (e) { gi.y = i_; gf.y = f_; gg.y = g_; go.y = o_; c.y = cn; h.y = hn; }
      else   { gi.x = i_; gf.x = f_; gg.x = g_; go.x = o_; c.x = cn; h.x = hn; }
    }
    float* row = &hbuf[(t + 1) & 1][ln * SW_ALD + ub];
    *reinterpret_cast<float2*>(row) = gi;
    *reinterpret_cast<float2*>(row + 64) = gf;
    *reinterpret_cast<float2*>(row + 128) = gg;
    *reinterpret_cast<float2*>(row + 192) = go;
    *reinterpret_cast<float2*>(row + 256) = c;
    *reinterpret_cast<float2*>(row + 320) = h;
    *xrow = xb;
    xrow += (size_t)B * 4;
    sw_barrier();
    {   // the saved rows of the step: 16 agents x 96 float4, three per thread, 1 KB of consecutive memory per wave instruction
      float* rows_t = act + (size_t)(t0 + t) * B * 384;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int f = k * 512 + (int)threadIdx.x, a = f / 96, c4 = f - a * 96;
        const f32x4 v = ld4(&hbuf[(t + 1) & 1][a * SW_ALD + 4 * c4]);
        st4g(rows_t + (size_t)min(a0 + a, B - 1) * 384 + 4 * c4, v);
      }
    }
    asm volatile("" : "+v"(xa), "+v"(xq));
  }
  *reinterpret_cast<float2*>(hT + (size_t)b * 64 + ub) = h;
  *reinterpret_cast<float2*>(cT + (size_t)b * 64 + ub) = c;
}

// BPTT.  Per step: elementwise gate gradients (lane-local) -> dgates row to HBM (for the
// deferred weight-gradient GEMM) and to LDS -> dh_{t-1} = W_hh^T dgates on the matrix cores.
// The saved rows of step t-1 are fetched while step t computes.  As in enc_lstm_fwd the loop body has NO conditional
// memory operation (DY is a template parameter, the two boundary steps are peeled, padding lanes of the last tile are
// replicas of agent B-1 and store the same values): with conditional loads / stores the compiler waited for
// everything in flight (s_waitcnt vmcnt(0)) behind every step's barrier - the dgates rows just stored included.
template <bool DY>
__global__ __launch_bounds__(SW_THREADS) void enc_lstm_bwd_kernel(
    const float* __restrict__ whh, const float* __restrict__ act, const float* __restrict__ c0,
    const float* __restrict__ dhT, const float* __restrict__ dcT, const float* __restrict__ dy, int B, int T,
    int t0, float* __restrict__ dgates, float* __restrict__ dh0, float* __restrict__ dc0, const float* __restrict__ gimg,
    const float* __restrict__ aux_src, float* __restrict__ aux_dst, const float* __restrict__ aux_mask, long long aux_n) {
  // Workgroups beyond the agent tiles run an auxiliary masked copy dst[i] = mask[i] > 0 ? src[i] : dst[i] (the training
  // step's D.load(backup), train.py:541-542, when the decode BPTT launch - its usual place - also reads D's weights)
  {
    const int tiles = (B + SW_TILE - 1) / SW_TILE;
    if ((int)blockIdx.x >= tiles) {
      const long long stride = (long long)(gridDim.x - tiles) * SW_THREADS;
      for (long long i = (long long)(blockIdx.x - tiles) * SW_THREADS + threadIdx.x; i < aux_n; i += stride)
        if (aux_mask[i] > 0.f) aux_dst[i] = aux_src[i];
      return;
    }
  }
  __shared__ __attribute__((aligned(16))) float dgbuf[2][SW_TILE * SW_GLD];
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16;
  const int a0 = blockIdx.x * SW_TILE;
  const int b = min(a0 + ln, B - 1);
  LstmWT W;
  if (gimg) {   // operand-layout image of W_hh^T of this step (swimg::OP_WHHT): 16 contiguous 1 KB loads per wave
#pragma unroll
    for (int j = 0; j < 16; ++j) W.whhT[j] = ld4(gimg + swimg::OP_WHHT + (((size_t)wave * 16 + j) * 64 + lane) * 4);
  } else {
    lstm_load_wT(W, whh, u0, ln, lg);
  }
  // optional inputs are read unconditionally from a selected address (a load under a branch costs the exact vmcnt
  // bookkeeping of everything behind it) and zeroed afterwards
  const float* act_b = act + ((size_t)t0 * B + b) * 384 + u0 + 4 * lg;
  f32x4 dh = ld4(dhT ? dhT + (size_t)b * 64 + u0 + 4 * lg : act_b);
  f32x4 dc = ld4(dcT ? dcT + (size_t)b * 64 + u0 + 4 * lg : act_b);
  if (!dhT) dh = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!dcT) dc = f32x4{0.f, 0.f, 0.f, 0.f};
  const size_t tstep = (size_t)B * 384;
  using T_ = std::true_type;
  using F_ = std::false_type;
  // rows of local step t; has_prev: row t-1 exists in `act` (always for t >= 1; for t = 0 only if t0 > 0)
  auto load_row = [&](int t, f32x4 g[4], f32x4& ct_, f32x4& cp_, auto has_prev) {
    const float* row = act_b + (size_t)t * tstep;
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = ld4(row + q * 64);
    ct_ = ld4(row + 256);
    if constexpr (decltype(has_prev)::value) {
      cp_ = ld4(row - tstep + 256);
    } else {   // the sequence start: c_{-1} = c0 or zero (or the row in front of t0)
      cp_ = ld4(t0 > 0 ? row - tstep + 256 : c0 ? c0 + (size_t)b * 64 + u0 + 4 * lg : row + 256);
      if (t0 <= 0 && !c0) cp_ = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x4 gate[4], ct, cprev;
  if (T > 1) load_row(T - 1, gate, ct, cprev, T_{});
  else load_row(0, gate, ct, cprev, F_{});
  const float* dyp = DY ? dy + ((size_t)b * T + T - 1) * 64 + u0 + 4 * lg : nullptr;
  auto step = [&](int t, auto pf, auto pp) {   // pf: prefetch the rows of step t-1 (pp: which have a predecessor row)
    f32x4 dgate[4];
    if constexpr (DY) {
      dh += ld4(dyp);
      dyp -= 64;
    }
    lstm_cell_bwd(gate, ct, cprev, dh, dc, dgate);
    // ROLLING prefetch (round 6, as in dec_rollout_bwd_kernel): the rows of step t - 1 are requested right behind the last
    // use of the rows of step t, into the same registers (no second row set)
    if constexpr (decltype(pf)::value) {
      load_row(t - 1, gate, ct, cprev, pp);
      asm volatile("" ::: "memory");
    }
    float* dgl = &dgbuf[t & 1][ln * SW_GLD + u0 + 4 * lg];
#pragma unroll
    for (int g = 0; g < 4; ++g) st4(dgl + g * 64, dgate[g]);
    sw_barrier();
    lstm_store_dgates_tile(dgbuf[t & 1], dgates + ((size_t)(t0 + t) * B + a0) * 256, nullptr, a0, B, wave, lane);
    dh = lstm_dh_prev(W, &dgbuf[t & 1][ln * SW_GLD + 4 * lg]);
    if constexpr (decltype(pf)::value) {
      // the prefetched rows are not touched before the matrix products above have been issued
      asm volatile("" : "+v"(gate[0]), "+v"(gate[1]), "+v"(gate[2]), "+v"(gate[3]), "+v"(ct), "+v"(cprev));
    }
  };
  for (int t = T - 1; t >= 2; --t) step(t, T_{}, T_{});
  if (T > 1) step(1, T_{}, F_{});
  step(0, F_{}, F_{});
  if (dh0) st4(dh0 + (size_t)b * 64 + u0 + 4 * lg, dh);
  if (dc0) st4(dc0 + (size_t)b * 64 + u0 + 4 * lg, dc);
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
  SW_LAUNCH(traj4d_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, obsv, pred, B,
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
  const float* gimg = sw_gen_images_for(enc_w, nullptr);
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  int extra = aux_n > 0 ? (int)((aux_n / 4 + SW_THREADS - 1) / SW_THREADS) : 0;
  if (extra > 64) extra = 64;
  // the 8-wave kernel (bit-identical results): per step -7.5 % when every tile has a CU to itself (m1: 1.67 -> 1.54 us per
  // step), +3.5 % once tiles queue for CUs (c4) - used up to one tile per CU; SW_ENC8=0 / 1 forces either kernel
  static const int enc8_env = getenv("SW_ENC8") ? atoi(getenv("SW_ENC8")) : -1;
  const bool enc8 = enc8_env >= 0 ? enc8_env != 0 : tiles <= 256;
  if (enc8 && gimg && x_mode == 0 && act && x4s && !y) {
    SW_LAUNCH(enc_lstm_fwd8_kernel, dim3(tiles + extra), dim3(512), 0, (hipStream_t)stream, x, h0, c0, B, T, hT, cT, act, x4s, t0,
              aux_src, aux_dst, aux_n, gimg);
    SW_CHECK_LAUNCH("enc_lstm_fwd8_kernel");
    return SW_OK;
  }
#define SW_ENC_FWD(XM, A, Y_, X4)                                                                                   \
  SW_LAUNCH((enc_lstm_fwd_kernel<XM, A, Y_, X4>), dim3(tiles + extra), dim3(SW_THREADS), 0, (hipStream_t)stream,    \
            x, enc_w, h0, c0, B, T, hT, cT, y, act, x4s, t0, aux_src, aux_dst, aux_n, gimg)
  switch ((x_mode ? 8 : 0) | (act ? 4 : 0) | (y ? 2 : 0) | (x4s ? 1 : 0)) {
    case 0: SW_ENC_FWD(0, false, false, false); break;
    case 1: SW_ENC_FWD(0, false, false, true); break;
    case 2: SW_ENC_FWD(0, false, true, false); break;
    case 3: SW_ENC_FWD(0, false, true, true); break;
    case 4: SW_ENC_FWD(0, true, false, false); break;
    case 5: SW_ENC_FWD(0, true, false, true); break;
    case 6: SW_ENC_FWD(0, true, true, false); break;
    case 7: SW_ENC_FWD(0, true, true, true); break;
    case 8: SW_ENC_FWD(1, false, false, false); break;
    case 9: SW_ENC_FWD(1, false, false, true); break;
    case 10: SW_ENC_FWD(1, false, true, false); break;
    case 11: SW_ENC_FWD(1, false, true, true); break;
    case 12: SW_ENC_FWD(1, true, false, false); break;
    case 13: SW_ENC_FWD(1, true, false, true); break;
    case 14: SW_ENC_FWD(1, true, true, false); break;
    default: SW_ENC_FWD(1, true, true, true); break;
  }
#undef SW_ENC_FWD
  SW_CHECK_LAUNCH("enc_lstm_fwd_kernel");
  return SW_OK;
}
extern "C" int sw_enc_lstm_fwd(const float* x, int x_mode, const float* enc_w, const float* h0,
                               const float* c0, int B, int T, float* hT, float* cT, float* y, float* act,
                               float* x4s, int t0, void* stream) {
  return sw_enc_lstm_fwd_aux(x, x_mode, enc_w, h0, c0, B, T, hT, cT, y, act, x4s, t0, nullptr, nullptr, 0, stream);
}

extern "C" int sw_enc_lstm_bwd_aux(const float* enc_w, const float* act, const float* c0, const float* dhT,
                                   const float* dcT, const float* dy, int B, int T, int t0, float* dgates,
                                   float* dh0, float* dc0, const float* aux_src, float* aux_dst, const float* aux_mask,
                                   long long aux_n, void* stream) {
  if (!enc_w || !act || !dgates || B < 0 || T < 1 || t0 < 0) return SW_EARG;
  if (aux_n < 0 || (aux_n > 0 && (!aux_src || !aux_dst || !aux_mask))) return SW_EARG;
  if (B == 0) return SW_OK;
  const float* gimg = sw_gen_images_for(enc_w, nullptr);
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  int extra = aux_n > 0 ? (int)((aux_n + SW_THREADS - 1) / SW_THREADS) : 0;
  if (extra > 64) extra = 64;
  if (dy)
    SW_LAUNCH(enc_lstm_bwd_kernel<true>, dim3(tiles + extra), dim3(SW_THREADS), 0, (hipStream_t)stream, enc_w + swp::ENC_WHH, act, c0,
              dhT, dcT, dy, B, T, t0, dgates, dh0, dc0, gimg, aux_src, aux_dst, aux_mask, aux_n);
  else
    SW_LAUNCH(enc_lstm_bwd_kernel<false>, dim3(tiles + extra), dim3(SW_THREADS), 0, (hipStream_t)stream, enc_w + swp::ENC_WHH, act, c0,
              dhT, dcT, dy, B, T, t0, dgates, dh0, dc0, gimg, aux_src, aux_dst, aux_mask, aux_n);
  SW_CHECK_LAUNCH("enc_lstm_bwd_kernel");
  return SW_OK;
}
extern "C" int sw_enc_lstm_bwd(const float* enc_w, const float* act, const float* c0, const float* dhT,
                               const float* dcT, const float* dy, int B, int T, int t0, float* dgates,
                               float* dh0, float* dc0, void* stream) {
  return sw_enc_lstm_bwd_aux(enc_w, act, c0, dhT, dcT, dy, B, T, t0, dgates, dh0, dc0, nullptr, nullptr, nullptr, 0, stream);
}
