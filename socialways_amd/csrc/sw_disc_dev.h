// sw_disc_dev.h - device code of Discriminator.forward (reference train.py:294-309) shared by sw_disc.hip and the launches
// that run a discriminator pass in front of their own work (sw_decoder.hip): LDS carves, save / delta layouts, the
// per-tile forward body.
#pragma once
#include "../../include/socialways_hip.h"
#include "sw_lstm_dev.h"
#ifndef SW_DSTAMP
#define SW_DSTAMP(k)
#define SW_DSTAMP_INIT
#endif

namespace {
constexpr int LD64 = sw_ld(64);  // 68
constexpr int LD32 = sw_ld(32);  // 36
constexpr int LD16 = sw_ld(16);  // 20
#define SW_DISC_MAXB 2

// time-major / branch-major save + delta layouts (floats)
struct DSave {
  size_t act, x4s, o1, both, q1, c1, l1, px, total;
};
__host__ __device__ inline DSave dsave_layout(int B, int To, int Tp, int nb) {
  DSave d;
  size_t b = (size_t)B;
  d.act = 0;                                // act / x4s do not depend on nb: sw_dec_rollout_fwd_aux writes them too
  d.x4s = d.act + (size_t)To * b * 384;
  d.o1 = d.x4s + (size_t)To * b * 4;
  d.both = d.o1 + b * 32;
  d.q1 = d.both + nb * b * 64;
  d.c1 = d.q1 + nb * b * 32;
  d.l1 = d.c1 + nb * b * 32;
  d.px = d.l1 + nb * b * 32;
  d.total = d.px + nb * b * 4 * Tp;
  return d;
}
struct DDelta {
  size_t dgates, do1, docode, dpcode, dq1, dc1, dl1, dlab, dcod, trash, total;
};
__host__ __device__ inline DDelta ddelta_layout(int B, int To, int Tp, int nb) {
  DDelta d;
  size_t b = (size_t)B;
  d.dgates = 0;
  d.do1 = d.dgates + (size_t)To * b * 256;
  d.docode = d.do1 + b * 32;
  d.dpcode = d.docode + b * 32;
  d.dq1 = d.dpcode + nb * b * 32;
  d.dc1 = d.dq1 + nb * b * 32;
  d.dl1 = d.dc1 + nb * b * 32;
  d.dlab = d.dl1 + nb * b * 32;
  d.dcod = d.dlab + nb * b * 4;
  d.trash = d.dcod + nb * b * 4;   // 16 x 256 floats nobody reads: where the padding lanes of the last tile store
  d.total = d.trash + 16 * 256;    //   (unconditional stores keep the s_waitcnt vmcnt bookkeeping of the loops exact)
  return d;
}

// runtime-K tile product (heads are tiny; KJ <= 4 normally)
__device__ __forceinline__ f32x4 tile_mm_rt(const float* wrow, const float* xrow, int KJ, f32x4 acc) {
  f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < KJ; j0 += 4) {  // groups of <= 4 k-steps: loads first, then the MFMAs
    f32x4 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (j0 + q < KJ) {
        a[q] = ld4(wrow + 16 * (j0 + q));
        b[q] = ld4(xrow + 16 * (j0 + q));
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (j0 + q < KJ) {
        acc = SW_MFMA(a[q][0], b[q][0], acc);
        acc1 = SW_MFMA(a[q][1], b[q][1], acc1);
        acc = SW_MFMA(a[q][2], b[q][2], acc);
        acc1 = SW_MFMA(a[q][3], b[q][3], acc1);
      }
    }
  }
  return acc + acc1;
}

struct HeadLds {  // forward LDS carve for a given padded pred width KP = roundup(4Tp,16)
  int of0, of1, pe0, pe1, cl0, la0, cl1, la1, bias, hlast, o1, x, q1, both, c1, l1, total, ldp;
};
__host__ __device__ inline HeadLds head_lds(int Tp, int base) {
  HeadLds L;
  int KP = ((4 * Tp + 15) / 16) * 16;
  L.ldp = KP + 4;
  int o = base;
  L.of0 = o; o += 32 * LD64;
  L.of1 = o; o += 32 * LD32;
  L.pe0 = o; o += 32 * L.ldp;
  L.pe1 = o; o += 32 * LD32;
  L.cl0 = o; o += 32 * LD64;
  L.la0 = o; o += 32 * LD64;
  L.cl1 = o; o += 16 * LD32;
  L.la1 = o; o += 16 * LD32;
  L.bias = o; o += 8 * 32;  // of0 of1 pe0 pe1 cl0 la0 cl1(16 used) la1(16 used)
  L.hlast = o; o += 16 * LD64;
  L.o1 = o; o += 16 * LD32;
  // the prediction rows are staged after the observation heads (the only readers of of0) are done: 3.3 KB less at the
  // usual horizons, and two workgroups of the plain forward pass fit one CU's 160 KB (dense crowds: 8+ tiles per CU)
  if (16 * L.ldp <= 32 * LD64) L.x = L.of0;
  else { L.x = o; o += 16 * L.ldp; }
  L.q1 = o; o += 16 * LD32;
  L.both = o; o += 16 * LD64;
  L.c1 = o; o += 16 * LD32;
  L.l1 = o; o += 16 * LD32;
  L.total = o;
  return L;
}
}  // namespace

namespace {
struct HeadLdsB {
  int of0T, of1T, pe0T, pe1T, cl0T, la0T, cl1T, la1T, dlab, dcod, dc1, dl1, dboth, docode, dq1, do1, total, kp;
};
__host__ __device__ inline HeadLdsB head_lds_b(int Tp, int base) {
  HeadLdsB L;
  L.kp = ((4 * Tp + 15) / 16) * 16;
  int o = base;
  L.of0T = o; o += 64 * LD32;       // [64][36]   of0T[m][k] = of0[k][m]
  L.of1T = o; o += 32 * LD32;
  L.pe0T = o; o += L.kp * LD32;     // [kp][36]
  L.pe1T = o; o += 32 * LD32;
  L.cl0T = o; o += 64 * LD32;
  L.la0T = o; o += 64 * LD32;
  L.cl1T = o; o += 32 * LD16;       // [32][20]  (K = 1)
  L.la1T = o; o += 32 * LD16;       // [32][20]  (K = 2)
  L.dlab = o; o += 16 * LD16;
  L.dcod = o; o += 16 * LD16;
  L.dc1 = o; o += 16 * LD32;
  L.dl1 = o; o += 16 * LD32;
  L.dboth = o; o += 16 * LD64;
  L.dq1 = o; o += 16 * LD32;
  L.docode = o; o += 16 * LD32;     // docode, do1 last: they are read after the heads (observation path), everything
  L.do1 = o; o += 16 * LD32;        // from pe0T up to here is dead by then and carries the BPTT's dgates tiles (below)
  L.total = o;
  return L;
}
// disc_bwd's double-buffered dgates tiles [2][16][SW_GLD] live where the prediction heads' transposed weight images and
// delta tiles were (all dead once the heads are done, one barrier earlier): 33 KB less LDS, and with <= 256 VGPRs two
// workgroups fit a CU - what dense crowds (8+ tiles per CU) need to hide one tile's latencies behind another's work
static_assert(7040 + 36 * 16 + 320 + 320 + 576 + 576 + 1088 + 576 >= 2 * 16 * SW_GLD, "dgates tiles fit the dead region");
}  // namespace

// GAN mode of the backward kernel: dlabel_* / dcode_* then carry the forward OUTPUTS (label, code) and
// the loss gradients are formed in the kernel, so no separate loss kernel sits on the critical path.
struct DiscLoss {
  const float* targets;  // device: label-noise scalars
  const float* z;        // [B][32] latent
  int t0, t1;            // target index of branch 0 / 1
  float g_label, g_code;
  int on;
  float* loss_part;      // [tiles][3] per-tile sums of the squared errors (reporting), or null
};

// The forward pass of D for the 16-agent tile(s) of workgroup index `bx` of `nbx` (see disc_fwd_kernel, sw_disc.hip); with
// `fuse` (generator phase, one branch) also the backward of the prediction heads down to d(loss)/d(pred).  A device function
// so that another launch can run the generator-phase pass of a tile in front of its own work (dec_rollout_bwd_kernel<true>,
// sw_decoder.hip: the pass is tile-local and its only consumer is that tile's decode BPTT).
__device__ __forceinline__ void disc_fwd_tile(float* smem, const unsigned bx, const unsigned nbx,
    const float* __restrict__ obsv, int To, int x_mode, const float* __restrict__ pred_a,
    const float* __restrict__ pred_b, int nb, const float* __restrict__ d_w, int B, int Tp, float* __restrict__ label_a, float* __restrict__ label_b,
    float* __restrict__ code_a, float* __restrict__ code_b, float* __restrict__ dsave, int save_lstm, int split,
    float* __restrict__ w_snap, int fuse, DiscLoss gl, float* __restrict__ dpred_out, const float* __restrict__ dimg) {
  // LSTM part
  float* hbuf = smem;                        // [2][16][68]
  const HeadLds L = head_lds(Tp, 2 * 16 * SW_HLD + 1280);   // (the 1280 floats in between: Wx | bx of disc_obs_lstm_tile)
  const swp::Disc O = swp::disc(Tp);
  const DSave ds = dsave_layout(B, To, Tp, nb);
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16;
  // split: the two branches of a tile run in two workgroups (each repeats the shared observation LSTM - free
  // while the launch leaves CUs idle - and does ONE head pass); the branch-1 workgroup saves no observation rows
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  const int bsel = split ? (int)(bx >= (unsigned)tiles) : -1;
  const int k_lo = bsel == 1 ? 1 : 0, k_hi = bsel == 0 ? 1 : nb;
  const bool save_obs = bsel != 1;
  const int a0 = (bx - (bsel == 1 ? tiles : 0)) * SW_TILE;
  const int b = min(a0 + ln, B - 1);
  const bool live = (a0 + ln) < B;
  const int K4 = 4 * Tp;

  // fuse (generator phase, one branch): the backward of the heads down to d(loss)/d(pred) runs in this kernel too -
  // its transposed weight images and delta buffers follow the forward carve in LDS, the activations never leave LDS
  const HeadLdsB LB = head_lds_b(Tp, L.total);
  SW_DSTAMP_INIT;
  // The prediction rows of this workgroup's branches (inputs of the pred_encoder heads) are requested NOW: staged
  // where the heads start they cost one global round trip per branch behind the observation LSTM.  Unconditional loads
  // from clamped addresses (4 per thread and branch cover the [16][4 Tp + pad] tile up to Tp = 12; longer horizons
  // load in place).
  const bool x_pre = 16 * L.ldp <= 4 * SW_THREADS;
  float xpre[2][4];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const float* pk = (k_lo + kk == 0 || nb == 1) ? pred_a : pred_b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = min((int)threadIdx.x + SW_THREADS * e, 16 * L.ldp - 1);
      const int a = i / L.ldp, cc = i - a * L.ldp;
      xpre[kk][e] = pk[(size_t)min(a0 + a, B - 1) * K4 + min(cc, K4 - 1)];
    }
  }
  // (generator phase) the label target and the agent's latent code for the loss gradients formed behind the heads
  const float* ztop = fuse ? gl.z + (size_t)b * SW_Z : d_w;
  const float ftg = (fuse ? gl.targets + gl.t0 : d_w)[0], fz0 = ztop[0], fz1 = ztop[1];
  if (fuse) {
    if (dimg) stage_zero(smem + LB.dlab, LB.dc1 - LB.dlab);   // dlab, dcod (the image block below carries its own zero padding)
    else stage_zero(smem + LB.of0T, LB.dc1 - LB.of0T);        // transposed images (zero padded) + dlab, dcod
  }
  const bool obs_pre = save_lstm == 2;   // LSTM rows already in dsave (sw_dec_rollout_fwd_aux ran the observation LSTM)
  LstmW W;
  if (!obs_pre) {   // global loads in flight during the LDS staging
    if (dimg) {     // operand-layout image: a wave's load = 1 KB of consecutive memory
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) W.whh[g][j] = ld4(dimg + swdimg::OP_WHH + ((((size_t)4 * g + wave) * 4 + j) * 64 + lane) * 4);
    } else {
      lstm_load_whh(W, d_w + O.whh, u0, ln, lg);
    }
    // input matrix W_ih [256][4] and b_ih + b_hh straight into their registers (no LDS staging, nothing behind a barrier)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      W.wx[g] = d_w[O.wih + (g * 64 + u0 + ln) * 4 + lg];
      W.bias[g] = ld4(d_w + O.bih + g * 64 + u0 + 4 * lg) + ld4(d_w + O.bhh + g * 64 + u0 + 4 * lg);
    }
  }
  if (w_snap)   // deepcopy(D) of train.py:499: the weights this pass runs with, a few floats per thread
    for (int i = bx * SW_THREADS + threadIdx.x; i < O.n; i += nbx * SW_THREADS) w_snap[i] = d_w[i];
  // ---- stage head weights / biases: ALL global loads first, then the LDS stores (one L2 round trip for the eight
  //      matrices instead of one each) --------------------------------------------------------------------------
  {
    f32x4 s_of0[3], s_of1[2], s_pe0[2], s_pe1[2], s_cl0[3], s_la0[3], s_cl1[1], s_la1[1];
    const bool pe0_small = 32 * (L.ldp >> 2) <= 2 * SW_THREADS;   // 4 Tp <= 48: the usual horizons
    stage_w_load<3>(s_of0, LD64, 32, d_w + O.of0w, 64, 32, 64);
    stage_w_load<2>(s_of1, LD32, 32, d_w + O.of1w, 32, 32, 32);
    if (pe0_small) stage_w_load<2>(s_pe0, L.ldp, 32, d_w + O.pe0w, K4, 32, K4);
    stage_w_load<2>(s_pe1, LD32, 32, d_w + O.pe1w, 32, 32, 32);
    stage_w_load<3>(s_cl0, LD64, 32, d_w + O.cl0w, 64, 32, 64);
    stage_w_load<3>(s_la0, LD64, 32, d_w + O.la0w, 64, 32, 64);
    stage_w_load<1>(s_cl1, LD32, 16, d_w + O.cl1w, 32, 1, 32);
    stage_w_load<1>(s_la1, LD32, 16, d_w + O.la1w, 32, 2, 32);
    stage_w_store<3>(s_of0, smem + L.of0, LD64, 32);
    stage_w_store<2>(s_of1, smem + L.of1, LD32, 32);
    if (pe0_small) stage_w_store<2>(s_pe0, smem + L.pe0, L.ldp, 32);
    else stage_w(smem + L.pe0, L.ldp, 32, d_w + O.pe0w, K4, 32, K4);
    stage_w_store<2>(s_pe1, smem + L.pe1, LD32, 32);
    stage_w_store<3>(s_cl0, smem + L.cl0, LD64, 32);
    stage_w_store<3>(s_la0, smem + L.la0, LD64, 32);
    stage_w_store<1>(s_cl1, smem + L.cl1, LD32, 16);
    stage_w_store<1>(s_la1, smem + L.la1, LD32, 16);
  }
  {   // the eight bias vectors: ONE unconditional load per thread from a selected offset (eight loads under lane branches
      // compiled to eight serial round trips, each behind an s_waitcnt vmcnt(0))
    const int i = threadIdx.x;  // 256 = 8 x 32
    const int q = i >> 5, k = i & 31;
    const int boff = q == 0 ? O.of0b : q == 1 ? O.of1b : q == 2 ? O.pe0b : q == 3 ? O.pe1b : q == 4 ? O.cl0b
                     : q == 5 ? O.la0b : q == 6 ? O.cl1b : O.la1b;
    const int lim = q < 6 ? 32 : (q == 6 ? 1 : 2);
    const float v = d_w[boff + min(k, lim - 1)];
    smem[L.bias + i] = k < lim ? v : 0.f;
  }
  if (fuse && dimg) {
    // transposed head images of the fused backward (swdimg::HEADT): the block pe0T .. la1T is one contiguous float4
    // copy, zero padding included - requested behind every other load of the prologue, so nothing waits for it alone
    constexpr int HB = 12;
    const int n4 = (LB.dlab - LB.pe0T) >> 2;
    const float* src = dimg + swdimg::HEADT + (LB.pe0T - LB.of0T);
    f32x4 hbv[HB];
#pragma unroll
    for (int e = 0; e < HB; ++e) hbv[e] = ld4(src + 4 * (size_t)min((int)threadIdx.x + SW_THREADS * e, n4 - 1));
#pragma unroll
    for (int e = 0; e < HB; ++e) {
      const int f = threadIdx.x + SW_THREADS * e;
      if (f < n4) st4(smem + LB.pe0T + 4 * f, hbv[e]);
    }
    for (int f = threadIdx.x + SW_THREADS * HB; f < n4; f += SW_THREADS) st4(smem + LB.pe0T + 4 * f, ld4(src + 4 * (size_t)f));
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};  // h0 = c0 = 0 (train.py:296-297)
  if (!obs_pre) {
    st4(&hbuf[ln * SW_HLD + u0 + 4 * lg], h);
  } else {   // h_T of the tile from the saved rows
    st4(&hbuf[(To & 1) * 16 * SW_HLD + ln * SW_HLD + u0 + 4 * lg],
        ld4(dsave + ds.act + ((size_t)(To - 1) * B + b) * 384 + 320 + u0 + 4 * lg));
  }
  sw_barrier();
  if (fuse && dimg) {
    // (copied from the image block in front of the barrier above)
  } else if (fuse) {
    f32x4 t_pe0[2], t_pe1[1], t_cl0[2], t_la0[2], t_cl1[1], t_la1[1];
    const bool pe0_small = 8 * K4 <= 2 * SW_THREADS;
    if (pe0_small) stage_wT_load<2>(t_pe0, d_w + O.pe0w, K4, 32, K4);
    stage_wT_load<1>(t_pe1, d_w + O.pe1w, 32, 32, 32);
    stage_wT_load<2>(t_cl0, d_w + O.cl0w, 64, 32, 64);
    stage_wT_load<2>(t_la0, d_w + O.la0w, 64, 32, 64);
    stage_wT_load<1>(t_cl1, d_w + O.cl1w, 32, 1, 32);
    stage_wT_load<1>(t_la1, d_w + O.la1w, 32, 2, 32);
    if (pe0_small) stage_wT_store<2>(t_pe0, smem + LB.pe0T, LD32, 32, K4);
    else stage_wT(smem + LB.pe0T, LD32, LB.kp, d_w + O.pe0w, K4, 32, K4);
    stage_wT_store<1>(t_pe1, smem + LB.pe1T, LD32, 32, 32);
    stage_wT_store<2>(t_cl0, smem + LB.cl0T, LD32, 32, 64);
    stage_wT_store<2>(t_la0, smem + LB.la0T, LD32, 32, 64);
    stage_wT_store<1>(t_cl1, smem + LB.cl1T, LD16, 1, 32);
    stage_wT_store<1>(t_la1, smem + LB.la1T, LD16, 2, 32);
  }

  // ---- LSTM over the observation (4-d state formed on the fly, train.py:130-133): lstm_obs_loop, chosen once ----
  SW_DSTAMP(0);
  if (!obs_pre) {
    const bool sv = dsave && save_lstm && save_obs;
    float* act = sv ? dsave + ds.act : nullptr;
    float* x4s = sv ? dsave + ds.x4s : nullptr;
    if (x_mode == 0) {
      if (sv) lstm_obs_loop<0, true>(W, hbuf, obsv, To, B, b, c, h, act, x4s);
      else lstm_obs_loop<0, false>(W, hbuf, obsv, To, B, b, c, h, act, x4s);
    } else {
      if (sv) lstm_obs_loop<1, true>(W, hbuf, obsv, To, B, b, c, h, act, x4s);
      else lstm_obs_loop<1, false>(W, hbuf, obsv, To, B, b, c, h, act, x4s);
    }
  }
  const float* hlast = &hbuf[(To & 1) * 16 * SW_HLD];
  SW_DSTAMP(1);

  // ---- heads ------------------------------------------------------------------------------------
  // pred branches into LDS rows [16][ldp] (zero padded), saved flat for the pe0 weight gradient
  // phase A: o1 = lrelu(of0 h + b)  (waves 0,1)
  if (wave < 2) {
    int m0 = 16 * wave;
    f32x4 acc = ld4(smem + L.bias + 0 * 32 + m0 + 4 * lg);
    acc = tile_mm_rt(smem + L.of0 + (m0 + ln) * LD64 + 4 * lg, hlast + ln * SW_HLD + 4 * lg, 4, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu(acc[r]);
    st4(smem + L.o1 + ln * LD32 + m0 + 4 * lg, acc);
    if (dsave && live && save_obs) st4(dsave + ds.o1 + (size_t)b * 32 + m0 + 4 * lg, acc);
  }
  sw_barrier();
  // phase B: obsv_code = of1 o1 + b  -> both[:, 0:32]  (waves 0,1)
  if (wave < 2) {
    int m0 = 16 * wave;
    f32x4 acc = ld4(smem + L.bias + 1 * 32 + m0 + 4 * lg);
    acc = tile_mm_rt(smem + L.of1 + (m0 + ln) * LD32 + 4 * lg, smem + L.o1 + ln * LD32 + 4 * lg, 2, acc);
    st4(smem + L.both + ln * LD64 + m0 + 4 * lg, acc);
  }
  SW_DSTAMP(2);
  for (int k = k_lo; k < k_hi; ++k) {
    const float* pred = k == 0 ? pred_a : pred_b;
    float* label = k == 0 ? label_a : label_b;
    float* code = k == 0 ? code_a : code_b;
    sw_barrier();
    if (x_pre) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = threadIdx.x + SW_THREADS * e;
        if (i < 16 * L.ldp) {
          const int a = i / L.ldp, cc = i - a * L.ldp;
          const int bb = min(a0 + a, B - 1);
          const float v = cc < K4 ? (k == k_lo ? xpre[0][e] : xpre[1][e]) : 0.f;
          smem[L.x + i] = v;
          if (dsave && cc < K4 && a0 + a < B) dsave[ds.px + ((size_t)k * B + bb) * K4 + cc] = v;
        }
      }
    } else {
      for (int i = threadIdx.x; i < 16 * L.ldp; i += blockDim.x) {
        int a = i / L.ldp, cc = i - a * L.ldp;
        int bb = min(a0 + a, B - 1);
        float v = cc < K4 ? pred[(size_t)bb * K4 + cc] : 0.f;
        smem[L.x + i] = v;
        if (dsave && cc < K4 && a0 + a < B) dsave[ds.px + ((size_t)k * B + bb) * K4 + cc] = v;
      }
    }
    sw_barrier();
    // q1 = lrelu(pe0 x + b)   (waves 0,1)
    if (wave < 2) {
      int m0 = 16 * wave;
      f32x4 acc = ld4(smem + L.bias + 2 * 32 + m0 + 4 * lg);
      acc = tile_mm_rt(smem + L.pe0 + (m0 + ln) * L.ldp + 4 * lg, smem + L.x + ln * L.ldp + 4 * lg, (L.ldp - 4) / 16, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu(acc[r]);
      st4(smem + L.q1 + ln * LD32 + m0 + 4 * lg, acc);
      if (dsave && live) st4(dsave + ds.q1 + ((size_t)k * B + b) * 32 + m0 + 4 * lg, acc);
    }
    sw_barrier();
    // pred_code = pe1 q1 + b -> both[:, 32:64]   (waves 0,1)
    if (wave < 2) {
      int m0 = 16 * wave;
      f32x4 acc = ld4(smem + L.bias + 3 * 32 + m0 + 4 * lg);
      acc = tile_mm_rt(smem + L.pe1 + (m0 + ln) * LD32 + 4 * lg, smem + L.q1 + ln * LD32 + 4 * lg, 2, acc);
      st4(smem + L.both + ln * LD64 + 32 + m0 + 4 * lg, acc);
    }
    sw_barrier();
    if (dsave && live) {  // both codes (64) saved by all 4 waves, 16 floats each
      st4(dsave + ds.both + ((size_t)k * B + b) * 64 + u0 + 4 * lg, ld4(smem + L.both + ln * LD64 + u0 + 4 * lg));
    }
    // c1 = lrelu(cl0 both + b) (waves 0,1) ; l1 = lrelu(la0 both + b) (waves 2,3)
    {
      int m0 = 16 * (wave & 1);
      bool cls = wave < 2;
      f32x4 acc = ld4(smem + L.bias + (cls ? 4 : 5) * 32 + m0 + 4 * lg);
      acc = tile_mm_rt(smem + (cls ? L.cl0 : L.la0) + (m0 + ln) * LD64 + 4 * lg, smem + L.both + ln * LD64 + 4 * lg, 4, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu(acc[r]);
      st4(smem + (cls ? L.c1 : L.l1) + ln * LD32 + m0 + 4 * lg, acc);
      if (dsave && live) st4(dsave + (cls ? ds.c1 : ds.l1) + ((size_t)k * B + b) * 32 + m0 + 4 * lg, acc);
    }
    sw_barrier();
    // label = cl1 c1 + b (wave 0) ; code_hat = la1 l1 + b (wave 1)
    if (wave < 2) {
      bool cls = wave == 0;
      f32x4 acc = ld4(smem + L.bias + (cls ? 6 : 7) * 32 + 4 * lg);
      acc = tile_mm_rt(smem + (cls ? L.cl1 : L.la1) + ln * LD32 + 4 * lg, smem + (cls ? L.c1 : L.l1) + ln * LD32 + 4 * lg, 2, acc);
      if (lg == 0 && live) {
        if (cls) { if (label) label[b] = acc[0]; }
        else if (code) { code[(size_t)b * 2] = acc[0]; code[(size_t)b * 2 + 1] = acc[1]; }
      }
      if (fuse) {   // LSGAN / InfoGAN loss gradients (train.py:512-523) and this tile's reported sums, lanes lg == 0
        float sl = 0.f;
        if (lg == 0) {
          if (cls) {
            const float e = acc[0] - ftg;
            smem[LB.dlab + ln * LD16] = 2.0f * e * gl.g_label;
            sl = live ? e * e : 0.f;
          } else {
            const float c0 = acc[0] - fz0, c1 = acc[1] - fz1;
            smem[LB.dcod + ln * LD16] = 2.0f * c0 * gl.g_code;
            smem[LB.dcod + ln * LD16 + 1] = 2.0f * c1 * gl.g_code;
            sl = live ? c0 * c0 + c1 * c1 : 0.f;
          }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sl += __shfl_xor(sl, o);
        if (gl.loss_part && lane == 0) gl.loss_part[(size_t)bx * 3 + (cls ? 0 : 1)] = sl;
      }
    }
  }
  SW_DSTAMP(3);
  if (!fuse) return;
  // ---- fused backward of the heads of branch 0: d(loss)/d(pred) only (generator phase) -----------------------
  sw_barrier();
  {  // dc1 = (cl1^T dlabel) * lrelu'(c1)  (waves 0,1) ; dl1 = (la1^T dcode) * lrelu'(l1)  (waves 2,3)
    int m0 = 16 * (wave & 1);
    bool cls = wave < 2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_mm_rt(smem + (cls ? LB.cl1T : LB.la1T) + (m0 + ln) * LD16 + 4 * lg,
                     smem + (cls ? LB.dlab : LB.dcod) + ln * LD16 + 4 * lg, 1, acc);
    f32x4 a = ld4(smem + (cls ? L.c1 : L.l1) + ln * LD32 + m0 + 4 * lg);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu_grad(a[r], acc[r]);
    st4(smem + (cls ? LB.dc1 : LB.dl1) + ln * LD32 + m0 + 4 * lg, acc);
  }
  sw_barrier();
  if (wave >= 2) {  // d(pred_code) = rows 32..63 of cl0^T dc1 + la0^T dl1
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_mm_rt(smem + LB.cl0T + (u0 + ln) * LD32 + 4 * lg, smem + LB.dc1 + ln * LD32 + 4 * lg, 2, acc);
    acc = tile_mm_rt(smem + LB.la0T + (u0 + ln) * LD32 + 4 * lg, smem + LB.dl1 + ln * LD32 + 4 * lg, 2, acc);
    st4(smem + LB.dboth + ln * LD64 + u0 + 4 * lg, acc);
  }
  sw_barrier();
  if (wave < 2) {  // dq1 = (pe1^T dpcode) * lrelu'(q1)
    int m0 = 16 * wave;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_mm_rt(smem + LB.pe1T + (m0 + ln) * LD32 + 4 * lg, smem + LB.dboth + ln * LD64 + 32 + 4 * lg, 2, acc);
    f32x4 a = ld4(smem + L.q1 + ln * LD32 + m0 + 4 * lg);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu_grad(a[r], acc[r]);
    st4(smem + LB.dq1 + ln * LD32 + m0 + 4 * lg, acc);
  }
  sw_barrier();
  for (int mt = wave; mt * 16 < K4; mt += 4) {  // dpred = pe0^T dq1   (4Tp rows)
    int m0 = 16 * mt;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_mm_rt(smem + LB.pe0T + (m0 + ln) * LD32 + 4 * lg, smem + LB.dq1 + ln * LD32 + 4 * lg, 2, acc);
    if (live && m0 + 4 * lg < K4) st4(dpred_out + (size_t)b * K4 + m0 + 4 * lg, acc);
  }
  SW_DSTAMP(4);
}
