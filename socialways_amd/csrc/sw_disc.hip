// sw_disc.hip - Discriminator.forward (reference train.py:294-309) and its backward:
//   LSTM(4->64) over the observed 4-d track, FC 64->32->32 on its last output, FC 4Tp->32->32 on
//   the flattened (predicted or real) future, classifier 64->32->1 (raw LSGAN score) and the
//   InfoGAN latent-code head 64->32->2.
// One workgroup per 16-agent tile.  The observation encoding does not depend on the future
// branch, so the fake and the real branch of a D update (train.py:482,487) share ONE LSTM pass:
// `nb` branches are evaluated per call.
#include "../../include/socialways_hip.h"
#include "sw_lstm_dev.h"
#include "sw_wgrad.h"
#include "sw_wgrad_dev.h"
#include <type_traits>
#include <cstdlib>
#ifdef SW_PHASE_STAMPS
__device__ long long sw_disc_stamps[16];
#define SW_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); long long _t = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) sw_disc_stamps[k] += _t - _tprev; _tprev = _t; __builtin_amdgcn_sched_barrier(0); } while (0)
#define SW_STAMP_INIT long long _tprev = clock64()
extern "C" int sw_debug_disc_stamps(long long* out, int reset) {
  if (reset) { long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(sw_disc_stamps), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sw_disc_stamps), 16 * sizeof(long long));
}
#else
#define SW_STAMP(k)
#define SW_STAMP_INIT
#endif

#define SW_DSTAMP(k) SW_STAMP(k)
#define SW_DSTAMP_INIT SW_STAMP_INIT
#include "sw_disc_dev.h"

__global__ __launch_bounds__(SW_THREADS, 2) void disc_fwd_kernel(
    const float* __restrict__ obsv, int To, int x_mode, const float* __restrict__ pred_a,
    const float* __restrict__ pred_b, int nb, const float* __restrict__ d_w, int B, int Tp, float* __restrict__ label_a, float* __restrict__ label_b,
    float* __restrict__ code_a, float* __restrict__ code_b, float* __restrict__ dsave, int save_lstm, int split,
    float* __restrict__ w_snap, int fuse, DiscLoss gl, float* __restrict__ dpred_out, const float* __restrict__ dimg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  disc_fwd_tile(smem, blockIdx.x, gridDim.x, obsv, To, x_mode, pred_a, pred_b, nb, d_w, B, Tp, label_a, label_b, code_a, code_b, dsave,
                save_lstm, split, w_snap, fuse, gl, dpred_out, dimg);
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
#define SW_SPLIT_MAX_WGS 256   // CUs of an MI355X: splitting only pays while the launch leaves some of them idle

__global__ __launch_bounds__(SW_THREADS, 2) void disc_bwd_kernel(
    const float* __restrict__ d_w, const float* __restrict__ dsave, const float* __restrict__ dlabel_a,
    const float* __restrict__ dlabel_b, const float* __restrict__ dcode_a, const float* __restrict__ dcode_b, int nb,
    int B, int To, int Tp, int want_w, float* __restrict__ ddelta, float* __restrict__ dpred_a,
    float* __restrict__ dpred_b, DiscLoss gl, const float* __restrict__ dimg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const HeadLdsB L = head_lds_b(Tp, 0);
  float* dgbuf = smem + L.pe0T;  // [2][16][260], aliasing the prediction heads' images / deltas (dead when the BPTT starts)
  const swp::Disc O = swp::disc(Tp);
  const DSave ds = dsave_layout(B, To, Tp, nb);
  const DDelta dd = ddelta_layout(B, To, Tp, nb);
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16;
  const int a0 = blockIdx.x * SW_TILE;
  const int b = min(a0 + ln, B - 1);
  const bool live = (a0 + ln) < B;
  const int K4 = 4 * Tp;

  SW_STAMP_INIT;
  // saved activations the head backward needs (post-lrelu c1 | l1, q1 per branch, o1): requested now, they arrive
  // under the weight staging instead of one L2 round trip in front of every head layer.  Unconditional loads (every
  // wave fetches a valid row, waves 2, 3 do not use q1 / o1): see the BPTT loop below.
  const int m0h = 16 * (wave & 1);
  f32x4 pc1[2], pq1[2], po1;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const size_t kb = (size_t)min(k, nb - 1) * B + b;
    pc1[k] = ld4(dsave + (wave < 2 ? ds.c1 : ds.l1) + kb * 32 + m0h + 4 * lg);
    pq1[k] = ld4(dsave + ds.q1 + kb * 32 + m0h + 4 * lg);
  }
  po1 = ld4(dsave + ds.o1 + (size_t)b * 32 + m0h + 4 * lg);
  // ... and the inputs of the loss gradients (forward outputs label / code_hat, z, the label targets): requested here,
  // not where the head backward starts (a global round trip of its own behind the staging barrier).  Element e of the
  // [16][LD16] delta tiles this thread fills, plus the per-agent values of the reported sums; unconditional loads.
  float hl[2][2], hc[2][2], hz[2], htg[2], rl[2], rc[2], rz[2];
  {
    const float* zsrc = gl.on ? gl.z : dsave;            // no GAN mode: any readable address, the value is unused
    const float* tsrc = gl.on ? gl.targets : dsave;
    htg[0] = tsrc[gl.on ? gl.t0 : 0];
    htg[1] = tsrc[gl.on ? gl.t1 : 0];
    const float* dl1 = nb > 1 ? dlabel_b : dlabel_a;
    const float* dc1p = nb > 1 ? dcode_b : dcode_a;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = min((int)threadIdx.x + SW_THREADS * e, 16 * LD16 - 1);
      const int a = i / LD16, c1 = min(i - a * LD16, 1);
      const int bb = min(a0 + a, B - 1);
      hl[e][0] = dlabel_a[bb];
      hl[e][1] = dl1[bb];
      hc[e][0] = dcode_a[(size_t)bb * 2 + c1];
      hc[e][1] = dc1p[(size_t)bb * 2 + c1];
      hz[e] = zsrc[(size_t)bb * SW_Z + c1];
    }
    const int bb = min(a0 + ln, B - 1);
    rl[0] = dlabel_a[bb];
    rl[1] = dl1[bb];
    rc[0] = dcode_a[(size_t)bb * 2];
    rc[1] = dcode_a[(size_t)bb * 2 + 1];
    rz[0] = zsrc[(size_t)bb * SW_Z];
    rz[1] = zsrc[(size_t)bb * SW_Z + 1];
  }
  if (dimg) {
    // registered images (swdimg::HEADT): the eight transposed, zero-padded head matrices are ONE contiguous block in
    // the LDS layout of this kernel - a float4 copy, no zero fill, no scalar scatter
    constexpr int HB = 12;
    const int n4 = L.dlab >> 2;
    f32x4 hbv[HB];
#pragma unroll
    for (int e = 0; e < HB; ++e) hbv[e] = ld4(dimg + swdimg::HEADT + 4 * (size_t)min((int)threadIdx.x + SW_THREADS * e, n4 - 1));
#pragma unroll
    for (int e = 0; e < HB; ++e) {
      const int f = threadIdx.x + SW_THREADS * e;
      if (f < n4) st4(smem + 4 * f, hbv[e]);
    }
    for (int f = threadIdx.x + SW_THREADS * HB; f < n4; f += SW_THREADS) st4(smem + 4 * f, ld4(dimg + swdimg::HEADT + 4 * (size_t)f));
  } else {
  // transposed weight images: all global loads are issued before the zero fill and its barrier
  f32x4 t_of0[2], t_of1[1], t_pe0[2], t_pe1[1], t_cl0[2], t_la0[2], t_cl1[1], t_la1[1];
  const bool pe0_small = 8 * K4 <= 2 * SW_THREADS;
  stage_wT_load<2>(t_of0, d_w + O.of0w, 64, 32, 64);
  stage_wT_load<1>(t_of1, d_w + O.of1w, 32, 32, 32);
  if (pe0_small) stage_wT_load<2>(t_pe0, d_w + O.pe0w, K4, 32, K4);
  stage_wT_load<1>(t_pe1, d_w + O.pe1w, 32, 32, 32);
  stage_wT_load<2>(t_cl0, d_w + O.cl0w, 64, 32, 64);
  stage_wT_load<2>(t_la0, d_w + O.la0w, 64, 32, 64);
  stage_wT_load<1>(t_cl1, d_w + O.cl1w, 32, 1, 32);
  stage_wT_load<1>(t_la1, d_w + O.la1w, 32, 2, 32);
  stage_zero(smem + L.of0T, L.dlab - L.of0T);  // transposed images are zero padded
  sw_barrier();
  stage_wT_store<2>(t_of0, smem + L.of0T, LD32, 32, 64);
  stage_wT_store<1>(t_of1, smem + L.of1T, LD32, 32, 32);
  if (pe0_small) stage_wT_store<2>(t_pe0, smem + L.pe0T, LD32, 32, K4);
  else stage_wT(smem + L.pe0T, LD32, L.kp, d_w + O.pe0w, K4, 32, K4);
  stage_wT_store<1>(t_pe1, smem + L.pe1T, LD32, 32, 32);
  stage_wT_store<2>(t_cl0, smem + L.cl0T, LD32, 32, 64);
  stage_wT_store<2>(t_la0, smem + L.la0T, LD32, 32, 64);
  stage_wT_store<1>(t_cl1, smem + L.cl1T, LD16, 1, 32);
  stage_wT_store<1>(t_la1, smem + L.la1T, LD16, 2, 32);
  }
  for (int i = threadIdx.x; i < 16 * LD32; i += blockDim.x) smem[L.docode + i] = 0.f;

  for (int k = 0; k < nb; ++k) {
    float* dpred = k == 0 ? dpred_a : dpred_b;
    sw_barrier();
    if (k == 0) SW_STAMP(8);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = threadIdx.x + SW_THREADS * e;
      if (i < 16 * LD16) {
        const int a = i / LD16, cc = i - a * LD16;
        const int bb = min(a0 + a, B - 1);
        const float xl = k == 0 ? hl[e][0] : hl[e][1], xc = k == 0 ? hc[e][0] : hc[e][1];
        float vl, vc;
        if (gl.on) {  // LSGAN / InfoGAN loss gradients formed here from the forward outputs (train.py:484-494, 512-523)
          const float tgt = k == 0 ? htg[0] : htg[1];
          vl = cc == 0 ? 2.0f * (xl - tgt) * gl.g_label : 0.f;
          vc = (k == 0 && cc < 2) ? 2.0f * (xc - hz[e]) * gl.g_code : 0.f;
        } else {
          vl = cc == 0 ? xl : 0.f;
          vc = cc < 2 ? xc : 0.f;
        }
        smem[L.dlab + i] = vl;
        smem[L.dcod + i] = vc;
        if (want_w && a0 + a < B && cc < 4) {
          ddelta[dd.dlab + ((size_t)k * B + bb) * 4 + cc] = vl;
          ddelta[dd.dcod + ((size_t)k * B + bb) * 4 + cc] = vc;
        }
      }
    }
    if (gl.on && gl.loss_part && wave == 3 && k < 2) {  // the reported MSE sums of this tile (train.py:484-488, 512-516)
      const bool lv = lane < 16 && a0 + lane < B;
      const float e = (k == 0 ? rl[0] : rl[1]) - (k == 0 ? htg[0] : htg[1]);
      float sl = lv ? e * e : 0.f, sc = 0.f;
      if (k == 0 && lv) {
        const float c0 = rc[0] - rz[0], c1 = rc[1] - rz[1];
        sc = c0 * c0 + c1 * c1;
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        sl += __shfl_xor(sl, o);
        sc += __shfl_xor(sc, o);
      }
      if (lane == 0) {
        gl.loss_part[(size_t)blockIdx.x * 3 + (k == 0 ? 0 : 2)] = sl;
        if (k == 0) gl.loss_part[(size_t)blockIdx.x * 3 + 1] = sc;
      }
    }
    sw_barrier();
    // dc1 = (cl1^T dlabel) * lrelu'(c1)  (waves 0,1) ; dl1 = (la1^T dcode) * lrelu'(l1)  (waves 2,3)
    {
      int m0 = 16 * (wave & 1);
      bool cls = wave < 2;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = tile_mm_rt(smem + (cls ? L.cl1T : L.la1T) + (m0 + ln) * LD16 + 4 * lg,
                       smem + (cls ? L.dlab : L.dcod) + ln * LD16 + 4 * lg, 1, acc);
      const f32x4 a = k == 0 ? pc1[0] : pc1[1];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu_grad(a[r], acc[r]);
      st4(smem + (cls ? L.dc1 : L.dl1) + ln * LD32 + m0 + 4 * lg, acc);
      if (want_w && live) st4(ddelta + (cls ? dd.dc1 : dd.dl1) + ((size_t)k * B + b) * 32 + m0 + 4 * lg, acc);
    }
    sw_barrier();
    // dboth = cl0^T dc1 + la0^T dl1   (wave w: rows 16w..)
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = tile_mm_rt(smem + L.cl0T + (u0 + ln) * LD32 + 4 * lg, smem + L.dc1 + ln * LD32 + 4 * lg, 2, acc);
      acc = tile_mm_rt(smem + L.la0T + (u0 + ln) * LD32 + 4 * lg, smem + L.dl1 + ln * LD32 + 4 * lg, 2, acc);
      st4(smem + L.dboth + ln * LD64 + u0 + 4 * lg, acc);
      if (wave < 2) {  // observation-code half: summed over branches
        float* p = smem + L.docode + ln * LD32 + u0 + 4 * lg;
        st4(p, ld4(p) + acc);
      } else if (want_w && live) {
        st4(ddelta + dd.dpcode + ((size_t)k * B + b) * 32 + (u0 - 32) + 4 * lg, acc);
      }
    }
    sw_barrier();
    // dq1 = (pe1^T dpcode) * lrelu'(q1)   (waves 0,1)
    if (wave < 2) {
      int m0 = 16 * wave;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = tile_mm_rt(smem + L.pe1T + (m0 + ln) * LD32 + 4 * lg, smem + L.dboth + ln * LD64 + 32 + 4 * lg, 2, acc);
      const f32x4 a = k == 0 ? pq1[0] : pq1[1];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu_grad(a[r], acc[r]);
      st4(smem + L.dq1 + ln * LD32 + m0 + 4 * lg, acc);
      if (want_w && live) st4(ddelta + dd.dq1 + ((size_t)k * B + b) * 32 + m0 + 4 * lg, acc);
    }
    sw_barrier();
    // dpred = pe0^T dq1   (4Tp rows)
    if (dpred) {
      for (int mt = wave; mt * 16 < K4; mt += 4) {
        int m0 = 16 * mt;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_mm_rt(smem + L.pe0T + (m0 + ln) * LD32 + 4 * lg, smem + L.dq1 + ln * LD32 + 4 * lg, 2, acc);
        if (live && m0 + 4 * lg < K4) st4(dpred + (size_t)b * K4 + m0 + 4 * lg, acc);
      }
    }
  }
  if (!want_w) return;
  sw_barrier();
  SW_STAMP(9);
  // ---- observation path: of1, of0, LSTM BPTT -------------------------------------------------
  if (live && wave < 2) st4(ddelta + dd.docode + (size_t)b * 32 + u0 + 4 * lg, ld4(smem + L.docode + ln * LD32 + u0 + 4 * lg));
  if (wave < 2) {
    int m0 = 16 * wave;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_mm_rt(smem + L.of1T + (m0 + ln) * LD32 + 4 * lg, smem + L.docode + ln * LD32 + 4 * lg, 2, acc);
    const f32x4 a = po1;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu_grad(a[r], acc[r]);
    st4(smem + L.do1 + ln * LD32 + m0 + 4 * lg, acc);
    if (live) st4(ddelta + dd.do1 + (size_t)b * 32 + m0 + 4 * lg, acc);
  }
  sw_barrier();
  f32x4 dh = {0.f, 0.f, 0.f, 0.f}, dc = {0.f, 0.f, 0.f, 0.f};
  dh = tile_mm_rt(smem + L.of0T + (u0 + ln) * LD32 + 4 * lg, smem + L.do1 + ln * LD32 + 4 * lg, 2, dh);
  LstmWT WT;
  if (dimg) {
#pragma unroll
    for (int j = 0; j < 16; ++j) WT.whhT[j] = ld4(dimg + swdimg::OP_WHHT + (((size_t)wave * 16 + j) * 64 + lane) * 4);
  } else {
    lstm_load_wT(WT, d_w + O.whh, u0, ln, lg);
  }
  // Saved rows are fetched one step ahead.  Every memory operation of the loop body is UNCONDITIONAL (the two
  // boundary steps are peeled; padding lanes of the last tile store to a trash row): with conditional loads or
  // stores the compiler cannot count what is in flight and waits for everything (s_waitcnt vmcnt(0)) right after
  // issuing the prefetch - that exposed one HBM round trip in every BPTT step.
  const float* act_b = dsave + ds.act + (size_t)b * 384 + u0 + 4 * lg;
  const size_t tstep = (size_t)B * 384;
  auto load_row = [&](int t, f32x4 g[4], f32x4& ct_, f32x4& cp_, auto has_prev) {
    const float* row = act_b + (size_t)t * tstep;
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = ld4(row + q * 64);
    ct_ = ld4(row + 256);
    if constexpr (decltype(has_prev)::value) cp_ = ld4(row - tstep + 256);
    else cp_ = f32x4{0.f, 0.f, 0.f, 0.f};   // c_{-1} = 0
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  f32x4 gate[4], ct, cprev;
  if (To > 1) load_row(To - 1, gate, ct, cprev, T_{});
  else load_row(0, gate, ct, cprev, F_{});
  SW_STAMP(10);
  // one BPTT step; pf: rows of step t-1 are prefetched (pp: they have a predecessor row), nx: dh_{t-1} is needed
  auto step = [&](int t, auto pf, auto pp, auto nx) {
    f32x4 dgate[4];
    lstm_cell_bwd(gate, ct, cprev, dh, dc, dgate);
    if constexpr (decltype(pf)::value) {      // rolling prefetch: the rows of step t - 1 into the registers just consumed
      load_row(t - 1, gate, ct, cprev, pp);
      asm volatile("" ::: "memory");
    }
    float* dgl = &dgbuf[(t & 1) * 16 * SW_GLD + ln * SW_GLD + u0 + 4 * lg];
#pragma unroll
    for (int g = 0; g < 4; ++g) st4(dgl + g * 64, dgate[g]);
    SW_STAMP(12);
    sw_barrier();
    lstm_store_dgates_tile(&dgbuf[(t & 1) * 16 * SW_GLD], ddelta + dd.dgates + ((size_t)t * B + a0) * 256, ddelta + dd.trash,
                           a0, B, wave, lane);
    SW_STAMP(13);
    if constexpr (decltype(nx)::value) dh = lstm_dh_prev(WT, &dgbuf[(t & 1) * 16 * SW_GLD + ln * SW_GLD + 4 * lg]);
    if constexpr (decltype(pf)::value)
      asm volatile("" : "+v"(gate[0]), "+v"(gate[1]), "+v"(gate[2]), "+v"(gate[3]), "+v"(ct), "+v"(cprev));
    SW_STAMP(14);
  };
  // every load issued so far (the saved rows of the first step, W_hh^T) is waited for HERE: behind the heads' conditional
  // stores the compiler cannot count what is pending on the way into the loop, and a loop header with an unknown state
  // gets s_waitcnt vmcnt(0) - which every BPTT step then pays as the round trip of the dgates rows it has just stored
  asm volatile("" : "+v"(gate[0]), "+v"(gate[1]), "+v"(gate[2]), "+v"(gate[3]), "+v"(ct), "+v"(cprev));
#pragma unroll
  for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(WT.whhT[j]));
  for (int t = To - 1; t >= 2; --t) step(t, T_{}, T_{}, T_{});
  if (To > 1) step(1, T_{}, F_{}, T_{});
  step(0, F_{}, F_{}, F_{});
  SW_STAMP(11);
}

// ---------------------------------------------------------------------------------------------
// One discriminator UPDATE pass in ONE launch (train.py:476-495): forward of D on the fake and the real branch, the LSGAN /
// InfoGAN loss gradients (per-agent local: means over the batch, scale known up front) and the whole backward down to the
// delta rows the weight-gradient GEMM contracts over.  One workgroup per 16-agent tile; the two-launch form costs a launch,
// a second prologue and the round trip of the gates / every head activation through the save buffer.  The two branches share the observation encoding; their heads run
// SIDE BY SIDE on the two wave pairs (waves 0, 1: fake; 2, 3: real) - a head layer is 2 row tiles, so the pair that used
// to idle now carries the other branch - forward and backward.  Rows for the weight gradients (saved activations, deltas)
// are written exactly where sw_disc_fwd / sw_disc_bwd put them.  Needs the registered weight images (sw_disc_images).
// ---------------------------------------------------------------------------------------------
namespace {
struct UpdLds {
  int hbuf, hw, bias, hwT, x, q1, both, c1, l1, o1, total;          // forward carve
  int dlab, dcod, dc1, dl1, dboth, docode, do1, dend;                // deltas: alias the forward head weights (dead by then)
  int ldp;
  HeadLds F;      // offsets of the forward head weights (relative to 0: add hw - F.of0)
  HeadLdsB T;     // offsets of the transposed block (relative to T.of0T: add hwT)
};
__host__ __device__ inline UpdLds upd_lds(int Tp) {
  UpdLds U;
  U.F = head_lds(Tp, 0);
  U.T = head_lds_b(Tp, 0);
  U.ldp = U.F.ldp;
  int o = 0;
  U.hbuf = o; o += 2 * 16 * SW_HLD;
  U.hw = o; o += U.F.bias - U.F.of0;          // of0 | of1 | pe0 | pe1 | cl0 | la0 | cl1 | la1 (row-major, padded)
  U.bias = o; o += 8 * 32;
  U.hwT = o; o += U.T.dlab - U.T.of0T;        // the transposed block of the images (HeadLdsB layout)
  U.x = o; o += 2 * 16 * U.ldp;
  U.q1 = o; o += 2 * 16 * LD32;
  U.both = o; o += 2 * 16 * LD64;
  U.c1 = o; o += 2 * 16 * LD32;
  U.l1 = o; o += 2 * 16 * LD32;
  U.o1 = o; o += 16 * LD32;
  U.total = o;
  int d = U.hw;
  U.dlab = d; d += 2 * 16 * LD16;
  U.dcod = d; d += 2 * 16 * LD16;
  U.dc1 = d; d += 2 * 16 * LD32;
  U.dl1 = d; d += 2 * 16 * LD32;
  U.dboth = d; d += 2 * 16 * LD64;
  U.docode = d; d += 16 * LD32;
  U.do1 = d; d += 16 * LD32;
  U.dend = d;
  return U;
}
}  // namespace

__global__ __launch_bounds__(SW_THREADS) void disc_update_kernel(
    const float* __restrict__ obsv, int To, const float* __restrict__ pred_a, const float* __restrict__ pred_b,
    const float* __restrict__ d_w, int B, int Tp, float* __restrict__ label_a, float* __restrict__ label_b,
    float* __restrict__ code_a, float* __restrict__ code_b, float* __restrict__ dsave, int obs_pre, float* __restrict__ w_snap,
    DiscLoss gl, float* __restrict__ ddelta, const float* __restrict__ dimg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const UpdLds U = upd_lds(Tp);
  const HeadLds& F = U.F;
  const HeadLdsB& T = U.T;
  const swp::Disc O = swp::disc(Tp);
  const DSave ds = dsave_layout(B, To, Tp, 2);
  const DDelta dd = ddelta_layout(B, To, Tp, 2);
  float* hbuf = smem + U.hbuf;
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16, br = wave >> 1, wp = wave & 1;       // br: the branch this wave's pair owns, wp: wave in pair
  const int a0 = blockIdx.x * SW_TILE;
  const int b = min(a0 + ln, B - 1);
  const bool live = (a0 + ln) < B;
  const int K4 = 4 * Tp, ldp = U.ldp;
  auto hwp = [&](int off) { return smem + U.hw + (off - F.of0); };      // forward head matrix at HeadLds offset `off`
  auto hwT = [&](int off) { return smem + U.hwT + (off - T.of0T); };    // transposed head matrix at HeadLdsB offset `off`

  // ---- prologue: every global load first --------------------------------------------------------------------------
  float xpre[2][4];       // prediction rows of both branches (unconditional loads from clamped addresses; Tp <= 12 here)
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const float* pk = kk == 0 ? pred_a : pred_b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = min((int)threadIdx.x + SW_THREADS * e, 16 * ldp - 1);
      const int a = i / ldp, cc = i - a * ldp;
      xpre[kk][e] = pk[(size_t)min(a0 + a, B - 1) * K4 + min(cc, K4 - 1)];
    }
  }
  const float tg0 = gl.targets[gl.t0], tg1 = gl.targets[gl.t1];
  const float z0 = gl.z[(size_t)b * SW_Z], z1 = gl.z[(size_t)b * SW_Z + 1];
  LstmW W;
  if (!obs_pre) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) W.whh[g][j] = ld4(dimg + swdimg::OP_WHH + ((((size_t)4 * g + wave) * 4 + j) * 64 + lane) * 4);
      W.wx[g] = d_w[O.wih + (g * 64 + u0 + ln) * 4 + lg];
      W.bias[g] = ld4(d_w + O.bih + g * 64 + u0 + 4 * lg) + ld4(d_w + O.bhh + g * 64 + u0 + 4 * lg);
    }
  }
  if (w_snap)   // deepcopy(D) of train.py:499: the weights this pass runs with
    for (int i = blockIdx.x * SW_THREADS + threadIdx.x; i < O.n; i += gridDim.x * SW_THREADS) w_snap[i] = d_w[i];
  {
    f32x4 s_of0[3], s_of1[2], s_pe0[2], s_pe1[2], s_cl0[3], s_la0[3], s_cl1[1], s_la1[1];
    stage_w_load<3>(s_of0, LD64, 32, d_w + O.of0w, 64, 32, 64);
    stage_w_load<2>(s_of1, LD32, 32, d_w + O.of1w, 32, 32, 32);
    stage_w_load<2>(s_pe0, ldp, 32, d_w + O.pe0w, K4, 32, K4);
    stage_w_load<2>(s_pe1, LD32, 32, d_w + O.pe1w, 32, 32, 32);
    stage_w_load<3>(s_cl0, LD64, 32, d_w + O.cl0w, 64, 32, 64);
    stage_w_load<3>(s_la0, LD64, 32, d_w + O.la0w, 64, 32, 64);
    stage_w_load<1>(s_cl1, LD32, 16, d_w + O.cl1w, 32, 1, 32);
    stage_w_load<1>(s_la1, LD32, 16, d_w + O.la1w, 32, 2, 32);
    stage_w_store<3>(s_of0, hwp(F.of0), LD64, 32);
    stage_w_store<2>(s_of1, hwp(F.of1), LD32, 32);
    stage_w_store<2>(s_pe0, hwp(F.pe0), ldp, 32);
    stage_w_store<2>(s_pe1, hwp(F.pe1), LD32, 32);
    stage_w_store<3>(s_cl0, hwp(F.cl0), LD64, 32);
    stage_w_store<3>(s_la0, hwp(F.la0), LD64, 32);
    stage_w_store<1>(s_cl1, hwp(F.cl1), LD32, 16);
    stage_w_store<1>(s_la1, hwp(F.la1), LD32, 16);
  }
  {
    const int i = threadIdx.x;  // 256 = 8 x 32 bias slots
    const int q = i >> 5, k = i & 31;
    const int boff = q == 0 ? O.of0b : q == 1 ? O.of1b : q == 2 ? O.pe0b : q == 3 ? O.pe1b : q == 4 ? O.cl0b
                     : q == 5 ? O.la0b : q == 6 ? O.cl1b : O.la1b;
    const int lim = q < 6 ? 32 : (q == 6 ? 1 : 2);
    const float v = d_w[boff + min(k, lim - 1)];
    smem[U.bias + i] = k < lim ? v : 0.f;
  }
  {   // the transposed block of the images: one contiguous float4 copy (zero padding included)
    constexpr int HB = 12;
    const int n4 = (T.dlab - T.of0T) >> 2;
    f32x4 hbv[HB];
#pragma unroll
    for (int e = 0; e < HB; ++e) hbv[e] = ld4(dimg + swdimg::HEADT + 4 * (size_t)min((int)threadIdx.x + SW_THREADS * e, n4 - 1));
#pragma unroll
    for (int e = 0; e < HB; ++e) {
      const int f = threadIdx.x + SW_THREADS * e;
      if (f < n4) st4(smem + U.hwT + 4 * f, hbv[e]);
    }
  }
  // prediction rows into LDS (zero padded) and into the save buffer (rows of the pe0 weight gradient)
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = threadIdx.x + SW_THREADS * e;
      if (i < 16 * ldp) {
        const int a = i / ldp, cc = i - a * ldp;
        const float v = cc < K4 ? xpre[kk][e] : 0.f;
        smem[U.x + kk * 16 * ldp + i] = v;
        if (cc < K4 && a0 + a < B) dsave[ds.px + ((size_t)kk * B + a0 + a) * K4 + cc] = v;
      }
    }
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};
  if (!obs_pre) {
    st4(&hbuf[ln * SW_HLD + u0 + 4 * lg], h);
  } else {   // h_T of the tile from the rows the decode launch left
    st4(&hbuf[(To & 1) * 16 * SW_HLD + ln * SW_HLD + u0 + 4 * lg],
        ld4(dsave + ds.act + ((size_t)(To - 1) * B + b) * 384 + 320 + u0 + 4 * lg));
  }
  sw_barrier();
  // The usual horizon (8 observed steps, train.py:44): the gates and cell states of the eight steps STAY IN REGISTERS
  // (160 of the 512 a wave owns at one wave per SIMD) for the BPTT below - only h (the operand of the LSTM's weight
  // gradient) and the step's input go to memory: a sixth of the saved bytes, no row loads in the BPTT.
  const bool reg8 = !obs_pre && To == 8;
  f32x4 sg[8][4], sc[8];
  if (reg8) {
    float xa, xq;
    obs_x4_load(obsv, b, 0, 8, lg, xa, xq);
    asm volatile("" : "+v"(xa), "+v"(xq));
    float* hrow_g = dsave + ds.act + (size_t)b * 384 + 320 + u0 + 4 * lg;
    float* xrow = dsave + ds.x4s + (size_t)b * 4 + lg;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float xb = xa - (lg >= 2 ? xq : 0.f);
      obs_x4_load(obsv, b, t + 1 < 8 ? t + 1 : 7, 8, lg, xa, xq);
      lstm_cell(W, xb, &hbuf[(t & 1) * 16 * SW_HLD + ln * SW_HLD + 4 * lg], sg[t], c, h);
      sc[t] = c;
      st4(&hbuf[((t + 1) & 1) * 16 * SW_HLD + ln * SW_HLD + u0 + 4 * lg], h);
      st4g(hrow_g, h);
      *xrow = xb;
      hrow_g += (size_t)B * 384;
      xrow += (size_t)B * 4;
      sw_barrier();
      asm volatile("" : "+v"(xa), "+v"(xq));
    }
  } else if (!obs_pre) {
    lstm_obs_loop<0, true>(W, hbuf, obsv, To, B, b, c, h, dsave + ds.act, dsave + ds.x4s);
  }
  const float* hlast = &hbuf[(To & 1) * 16 * SW_HLD];

  // ---- observation fc (waves 0, 1): o1 = lrelu(of0 h + b), obsv_code = of1 o1 + b -> both[0 | 1][:, 0:32] --------------
  f32x4 o1reg = {0.f, 0.f, 0.f, 0.f};
  if (wave < 2) {
    const int m0 = 16 * wave;
    f32x4 acc = ld4(smem + U.bias + 0 * 32 + m0 + 4 * lg);
    acc = tile_mm_rt(hwp(F.of0) + (m0 + ln) * LD64 + 4 * lg, hlast + ln * SW_HLD + 4 * lg, 4, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu(acc[r]);
    o1reg = acc;
    st4(smem + U.o1 + ln * LD32 + m0 + 4 * lg, acc);
    if (live) st4(dsave + ds.o1 + (size_t)b * 32 + m0 + 4 * lg, acc);
  }
  sw_barrier();
  if (wave < 2) {
    const int m0 = 16 * wave;
    f32x4 acc = ld4(smem + U.bias + 1 * 32 + m0 + 4 * lg);
    acc = tile_mm_rt(hwp(F.of1) + (m0 + ln) * LD32 + 4 * lg, smem + U.o1 + ln * LD32 + 4 * lg, 2, acc);
    st4(smem + U.both + ln * LD64 + m0 + 4 * lg, acc);
    st4(smem + U.both + 16 * LD64 + ln * LD64 + m0 + 4 * lg, acc);
  }
  // ---- prediction heads, the two branches side by side: wave pair br, wave wp of the pair ------------------------------
  float* xb_ = smem + U.x + br * 16 * ldp;
  float* q1_ = smem + U.q1 + br * 16 * LD32;
  float* both_ = smem + U.both + br * 16 * LD64;
  float* c1_ = smem + U.c1 + br * 16 * LD32;
  float* l1_ = smem + U.l1 + br * 16 * LD32;
  const size_t kb = (size_t)br * B + b;
  f32x4 q1reg, c1reg, l1reg;
  {   // q1 = lrelu(pe0 x + b): row tile wp
    const int m0 = 16 * wp;
    f32x4 acc = ld4(smem + U.bias + 2 * 32 + m0 + 4 * lg);
    acc = tile_mm_rt(hwp(F.pe0) + (m0 + ln) * ldp + 4 * lg, xb_ + ln * ldp + 4 * lg, (ldp - 4) / 16, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu(acc[r]);
    q1reg = acc;
    st4(q1_ + ln * LD32 + m0 + 4 * lg, acc);
    if (live) st4(dsave + ds.q1 + kb * 32 + m0 + 4 * lg, acc);
  }
  sw_barrier();
  {   // pred_code = pe1 q1 + b -> both[br][:, 32:64]
    // (of0 / of1 are dead behind the barrier above: the dlab | dcod delta tiles that alias them are zeroed here - only
    //  their columns 0 / 0..1 are written later)
    stage_zero(smem + U.dlab, U.dc1 - U.dlab);
    const int m0 = 16 * wp;
    f32x4 acc = ld4(smem + U.bias + 3 * 32 + m0 + 4 * lg);
    acc = tile_mm_rt(hwp(F.pe1) + (m0 + ln) * LD32 + 4 * lg, q1_ + ln * LD32 + 4 * lg, 2, acc);
    st4(both_ + ln * LD64 + 32 + m0 + 4 * lg, acc);
  }
  sw_barrier();
  {   // both codes saved (wave wp: columns 32 wp ..), then c1 = lrelu(cl0 both + b), l1 = lrelu(la0 both + b): row tile wp each
    if (live) {
      st4(dsave + ds.both + kb * 64 + 32 * wp + 4 * lg, ld4(both_ + ln * LD64 + 32 * wp + 4 * lg));
      st4(dsave + ds.both + kb * 64 + 32 * wp + 16 + 4 * lg, ld4(both_ + ln * LD64 + 32 * wp + 16 + 4 * lg));
    }
    const int m0 = 16 * wp;
    f32x4 ac = ld4(smem + U.bias + 4 * 32 + m0 + 4 * lg), al = ld4(smem + U.bias + 5 * 32 + m0 + 4 * lg);
    ac = tile_mm_rt(hwp(F.cl0) + (m0 + ln) * LD64 + 4 * lg, both_ + ln * LD64 + 4 * lg, 4, ac);
    al = tile_mm_rt(hwp(F.la0) + (m0 + ln) * LD64 + 4 * lg, both_ + ln * LD64 + 4 * lg, 4, al);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ac[r] = sw_lrelu(ac[r]);
      al[r] = sw_lrelu(al[r]);
    }
    c1reg = ac;
    l1reg = al;
    st4(c1_ + ln * LD32 + m0 + 4 * lg, ac);
    st4(l1_ + ln * LD32 + m0 + 4 * lg, al);
    if (live) {
      st4(dsave + ds.c1 + kb * 32 + m0 + 4 * lg, ac);
      st4(dsave + ds.l1 + kb * 32 + m0 + 4 * lg, al);
    }
  }
  sw_barrier();     // every reader of the forward head matrices up to la0 is done: the delta tiles overwrite them from here on
  // ---- label = cl1 c1 + b (wave wp = 0), code_hat = la1 l1 + b (wp = 1); loss gradients (train.py:484-494) + reported sums
  {
    const bool cls = wp == 0;
    f32x4 acc = ld4(smem + U.bias + (cls ? 6 : 7) * 32 + 4 * lg);
    acc = tile_mm_rt(hwp(cls ? F.cl1 : F.la1) + ln * LD32 + 4 * lg, (cls ? c1_ : l1_) + ln * LD32 + 4 * lg, 2, acc);
    float* label = br == 0 ? label_a : label_b;
    float* code = br == 0 ? code_a : code_b;
    float sl = 0.f;
    if (lg == 0) {
      if (cls) {
        if (live) label[b] = acc[0];
        const float e = acc[0] - (br == 0 ? tg0 : tg1);
        const float g = 2.0f * e * gl.g_label;
        smem[U.dlab + br * 16 * LD16 + ln * LD16] = g;
        if (live) st4(ddelta + dd.dlab + kb * 4, f32x4{g, 0.f, 0.f, 0.f});
        sl = live ? e * e : 0.f;
      } else {
        if (live) { code[(size_t)b * 2] = acc[0]; code[(size_t)b * 2 + 1] = acc[1]; }
        const float e0 = acc[0] - z0, e1 = acc[1] - z1;
        const float g0 = br == 0 ? 2.0f * e0 * gl.g_code : 0.f, g1 = br == 0 ? 2.0f * e1 * gl.g_code : 0.f;   // info term: fake branch only
        smem[U.dcod + br * 16 * LD16 + ln * LD16] = g0;
        smem[U.dcod + br * 16 * LD16 + ln * LD16 + 1] = g1;
        if (live) st4(ddelta + dd.dcod + kb * 4, f32x4{g0, g1, 0.f, 0.f});
        sl = live ? e0 * e0 + e1 * e1 : 0.f;
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sl += __shfl_xor(sl, o);
    if (gl.loss_part && lane == 0) {
      if (cls) gl.loss_part[(size_t)blockIdx.x * 3 + (br == 0 ? 0 : 2)] = sl;
      else if (br == 0) gl.loss_part[(size_t)blockIdx.x * 3 + 1] = sl;
    }
  }
  sw_barrier();
  // ---- backward of the heads, branches side by side ------------------------------------------------------------------
  float* dlab_ = smem + U.dlab + br * 16 * LD16;
  float* dcod_ = smem + U.dcod + br * 16 * LD16;
  float* dc1_ = smem + U.dc1 + br * 16 * LD32;
  float* dl1_ = smem + U.dl1 + br * 16 * LD32;
  float* dboth_ = smem + U.dboth + br * 16 * LD64;
  {   // dc1 = (cl1^T dlabel) * lrelu'(c1), dl1 = (la1^T dcode) * lrelu'(l1): row tile wp each
    const int m0 = 16 * wp;
    f32x4 ac = {0.f, 0.f, 0.f, 0.f}, al = ac;
    ac = tile_mm_rt(hwT(T.cl1T) + (m0 + ln) * LD16 + 4 * lg, dlab_ + ln * LD16 + 4 * lg, 1, ac);
    al = tile_mm_rt(hwT(T.la1T) + (m0 + ln) * LD16 + 4 * lg, dcod_ + ln * LD16 + 4 * lg, 1, al);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ac[r] = sw_lrelu_grad(c1reg[r], ac[r]);
      al[r] = sw_lrelu_grad(l1reg[r], al[r]);
    }
    st4(dc1_ + ln * LD32 + m0 + 4 * lg, ac);
    st4(dl1_ + ln * LD32 + m0 + 4 * lg, al);
    if (live) {
      st4(ddelta + dd.dc1 + kb * 32 + m0 + 4 * lg, ac);
      st4(ddelta + dd.dl1 + kb * 32 + m0 + 4 * lg, al);
    }
  }
  sw_barrier();
  {   // dboth = cl0^T dc1 + la0^T dl1 (64 rows): row tiles 2 wp, 2 wp + 1
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int m0 = 16 * (2 * wp + q);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = tile_mm_rt(hwT(T.cl0T) + (m0 + ln) * LD32 + 4 * lg, dc1_ + ln * LD32 + 4 * lg, 2, acc);
      acc = tile_mm_rt(hwT(T.la0T) + (m0 + ln) * LD32 + 4 * lg, dl1_ + ln * LD32 + 4 * lg, 2, acc);
      st4(dboth_ + ln * LD64 + m0 + 4 * lg, acc);
      if (wp == 1 && live) st4(ddelta + dd.dpcode + kb * 32 + (m0 - 32) + 4 * lg, acc);   // the prediction-code half
    }
  }
  sw_barrier();
  {   // dq1 = (pe1^T dpcode) * lrelu'(q1): row tile wp (only the weight gradient needs it: D updates want no d/dpred)
    const int m0 = 16 * wp;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_mm_rt(hwT(T.pe1T) + (m0 + ln) * LD32 + 4 * lg, dboth_ + ln * LD64 + 32 + 4 * lg, 2, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu_grad(q1reg[r], acc[r]);
    if (live) st4(ddelta + dd.dq1 + kb * 32 + m0 + 4 * lg, acc);
  }
  if (wave < 2) {   // observation-code half summed over the branches (fixed order), then do1 = (of1^T docode) * lrelu'(o1)
    const int m0 = 16 * wave;
    const f32x4 v = ld4(smem + U.dboth + ln * LD64 + m0 + 4 * lg) + ld4(smem + U.dboth + 16 * LD64 + ln * LD64 + m0 + 4 * lg);
    st4(smem + U.docode + ln * LD32 + m0 + 4 * lg, v);
    if (live) st4(ddelta + dd.docode + (size_t)b * 32 + m0 + 4 * lg, v);
  }
  sw_barrier();
  if (wave < 2) {
    const int m0 = 16 * wave;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_mm_rt(hwT(T.of1T) + (m0 + ln) * LD32 + 4 * lg, smem + U.docode + ln * LD32 + 4 * lg, 2, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu_grad(o1reg[r], acc[r]);
    st4(smem + U.do1 + ln * LD32 + m0 + 4 * lg, acc);
    if (live) st4(ddelta + dd.do1 + (size_t)b * 32 + m0 + 4 * lg, acc);
  }
  sw_barrier();
  // ---- observation LSTM backward (as disc_bwd_kernel): dh_T = of0^T do1, then BPTT over the saved rows ------------------
  f32x4 dh = {0.f, 0.f, 0.f, 0.f}, dc = {0.f, 0.f, 0.f, 0.f};
  dh = tile_mm_rt(hwT(T.of0T) + (u0 + ln) * LD32 + 4 * lg, smem + U.do1 + ln * LD32 + 4 * lg, 2, dh);
  LstmWT WT;
#pragma unroll
  for (int j = 0; j < 16; ++j) WT.whhT[j] = ld4(dimg + swdimg::OP_WHHT + (((size_t)wave * 16 + j) * 64 + lane) * 4);
  float* dgbuf = smem + U.hwT + (T.pe0T - T.of0T);   // [2][16][SW_GLD] over the prediction heads' transposed images (dead now)
  const float* act_b = dsave + ds.act + (size_t)b * 384 + u0 + 4 * lg;
  const size_t tstep = (size_t)B * 384;
  using T_ = std::true_type;
  using F_ = std::false_type;
  auto load_row = [&](int t, f32x4 g[4], f32x4& ct_, f32x4& cp_, auto has_prev) {
    const float* row = act_b + (size_t)t * tstep;
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = ld4(row + q * 64);
    ct_ = ld4(row + 256);
    if constexpr (decltype(has_prev)::value) cp_ = ld4(row - tstep + 256);
    else cp_ = f32x4{0.f, 0.f, 0.f, 0.f};   // c_{-1} = 0
  };
  if (reg8) {        // the eight steps' gates / cell states are in registers: no row traffic at all
    sw_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(WT.whhT[j]));
#pragma unroll
    for (int t = 7; t >= 0; --t) {
      f32x4 dgate[4];
      lstm_cell_bwd(sg[t], sc[t], t > 0 ? sc[t > 0 ? t - 1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f}, dh, dc, dgate);
      float* dgl = &dgbuf[(t & 1) * 16 * SW_GLD + ln * SW_GLD + u0 + 4 * lg];
#pragma unroll
      for (int g = 0; g < 4; ++g) st4(dgl + g * 64, dgate[g]);
      sw_barrier();
      lstm_store_dgates_tile(&dgbuf[(t & 1) * 16 * SW_GLD], ddelta + dd.dgates + ((size_t)t * B + a0) * 256, ddelta + dd.trash,
                             a0, B, wave, lane);
      if (t > 0) dh = lstm_dh_prev(WT, &dgbuf[(t & 1) * 16 * SW_GLD + ln * SW_GLD + 4 * lg]);
    }
    return;
  }
  f32x4 gate[4], ct, cprev;
  if (To > 1) load_row(To - 1, gate, ct, cprev, T_{});
  else load_row(0, gate, ct, cprev, F_{});
  sw_barrier();      // every wave has read of0T / do1: (the dgates tiles start behind of0T / of1T; kept for symmetry with disc_bwd)
  asm volatile("" : "+v"(gate[0]), "+v"(gate[1]), "+v"(gate[2]), "+v"(gate[3]), "+v"(ct), "+v"(cprev));
#pragma unroll
  for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(WT.whhT[j]));
  auto step = [&](int t, auto pf, auto pp, auto nx) {
    f32x4 dgate[4];
    lstm_cell_bwd(gate, ct, cprev, dh, dc, dgate);
    if constexpr (decltype(pf)::value) {      // rolling prefetch: the rows of step t - 1 into the registers just consumed
      load_row(t - 1, gate, ct, cprev, pp);
      asm volatile("" ::: "memory");
    }
    float* dgl = &dgbuf[(t & 1) * 16 * SW_GLD + ln * SW_GLD + u0 + 4 * lg];
#pragma unroll
    for (int g = 0; g < 4; ++g) st4(dgl + g * 64, dgate[g]);
    sw_barrier();
    lstm_store_dgates_tile(&dgbuf[(t & 1) * 16 * SW_GLD], ddelta + dd.dgates + ((size_t)t * B + a0) * 256, ddelta + dd.trash,
                           a0, B, wave, lane);
    if constexpr (decltype(nx)::value) dh = lstm_dh_prev(WT, &dgbuf[(t & 1) * 16 * SW_GLD + ln * SW_GLD + 4 * lg]);
    if constexpr (decltype(pf)::value)
      asm volatile("" : "+v"(gate[0]), "+v"(gate[1]), "+v"(gate[2]), "+v"(gate[3]), "+v"(ct), "+v"(cprev));
  };
  for (int t = To - 1; t >= 2; --t) step(t, T_{}, T_{}, T_{});
  if (To > 1) step(1, T_{}, F_{}, T_{});
  step(0, F_{}, F_{}, F_{});
}

static int set_lds(const void* fn, int bytes) {
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    sw_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
    return SW_EHIP;
  }
  return SW_OK;
}

static int g_disc_fwd_lds = 0;   // largest dynamic-LDS size disc_fwd_kernel has been enabled for (plain + fused launches)
static int disc_fwd_lds(int bytes) {
  if (g_disc_fwd_lds < bytes) {
    if (int rc = set_lds((const void*)disc_fwd_kernel, bytes)) return rc;
    g_disc_fwd_lds = bytes;
  }
  return SW_OK;
}

size_t sw_dsave_floats(int B, int To, int Tp, int nb) { return dsave_layout(B, To, Tp, nb).total; }
size_t sw_ddelta_floats(int B, int To, int Tp, int nb) { return ddelta_layout(B, To, Tp, nb).total; }

extern "C" int sw_disc_fwd(const float* obsv, int To, int x_mode, const float* const* pred4, int nb,
                           const float* d_w, int B, int Tp, float* const* label, float* const* code, float* dsave,
                           int save_lstm, float* w_snapshot, void* stream) {
  if (!obsv || !pred4 || !d_w || !label || !code || nb < 1 || nb > SW_DISC_MAXB || B < 0 || To < 1 || Tp < 1 ||
      (x_mode != 0 && x_mode != 1) || (x_mode == 0 && To < 2) || save_lstm < 0 || save_lstm > 2 || (save_lstm == 2 && !dsave))
    return SW_EARG;
  for (int k = 0; k < nb; ++k)
    if (!pred4[k] || !label[k] || !code[k]) return SW_EARG;
  if (Tp > 64) return SW_ESHAPE;
  if (B == 0) return SW_OK;
  int lds = head_lds(Tp, 2 * 16 * SW_HLD + 1280).total * 4;
  if (lds > 163840) return SW_ESHAPE;
  if (int rc = disc_fwd_lds(lds)) return rc;
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  const int split = (nb == 2 && 2 * tiles <= SW_SPLIT_MAX_WGS) ? 1 : 0;   // idle CUs: one workgroup per (tile, branch)
  SW_LAUNCH(disc_fwd_kernel, dim3(split ? 2 * tiles : tiles), dim3(SW_THREADS), lds, (hipStream_t)stream,
                     obsv, To, x_mode, pred4[0], nb > 1 ? pred4[1] : nullptr, nb, d_w, B, Tp, label[0],
                     nb > 1 ? label[1] : nullptr, code[0], nb > 1 ? code[1] : nullptr, dsave, save_lstm, split, w_snapshot, 0, DiscLoss{}, nullptr,
                     sw_disc_images_for(d_w, Tp).img);
  SW_CHECK_LAUNCH("disc_fwd_kernel");
  return SW_OK;
}

// Generator phase in ONE launch: forward of D on (obsv, pred_hat) and the backward of its prediction heads down to
// d(g_loss)/d(pred_hat) (train.py:510-523, 538): no saves, no second prologue, activations stay in LDS.
extern "C" int sw_disc_dpred(const float* obsv, int To, int x_mode, const float* pred4, const float* d_w, int B, int Tp,
                             const float* targets, int t_idx, const float* z, float g_label, float g_code,
                             float* dpred4, float* label, float* code, float* loss_part, void* stream) {
  if (!obsv || !pred4 || !d_w || !targets || !z || !dpred4 || B < 0 || To < 1 || Tp < 1 || t_idx < 0 ||
      (x_mode != 0 && x_mode != 1) || (x_mode == 0 && To < 2))
    return SW_EARG;
  if (Tp > 64) return SW_ESHAPE;
  if (B == 0) return SW_OK;
  const int lds = head_lds_b(Tp, head_lds(Tp, 2 * 16 * SW_HLD + 1280).total).total * 4;
  if (lds > 163840) return SW_ESHAPE;
  if (int rc = disc_fwd_lds(lds)) return rc;
  DiscLoss gl{targets, z, t_idx, t_idx, g_label, g_code, 1, loss_part};
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  SW_LAUNCH(disc_fwd_kernel, dim3(tiles), dim3(SW_THREADS), lds, (hipStream_t)stream, obsv, To, x_mode, pred4,
                     (const float*)nullptr, 1, d_w, B, Tp, label, (float*)nullptr, code, (float*)nullptr, (float*)nullptr, 0, 0,
                     (float*)nullptr, 1, gl, dpred4, sw_disc_images_for(d_w, Tp).img);
  SW_CHECK_LAUNCH("disc_fwd_kernel");
  return SW_OK;
}

// the weight-gradient problems of a discriminator pass over its saved / delta rows (dW = delta^T act per layer)
static int disc_wgrad_problems(WgBatch& wb, const float* dsave, float* ddelta, int nb, int B, int To, int Tp, float* d_d_w) {
  const swp::Disc O = swp::disc(Tp);
  const DSave ds = dsave_layout(B, To, Tp, nb);
  const DDelta dd = ddelta_layout(B, To, Tp, nb);
  const int R = nb * B, K4 = 4 * Tp;
  int rc_add = 0;
  // LSTM: dW_hh over rows t >= 1 against h_{t-1}; dW_ih / biases over all rows against x4
  rc_add |= wg_add_tail(wb, ddelta + dd.dgates, 256, dsave + ds.act + 320 - (ptrdiff_t)B * 384, 384, To * B, 256, 64,
                        d_d_w + O.whh, 64, dsave + ds.x4s, 4, 4, d_d_w + O.wih, 4, B /*h_{t-1}: rows t >= 1*/,
                        d_d_w + O.bih, d_d_w + O.bhh, 0);
  rc_add |= wg_add(wb, ddelta + dd.do1, 32, dsave + ds.act + (size_t)(To - 1) * B * 384 + 320, 384, B, 32, 64, d_d_w + O.of0w,
                   64, d_d_w + O.of0b, nullptr, 0);
  rc_add |= wg_add(wb, ddelta + dd.docode, 32, dsave + ds.o1, 32, B, 32, 32, d_d_w + O.of1w, 32, d_d_w + O.of1b, nullptr, 0);
  rc_add |= wg_add(wb, ddelta + dd.dq1, 32, dsave + ds.px, K4, R, 32, K4, d_d_w + O.pe0w, K4, d_d_w + O.pe0b, nullptr, 0);
  rc_add |= wg_add(wb, ddelta + dd.dpcode, 32, dsave + ds.q1, 32, R, 32, 32, d_d_w + O.pe1w, 32, d_d_w + O.pe1b, nullptr, 0);
  rc_add |= wg_add(wb, ddelta + dd.dc1, 32, dsave + ds.both, 64, R, 32, 64, d_d_w + O.cl0w, 64, d_d_w + O.cl0b, nullptr, 0);
  rc_add |= wg_add(wb, ddelta + dd.dlab, 4, dsave + ds.c1, 32, R, 1, 32, d_d_w + O.cl1w, 32, d_d_w + O.cl1b, nullptr, 0);
  rc_add |= wg_add(wb, ddelta + dd.dl1, 32, dsave + ds.both, 64, R, 32, 64, d_d_w + O.la0w, 64, d_d_w + O.la0b, nullptr, 0);
  rc_add |= wg_add(wb, ddelta + dd.dcod, 4, dsave + ds.l1, 32, R, 2, 32, d_d_w + O.la1w, 32, d_d_w + O.la1b, nullptr, 0);
  return rc_add ? SW_ESHAPE : SW_OK;
}

static int disc_bwd_impl(const float* d_w, const float* dsave, const float* const* dlabel, const float* const* dcode,
                         int nb, int B, int To, int Tp, float* ddelta, float* d_d_w, float* const* dpred4,
                         float* wgrad_ws, void* stream, DiscLoss gl, const WgAdam& adam = WgAdam()) {
  if (!d_w || !dsave || !dlabel || !dcode || nb < 1 || nb > SW_DISC_MAXB || B < 0 || To < 1 || Tp < 1) return SW_EARG;
  for (int k = 0; k < nb; ++k)
    if (!dlabel[k] || !dcode[k]) return SW_EARG;
  if (d_d_w && (!ddelta || !wgrad_ws)) return SW_EARG;
  if (Tp > 64) return SW_ESHAPE;
  if (B == 0) return SW_OK;
  int lds = head_lds_b(Tp, 0).total * 4;
  if (lds > 163840) return SW_ESHAPE;
  static int attr = 0;
  if (attr < lds) {
    if (int rc = set_lds((const void*)disc_bwd_kernel, lds)) return rc;
    attr = lds;
  }
  hipStream_t st = (hipStream_t)stream;
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  WgBatch wb;
  if (d_d_w)
    if (int rc = disc_wgrad_problems(wb, dsave, ddelta, nb, B, To, Tp, d_d_w)) return rc;
  SW_LAUNCH(disc_bwd_kernel, dim3(tiles), dim3(SW_THREADS), lds, st, d_w, dsave,
                     dlabel[0], nb > 1 ? dlabel[1] : nullptr, dcode[0], nb > 1 ? dcode[1] : nullptr, nb, B, To, Tp,
                     d_d_w ? 1 : 0, ddelta, dpred4 ? dpred4[0] : nullptr, (dpred4 && nb > 1) ? dpred4[1] : nullptr, gl,
                     sw_disc_images_for(d_w, Tp).img);
  SW_CHECK_LAUNCH("disc_bwd_kernel");
  if (!d_d_w) return SW_OK;
  WgAdam ad = adam;
  return wg_launch_adam(wb, wgrad_ws, ad, st);
}

// Can sw_disc_update run this pass (else: sw_disc_fwd + sw_disc_bwd_gan*)?  Built in round 3 for the shapes that leave CUs
// idle (<= 128 tiles); measured in round 4 at 160 .. 2 048 tiles: one workgroup per CU without the round trip of the gates
// through the save buffer beats the two launches at two workgroups per CU by 1-5 % of the step everywhere.
extern "C" int sw_disc_update_supported(const float* d_w, int B, int To, int Tp) {
  if (!d_w || B < 1 || To < 1 || Tp < 1 || Tp > 12) return 0;
  if (!sw_disc_images_for(d_w, Tp).img) return 0;
  return upd_lds(Tp).total * 4 <= 163840 ? 1 : 0;
}
// One discriminator update pass (train.py:476-495) in ONE launch + its weight-gradient GEMM (+ the Adam update when
// adam_w = d_w): what sw_disc_fwd(nb = 2, x_mode 0, save_lstm 1 | 2) followed by sw_disc_bwd_gan[_adam] computes, to the
// same buffers.  obs_pre = 1: the observation-LSTM rows are already in dsave (sw_dec_rollout_fwd_aux).  Requires
// sw_disc_update_supported() (SW_ESHAPE otherwise).
extern "C" int sw_disc_update(const float* obsv, int To, const float* const* pred4, const float* d_w, int B, int Tp,
                              float* const* label, float* const* code, float* dsave, int obs_pre, float* w_snapshot,
                              const float* targets, int t0, int t1, const float* z, float g_label, float g_code, float* ddelta,
                              float* d_d_w, float* wgrad_ws, float* loss_part, float* adam_w, float* adam_m, float* adam_v,
                              const float* adam_step, double lr, double beta1, double beta2, double eps, void* stream) {
  if (!obsv || !pred4 || !pred4[0] || !pred4[1] || !d_w || !label || !label[0] || !label[1] || !code || !code[0] || !code[1] ||
      !dsave || !targets || !z || !ddelta || !d_d_w || !wgrad_ws || t0 < 0 || t1 < 0 || To < 2)
    return SW_EARG;
  if (adam_w && (!adam_m || !adam_v || !adam_step || adam_w != d_w)) return SW_EARG;
  if (B == 0) return SW_OK;
  if (!sw_disc_update_supported(d_w, B, To, Tp)) return SW_ESHAPE;
  const int lds = upd_lds(Tp).total * 4;
  static int attr = 0;
  if (attr < lds) {
    if (int rc = set_lds((const void*)disc_update_kernel, lds)) return rc;
    attr = lds;
  }
  hipStream_t st = (hipStream_t)stream;
  const DiscImages di = sw_disc_images_for(d_w, Tp);
  DiscLoss gl{targets, z, t0, t1, g_label, g_code, 1, loss_part};
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  WgBatch wb;
  if (int rc = disc_wgrad_problems(wb, dsave, ddelta, 2, B, To, Tp, d_d_w)) return rc;
  SW_LAUNCH(disc_update_kernel, dim3(tiles), dim3(SW_THREADS), lds, st, obsv, To, pred4[0], pred4[1], d_w, B, Tp, label[0],
            label[1], code[0], code[1], dsave, obs_pre ? 1 : 0, w_snapshot, gl, ddelta, di.img);
  SW_CHECK_LAUNCH("disc_update_kernel");
  WgAdam ad;
  if (adam_w) {
    ad.w = adam_w; ad.m = adam_m; ad.v = adam_v; ad.g0 = d_d_w; ad.step = adam_step;
    ad.lr = lr; ad.beta1 = beta1; ad.beta2 = beta2; ad.eps = eps;
    ad.img = const_cast<float*>(di.img);
    ad.tab = di.tab;
  }
  return wg_launch_adam(wb, wgrad_ws, ad, st);
}

extern "C" int sw_disc_bwd(const float* d_w, const float* dsave, const float* const* dlabel,
                           const float* const* dcode, int nb, int B, int To, int Tp, float* ddelta, float* d_d_w,
                           float* const* dpred4, float* wgrad_ws, void* stream) {
  DiscLoss gl{};
  gl.on = 0;
  gl.loss_part = nullptr;
  return disc_bwd_impl(d_w, dsave, dlabel, dcode, nb, B, To, Tp, ddelta, d_d_w, dpred4, wgrad_ws, stream, gl);
}

extern "C" int sw_disc_bwd_gan_adam(const float* d_w, const float* dsave, const float* const* label,
                                    const float* const* code, const float* targets, int t0, int t1, const float* z,
                                    float g_label, float g_code, int nb, int B, int To, int Tp, float* ddelta,
                                    float* d_d_w, float* const* dpred4, float* wgrad_ws, float* loss_part,
                                    float* adam_w, float* adam_m, float* adam_v, const float* adam_step, double lr,
                                    double beta1, double beta2, double eps, void* stream) {
  if (!targets || !z || t0 < 0 || t1 < 0) return SW_EARG;
  if (adam_w && (!adam_m || !adam_v || !adam_step || !d_d_w || adam_w != d_w)) return SW_EARG;
  DiscLoss gl{targets, z, t0, t1, g_label, g_code, 1, loss_part};
  WgAdam ad;
  if (adam_w) {
    ad.w = adam_w; ad.m = adam_m; ad.v = adam_v; ad.g0 = d_d_w; ad.step = adam_step;
    ad.lr = lr; ad.beta1 = beta1; ad.beta2 = beta2; ad.eps = eps;
    const DiscImages di = sw_disc_images_for(adam_w, Tp);   // registered images follow the update element by element
    ad.img = const_cast<float*>(di.img);
    ad.tab = di.tab;
  }
  return disc_bwd_impl(d_w, dsave, label, code, nb, B, To, Tp, ddelta, d_d_w, dpred4, wgrad_ws, stream, gl, ad);
}
extern "C" int sw_disc_bwd_gan(const float* d_w, const float* dsave, const float* const* label,
                               const float* const* code, const float* targets, int t0, int t1, const float* z,
                               float g_label, float g_code, int nb, int B, int To, int Tp, float* ddelta,
                               float* d_d_w, float* const* dpred4, float* wgrad_ws, float* loss_part, void* stream) {
  return sw_disc_bwd_gan_adam(d_w, dsave, label, code, targets, t0, t1, z, g_label, g_code, nb, B, To, Tp, ddelta, d_d_w,
                              dpred4, wgrad_ws, loss_part, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, stream);
}

// ---- derived images of the discriminator's weights (swdimg, sw_common.h) --------------------------------------------
extern "C" int sw_disc_image_floats(int Tp) {
  if (Tp < 1 || Tp > 64) return SW_EARG;
  return swdimg::HEADT + head_lds_b(Tp, 0).dlab;
}
// tab[2 i], tab[2 i + 1] = image offsets of packed float i (or -1); host memory, sw_param_count(SW_GRP_DISC, Tp) pairs
extern "C" int sw_disc_image_table(int Tp, int* tab) {
  if (Tp < 1 || Tp > 64 || !tab) return SW_EARG;
  const swp::Disc O = swp::disc(Tp);
  const HeadLdsB L = head_lds_b(Tp, 0);
  for (int i = 0; i < 2 * O.n; ++i) tab[i] = -1;
  for (int R = 0; R < 256; ++R)
    for (int C = 0; C < 64; ++C) {
      const int i = O.whh + R * 64 + C;
      tab[2 * i] = swdimg::OP_WHH + ((((R >> 4) * 4 + (C >> 4)) * 64 + ((C & 15) >> 2) * 16 + (R & 15)) * 4) + (C & 3);
      tab[2 * i + 1] = swdimg::OP_WHHT + ((((C >> 4) * 16 + (R >> 4)) * 64 + ((R & 15) >> 2) * 16 + (C & 15)) * 4) + (R & 3);
    }
  auto head = [&](int w_off, int M, int K, int img_off, int ld) {   // XT[c][r] = X[r][c]
    for (int r = 0; r < M; ++r)
      for (int c = 0; c < K; ++c) tab[2 * (w_off + r * K + c)] = swdimg::HEADT + img_off + c * ld + r;
  };
  head(O.of0w, 32, 64, L.of0T, LD32);
  head(O.of1w, 32, 32, L.of1T, LD32);
  head(O.pe0w, 32, 4 * Tp, L.pe0T, LD32);
  head(O.pe1w, 32, 32, L.pe1T, LD32);
  head(O.cl0w, 32, 64, L.cl0T, LD32);
  head(O.la0w, 32, 64, L.la0T, LD32);
  head(O.cl1w, 1, 32, L.cl1T, LD16);
  head(O.la1w, 2, 32, L.la1T, LD16);
  return SW_OK;
}
static thread_local const float* g_dimg_w = nullptr;     // per host thread, like the generator's registration (sw_misc.hip)
static thread_local DiscImages g_dimg;
static thread_local int g_dimg_tp = 0;
DiscImages sw_disc_images_for(const float* d_w, int Tp) {
  return (g_dimg.img && d_w == g_dimg_w && Tp == g_dimg_tp) ? g_dimg : DiscImages();
}
void sw_disc_images_register(const float* d_w, const float* img, const int* tab, int Tp) {
  g_dimg_w = d_w; g_dimg.img = img; g_dimg.tab = tab; g_dimg_tp = Tp;
}
__global__ __launch_bounds__(256) void disc_images_kernel(const float* __restrict__ d_w, float* __restrict__ img,
                                                           const int* __restrict__ tab, int n) {
  disc_images_scatter(d_w, img, tab, n, blockIdx.x, gridDim.x);
}
// Scatter the packed weights d_w into img (sw_disc_image_floats(Tp) floats, ZERO-FILLED by the caller once: padding is
// never written) through the device copy `tab` of sw_disc_image_table(Tp), and register the images for d_w: until the
// registration is dropped - sw_disc_images(NULL, NULL, NULL, 0, NULL) - sw_disc_fwd / sw_disc_dpred / sw_disc_bwd* called with
// these weights read the images, and sw_disc_bwd_gan_adam keeps them current while it updates the weights.  Whoever
// changes the weights by other means re-scatters or drops the registration.
extern "C" int sw_disc_images(const float* d_w, float* img, const int* tab, int Tp, void* stream) {
  if (!img) {
    sw_disc_images_register(nullptr, nullptr, nullptr, 0);
    return SW_OK;
  }
  if (!d_w || !tab || Tp < 1 || Tp > 64) return SW_EARG;
  SW_LAUNCH(disc_images_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, d_w, img, tab, swp::disc(Tp).n);
  SW_CHECK_LAUNCH("disc_images_kernel");
  sw_disc_images_register(d_w, img, tab, Tp);
  return SW_OK;
}
