// sw_lstm_dev.h - the time-unrolled LSTM cell shared by the encoder, decode-loop and
// discriminator kernels (reference: nn.LSTM in EncoderLstm train.py:254,268 and Discriminator
// train.py:278,299; gate order i,f,g,o, c' = f c + i g, h' = o tanh(c')).
//
// A workgroup = 4 waves owns a tile of 16 agents.  Wave w owns hidden units [16w, 16w+16) of all
// four gates, so the cell update is lane-local.  W_hh lives in registers for the whole kernel
// (64 VGPRs/lane); the 4-d input goes through a 256x4 input matrix Wx which for the encoder is
// the composition W_ih * W_embed (no non-linearity sits between embed and the LSTM,
// train.py:266-268), for the discriminator W_ih itself.
#pragma once
#include "sw_common.h"

#define SW_HLD 68    // LDS row stride of a 64-wide h tile
#define SW_GLD 260   // LDS row stride of a 256-wide dgates tile

struct LstmW {
  f32x4 whh[4][4];  // [gate][j] = Whh[gate*64 + u0 + ln][16j + 4lg .. +3]
  float wx[4];      // Wx[gate*64 + u0 + ln][lg]
  f32x4 bias[4];    // rows gate*64 + u0 + 4lg + r
};

// Cooperative prologue: every thread computes one row of the input matrix / bias into LDS
// (wx_lds[256][4], bx_lds[256]); caller must __syncthreads() afterwards.
//   composed (encoder): Wx = Wih * We, bx = Wih * be + bih + bhh
//   direct (discriminator): Wx = Wih (256x4), bx = bih + bhh
__device__ __forceinline__ void lstm_prep_rows(const float* We, const float* be, const float* Wih,
                                               const float* bih, const float* bhh, bool composed,
                                               float* wx_lds, float* bx_lds) {
  int row = threadIdx.x;  // 256 threads = 256 gate rows
  if (composed) {
    // the whole W_ih row first (16 independent float4 loads: ONE L2 round trip, not one per k-step),
    // W_embed / b_embed are tiny and L1-resident after the first touch
    f32x4 w[16];
    const float* wr = Wih + (size_t)row * 64;
#pragma unroll
    for (int e = 0; e < 16; ++e) w[e] = ld4(wr + 4 * e);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, ab = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      f32x4 bq = ld4(be + 4 * e);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 em = ld4(We + (4 * e + q) * 4);
        a0 = fmaf(w[e][q], em[0], a0);
        a1 = fmaf(w[e][q], em[1], a1);
        a2 = fmaf(w[e][q], em[2], a2);
        a3 = fmaf(w[e][q], em[3], a3);
        ab = fmaf(w[e][q], bq[q], ab);
      }
    }
    wx_lds[row * 4 + 0] = a0;
    wx_lds[row * 4 + 1] = a1;
    wx_lds[row * 4 + 2] = a2;
    wx_lds[row * 4 + 3] = a3;
    bx_lds[row] = ab + bih[row] + bhh[row];
  } else {
    f32x4 w = ld4(Wih + row * 4);
    wx_lds[row * 4 + 0] = w[0];
    wx_lds[row * 4 + 1] = w[1];
    wx_lds[row * 4 + 2] = w[2];
    wx_lds[row * 4 + 3] = w[3];
    bx_lds[row] = bih[row] + bhh[row];
  }
}

// W_hh rows straight from global memory: issued at the very top of a kernel so the loads overlap the
// LDS staging of the other weights; the composed input matrix / bias come from LDS after the barrier.
__device__ __forceinline__ void lstm_load_whh(LstmW& W, const float* Whh, int u0, int ln, int lg) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float* wr = Whh + (size_t)(g * 64 + u0 + ln) * 64 + 4 * lg;
#pragma unroll
    for (int j = 0; j < 4; ++j) W.whh[g][j] = ld4(wr + 16 * j);
  }
}
__device__ __forceinline__ void lstm_load_wx(LstmW& W, const float* wx_lds, const float* bx_lds, int u0, int ln, int lg) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    W.wx[g] = wx_lds[(g * 64 + u0 + ln) * 4 + lg];
    W.bias[g] = ld4(bx_lds + g * 64 + u0 + 4 * lg);
  }
}

// The generator's LSTM weights from the image buffer of the step (swimg, sw_common.h): W_hh as an operand-layout image
// (a wave's load instruction = 1 KB of consecutive memory; the row-per-lane loads of lstm_load_whh touch 64 cache lines
// per instruction), the composed input matrix and bias straight into their registers - no LDS staging, no barrier.
__device__ __forceinline__ void lstm_load_img(LstmW& W, const float* __restrict__ gimg, int wave, int lane) {
  const int ln = lane & 15, lg = lane >> 4, u0 = wave * 16;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) W.whh[g][j] = ld4(gimg + swimg::OP_WHH + ((((size_t)4 * g + wave) * 4 + j) * 64 + lane) * 4);
    W.wx[g] = gimg[swimg::WX + (g * 64 + u0 + ln) * 4 + lg];
    W.bias[g] = ld4(gimg + swimg::BX + g * 64 + u0 + 4 * lg);
  }
}

// One cell step.  xb = x4[agent ln][component lg]; hrow = &h_lds[ln*SW_HLD + 4*lg] (previous h).
// On return gate[] holds the post-activation gates i,f,g,o, c the new cell state, h the new h.
__device__ __forceinline__ void lstm_cell(const LstmW& W, float xb, const float* hrow, f32x4 gate[4],
                                          f32x4& c, f32x4& h) {
  f32x4 acc[4], b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = ld4(hrow + 16 * j);   // one LDS round trip for the whole step
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = SW_MFMA(W.wx[g], xb, W.bias[g]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = SW_MFMA(W.whh[g][j][r], b[j][r], acc[g]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float i = sw_sigmoid(acc[0][r]);
    float f = sw_sigmoid(acc[1][r]);
    float g = sw_tanh(acc[2][r]);
    float o = sw_sigmoid(acc[3][r]);
    float cn = fmaf(f, c[r], i * g);
    gate[0][r] = i;
    gate[1][r] = f;
    gate[2][r] = g;
    gate[3][r] = o;
    c[r] = cn;
    h[r] = o * sw_tanh(cn);
  }
}

// Backward of one cell step (elementwise part).  In: dh, dc (gradients w.r.t. h_t, c_t), saved
// gates/c_t/c_{t-1}.  Out: dgate[] = gradients w.r.t. the four PRE-activation gate rows, dc :=
// gradient w.r.t. c_{t-1}.
__device__ __forceinline__ void lstm_cell_bwd(const f32x4 gate[4], f32x4 ct, f32x4 cprev, f32x4 dh,
                                              f32x4& dc, f32x4 dgate[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float i = gate[0][r], f = gate[1][r], g = gate[2][r], o = gate[3][r];
    float tc = sw_tanh(ct[r]);
    float d_o = dh[r] * tc;
    float dct = fmaf(dh[r] * o, 1.0f - tc * tc, dc[r]);
    dgate[0][r] = dct * g * i * (1.0f - i);
    dgate[1][r] = dct * cprev[r] * f * (1.0f - f);
    dgate[2][r] = dct * i * (1.0f - g * g);
    dgate[3][r] = d_o * o * (1.0f - o);
    dc[r] = dct * f;
  }
}

// The dgates rows of a BPTT step go to memory FROM THE LDS TILE the step builds anyway ([16 agents][SW_GLD]), behind the
// step's barrier and one agent per store instruction: a wave writes 4 agents' rows, lane l the l-th float4 of the 256-float
// row - 1 KB of consecutive memory per instruction (8 full cache lines).  Stored from the MFMA result registers a lane
// holds 4 units of one agent, i.e. an instruction scatters sixteen 64-byte pieces over sixteen rows - half-line writes,
// which the store path charges for: at the dense-crowd shape the dgates stores were a third of disc_bwd (round 3,
// SW_EXP_NOSTORE).  Rows beyond agent B-1 (padding of the last tile) go to `trash` (16 x 256 floats).
__device__ __forceinline__ void lstm_store_dgates_tile(const float* dgtile, float* __restrict__ rows /*row of agent a0*/,
                                                       float* __restrict__ trash, int a0, int B, int wave, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int a = 4 * wave + q;
    const f32x4 v = ld4(dgtile + a * SW_GLD + 4 * lane);
    // trash == nullptr: the padding rows of the tile are exact replicas of agent B-1 (every load of the kernel is clamped
    // to it) and are stored over its row - the same values to the same place
    float* dst = (a0 + a < B) ? rows + (size_t)a * 256 + 4 * lane
                 : trash      ? trash + a * 256 + 4 * lane
                              : rows + (size_t)(B - 1 - a0) * 256 + 4 * lane;
    st4g(dst, v);
  }
}

// Forward counterpart: the saved row of an LSTM step (gates i f g o | c | h = 384 floats per agent) is assembled in an LDS
// tile [16 agents][SW_ALD] - whose h columns double as the B operand of the next step's products - and written out behind
// the step's barrier, a wave storing 4 agents' rows as six 1 KB instructions of consecutive memory.
#define SW_ALD 388   // LDS row stride of a 384-wide saved row
__device__ __forceinline__ void lstm_put_act_tile(float* tile, const f32x4 gate[4], f32x4 c, f32x4 h, int ln, int lg, int u0) {
  float* row = tile + ln * SW_ALD + u0 + 4 * lg;
#pragma unroll
  for (int g = 0; g < 4; ++g) st4(row + g * 64, gate[g]);
  st4(row + 256, c);
  st4(row + 320, h);
}
__device__ __forceinline__ void lstm_store_act_tile(const float* tile, float* __restrict__ rows_t /*row of agent 0 at this step*/,
                                                    int a0, int B, int wave, int lane) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int f = k * 64 + lane, q = f / 96, c4 = f - q * 96;      // float4 c4 of agent 4 wave + q
    const int a = 4 * wave + q;
    const f32x4 v = ld4(tile + a * SW_ALD + 4 * c4);
    st4g(rows_t + (size_t)min(a0 + a, B - 1) * 384 + 4 * c4, v);     // padding rows: replicas of agent B-1, same values
  }
}

// W_hh^T in registers for dh_{t-1} = W_hh^T dgates: whhT[j][r] = Whh[16j + 4lg + r][u0 + ln].
struct LstmWT {
  f32x4 whhT[16];
};
__device__ __forceinline__ void lstm_load_wT(LstmWT& W, const float* Whh, int u0, int ln, int lg) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) W.whhT[j][r] = Whh[(size_t)(16 * j + 4 * lg + r) * 64 + u0 + ln];
  }
}
// dgrow = &dg_lds[ln*SW_GLD + 4*lg]
__device__ __forceinline__ f32x4 lstm_dh_prev(const LstmWT& W, const float* dgrow) {
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  f32x4 b[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) b[j] = ld4(dgrow + 16 * j);
#pragma unroll
  for (int j = 0; j < 16; j += 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a0 = SW_MFMA(W.whhT[j][r], b[j][r], a0);
      a1 = SW_MFMA(W.whhT[j + 1][r], b[j + 1][r], a1);
    }
  }
  return a0 + a1;
}

// x4[agent][t][comp] for the observation rule of get_traj_4d (train.py:131-133): v_0 := v_1, as two raw loads
// (a, q) with x = a - (comp >= 2 ? q : 0).  Branch-free and split from the arithmetic on purpose: memory operations
// under lane-dependent branches make the compiler lose count of what is in flight (it then waits for everything,
// s_waitcnt vmcnt(0)), and arithmetic on a prefetched value gets scheduled right behind its load.
__device__ __forceinline__ void obs_x4_load(const float* pos, int b, int t, int T, int comp, float& a, float& q) {
  const float* p = pos + (size_t)b * T * 2;
  const int c = comp & 1, tt = t == 0 ? 1 : t;
  const bool vel = comp >= 2;
  a = p[(vel ? tt : t) * 2 + c];
  q = p[(vel ? tt - 1 : t) * 2 + c];
}
__device__ __forceinline__ float obs_x4(const float* pos, int b, int t, int T, int comp) {
  float a, q;
  obs_x4_load(pos, b, t, T, comp, a, q);
  return a - (comp >= 2 ? q : 0.f);
}

// The time loop of an LSTM over a 4-d input sequence (h0 = c0 given in hbuf[0] / c) for the 16-agent tile of agent
// row b (clamped), W loaded.  XMODE 0: x = positions [B][T][2]; 1: x = [B][T][4].  SAVE: per step t and agent b,
// act + (t B + b) 384 = gates i|f|g|o [256], c [64], h [64] and the step's input at x4s + (t B + b) 4.  No
// conditional memory operation inside (see enc_lstm_fwd_kernel); padding lanes of the last tile are replicas of
// agent B-1.  On return h_T sits in hbuf[T & 1].
template <int XMODE, bool SAVE>
__device__ __forceinline__ void lstm_obs_loop(const LstmW& W, float* hbuf, const float* __restrict__ x, int T, int B, int b,
                                              f32x4& c, f32x4& h, float* __restrict__ act, float* __restrict__ x4s) {
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16;
  float xa, xq = 0.f;
  auto load_x = [&](int t) {
    if constexpr (XMODE == 0) obs_x4_load(x, b, t, T, lg, xa, xq);
    else xa = x[((size_t)b * T + t) * 4 + lg];
  };
  load_x(0);
  asm volatile("" : "+v"(xa), "+v"(xq));   // waited for HERE: the loop header must see no pending load on any path in
  float* arow = SAVE ? act + (size_t)b * 384 + u0 + 4 * lg : nullptr;
  float* xrow = SAVE ? x4s + (size_t)b * 4 + lg : nullptr;
  for (int t = 0; t < T; ++t) {
    const float xb = XMODE == 0 ? xa - (lg >= 2 ? xq : 0.f) : xa;
    load_x(min(t + 1, T - 1));   // the input of step t+1 is fetched while step t computes
    f32x4 gate[4];
    lstm_cell(W, xb, &hbuf[(t & 1) * 16 * SW_HLD + ln * SW_HLD + 4 * lg], gate, c, h);
    st4(&hbuf[((t + 1) & 1) * 16 * SW_HLD + ln * SW_HLD + u0 + 4 * lg], h);
    if constexpr (SAVE) {
#pragma unroll
      for (int g = 0; g < 4; ++g) st4g(arow + g * 64, gate[g]);
      st4g(arow + 256, c);
      st4g(arow + 320, h);
      *xrow = xb;   // all four waves hold the same x_t and all store it
      arow += (size_t)B * 384;
      xrow += (size_t)B * 4;
    }
    sw_barrier();
    asm volatile("" : "+v"(xa), "+v"(xq));   // the prefetched input is not touched before this point
  }
}

// Observation LSTM of the discriminator (train.py:296-299) for the 16-agent tile at a0, leaving the rows its
// backward needs (see lstm_obs_loop).  It exists as a function so that idle workgroups of ANOTHER launch can run it
// (the first D pass of a step does not depend on the generator: sw_dec_rollout_fwd_aux).
// smem: [2][16][SW_HLD] h tiles | [256][4] Wx | [256] bx  (2 * 16 * SW_HLD + 1280 floats).
__device__ __forceinline__ void disc_obs_lstm_tile(float* smem, const float* __restrict__ obsv, int To, int x_mode,
                                                   const float* wih, const float* whh, const float* bih, const float* bhh,
                                                   int B, int a0, float* __restrict__ act, float* __restrict__ x4s) {
  float* hbuf = smem;
  float* wx_lds = smem + 2 * 16 * SW_HLD;
  float* bx_lds = wx_lds + 1024;
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16;
  const int b = min(a0 + ln, B - 1);
  LstmW W;
  lstm_load_whh(W, whh, u0, ln, lg);
  lstm_prep_rows(nullptr, nullptr, wih, bih, bhh, false, wx_lds, bx_lds);
  f32x4 c = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};
  st4(&hbuf[ln * SW_HLD + u0 + 4 * lg], h);
  sw_barrier();
  lstm_load_wx(W, wx_lds, bx_lds, u0, ln, lg);
  if (x_mode == 0) lstm_obs_loop<0, true>(W, hbuf, obsv, To, B, b, c, h, act, x4s);
  else lstm_obs_loop<1, true>(W, hbuf, obsv, To, B, b, c, h, act, x4s);
}
