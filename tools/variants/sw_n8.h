// sw_n8.h - building blocks of the NARROW serial kernels: 8 agents per workgroup on v_mfma_f32_4x4x1_16B_f32.
//
// Why a second tiling.  The 16x16x4 tiling of sw_common.h needs 16 agents in the MFMA's N dimension, so the metric
// shape (2048 agents) is 128 workgroups for 256 CUs and every serial kernel leaves half of the chip idle.  An
// 8-agent tile on the same instruction would issue the same number of MFMAs per workgroup (half of each wasted).
// v_mfma_f32_4x4x1_16B_f32 - 16 independent 4x4 outer products, K = 1, 8 cycles, the same 64 FLOP/clk/SIMD - has
// only FOUR columns per block: with the 4 agents of a group as the A rows and 64 weight rows spread over the 16
// blocks' B columns, one instruction computes 64 output rows x 4 agents x 1 k with nothing wasted, and a workgroup
// of 8 agents does half the matrix work of a 16-agent one.  Measured (tools/mb/mfma4x4.hip, LSTM step, rows saved):
// 2.24 us per step for 128 sixteen-agent tiles -> 1.49 us for 256 eight-agent tiles.
//
// Operand roles (checked against a host model by the micro-benchmark):
//   B operand, lane l            = W[row(l)][k]                         one weight ROW per lane, one VGPR per k
//   A operand, lane l = 4 blk + i = X[agent i][k(v, blk)]               activations, broadcast: `cbsz = 4, abid = b`
//                                                                       makes every block use block b's A rows, so ONE
//                                                                       VGPR holds 16 k's of the 4 agents
//   D, lane l, VGPR r            = Y[row(l)][agent r]
// With cbsz = 3 the upper 8 blocks take their A rows from block 8 + abid: lanes 32..63 can then work on a second
// k-range (K split inside the wave, halves summed with one lane swap) - used where a job has only 32 rows.
#pragma once
#include "sw_common.h"
#include <type_traits>

#define SW8_TILE 8       // agents per narrow workgroup: two groups of 4
#define SW8_LD64 80      // LDS row strides == 16 (mod 64): the b128 A-operand reads of lanes (blk, i) - row i, column
#define SW8_LD80 80      // 4 blk - are then bank-conflict free (searched: strides 16 / 48 mod 64)
#define SW8_LD160 176
#define SW8_LD256 272

template <int CBSZ, int ABID>
__device__ __forceinline__ f32x4 sw_m4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CBSZ, ABID, 0);
}
template <int I, int N, class F>
__device__ __forceinline__ void sw_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sw_static_for<I + 1, N>(f);
  }
}

// value of x in lane l ^ 32 (v_permlane32_swap: a VALU move, no LDS crossbar)
__device__ __forceinline__ float sw_xor32(float x, bool up) {
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
  const unsigned u = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[0] = (x.lo, x.lo), r[1] = (x.hi, x.hi)
  return __uint_as_float(up ? r[0] : r[1]);
#else
  return __shfl_xor(x, 32);
#endif
}

// value of x in lane l ^ 16 (v_permlane16_swap); odd = (lane >> 4) & 1
__device__ __forceinline__ float sw_xor16(float x, int odd) {
  const unsigned u = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // r[0] = rows (0,0,2,2), r[1] = rows (1,1,3,3)
  return __uint_as_float(odd ? r[0] : r[1]);
}
// value of x in lane l ^ 8 / l ^ 4 inside its row of 16: after x += rot(x) all lanes of an orbit hold the same sum,
// so rotations serve as butterflies (DPP row_ror)
__device__ __forceinline__ float sw_ror8(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));
}
__device__ __forceinline__ float sw_ror4(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, false));
}
// sums over the lanes that differ in lane bits {5}, {4,5}, {3,4,5}, {2,3,4,5}: the k-parts of a product whose K was
// split over 2 / 4 / 8 / 16 lane groups (cbsz = 3 / 2 / 1 / 0)
__device__ __forceinline__ float sw8_sum2(float x, int lane) { return x + sw_xor32(x, lane >> 5); }
__device__ __forceinline__ float sw8_sum4(float x, int lane) {
  x += sw_xor16(x, (lane >> 4) & 1);
  return x + sw_xor32(x, lane >> 5);
}
__device__ __forceinline__ float sw8_sum8(float x, int lane) {
  x += sw_ror8(x);
  return sw8_sum4(x, lane);
}
__device__ __forceinline__ float sw8_sum16(float x, int lane) {
  x += sw_ror4(x);
  return sw8_sum8(x, lane);
}

// acc0 / acc1 += W0 / W1 (64 k each, registers) x the 64-long activation vector held as 4 broadcast VGPRs
// (hv[v], lane (blk, i) = X[agent i][16 v + blk]): 128 instructions on two independent accumulators.
__device__ __forceinline__ void sw8_mm64x2(f32x4& acc0, f32x4& acc1, const float (&w0)[64], const float (&w1)[64],
                                           const float (&hv)[4]) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    sw_static_for<0, 16>([&](auto ic) {
      constexpr int b = decltype(ic)::value;
      acc0 = sw_m4<4, b>(hv[v], w0[16 * v + b], acc0);
      acc1 = sw_m4<4, b>(hv[v], w1[16 * v + b], acc1);
    });
  }
}

// position of unit k inside a 64-wide LDS row written for the broadcast A operand: lane (blk, i) fetches its four
// values X[agent i][16 v + blk], v = 0..3, as ONE 16-byte read at column 4 blk
__device__ __forceinline__ int sw8_pos64(int k) { return ((k & 15) << 2) | (k >> 4); }

// ---------------------------------------------------------------------------------------------------------------
// LSTM cell on an 8-agent tile.  wave w: agent group ag = w & 1 (agents a0 + 4 ag .. + 3), unit half uh = w >> 1.
// lane l: lower half (l < 32) owns the rows of gates i (pass 0) and g (pass 1) of unit u = 32 uh + (l & 31), the
// upper half those of gates f and o of the same unit; the halves exchange their activated gates (sw_xor32) and BOTH
// compute c' and h' (the SIMD executes both halves anyway), so every store below is unconditional.
// ---------------------------------------------------------------------------------------------------------------
struct Lstm8W {
  float w0[64], w1[64];   // W_hh rows of pass 0 / pass 1
  float wx0[4], wx1[4];   // input matrix rows (composed W_ih W_embed for the encoder, W_ih for the discriminator)
  float b0, b1;
};
struct Lstm8Lane {
  int ag, uh, up, u, row0, row1, ai, blk;
  float sc;               // pass 1: tanh(x) = 2 sigmoid(2x) - 1 on the lower half (sc = 2), sigmoid on the upper (sc = 1)
  __device__ __forceinline__ Lstm8Lane() {
    const int lane = sw_lane(), wave = sw_wave();
    ag = wave & 1;
    uh = wave >> 1;
    up = lane >> 5;
    u = 32 * uh + (lane & 31);
    row0 = (up ? 64 : 0) + u;
    row1 = (up ? 192 : 128) + u;
    ai = lane & 3;
    blk = lane >> 2;
    sc = up ? 1.0f : 2.0f;
  }
};
// One weight ROW per lane is the worst global access pattern there is (64 lanes x 16 bytes from 64 different 256-byte
// rows per instruction; measured: a 12 us prologue).  A [256][64] matrix is therefore fetched coalesced by the whole
// workgroup (16 float4 per thread), parked in LDS as [256][SW8_WLD] and read back row-wise (conflict free:
// SW8_WLD == 4 mod 64).
#define SW8_WLD 68
__device__ __forceinline__ void sw8_stage256_load(f32x4 (&v)[16], const float* __restrict__ src) {
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = ld4(src + 4 * (threadIdx.x + SW_THREADS * j));
}
__device__ __forceinline__ void sw8_stage256_store(const f32x4 (&v)[16], float* dst) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int f = threadIdx.x + SW_THREADS * j;
    st4(dst + (f >> 4) * SW8_WLD + 4 * (f & 15), v[j]);
  }
}
// lstm_prep_rows(composed) with the W_ih rows read from their LDS image: Wx = W_ih W_e, bx = W_ih b_e + b_ih + b_hh
__device__ __forceinline__ void lstm8_prep_rows_lds(const float* We, const float* be, const float* wih_lds, const float* bih,
                                                    const float* bhh, float* wx_lds, float* bx_lds) {
  const int row = threadIdx.x;
  f32x4 w[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) w[e] = ld4(wih_lds + row * SW8_WLD + 4 * e);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, ab = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const f32x4 bq = ld4(be + 4 * e);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 em = ld4(We + (4 * e + q) * 4);
      a0 = fmaf(w[e][q], em[0], a0);
      a1 = fmaf(w[e][q], em[1], a1);
      a2 = fmaf(w[e][q], em[2], a2);
      a3 = fmaf(w[e][q], em[3], a3);
      ab = fmaf(w[e][q], bq[q], ab);
    }
  }
  st4(wx_lds + row * 4, f32x4{a0, a1, a2, a3});
  bx_lds[row] = ab + bih[row] + bhh[row];
}
__device__ __forceinline__ void lstm8_load_whh(Lstm8W& W, const float* Whh, const Lstm8Lane& L, int ld = 64) {
  const float* r0 = Whh + (size_t)L.row0 * ld;
  const float* r1 = Whh + (size_t)L.row1 * ld;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const f32x4 a = ld4(r0 + 4 * j), b = ld4(r1 + 4 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      W.w0[4 * j + e] = a[e];
      W.w1[4 * j + e] = b[e];
    }
  }
}
__device__ __forceinline__ void lstm8_load_wx(Lstm8W& W, const float* wx_lds, const float* bx_lds, const Lstm8Lane& L) {
  const f32x4 a = ld4(wx_lds + L.row0 * 4), b = ld4(wx_lds + L.row1 * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    W.wx0[e] = a[e];
    W.wx1[e] = b[e];
  }
  W.b0 = bx_lds[L.row0];
  W.b1 = bx_lds[L.row1];
}

// One cell step for the wave's 4 agents.  xv: lane (blk, i) = x4[agent i][component blk & 3]; hrow = the lane's row
// of the h tile (&hbuf[(4 ag + i) * SW8_LD64 + 4 blk], positions sw8_pos64).  On return a0 / a1 hold the lane's own
// activated gates (i | f, g | o: what the save rows take), c / h the new state of unit u for the 4 agents.
__device__ __forceinline__ void lstm8_cell(const Lstm8W& W, const Lstm8Lane& L, float xv, const float* hrow, f32x4& a0,
                                           f32x4& a1, f32x4& c, f32x4& h) {
  const f32x4 hq = ld4(hrow);
  const float hv[4] = {hq[0], hq[1], hq[2], hq[3]};
  f32x4 acc0 = {W.b0, W.b0, W.b0, W.b0}, acc1 = {W.b1, W.b1, W.b1, W.b1};
  sw_static_for<0, 4>([&](auto ic) {
    constexpr int k = decltype(ic)::value;
    acc0 = sw_m4<4, k>(xv, W.wx0[k], acc0);
    acc1 = sw_m4<4, k>(xv, W.wx1[k], acc1);
  });
  sw8_mm64x2(acc0, acc1, W.w0, W.w1, hv);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    a0[r] = sw_sigmoid(acc0[r]);
    a1[r] = fmaf(L.sc, sw_sigmoid(L.sc * acc1[r]), 1.0f - L.sc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float p0 = sw_xor32(a0[r], L.up), p1 = sw_xor32(a1[r], L.up);
    const float gi = L.up ? p0 : a0[r], gf = L.up ? a0[r] : p0;
    const float gg = L.up ? p1 : a1[r], go = L.up ? a1[r] : p1;
    const float cn = fmaf(gf, c[r], gi * gg);
    c[r] = cn;
    h[r] = go * sw_tanh(cn);
  }
}
// h of the wave's 4 agents -> the h tile (lower lanes write agents 0, 1, upper lanes agents 2, 3: no duplicate writes)
__device__ __forceinline__ void lstm8_put_h(float* htile, const Lstm8Lane& L, const f32x4& h) {
  const int r0 = L.up ? 2 : 0;
  float* p = htile + (4 * L.ag + r0) * SW8_LD64 + sw8_pos64(L.u);
  p[0] = L.up ? h[2] : h[0];
  p[SW8_LD64] = L.up ? h[3] : h[1];
}
