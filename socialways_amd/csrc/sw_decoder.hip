// sw_decoder.hip - the decode loop of predict() (reference train.py:415-432): Tp times
// { DecoderFC on cat[h, S, z] (train.py:320-335) -> v ; p += v ; EncoderLstm step on (p, v) },
// as ONE persistent kernel per 16-agent tile, forward and backward.
//
// Layout of the work inside a workgroup (4 waves, one per SIMD):
//   * decoder weights live in LDS for the whole kernel (115 KB, zero padded to MFMA tiles);
//     LSTM W_hh lives in registers (sw_lstm_dev.h);
//   * cat[h,S,z] W1^T is split: u = W1[:,64:160] [S;z] + b1 is constant over the Tp steps
//     (train.py:411,421: S and z do not change inside the loop) and computed once; per step only
//     the 64-wide h part of layer 1 is multiplied;
//   * every layer's 16-row output tiles are dealt round-robin to the waves; activations move
//     between layers through small LDS tiles (one barrier per layer).
#include "../../include/socialways_hip.h"
#include "sw_lstm_dev.h"
#include "sw_disc_dev.h"
#include <type_traits>
#include <stdlib.h>

namespace {     // (LD64 = sw_ld(64) = 68 comes with sw_disc_dev.h)
constexpr int LD160 = sw_ld(160);  // 164
constexpr int LD80 = sw_ld(80);    // 84
constexpr int LD96 = sw_ld(96);    // 100

}  // namespace

#ifdef SW_PHASE_STAMPS
__device__ long long sw_stamps[16];
#define SW_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); long long _t = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) sw_stamps[k] += _t - _tprev; _tprev = _t; __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int sw_debug_stamps(long long* out, int reset) {
  if (reset) { long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(sw_stamps), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sw_stamps), 16 * sizeof(long long));
}
#else
#define SW_STAMP(k)
#endif
// Forward.  EVERY weight of the decode step lives in registers (A operands of the MFMAs: ~220 of the 512 VGPRs + AGPRs a
// wave owns at one wave per SIMD), loaded once from global memory in operand layout; LDS holds activations only
// (31 KB).  Round 2 had the 110 KB of decoder weights in LDS: every layer then began with a burst of ds_read_b128 for
// its weight tiles (112 KB per step through a 128 B/clk port shared by the four waves) before its first MFMA could
// issue.  The layers are also cut so that the four waves carry equal MFMA counts:
//   layer 1 (160 x 64):  wave w owns row tiles 2w, 2w+1 and ONE K-half of tile 8 + (w >> 1)        (40 MFMAs each; was 48/32)
//   layer 2 (80 x 160):  wave w owns row tile w and 3 / 3 / 2 / 2 of the 10 k-steps of tile 4       (52/52/48/48; was 80/40)
//   fc4 . fc3 (2 x 80):  every wave for itself (20), then its 16 hidden units of the LSTM step       (68)
// The K-split tiles leave PARTIAL sums in LDS; the consumer of the layer (every wave reads the whole activation row as
// its B operand anyway) adds the partials, the bias and applies the LeakyReLU while loading.
namespace {
struct FwdLds {
  static constexpr int LD128 = sw_ld(128);  // 132
  static constexpr int LD32 = sw_ld(32);    // 36
  static constexpr int LD16 = sw_ld(16);    // 20
  static constexpr int hbuf = 0;                        // [2][16][SW_ALD]: the saved LSTM row of a step (i f g o | c | h), assembled
                                                        //   here and stored row-wise behind the barrier; its h columns (320..) are
                                                        //   the B operand of the next step's products (lstm_put_act_tile)
  static constexpr int a1buf = hbuf + 2 * 16 * SW_ALD;  // [16][132]  a1[:, :128]            (prologue: [S|z] tile [16][100])
  static constexpr int p1 = a1buf + 16 * LD128;         // [2][16][36] K-halves of z1[:, 128:160] (u in half 0)
  static constexpr int a2buf = p1 + 2 * 16 * LD32;      // [16][68]   a2[:, :64]             (prologue: wx | bx | W43, 1456 floats)
  static constexpr int q2 = a2buf + 16 * LD64;          // [4][16][20] K-quarters of z2[:, 64:80] (no bias)
  static constexpr int total = q2 + 4 * 16 * LD16;
};
static_assert(16 * FwdLds::LD128 >= 16 * LD96, "prologue alias [S|z]");
static_assert(16 * LD64 + 4 * 16 * FwdLds::LD16 >= 1280 + 176, "prologue alias wx | bx | W43");
static_assert(FwdLds::total >= 2 * 16 * SW_HLD + 1280, "LDS of the observation-LSTM workgroups");
static_assert(FwdLds::total * 4 <= 160 * 1024, "LDS of a CU");

// the k-steps J0 .. J0+NJ-1 of layer 2's tile 4 (rows 64..79): acc += W2[64 + ln][16 j + 4 lg + r] a1[ln][16 j + 4 lg + r]
template <int J0, int NJ>
__device__ __forceinline__ f32x4 fwd_l2_part(const f32x4 (&w2p)[3], const f32x4 (&b1)[10]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = SW_MFMA(w2p[jj][r], b1[J0 + jj][r], acc);
  return acc;
}
}  // namespace

// SAVE (gsave given) and ADE (displacement-error sums wanted) are template parameters, the last decode step is peeled and
// nothing in the step is stored under a lane- or wave-dependent branch (padding lanes of the last tile are replicas of
// agent B-1 and store the same values to the same rows; values every wave holds are stored by every wave): with a
// conditional memory operation in the loop the compiler cannot count what is in flight and waited for EVERYTHING
// (s_waitcnt vmcnt(0)) at the head of the third layer of every step - the round trip of the ~40 KB of rows the step had
// just stored (found in the ISA in round 3; the backward kernels had been cleaned of this in round 1).
template <bool SAVE, bool ADE>
__global__ __launch_bounds__(SW_THREADS) void dec_rollout_fwd_kernel(
    const float* __restrict__ obsv, int To, const float* __restrict__ z, const float* __restrict__ S_pool,
    const float* __restrict__ hT, const float* __restrict__ cT, const float* __restrict__ enc_w,
    const float* __restrict__ dec_w, int B, int Tp, float* __restrict__ pred4, float* __restrict__ h_end,
    float* __restrict__ c_end, float* __restrict__ gsave, const float* __restrict__ gt, float inv_ss,
    float* __restrict__ ade_part, const float* __restrict__ dobs_w, float* __restrict__ dobs_act,
    float* __restrict__ dobs_x4s, const float* __restrict__ gimg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // Workgroups beyond the agent tiles (only launched while the rollout leaves CUs idle): the observation LSTM of
  // the discriminator's first pass of this step - it does not depend on the generator - with the rows disc_bwd needs
  if (blockIdx.x * SW_TILE >= (unsigned)B) {
    const swp::Disc O = swp::disc(Tp);
    disc_obs_lstm_tile(smem, obsv, To, 0, dobs_w + O.wih, dobs_w + O.whh, dobs_w + O.bih, dobs_w + O.bhh, B,
                       (int)(blockIdx.x * SW_TILE) - ((B + SW_TILE - 1) / SW_TILE) * SW_TILE, dobs_act, dobs_x4s);
    return;
  }
  constexpr int LD128 = FwdLds::LD128, LD32 = FwdLds::LD32, LD16 = FwdLds::LD16;
  float* hbuf = smem + FwdLds::hbuf;
  float* a1buf = smem + FwdLds::a1buf;
  float* p1 = smem + FwdLds::p1;
  float* a2buf = smem + FwdLds::a2buf;
  float* q2 = smem + FwdLds::q2;
  float* szbuf = a1buf;            // prologue alias [16][100]
  float* wx_lds = a2buf;           // prologue alias (1024)
  float* bx_lds = a2buf + 1024;    // prologue alias (256)
  float* w43_lds = a2buf + 1280;   // prologue alias: [2][80] | b43[2] when no image buffer is registered

  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16;
  const int a0 = blockIdx.x * SW_TILE;
  const int b = min(a0 + ln, B - 1);
  const bool live = (a0 + ln) < B;
  const GSave gs = gsave_layout(B, To, Tp);
  // this wave's share of the layers (see above)
  const int m1a = 32 * wave, m1b = m1a + 16;              // layer-1 row tiles it owns outright
  const int hf = wave & 1, t1p = wave >> 1;               // ... and K-half hf of tile 8 + t1p (rows 128 + 16 t1p ..)
  const int m2 = 16 * wave;                               // layer-2 row tile; of tile 4 the k-steps J0 ..
  const int J0 = wave < 2 ? 3 * wave : 2 + 2 * wave;      // {0,1,2} {3,4,5} {6,7} {8,9}

  // ---- prologue: every weight into registers, straight from global memory in A-operand layout ------------------
  // ALL global loads of the prologue are issued before anything waits on one of them: one L2 round trip (bounded by
  // the ~200 KB a workgroup pulls in), not one per matrix.
#ifdef SW_PHASE_STAMPS
  long long _tprev = clock64();
#endif
  LstmW W;
  f32x4 w1a[4], w1b[4], w1p[2], w2f[10], w2p[3];
  f32x4 wu[3][6], ua, ub, up;    // W1[:, 64:160] rows of this wave's layer-1 tiles: operands of u (below), prologue only
  const int m1p = 128 + 16 * t1p;
  if (gimg) {
    // operand-layout images of this step (swimg::OP_*): the float4 of (row tile, k-step, lane) is one contiguous
    // 16-byte piece, so a wave's load instruction reads 1 KB of consecutive memory.  The row-per-lane loads of the
    // fallback below touch 64 cache lines per instruction - 95 of them kept the four waves' address units busy for
    // ~6.5 us (cycle stamps)
    auto op = [&](int base, int KJ, int tile, int j) { return ld4(gimg + base + (((size_t)tile * KJ + j) * 64 + lane) * 4); };
    lstm_load_img(W, gimg, wave, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w1a[j] = op(swimg::OP_W1H, 4, 2 * wave, j);
      w1b[j] = op(swimg::OP_W1H, 4, 2 * wave + 1, j);
    }
    w1p[0] = op(swimg::OP_W1H, 4, 8 + t1p, 2 * hf);
    w1p[1] = op(swimg::OP_W1H, 4, 8 + t1p, 2 * hf + 1);
#pragma unroll
    for (int j = 0; j < 10; ++j) w2f[j] = op(swimg::OP_W2, 10, wave, j);
    w2p[0] = op(swimg::OP_W2, 10, 4, J0);
    w2p[1] = op(swimg::OP_W2, 10, 4, J0 + 1);
    w2p[2] = op(swimg::OP_W2, 10, 4, J0 + (wave < 2 ? 2 : 1));   // waves 2, 3 have two k-steps: the third set is unused
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      wu[0][j] = op(swimg::OP_W1SZ, 6, 2 * wave, j);
      wu[1][j] = op(swimg::OP_W1SZ, 6, 2 * wave + 1, j);
      wu[2][j] = op(swimg::OP_W1SZ, 6, 8 + t1p, j);
    }
  } else {
    lstm_load_whh(W, enc_w + swp::ENC_WHH, u0, ln, lg);
    const float* ra = dec_w + swp::DEC_W1 + (size_t)(m1a + ln) * 160 + 4 * lg;
    const float* rp = dec_w + swp::DEC_W1 + (size_t)(m1p + ln) * 160 + 4 * lg;
    const float* r2 = dec_w + swp::DEC_W2 + (size_t)(m2 + ln) * 160 + 4 * lg;
    const float* rq = dec_w + swp::DEC_W2 + (size_t)(64 + ln) * 160 + 16 * J0 + 4 * lg;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w1a[j] = ld4(ra + 16 * j);
      w1b[j] = ld4(ra + 16 * 160 + 16 * j);
    }
    w1p[0] = ld4(rp + 32 * hf);
    w1p[1] = ld4(rp + 32 * hf + 16);
#pragma unroll
    for (int j = 0; j < 10; ++j) w2f[j] = ld4(r2 + 16 * j);
    w2p[0] = ld4(rq);
    w2p[1] = ld4(rq + 16);
    w2p[2] = ld4(rq + (wave < 2 ? 32 : 16));
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      wu[0][j] = ld4(ra + 64 + 16 * j);
      wu[1][j] = ld4(ra + 16 * 160 + 64 + 16 * j);
      wu[2][j] = ld4(rp + 64 + 16 * j);
    }
  }
  ua = ld4(dec_w + swp::DEC_B1 + m1a + 4 * lg);
  ub = ld4(dec_w + swp::DEC_B1 + m1b + 4 * lg);
  up = ld4(dec_w + swp::DEC_B1 + m1p + 4 * lg);
  const f32x4 b2f = ld4(dec_w + swp::DEC_B2 + m2 + 4 * lg), b2p = ld4(dec_w + swp::DEC_B2 + 64 + 4 * lg);
  f32x4 c = ld4(cT + (size_t)b * 64 + u0 + 4 * lg);
  f32x4 h = ld4(hT + (size_t)b * 64 + u0 + 4 * lg);
  // running position of agent ln (every lane keeps a copy)
  float px = obsv[((size_t)b * To + To - 1) * 2 + 0];
  float py = obsv[((size_t)b * To + To - 1) * 2 + 1];
  // [S | z] tile: both sources are read unconditionally from clamped addresses and selected afterwards - a load under a
  // lane-dependent branch makes the compiler wait for EVERYTHING in flight right behind it (six serial round trips here)
  float szs[6], szz[6];
  {
    const float* sp = S_pool ? S_pool : z;     // no social block: any readable address, the value is discarded
    const int sld = S_pool ? 64 : 32;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int i = threadIdx.x + q * SW_THREADS, a = i / 96, cc = i - a * 96;
      const int bb = min(a0 + a, B - 1);
      szs[q] = sp[(size_t)bb * sld + min(cc, sld - 1)];
      szz[q] = z[(size_t)bb * 32 + max(cc - 64, 0)];
    }
  }
  // fc3 (80 -> 40) has NO activation in front of fc4 (40 -> 2) (train.py:327-330): the two are ONE 2 x 80 map
  //   v = W4 (W3 a2 + b3) + b4 = W43 a2 + b43
  // (one layer less on the serial chain of every decode step; a3 itself is never needed: its weight gradients are
  // recovered from dv^T [a2 | 1], sw_misc.hip).  W43 and the composed LSTM input matrix come from the image buffer of
  // this step when one is registered (swimg: straight into operand registers), else they are derived here through LDS
  // with the very same arithmetic.
  // fc4 . fc3 is 2 x 80: on the matrix cores it is 20 MFMAs per wave with 2 live rows of 16; on the VALU the lane that
  // holds a2[agent][16 j + 4 lg + r] (its B operand of a matrix product) multiplies it with the two weights of that
  // column - 40 FMAs - and the four lane groups of an agent meet in two shuffles: ~400 cycles per decode step less
  f32x4 w43[2][5];      // [output c][j] = W43[c][16 j + 4 lg .. + 3]
  float b43i[2] = {0.f, 0.f};
  if (gimg) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      w43[0][j] = ld4(gimg + swimg::W43 + 16 * j + 4 * lg);
      w43[1][j] = ld4(gimg + swimg::W43 + 80 + 16 * j + 4 * lg);
    }
    b43i[0] = gimg[swimg::W43 + 160];
    b43i[1] = gimg[swimg::W43 + 161];
  } else {
    lstm_prep_rows(enc_w + swp::ENC_EMB_W, enc_w + swp::ENC_EMB_B, enc_w + swp::ENC_WIH, enc_w + swp::ENC_BIH,
                   enc_w + swp::ENC_BHH, true, wx_lds, bx_lds);
    const int t = threadIdx.x;
    if (t < 160) {
      const int cc = t / 80, k = t - cc * 80;
      const float* w4 = dec_w + swp::DEC_W4 + cc * 40;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll
      for (int m = 0; m < 40; m += 4) {
        v0 = fmaf(w4[m], dec_w[swp::DEC_W3 + m * 80 + k], v0);
        v1 = fmaf(w4[m + 1], dec_w[swp::DEC_W3 + (m + 1) * 80 + k], v1);
        v2 = fmaf(w4[m + 2], dec_w[swp::DEC_W3 + (m + 2) * 80 + k], v2);
        v3 = fmaf(w4[m + 3], dec_w[swp::DEC_W3 + (m + 3) * 80 + k], v3);
      }
      w43_lds[t] = (v0 + v1) + (v2 + v3);
    } else if (t < 162) {
      const int cc = t - 160;
      float v = dec_w[swp::DEC_B4 + cc];
#pragma unroll
      for (int m = 0; m < 40; m += 4) {
        const f32x4 w = ld4(dec_w + swp::DEC_W4 + cc * 40 + m), bb = ld4(dec_w + swp::DEC_B3 + m);
        v = fmaf(w[0], bb[0], fmaf(w[1], bb[1], fmaf(w[2], bb[2], fmaf(w[3], bb[3], v))));
      }
      w43_lds[t] = v;
    }
  }
  SW_STAMP(13);
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int i = threadIdx.x + q * SW_THREADS, a = i / 96, cc = i - a * 96;
    szbuf[a * LD96 + cc] = cc < 64 ? (S_pool ? szs[q] : 0.f) : szz[q];
  }
  st4(&hbuf[ln * SW_ALD + 320 + u0 + 4 * lg], h);
  sw_barrier();
  SW_STAMP(14);
  if (!gimg) {
    lstm_load_wx(W, wx_lds, bx_lds, u0, ln, lg);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      w43[0][j] = ld4(w43_lds + 16 * j + 4 * lg);
      w43[1][j] = ld4(w43_lds + 80 + 16 * j + 4 * lg);
    }
    b43i[0] = w43_lds[160];
    b43i[1] = w43_lds[161];
  }
  // u = W1[:, 64:160] [S; z] + b1 is constant over the steps (train.py:411,421): it is the initial accumulator of
  // this wave's layer-1 tiles (K-half 0 carries it for the split tiles)
  {
    f32x4 bz[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) bz[j] = ld4(&szbuf[ln * LD96 + 16 * j + 4 * lg]);
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ua = SW_MFMA(wu[0][j][r], bz[j][r], ua);
        ub = SW_MFMA(wu[1][j][r], bz[j][r], ub);
        up = SW_MFMA(wu[2][j][r], bz[j][r], up);
      }
    if (hf != 0) up = f32x4{0.f, 0.f, 0.f, 0.f};   // K-half 1 of the split tile starts from zero
  }
  SW_STAMP(15);
  sw_barrier();  // the prologue aliases are dead from here on
  SW_STAMP(8);

  // displacement-error sums of this tile (train.py:546-551), lanes lg == 0 of wave 0
  float e_sum = 0.f, e_last = 0.f, e_sq = 0.f;
  const bool ade_lane = ADE && wave == 0 && lg == 0 && live;
  int cur = 0;
  // everything the prologue requested is waited for HERE, in front of the loop (see the note above the kernel)
#pragma unroll
  for (int j = 0; j < 5; ++j) asm volatile("" : "+v"(w43[0][j]), "+v"(w43[1][j]));
  asm volatile("" : "+v"(b43i[0]), "+v"(b43i[1]), "+v"(px), "+v"(py), "+v"(c), "+v"(h));
  using T_ = std::true_type;
  using F_ = std::false_type;
  auto step = [&](int i, auto last_) {
    constexpr bool LAST = decltype(last_)::value;
    float2 gti = {0.f, 0.f};
    if constexpr (ADE) gti = *reinterpret_cast<const float2*>(gt + ((size_t)b * Tp + i) * 2);   // in flight under the layers
    const float* hrow = &hbuf[cur * 16 * SW_ALD + ln * SW_ALD + 320 + 4 * lg];
    // ---- layer 1: z1 = W1h h + u ; a1 = lrelu(z1) --------------------------------------------------
    {
      f32x4 bh[4], bp[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) bh[j] = ld4(hrow + 16 * j);
      bp[0] = ld4(hrow + 32 * hf);
      bp[1] = ld4(hrow + 32 * hf + 16);
      f32x4 acc_a = ua, acc_b = ub, acc_p = up;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc_a = SW_MFMA(w1a[j][r], bh[j][r], acc_a);
          acc_b = SW_MFMA(w1b[j][r], bh[j][r], acc_b);
          if (j < 2) acc_p = SW_MFMA(w1p[j][r], bp[j][r], acc_p);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc_a[r] = sw_lrelu(acc_a[r]);
        acc_b[r] = sw_lrelu(acc_b[r]);
      }
      st4(&a1buf[ln * LD128 + m1a + 4 * lg], acc_a);
      st4(&a1buf[ln * LD128 + m1b + 4 * lg], acc_b);
      st4(&p1[hf * 16 * LD32 + ln * LD32 + 16 * t1p + 4 * lg], acc_p);
      if constexpr (SAVE) {
        float* row = gsave + gs.a1 + ((size_t)i * B + b) * 160 + 4 * lg;
        st4g(row + m1a, acc_a);
        st4g(row + m1b, acc_b);
      }
    }
    sw_barrier();
    SW_STAMP(9);
    // ---- layer 2: a2 = lrelu(W2 a1 + b2) ----------------------------------------------------------
    {
      f32x4 b1[10];
#pragma unroll
      for (int j = 0; j < 8; ++j) b1[j] = ld4(&a1buf[ln * LD128 + 16 * j + 4 * lg]);
#pragma unroll
      for (int j = 8; j < 10; ++j) {     // the K-split tiles of layer 1: sum of the halves, then the LeakyReLU
        const f32x4 s = ld4(&p1[ln * LD32 + 16 * (j - 8) + 4 * lg]) + ld4(&p1[16 * LD32 + ln * LD32 + 16 * (j - 8) + 4 * lg]);
#pragma unroll
        for (int r = 0; r < 4; ++r) b1[j][r] = sw_lrelu(s[r]);
      }
      if constexpr (SAVE)    // ... whose rows of the save buffer every wave writes (even waves tile 8, odd waves tile 9)
        st4g(gsave + gs.a1 + ((size_t)i * B + b) * 160 + 128 + 16 * (wave & 1) + 4 * lg, (wave & 1) ? b1[9] : b1[8]);
      f32x4 acc = b2f, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        acc = SW_MFMA(w2f[j][0], b1[j][0], acc);
        acc1 = SW_MFMA(w2f[j][1], b1[j][1], acc1);
        acc = SW_MFMA(w2f[j][2], b1[j][2], acc);
        acc1 = SW_MFMA(w2f[j][3], b1[j][3], acc1);
      }
      f32x4 accq;
      if (wave == 0) accq = fwd_l2_part<0, 3>(w2p, b1);
      else if (wave == 1) accq = fwd_l2_part<3, 3>(w2p, b1);
      else if (wave == 2) accq = fwd_l2_part<6, 2>(w2p, b1);
      else accq = fwd_l2_part<8, 2>(w2p, b1);
      acc = acc + acc1;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu(acc[r]);
      st4(&a2buf[ln * LD64 + m2 + 4 * lg], acc);
      st4(&q2[wave * 16 * LD16 + ln * LD16 + 4 * lg], accq);
      if constexpr (SAVE) st4g(gsave + gs.a2 + ((size_t)i * B + b) * 80 + m2 + 4 * lg, acc);
    }
    sw_barrier();
    SW_STAMP(10);
    // ---- layers 3+4 composed (v = W43 a2 + b43 ; p += v) and the re-fed encoder step (train.py:422-430) ----
    // Every wave computes the 2-row map itself and keeps its own copy of the running position, so the LSTM step needs
    // no barrier / LDS hop for its input.
    {
      f32x4 b2v[5];
#pragma unroll
      for (int j = 0; j < 4; ++j) b2v[j] = ld4(&a2buf[ln * LD64 + 16 * j + 4 * lg]);
      {
        const float* q = &q2[ln * LD16 + 4 * lg];
        const f32x4 s = ((b2p + ld4(q)) + ld4(q + 16 * LD16)) + (ld4(q + 2 * 16 * LD16) + ld4(q + 3 * 16 * LD16));
#pragma unroll
        for (int r = 0; r < 4; ++r) b2v[4][r] = sw_lrelu(s[r]);
      }
      if constexpr (SAVE) st4g(gsave + gs.a2 + ((size_t)i * B + b) * 80 + 64 + 4 * lg, b2v[4]);   // every wave holds it
      float vx0 = 0.f, vx1 = 0.f, vy0 = 0.f, vy1 = 0.f;      // this lane's 20 columns, two chains per output
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        vx0 = fmaf(w43[0][j][0], b2v[j][0], vx0);
        vx1 = fmaf(w43[0][j][1], b2v[j][1], vx1);
        vx0 = fmaf(w43[0][j][2], b2v[j][2], vx0);
        vx1 = fmaf(w43[0][j][3], b2v[j][3], vx1);
        vy0 = fmaf(w43[1][j][0], b2v[j][0], vy0);
        vy1 = fmaf(w43[1][j][1], b2v[j][1], vy1);
        vy0 = fmaf(w43[1][j][2], b2v[j][2], vy0);
        vy1 = fmaf(w43[1][j][3], b2v[j][3], vy1);
      }
      float vx = vx0 + vx1, vy = vy0 + vy1;
      vx += __shfl_xor(vx, 16);      // the four lane groups (k quarters) of agent ln: every lane ends with the sum
      vy += __shfl_xor(vy, 16);
      vx += __shfl_xor(vx, 32);
      vy += __shfl_xor(vy, 32);
      vx += b43i[0];
      vy += b43i[1];
      px += vx;
      py += vy;
      SW_STAMP(11);
      if constexpr (ADE) {
        const float dx = (px - gti.x) * inv_ss, dy = (py - gti.y) * inv_ss;
        const float q = dx * dx + dy * dy;
        const float e = sqrtf(q);
        if (ade_lane) {      // arithmetic only: no memory operation under this branch
          e_sum += e;
          e_sq += q;
          if (LAST) e_last = e;
        }
      }
      {   // every lane of agent ln holds the same (p, v): all of them store it (no lane-dependent store)
        const f32x4 x4 = {px, py, vx, vy};
        st4(pred4 + ((size_t)b * Tp + i) * 4, x4);
        if constexpr (SAVE && !LAST) st4(gsave + gs.x4s + ((size_t)(To + i) * B + b) * 4, x4);
      }
      auto lstm_step = [&](auto save_) {
        const float xb = lg == 0 ? px : (lg == 1 ? py : (lg == 2 ? vx : vy));
        f32x4 gate[4];
        lstm_cell(W, xb, hrow, gate, c, h);
        if constexpr (decltype(save_)::value) lstm_put_act_tile(&hbuf[(cur ^ 1) * 16 * SW_ALD], gate, c, h, ln, lg, u0);
        else st4(&hbuf[(cur ^ 1) * 16 * SW_ALD + ln * SW_ALD + 320 + u0 + 4 * lg], h);
        cur ^= 1;
      };
      if constexpr (!LAST) {
        if constexpr (SAVE) lstm_step(T_{});
        else lstm_step(F_{});
      } else {
        if (h_end) lstm_step(F_{});     // the step after the last decode is dead compute (train.py:430) unless the state is wanted
      }
      sw_barrier();
      if constexpr (SAVE && !LAST)      // the saved row of LSTM step To + i, from the tile just completed
        lstm_store_act_tile(&hbuf[cur * 16 * SW_ALD], gsave + gs.act + (size_t)(To + i) * B * 384, a0, B, wave, lane);
      SW_STAMP(12);
    }
  };
  for (int i = 0; i < Tp - 1; ++i) step(i, F_{});
  step(Tp - 1, T_{});
  if (h_end && live) {
    st4(h_end + (size_t)b * 64 + u0 + 4 * lg, h);
    if (c_end) st4(c_end + (size_t)b * 64 + u0 + 4 * lg, c);
  }
  if (ADE && wave == 0) {   // fixed shuffle tree over the tile's 16 agents -> one partial triple per workgroup
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      e_sum += __shfl_xor(e_sum, o);
      e_last += __shfl_xor(e_last, o);
      e_sq += __shfl_xor(e_sq, o);
    }
    if (lane == 0) {
      ade_part[(size_t)blockIdx.x * 3 + 0] = e_sum / (float)Tp;
      ade_part[(size_t)blockIdx.x * 3 + 1] = e_last;
      ade_part[(size_t)blockIdx.x * 3 + 2] = e_sq;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same kernel for TWO 16-agent column blocks per workgroup (round 6): for batches with more tiles than CUs (dense
// crowds, large shards).  At 394 registers the kernel above runs one workgroup per CU, so with eight tiles queued per CU
// nothing fills a tile's barrier / LDS turn-arounds (4 per decode step, ~0.9 K cycles each of a 10.9 K-cycle step).  Here
// every register-resident A operand (weight) is issued against two B operands - the blocks' activation tiles - so the
// four turn-arounds, the prologue and the weight loads are paid once per 32 agents, and each wave carries two independent
// chains.  Every agent's arithmetic is the 16-agent kernel's, operation for operation (same accumulator order): results
// are bit-identical.  Workgroup v owns the tiles 2v, 2v + 1; a second block beyond the batch is a replica of agent B - 1
// (every load is clamped) and stores the same values to the same rows.  No observation-LSTM riders (only launched when
// CUs are idle).
template <bool SAVE, bool ADE>
__global__ __launch_bounds__(SW_THREADS) void dec_rollout_fwd2_kernel(
    const float* __restrict__ obsv, int To, const float* __restrict__ z, const float* __restrict__ S_pool,
    const float* __restrict__ hT, const float* __restrict__ cT, const float* __restrict__ enc_w,
    const float* __restrict__ dec_w, int B, int Tp, float* __restrict__ pred4, float* __restrict__ h_end,
    float* __restrict__ c_end, float* __restrict__ gsave, const float* __restrict__ gt, float inv_ss,
    float* __restrict__ ade_part, const float* __restrict__ gimg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD128 = FwdLds::LD128, LD32 = FwdLds::LD32, LD16 = FwdLds::LD16;
  constexpr int NB = 2;
  float *hbuf[NB], *a1buf[NB], *p1[NB], *a2buf[NB], *q2[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    float* base = smem + k * FwdLds::total;
    hbuf[k] = base + FwdLds::hbuf;
    a1buf[k] = base + FwdLds::a1buf;
    p1[k] = base + FwdLds::p1;
    a2buf[k] = base + FwdLds::a2buf;
    q2[k] = base + FwdLds::q2;
  }
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16;
  const int tiles16 = (B + SW_TILE - 1) / SW_TILE;
  int a0[NB], b[NB];
  bool live[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    a0[k] = (2 * (int)blockIdx.x + k) * SW_TILE;
    b[k] = min(a0[k] + ln, B - 1);
    live[k] = (a0[k] + ln) < B;
  }
  const GSave gs = gsave_layout(B, To, Tp);
  const int m1a = 32 * wave, m1b = m1a + 16;
  const int hf = wave & 1, t1p = wave >> 1;
  const int m2 = 16 * wave;
  const int J0 = wave < 2 ? 3 * wave : 2 + 2 * wave;
  const int m1p = 128 + 16 * t1p;

  // ---- prologue: the weights once (operand-layout images of the step: this kernel is only launched with them) ----
  LstmW W;
  f32x4 w1a[4], w1b[4], w1p[2], w2f[10], w2p[3];
  f32x4 wu[3][6];
  f32x4 ua[NB], ub[NB], up[NB];
  auto op = [&](int base, int KJ, int tile, int j) { return ld4(gimg + base + (((size_t)tile * KJ + j) * 64 + lane) * 4); };
  lstm_load_img(W, gimg, wave, lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    w1a[j] = op(swimg::OP_W1H, 4, 2 * wave, j);
    w1b[j] = op(swimg::OP_W1H, 4, 2 * wave + 1, j);
  }
  w1p[0] = op(swimg::OP_W1H, 4, 8 + t1p, 2 * hf);
  w1p[1] = op(swimg::OP_W1H, 4, 8 + t1p, 2 * hf + 1);
#pragma unroll
  for (int j = 0; j < 10; ++j) w2f[j] = op(swimg::OP_W2, 10, wave, j);
  w2p[0] = op(swimg::OP_W2, 10, 4, J0);
  w2p[1] = op(swimg::OP_W2, 10, 4, J0 + 1);
  w2p[2] = op(swimg::OP_W2, 10, 4, J0 + (wave < 2 ? 2 : 1));
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    wu[0][j] = op(swimg::OP_W1SZ, 6, 2 * wave, j);
    wu[1][j] = op(swimg::OP_W1SZ, 6, 2 * wave + 1, j);
    wu[2][j] = op(swimg::OP_W1SZ, 6, 8 + t1p, j);
  }
  const f32x4 ua0 = ld4(dec_w + swp::DEC_B1 + m1a + 4 * lg);
  const f32x4 ub0 = ld4(dec_w + swp::DEC_B1 + m1b + 4 * lg);
  const f32x4 up0 = ld4(dec_w + swp::DEC_B1 + m1p + 4 * lg);
  const f32x4 b2f = ld4(dec_w + swp::DEC_B2 + m2 + 4 * lg), b2p = ld4(dec_w + swp::DEC_B2 + 64 + 4 * lg);
  f32x4 c[NB], h[NB];
  float px[NB], py[NB];
  float szs[NB][6], szz[NB][6];
  {
    const float* sp = S_pool ? S_pool : z;
    const int sld = S_pool ? 64 : 32;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      c[k] = ld4(cT + (size_t)b[k] * 64 + u0 + 4 * lg);
      h[k] = ld4(hT + (size_t)b[k] * 64 + u0 + 4 * lg);
      px[k] = obsv[((size_t)b[k] * To + To - 1) * 2 + 0];
      py[k] = obsv[((size_t)b[k] * To + To - 1) * 2 + 1];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int i = threadIdx.x + q * SW_THREADS, a = i / 96, cc = i - a * 96;
        const int bb = min(a0[k] + a, B - 1);
        szs[k][q] = sp[(size_t)bb * sld + min(cc, sld - 1)];
        szz[k][q] = z[(size_t)bb * 32 + max(cc - 64, 0)];
      }
    }
  }
  f32x4 w43[2][5];
  float b43i[2];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    w43[0][j] = ld4(gimg + swimg::W43 + 16 * j + 4 * lg);
    w43[1][j] = ld4(gimg + swimg::W43 + 80 + 16 * j + 4 * lg);
  }
  b43i[0] = gimg[swimg::W43 + 160];
  b43i[1] = gimg[swimg::W43 + 161];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    float* szbuf = a1buf[k];      // prologue alias [16][100]
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int i = threadIdx.x + q * SW_THREADS, a = i / 96, cc = i - a * 96;
      szbuf[a * LD96 + cc] = cc < 64 ? (S_pool ? szs[k][q] : 0.f) : szz[k][q];
    }
    st4(&hbuf[k][ln * SW_ALD + 320 + u0 + 4 * lg], h[k]);
  }
  sw_barrier();
  // u = W1[:, 64:160] [S; z] + b1: the initial accumulators of this wave's layer-1 tiles, per block
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    f32x4 bz[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) bz[j] = ld4(&a1buf[k][ln * LD96 + 16 * j + 4 * lg]);
    ua[k] = ua0;
    ub[k] = ub0;
    up[k] = up0;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ua[k] = SW_MFMA(wu[0][j][r], bz[j][r], ua[k]);
        ub[k] = SW_MFMA(wu[1][j][r], bz[j][r], ub[k]);
        up[k] = SW_MFMA(wu[2][j][r], bz[j][r], up[k]);
      }
    if (hf != 0) up[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  sw_barrier();  // the prologue aliases are dead from here on

  float e_sum[NB] = {0.f, 0.f}, e_last[NB] = {0.f, 0.f}, e_sq[NB] = {0.f, 0.f};
  int cur = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) asm volatile("" : "+v"(w43[0][j]), "+v"(w43[1][j]));
  asm volatile("" : "+v"(b43i[0]), "+v"(b43i[1]));
#pragma unroll
  for (int k = 0; k < NB; ++k) asm volatile("" : "+v"(px[k]), "+v"(py[k]), "+v"(c[k]), "+v"(h[k]));
  using T_ = std::true_type;
  using F_ = std::false_type;
  auto step = [&](int i, auto last_) {
    constexpr bool LAST = decltype(last_)::value;
    float2 gti[NB];
    const float* hrow[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      gti[k] = float2{0.f, 0.f};
      if constexpr (ADE) gti[k] = *reinterpret_cast<const float2*>(gt + ((size_t)b[k] * Tp + i) * 2);
      hrow[k] = &hbuf[k][cur * 16 * SW_ALD + ln * SW_ALD + 320 + 4 * lg];
    }
    // ---- layer 1: z1 = W1h h + u ; a1 = lrelu(z1) --------------------------------------------------
    {
      f32x4 bh[NB][4], bp[NB][2];
      f32x4 acc_a[NB], acc_b[NB], acc_p[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bh[k][j] = ld4(hrow[k] + 16 * j);
        bp[k][0] = ld4(hrow[k] + 32 * hf);
        bp[k][1] = ld4(hrow[k] + 32 * hf + 16);
        acc_a[k] = ua[k];
        acc_b[k] = ub[k];
        acc_p[k] = up[k];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int k = 0; k < NB; ++k) {
            acc_a[k] = SW_MFMA(w1a[j][r], bh[k][j][r], acc_a[k]);
            acc_b[k] = SW_MFMA(w1b[j][r], bh[k][j][r], acc_b[k]);
            if (j < 2) acc_p[k] = SW_MFMA(w1p[j][r], bp[k][j][r], acc_p[k]);
          }
        }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc_a[k][r] = sw_lrelu(acc_a[k][r]);
          acc_b[k][r] = sw_lrelu(acc_b[k][r]);
        }
        st4(&a1buf[k][ln * LD128 + m1a + 4 * lg], acc_a[k]);
        st4(&a1buf[k][ln * LD128 + m1b + 4 * lg], acc_b[k]);
        st4(&p1[k][hf * 16 * LD32 + ln * LD32 + 16 * t1p + 4 * lg], acc_p[k]);
        if constexpr (SAVE) {
          float* row = gsave + gs.a1 + ((size_t)i * B + b[k]) * 160 + 4 * lg;
          st4g(row + m1a, acc_a[k]);
          st4g(row + m1b, acc_b[k]);
        }
      }
    }
    sw_barrier();
    // ---- layer 2: a2 = lrelu(W2 a1 + b2) ----------------------------------------------------------
    {
      f32x4 b1[NB][10];
      f32x4 acc[NB], acc1[NB], accq[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) b1[k][j] = ld4(&a1buf[k][ln * LD128 + 16 * j + 4 * lg]);
#pragma unroll
        for (int j = 8; j < 10; ++j) {
          const f32x4 s = ld4(&p1[k][ln * LD32 + 16 * (j - 8) + 4 * lg]) + ld4(&p1[k][16 * LD32 + ln * LD32 + 16 * (j - 8) + 4 * lg]);
#pragma unroll
          for (int r = 0; r < 4; ++r) b1[k][j][r] = sw_lrelu(s[r]);
        }
        if constexpr (SAVE)
          st4g(gsave + gs.a1 + ((size_t)i * B + b[k]) * 160 + 128 + 16 * (wave & 1) + 4 * lg, (wave & 1) ? b1[k][9] : b1[k][8]);
        acc[k] = b2f;
        acc1[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 10; ++j) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          acc[k] = SW_MFMA(w2f[j][0], b1[k][j][0], acc[k]);
          acc1[k] = SW_MFMA(w2f[j][1], b1[k][j][1], acc1[k]);
          acc[k] = SW_MFMA(w2f[j][2], b1[k][j][2], acc[k]);
          acc1[k] = SW_MFMA(w2f[j][3], b1[k][j][3], acc1[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        if (wave == 0) accq[k] = fwd_l2_part<0, 3>(w2p, b1[k]);
        else if (wave == 1) accq[k] = fwd_l2_part<3, 3>(w2p, b1[k]);
        else if (wave == 2) accq[k] = fwd_l2_part<6, 2>(w2p, b1[k]);
        else accq[k] = fwd_l2_part<8, 2>(w2p, b1[k]);
        acc[k] = acc[k] + acc1[k];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[k][r] = sw_lrelu(acc[k][r]);
        st4(&a2buf[k][ln * LD64 + m2 + 4 * lg], acc[k]);
        st4(&q2[k][wave * 16 * LD16 + ln * LD16 + 4 * lg], accq[k]);
        if constexpr (SAVE) st4g(gsave + gs.a2 + ((size_t)i * B + b[k]) * 80 + m2 + 4 * lg, acc[k]);
      }
    }
    sw_barrier();
    // ---- layers 3+4 composed (v = W43 a2 + b43 ; p += v) and the re-fed encoder step (train.py:422-430) ----
    {
      float vx[NB], vy[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        f32x4 b2v[5];
#pragma unroll
        for (int j = 0; j < 4; ++j) b2v[j] = ld4(&a2buf[k][ln * LD64 + 16 * j + 4 * lg]);
        {
          const float* q = &q2[k][ln * LD16 + 4 * lg];
          const f32x4 s = ((b2p + ld4(q)) + ld4(q + 16 * LD16)) + (ld4(q + 2 * 16 * LD16) + ld4(q + 3 * 16 * LD16));
#pragma unroll
          for (int r = 0; r < 4; ++r) b2v[4][r] = sw_lrelu(s[r]);
        }
        if constexpr (SAVE) st4g(gsave + gs.a2 + ((size_t)i * B + b[k]) * 80 + 64 + 4 * lg, b2v[4]);
        float vx0 = 0.f, vx1 = 0.f, vy0 = 0.f, vy1 = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          vx0 = fmaf(w43[0][j][0], b2v[j][0], vx0);
          vx1 = fmaf(w43[0][j][1], b2v[j][1], vx1);
          vx0 = fmaf(w43[0][j][2], b2v[j][2], vx0);
          vx1 = fmaf(w43[0][j][3], b2v[j][3], vx1);
          vy0 = fmaf(w43[1][j][0], b2v[j][0], vy0);
          vy1 = fmaf(w43[1][j][1], b2v[j][1], vy1);
          vy0 = fmaf(w43[1][j][2], b2v[j][2], vy0);
          vy1 = fmaf(w43[1][j][3], b2v[j][3], vy1);
        }
        float x = vx0 + vx1, y = vy0 + vy1;
        x += __shfl_xor(x, 16);
        y += __shfl_xor(y, 16);
        x += __shfl_xor(x, 32);
        y += __shfl_xor(y, 32);
        x += b43i[0];
        y += b43i[1];
        vx[k] = x;
        vy[k] = y;
        px[k] += x;
        py[k] += y;
        if constexpr (ADE) {
          const float dx = (px[k] - gti[k].x) * inv_ss, dy = (py[k] - gti[k].y) * inv_ss;
          const float q = dx * dx + dy * dy;
          const float e = sqrtf(q);
          if (wave == 0 && lg == 0 && live[k]) {      // arithmetic only: no memory operation under this branch
            e_sum[k] += e;
            e_sq[k] += q;
            if (LAST) e_last[k] = e;
          }
        }
        {
          const f32x4 x4 = {px[k], py[k], x, y};
          st4(pred4 + ((size_t)b[k] * Tp + i) * 4, x4);
          if constexpr (SAVE && !LAST) st4(gsave + gs.x4s + ((size_t)(To + i) * B + b[k]) * 4, x4);
        }
      }
      auto lstm_step = [&](auto save_) {
        float xb[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) xb[k] = lg == 0 ? px[k] : (lg == 1 ? py[k] : (lg == 2 ? vx[k] : vy[k]));
        // the cell of both blocks: every W_hh operand against the two h tiles (lstm_cell, sw_lstm_dev.h, per block)
        f32x4 acc[NB][4], bb[NB][4];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
#pragma unroll
          for (int j = 0; j < 4; ++j) bb[k][j] = ld4(hrow[k] + 16 * j);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[k][g] = SW_MFMA(W.wx[g], xb[k], W.bias[g]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int k = 0; k < NB; ++k) acc[k][g] = SW_MFMA(W.whh[g][j][r], bb[k][j][r], acc[k][g]);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          f32x4 gate[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float ig = sw_sigmoid(acc[k][0][r]);
            const float fg = sw_sigmoid(acc[k][1][r]);
            const float gg = sw_tanh(acc[k][2][r]);
            const float og = sw_sigmoid(acc[k][3][r]);
            const float cn = fmaf(fg, c[k][r], ig * gg);
            gate[0][r] = ig;
            gate[1][r] = fg;
            gate[2][r] = gg;
            gate[3][r] = og;
            c[k][r] = cn;
            h[k][r] = og * sw_tanh(cn);
          }
          if constexpr (decltype(save_)::value) lstm_put_act_tile(&hbuf[k][(cur ^ 1) * 16 * SW_ALD], gate, c[k], h[k], ln, lg, u0);
          else st4(&hbuf[k][(cur ^ 1) * 16 * SW_ALD + ln * SW_ALD + 320 + u0 + 4 * lg], h[k]);
        }
        cur ^= 1;
      };
      if constexpr (!LAST) {
        if constexpr (SAVE) lstm_step(T_{});
        else lstm_step(F_{});
      } else {
        if (h_end) lstm_step(F_{});
      }
      sw_barrier();
      if constexpr (SAVE && !LAST) {
#pragma unroll
        for (int k = 0; k < NB; ++k)
          lstm_store_act_tile(&hbuf[k][cur * 16 * SW_ALD], gsave + gs.act + (size_t)(To + i) * B * 384, a0[k], B, wave, lane);
      }
    }
  };
  for (int i = 0; i < Tp - 1; ++i) step(i, F_{});
  step(Tp - 1, T_{});
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    if (h_end && live[k]) {
      st4(h_end + (size_t)b[k] * 64 + u0 + 4 * lg, h[k]);
      if (c_end) st4(c_end + (size_t)b[k] * 64 + u0 + 4 * lg, c[k]);
    }
    if (ADE && wave == 0) {   // fixed shuffle tree over the block's 16 agents -> one partial triple per 16-agent tile
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        e_sum[k] += __shfl_xor(e_sum[k], o);
        e_last[k] += __shfl_xor(e_last[k], o);
        e_sq[k] += __shfl_xor(e_sq[k], o);
      }
      const int t16 = 2 * (int)blockIdx.x + k;
      if (lane == 0 && t16 < tiles16) {
        ade_part[(size_t)t16 * 3 + 0] = e_sum[k] / (float)Tp;
        ade_part[(size_t)t16 * 3 + 1] = e_last[k];
        ade_part[(size_t)t16 * 3 + 2] = e_sq[k];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward of the decode loop.  Propagates data gradients only; every weight gradient is a
// deferred GEMM over the time-major delta / activation rows written here (sw_wgrad.hip).
//
// Like the forward kernel every weight (transposed) lives in registers as MFMA A operands - W_hh^T, Wx^T, W2^T,
// W1h^T, (fc4 . fc3)^T: ~180 registers per lane - loaded from the operand-layout images of the step, and LDS holds
// delta tiles only.  Work per wave and decode step: dh_prev = W_hh^T dgates (64 MFMAs) + its K-quarter of dx4 (16);
// dz2 (2 tiles x 2); dz1 = W2^T dz2 with row tiles 2w, 2w+1 outright and K-part (3 or 2 of 5 k-steps) of tile
// 8 + (w >> 1) (52 / 48; was 60 with tile 9 repeated by the waves that would have idled); dh += W1h^T dz1 (40).  The
// consumer of dz1 adds the partial sums of the split tiles and applies LeakyReLU' while loading its B operand.
// ---------------------------------------------------------------------------------------------
namespace {
struct BwdLds {
  static constexpr int LD128 = sw_ld(128);  // 132
  static constexpr int LD16 = sw_ld(16);    // 20
  static constexpr int dgbuf = 0;                         // [16][260]  (prologue without images: wx | bx | W43; epilogue: du [16][164])
  static constexpr int dz1buf = dgbuf + 16 * SW_GLD;      // [16][132]  dz1[:, :128]
  static constexpr int pz1 = dz1buf + 16 * LD128;         // [4][16][20] K-parts of (W2^T dz2)[:, 128:160]: slot 2 (tile - 8) + part
  static constexpr int dz2buf = pz1 + 4 * 16 * LD16;      // [16][84]
  static constexpr int dxpart = dz2buf + 16 * LD80;       // [4 waves][16][4]
  static constexpr int total = dxpart + 256;
};
static_assert(16 * SW_GLD >= 16 * LD160 && 16 * SW_GLD >= 1280 + 176, "aliases of dgbuf");
}  // namespace

// DFUSE: the generator-phase discriminator pass of the tile (D forward on the tile's prediction + the backward of its
// prediction heads down to d(g_loss)/d(pred_hat), train.py:510-523) runs IN FRONT of the tile's decode BPTT, in the same
// workgroup: the pass is tile-local and this BPTT is its only consumer - one graph node (~4 us of boundary) less per step.
struct DecDiscFuse {
  const float* obsv;      // [B][To][2]
  const float* pred_hat;  // [B][Tp][4]
  const float* d_w;       // packed D weights (after its last update of the step)
  const float* dimg;      // D's weight images or null
  DiscLoss gl;
  float* dpred;           // [B][Tp][4] scratch: d(g_loss)/d(pred_hat), written and re-read by the same workgroup
};
template <bool DFUSE>
__global__ __launch_bounds__(SW_THREADS) void dec_rollout_bwd_kernel(
    const float* __restrict__ dpred4, const float* __restrict__ enc_w, const float* __restrict__ dec_w,
    const float* __restrict__ gsave, int B, int To, int Tp, float* __restrict__ gdelta,
    float* __restrict__ dhT, float* __restrict__ dcT, float* __restrict__ dS_pool, const float* __restrict__ aux_src,
    float* __restrict__ aux_dst, const float* __restrict__ aux_mask, long long aux_n, const float* __restrict__ gimg,
    DecDiscFuse df) {
  // Workgroups beyond the agent tiles run an auxiliary masked copy dst[i] = mask[i] > 0 ? src[i] : dst[i]
  // (the training step's D.load(backup), train.py:541-542, on CUs this latency-bound launch leaves idle)
  {
    const int tiles = (B + SW_TILE - 1) / SW_TILE;
    if ((int)blockIdx.x >= tiles) {
      const long long stride = (long long)(gridDim.x - tiles) * SW_THREADS;
      for (long long i = (long long)(blockIdx.x - tiles) * SW_THREADS + threadIdx.x; i < aux_n; i += stride)
        if (aux_mask[i] > 0.f) aux_dst[i] = aux_src[i];
      return;
    }
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if constexpr (DFUSE) {
    disc_fwd_tile(smem, blockIdx.x, gridDim.x, df.obsv, To, 0, df.pred_hat, nullptr, 1, df.d_w, B, Tp, nullptr, nullptr, nullptr,
                  nullptr, nullptr, 0, 0, nullptr, 1, df.gl, df.dpred, df.dimg);
    __syncthreads();      // the tile's d/d(pred) rows are in memory (this workgroup wrote them) and the LDS is free again
  }
  constexpr int LD128 = BwdLds::LD128, LD16 = BwdLds::LD16;
  float* dgbuf = smem + BwdLds::dgbuf;
  float* dz1buf = smem + BwdLds::dz1buf;
  float* pz1 = smem + BwdLds::pz1;
  float* dz2buf = smem + BwdLds::dz2buf;
  float* dxpart = smem + BwdLds::dxpart;
  float* wx_lds = dgbuf;            // prologue alias (1024), only without an image buffer
  float* bx_lds = dgbuf + 1024;     // prologue alias (256)
  float* w43_lds = dgbuf + 1280;    // prologue alias [2][80]

  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16;
  const int a0 = blockIdx.x * SW_TILE;
  const int b = min(a0 + ln, B - 1);
  const bool live = (a0 + ln) < B;
  const GSave gs = gsave_layout(B, To, Tp);
  const GDelta gd = gdelta_layout(B, To, Tp);
  const int hf = wave & 1, t1p = wave >> 1;     // of dz1's split tile 8 + t1p this wave sums k-steps {0,1,2} (hf 0) or {3,4}
  const int m2q[2] = {min(wave, 4) * 16, 64};   // its dz2 row tiles: wave and 4 (every wave repeats tile 4: same values, same places)

  // ---- prologue: every (transposed) weight into registers ------------------------------------------------------
#ifdef SW_PHASE_STAMPS
  long long _tprev = clock64();
#endif
  LstmWT WT;
  f32x4 wxT[4], w2t[2][5], w2p[3], w1t[10];
  f32x4 w43c[2][2];     // (fc4 . fc3) columns of this wave's two dz2 tiles in C layout: [tile][c][r] = W43[c][m0 + 4 lg + r]
  if (gimg) {
    auto op = [&](int base, int KJ, int tile, int j) { return ld4(gimg + base + (((size_t)tile * KJ + j) * 64 + lane) * 4); };
#pragma unroll
    for (int j = 0; j < 16; ++j) WT.whhT[j] = op(swimg::OP_WHHT, 16, wave, j);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      w2t[0][j] = op(swimg::OP_W2T, 5, 2 * wave, j);
      w2t[1][j] = op(swimg::OP_W2T, 5, 2 * wave + 1, j);
    }
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) w2p[jj] = op(swimg::OP_W2T, 5, 8 + t1p, hf ? 3 + min(jj, 1) : jj);
#pragma unroll
    for (int j = 0; j < 10; ++j) w1t[j] = op(swimg::OP_W1HT, 10, wave, j);
    // Wx^T slice of this wave's K-quarter as A operands: row c = ln (< 4 live), k = 64 wave + 16 j + 4 lg + r
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) wxT[j][r] = gimg[swimg::WX + (64 * wave + 16 * j + 4 * lg + r) * 4 + (ln & 3)];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      w43c[q][0] = ld4(gimg + swimg::W43 + m2q[q] + 4 * lg);
      w43c[q][1] = ld4(gimg + swimg::W43 + 80 + m2q[q] + 4 * lg);
    }
  } else {
    // no image buffer registered (stand-alone module calls): strided loads from the weights themselves, the composed
    // maps derived here with the arithmetic of sw_gen_images
    lstm_load_wT(WT, enc_w + swp::ENC_WHH, u0, ln, lg);
    lstm_prep_rows(enc_w + swp::ENC_EMB_W, enc_w + swp::ENC_EMB_B, enc_w + swp::ENC_WIH, enc_w + swp::ENC_BIH,
                   enc_w + swp::ENC_BHH, true, wx_lds, bx_lds);
    {
      const int t = threadIdx.x;
      if (t < 160) {
        const int cc = t / 80, k = t - cc * 80;
        const float* w4 = dec_w + swp::DEC_W4 + cc * 40;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll
        for (int m = 0; m < 40; m += 4) {
          v0 = fmaf(w4[m], dec_w[swp::DEC_W3 + m * 80 + k], v0);
          v1 = fmaf(w4[m + 1], dec_w[swp::DEC_W3 + (m + 1) * 80 + k], v1);
          v2 = fmaf(w4[m + 2], dec_w[swp::DEC_W3 + (m + 2) * 80 + k], v2);
          v3 = fmaf(w4[m + 3], dec_w[swp::DEC_W3 + (m + 3) * 80 + k], v3);
        }
        w43_lds[t] = (v0 + v1) + (v2 + v3);
      }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* col = dec_w + swp::DEC_W2 + (size_t)(16 * j + 4 * lg + r) * 160;
        w2t[0][j][r] = col[32 * wave + ln];
        w2t[1][j][r] = col[32 * wave + 16 + ln];
      }
#pragma unroll
    for (int jj = 0; jj < 3; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        w2p[jj][r] = dec_w[swp::DEC_W2 + (size_t)(16 * (hf ? 3 + min(jj, 1) : jj) + 4 * lg + r) * 160 + 128 + 16 * t1p + ln];
#pragma unroll
    for (int j = 0; j < 10; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) w1t[j][r] = dec_w[swp::DEC_W1 + (size_t)(16 * j + 4 * lg + r) * 160 + u0 + ln];
    sw_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) wxT[j][r] = wx_lds[(64 * wave + 16 * j + 4 * lg + r) * 4 + (ln & 3)];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      w43c[q][0] = ld4(w43_lds + m2q[q] + 4 * lg);
      w43c[q][1] = ld4(w43_lds + 80 + m2q[q] + 4 * lg);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ln >= 4) wxT[j][r] = 0.f;
  sw_barrier();   // the prologue aliases of dgbuf are dead
  SW_STAMP(5);
  SW_STAMP(6);
  SW_STAMP(2);

  f32x4 dh = {0.f, 0.f, 0.f, 0.f}, dc = {0.f, 0.f, 0.f, 0.f};
  f32x4 du_a = {0.f, 0.f, 0.f, 0.f}, du_b = du_a, du_s[2] = {du_a, du_a};   // sum over the steps of dz1: own tiles | split tiles 8, 9
  // per-agent running gradient w.r.t. the position (every lane keeps a copy)
  float dpx = 0.f, dpy = 0.f;

  // Everything an iteration reads from HBM/L2 (saved LSTM rows, saved a1 / a2 tiles, the upstream gradient) is
  // fetched ONE ITERATION AHEAD, and the loop body has no conditional memory operation: clamped tile indices
  // instead of `mt < n ? load : 0`, the first / last iteration peeled instead of `if (i < Tp - 1)`, stores of the
  // padding lanes of the last tile kept (exact replicas of agent B-1: every load is clamped to it), values that
  // several waves hold stored by all of them.  With conditional loads or stores the compiler cannot count what is in
  // flight and waits for EVERYTHING (s_waitcnt vmcnt(0)) where a loaded value is first used.
  using T_ = std::true_type;
  using F_ = std::false_type;
  struct Rows {
    f32x4 gate[4], ct, cprev;     // LSTM step To + i (consumed x4_i)
    f32x4 a2[2], a1[2], a1s[2], g4;   // decoder step i: a2 tiles wave, 4; a1 tiles 2w, 2w+1; a1[:, 128:160] (every wave)
  };
  const float* act_b = gsave + gs.act + ((size_t)To * B + b) * 384 + u0 + 4 * lg;
  const float* a2_b = gsave + gs.a2 + (size_t)b * 80 + 4 * lg;
  const float* a1_b = gsave + gs.a1 + (size_t)b * 160 + 4 * lg;
  auto load_rows = [&](int i, Rows& R, auto lstm) {
    if constexpr (decltype(lstm)::value) {
      const float* row = act_b + (size_t)i * B * 384;
#pragma unroll
      for (int g = 0; g < 4; ++g) R.gate[g] = ld4(row + g * 64);
      R.ct = ld4(row + 256);
      R.cprev = ld4(row - (size_t)B * 384 + 256);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      R.a2[q] = ld4(a2_b + (size_t)i * B * 80 + m2q[q]);
      R.a1[q] = ld4(a1_b + (size_t)i * B * 160 + 32 * wave + 16 * q);
      R.a1s[q] = ld4(a1_b + (size_t)i * B * 160 + 128 + 16 * q);
    }
    // every wave keeps its own copy of the p/v gradient state.  DFUSE: the rows were written by THIS workgroup through df.dpred
    // (disc_fwd_tile above) - they are read back through the same unqualified pointer, never through the __restrict__
    // const parameter that aliases it (the compiler may treat such loads as invariant across the barrier)
    if constexpr (DFUSE) R.g4 = ld4(df.dpred + ((size_t)b * Tp + i) * 4);
    else R.g4 = ld4(dpred4 + ((size_t)b * Tp + i) * 4);
  };
  // ROLLING prefetch (round 6): every group of saved rows is re-requested for step i - 1 right behind its last use in
  // step i, into the SAME registers.  Rounds 1-5 fetched the whole row set of step i - 1 at the head of step i into a
  // second set (52 registers: 490 in all, 234 of them AGPRs the vector ALU cannot read without a copy); in place the kernel
  // needs 451 and the step is 10 % shorter (82.9 -> 75 us at m1, c4 -1.8 % per step; profiles/r06_bwd_roll_ab.txt)
  auto roll_lstm = [&](int i, Rows& Q) {
    const float* row = act_b + (size_t)i * B * 384;
#pragma unroll
    for (int g = 0; g < 4; ++g) Q.gate[g] = ld4(row + g * 64);
    Q.ct = ld4(row + 256);
    Q.cprev = ld4(row - (size_t)B * 384 + 256);
    asm volatile("" ::: "memory");
  };
  auto roll_g4 = [&](int i, Rows& Q) {
    if constexpr (DFUSE) Q.g4 = ld4(df.dpred + ((size_t)b * Tp + i) * 4);
    else Q.g4 = ld4(dpred4 + ((size_t)b * Tp + i) * 4);
    asm volatile("" ::: "memory");
  };
  auto roll_a2 = [&](int i, Rows& Q) {
#pragma unroll
    for (int q = 0; q < 2; ++q) Q.a2[q] = ld4(a2_b + (size_t)i * B * 80 + m2q[q]);
    asm volatile("" ::: "memory");
  };
  auto roll_a1 = [&](int i, Rows& Q) {
#pragma unroll
    for (int q = 0; q < 2; ++q) Q.a1[q] = ld4(a1_b + (size_t)i * B * 160 + 32 * wave + 16 * q);
    asm volatile("" ::: "memory");
  };
  auto roll_a1s = [&](int i, Rows& Q) {
#pragma unroll
    for (int q = 0; q < 2; ++q) Q.a1s[q] = ld4(a1_b + (size_t)i * B * 160 + 128 + 16 * q);
    asm volatile("" ::: "memory");
  };
  Rows R;
  load_rows(Tp - 1, R, F_{});
  // one decode step backwards; lstm: the step has an LSTM step behind it (all but i = Tp-1); pf: prefetch i-1
  auto step = [&](int i, auto lstm, auto pf) {
    SW_STAMP(7);
    constexpr bool PF = decltype(pf)::value;      // the rows of step i - 1 are prefetched (all steps but i = 0)
    if constexpr (PF && !decltype(lstm)::value) roll_lstm(i - 1, R);      // (step Tp - 1 has no LSTM step of its own)
    f32x4 dx4 = {0.f, 0.f, 0.f, 0.f};  // gradient through the LSTM input (p_i, v_i)
    if constexpr (decltype(lstm)::value) {
      // ---- LSTM step t = To+i (consumed x4_i, produced h_t) --------------------------------
      const int t = To + i;
      f32x4 dgate[4];
      lstm_cell_bwd(R.gate, R.ct, R.cprev, dh, dc, dgate);
      if constexpr (PF) roll_lstm(i - 1, R);
#pragma unroll
      for (int g = 0; g < 4; ++g) st4(&dgbuf[ln * SW_GLD + g * 64 + u0 + 4 * lg], dgate[g]);
      sw_barrier();
      lstm_store_dgates_tile(dgbuf, gdelta + gd.dgates + ((size_t)t * B + a0) * 256, nullptr, a0, B, wave, lane);
      SW_STAMP(0);
      dh = lstm_dh_prev(WT, &dgbuf[ln * SW_GLD + 4 * lg]);
      // dx4 = Wx^T dgates: each wave reduces its own quarter of K, partials through LDS
      {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_mm_reg<4>(wxT, &dgbuf[ln * SW_GLD + 64 * wave + 4 * lg], acc);
        if (lg == 0) st4(&dxpart[(wave * 16 + ln) * 4], acc);
      }
      sw_barrier();
      SW_STAMP(1);
      dx4 = ld4(&dxpart[ln * 4]) + ld4(&dxpart[(16 + ln) * 4]) + ld4(&dxpart[(32 + ln) * 4]) +
            ld4(&dxpart[(48 + ln) * 4]);
    }
    // ---- decoder step i: dv (registers, every wave) ------------------------------------------------
    const f32x4 g4 = R.g4;
    if constexpr (PF) roll_g4(i - 1, R);
    dpx += g4[0] + dx4[0];  // dL/dp_i  (p_i also feeds p_{i+1}: carried in dpx)
    dpy += g4[1] + dx4[1];
    const float dvx = g4[2] + dx4[2] + dpx;  // p_i = p_{i-1} + v_i
    const float dvy = g4[3] + dx4[3] + dpy;
    {   // every lane of agent ln holds the same pair: all of them store it (no lane-dependent store)
      f32x4 o = {dvx, dvy, 0.f, 0.f};
      st4(gdelta + gd.dv + ((size_t)i * B + b) * 4, o);
    }
    // dz2 = (W43^T dv) * lrelu'(a2)   (80): fc3 and fc4 are one linear map, so d(a2) comes straight from dv - a rank-2
    // product: two FMAs per element on the VALU (every lane knows its agent's dv)
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      f32x4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = fmaf(w43c[q2][0][r], dvx, w43c[q2][1][r] * dvy);
      const f32x4 a2 = R.a2[q2];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = sw_lrelu_grad(a2[r], acc[r]);
      st4(&dz2buf[ln * LD80 + m2q[q2] + 4 * lg], acc);
      st4g(gdelta + gd.dz2 + ((size_t)i * B + b) * 80 + m2q[q2] + 4 * lg, acc);
    }
    if constexpr (PF) roll_a2(i - 1, R);
    sw_barrier();
    SW_STAMP(3);
    // dz1 = (W2^T dz2) * lrelu'(a1)   (160): row tiles 2w, 2w+1 and this wave's K-part of the split tile
    {
      f32x4 b2[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) b2[j] = ld4(&dz2buf[ln * LD80 + 16 * j + 4 * lg]);
      f32x4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = acc_a, acc_p = acc_a;
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc_a = SW_MFMA(w2t[0][j][r], b2[j][r], acc_a);
          acc_b = SW_MFMA(w2t[1][j][r], b2[j][r], acc_b);
        }
      if (hf == 0) {
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc_p = SW_MFMA(w2p[jj][r], b2[jj][r], acc_p);
      } else {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc_p = SW_MFMA(w2p[jj][r], b2[3 + jj][r], acc_p);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc_a[r] = sw_lrelu_grad(R.a1[0][r], acc_a[r]);
        acc_b[r] = sw_lrelu_grad(R.a1[1][r], acc_b[r]);
      }
      st4(&dz1buf[ln * LD128 + 32 * wave + 4 * lg], acc_a);
      st4(&dz1buf[ln * LD128 + 32 * wave + 16 + 4 * lg], acc_b);
      st4(&pz1[(2 * t1p + hf) * 16 * LD16 + ln * LD16 + 4 * lg], acc_p);
      float* row = gdelta + gd.dz1 + ((size_t)i * B + b) * 160 + 32 * wave + 4 * lg;
      st4g(row, acc_a);
      st4g(row + 16, acc_b);
      du_a += acc_a;
      du_b += acc_b;
    }
    if constexpr (PF) roll_a1(i - 1, R);
    sw_barrier();
    SW_STAMP(4);
    // dh_{To+i-1} += W1h^T dz1   (wave w owns units 16w.. : same layout as dh)
    {
      f32x4 b1[10];
#pragma unroll
      for (int j = 0; j < 8; ++j) b1[j] = ld4(&dz1buf[ln * LD128 + 16 * j + 4 * lg]);
#pragma unroll
      for (int q = 0; q < 2; ++q) {   // the split tiles: sum of the K-parts, then LeakyReLU' - every wave, and every wave
        const float* pp = &pz1[2 * q * 16 * LD16 + ln * LD16 + 4 * lg];   // stores them (the same values to the same places)
        f32x4 v = ld4(pp) + ld4(pp + 16 * LD16);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = sw_lrelu_grad(R.a1s[q][r], v[r]);
        b1[8 + q] = v;
        st4g(gdelta + gd.dz1 + ((size_t)i * B + b) * 160 + 128 + 16 * q + 4 * lg, v);
        du_s[q] += v;
      }
      if constexpr (PF) roll_a1s(i - 1, R);
      f32x4 acc = decltype(lstm)::value ? dh : f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        acc = SW_MFMA(w1t[j][0], b1[j][0], acc);
        acc1 = SW_MFMA(w1t[j][1], b1[j][1], acc1);
        acc = SW_MFMA(w1t[j][2], b1[j][2], acc);
        acc1 = SW_MFMA(w1t[j][3], b1[j][3], acc1);
      }
      dh = acc + acc1;
    }
    // (next iteration's first LDS writes are to dgbuf, whose readers are behind barriers)
  };
  if (Tp > 1) {
    step(Tp - 1, F_{}, T_{});
    for (int i = Tp - 2; i >= 1; --i) step(i, T_{}, T_{});
    step(0, T_{}, F_{});
  } else {
    step(0, F_{}, F_{});
  }
  // ---- epilogue: state gradients, du and dS = W1[:,64:128]^T du ---------------------------------
  if (live) {
    st4(dhT + (size_t)b * 64 + u0 + 4 * lg, dh);
    st4(dcT + (size_t)b * 64 + u0 + 4 * lg, dc);
  }
  sw_barrier();
  float* dub = dgbuf;   // [16][164]
  {
    st4(&dub[ln * LD160 + 32 * wave + 4 * lg], du_a);
    st4(&dub[ln * LD160 + 32 * wave + 16 + 4 * lg], du_b);
    st4(&dub[ln * LD160 + 128 + 4 * lg], du_s[0]);      // every wave holds the split tiles' sums
    st4(&dub[ln * LD160 + 144 + 4 * lg], du_s[1]);
    if (live) {
      float* row = gdelta + gd.du + (size_t)b * 160 + 4 * lg;
      st4(row + 32 * wave, du_a);
      st4(row + 32 * wave + 16, du_b);
      st4(row + 128, du_s[0]);
      st4(row + 144, du_s[1]);
    }
  }
  sw_barrier();
  if (dS_pool) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* w1 = dec_w + swp::DEC_W1 + 64 + u0 + ln;  // column 64+unit of fc1.0.weight
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      f32x4 bb = ld4(&dub[ln * LD160 + 16 * j + 4 * lg]);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = SW_MFMA(w1[(size_t)(16 * j + 4 * lg + r) * 160], bb[r], acc);
    }
    if (live) st4(dS_pool + (size_t)b * 64 + u0 + 4 * lg, acc);
  }
}

static int set_lds(const void* fn, int bytes) {
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    sw_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
    return SW_EHIP;
  }
  return SW_OK;
}

extern "C" int sw_dec_rollout_fwd_aux(const float* obsv, int To, const float* z, const float* S_pool,
                                      const float* hT, const float* cT, const float* enc_w, const float* dec_w,
                                      int B, int Tp, float* pred4, float* h_end, float* c_end, float* gsave,
                                      const float* gt, float inv_ss, float* ade_part, const float* d_w, float* dsave,
                                      void* stream) {
  if (!obsv || !z || !hT || !cT || !enc_w || !dec_w || !pred4 || B < 0 || To < 2 || Tp < 1) return SW_EARG;
  if (ade_part && !gt) return SW_EARG;
  if ((d_w != nullptr) != (dsave != nullptr)) return SW_EARG;
  if (B == 0) return SW_OK;
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  // rows of the D observation LSTM inside the save buffer of sw_disc_fwd (independent of its branch count)
  float* act = dsave;
  float* x4s = dsave ? dsave + (size_t)To * B * 384 : nullptr;
  const float* gimg = sw_gen_images_for(enc_w, dec_w);
  // More tiles than CUs (dense crowds, large shards): two 16-agent column blocks per workgroup - every resident weight
  // operand issued against both, the step's barrier / LDS turn-arounds paid once per 32 agents (dec_rollout_fwd2_kernel,
  // bit-identical).  Needs the step's weight images; SW_DEC_FWD2=0 / 1 forces either kernel (A/B runs, tests).
  static const int fwd2_env = getenv("SW_DEC_FWD2") ? atoi(getenv("SW_DEC_FWD2")) : -1;
  if (gimg && !d_w && (fwd2_env >= 0 ? fwd2_env != 0 : tiles > 256)) {
#define SW_DEC_FWD2(SV, AD)                                                                                          \
  do {                                                                                                               \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      if (int rc = set_lds((const void*)dec_rollout_fwd2_kernel<SV, AD>, 2 * FwdLds::total * 4)) return rc;          \
      attr = true;                                                                                                   \
    }                                                                                                                \
    SW_LAUNCH((dec_rollout_fwd2_kernel<SV, AD>), dim3((tiles + 1) / 2), dim3(SW_THREADS), 2 * FwdLds::total * 4,     \
              (hipStream_t)stream, obsv, To, z, S_pool, hT, cT, enc_w, dec_w, B, Tp, pred4, h_end, c_end, gsave, gt,  \
              inv_ss, ade_part, gimg);                                                                               \
  } while (0)
    if (gsave) {
      if (ade_part) SW_DEC_FWD2(true, true);
      else SW_DEC_FWD2(true, false);
    } else {
      if (ade_part) SW_DEC_FWD2(false, true);
      else SW_DEC_FWD2(false, false);
    }
#undef SW_DEC_FWD2
    SW_CHECK_LAUNCH("dec_rollout_fwd2_kernel");
    return SW_OK;
  }
#define SW_DEC_FWD(SV, AD)                                                                                           \
  do {                                                                                                               \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      if (int rc = set_lds((const void*)dec_rollout_fwd_kernel<SV, AD>, FwdLds::total * 4)) return rc;               \
      attr = true;                                                                                                   \
    }                                                                                                                \
    SW_LAUNCH((dec_rollout_fwd_kernel<SV, AD>), dim3(d_w ? 2 * tiles : tiles), dim3(SW_THREADS), FwdLds::total * 4,  \
              (hipStream_t)stream, obsv, To, z, S_pool, hT, cT, enc_w, dec_w, B, Tp, pred4, h_end, c_end, gsave, gt,  \
              inv_ss, ade_part, d_w, act, x4s, gimg);                                                                \
  } while (0)
  if (gsave) {
    if (ade_part) SW_DEC_FWD(true, true);
    else SW_DEC_FWD(true, false);
  } else {
    if (ade_part) SW_DEC_FWD(false, true);
    else SW_DEC_FWD(false, false);
  }
#undef SW_DEC_FWD
  SW_CHECK_LAUNCH("dec_rollout_fwd_kernel");
  return SW_OK;
}

extern "C" int sw_dec_rollout_fwd(const float* obsv, int To, const float* z, const float* S_pool,
                                  const float* hT, const float* cT, const float* enc_w, const float* dec_w,
                                  int B, int Tp, float* pred4, float* h_end, float* c_end, float* gsave,
                                  const float* gt, float inv_ss, float* ade_part, void* stream) {
  return sw_dec_rollout_fwd_aux(obsv, To, z, S_pool, hT, cT, enc_w, dec_w, B, Tp, pred4, h_end, c_end, gsave, gt, inv_ss,
                                ade_part, nullptr, nullptr, stream);
}

static int set_lds_bwd(const void* fn, int bytes, int& have) {
  if (have >= bytes) return SW_OK;
  if (int rc = set_lds(fn, bytes)) return rc;
  have = bytes;
  return SW_OK;
}
extern "C" int sw_dec_rollout_bwd_aux(const float* dpred4, const float* enc_w, const float* dec_w,
                                      const float* gsave, int B, int To, int Tp, float* gdelta, float* dhT,
                                      float* dcT, float* dS_pool, const float* aux_src, float* aux_dst,
                                      const float* aux_mask, long long aux_n, void* stream) {
  if (!dpred4 || !enc_w || !dec_w || !gsave || !gdelta || !dhT || !dcT || B < 0 || To < 2 || Tp < 1)
    return SW_EARG;
  if (aux_n < 0 || (aux_n > 0 && (!aux_src || !aux_dst || !aux_mask))) return SW_EARG;
  if (B == 0) return SW_OK;
  static int have = 0;
  if (int rc = set_lds_bwd((const void*)dec_rollout_bwd_kernel<false>, BwdLds::total * 4, have)) return rc;
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  int extra = aux_n > 0 ? (int)((aux_n + SW_THREADS - 1) / SW_THREADS) : 0;
  if (extra > 64) extra = 64;
  SW_LAUNCH(dec_rollout_bwd_kernel<false>, dim3(tiles + extra), dim3(SW_THREADS), BwdLds::total * 4, (hipStream_t)stream,
                     dpred4, enc_w, dec_w, gsave, B, To, Tp, gdelta, dhT, dcT, dS_pool, aux_src, aux_dst, aux_mask, aux_n,
                     sw_gen_images_for(enc_w, dec_w), DecDiscFuse{});
  SW_CHECK_LAUNCH("dec_rollout_bwd_kernel");
  return SW_OK;
}

// sw_disc_dpred + sw_dec_rollout_bwd in ONE launch (generator phase of a training step, train.py:510-538): every
// workgroup first runs D's forward on its tile's prediction and the backward of the prediction heads (as sw_disc_dpred:
// loss gradients of train.py:512-523 formed in the kernel, per-tile loss sums to loss_part), then the decode BPTT of that
// tile from the d/d(pred) rows it has just written to `dpred4` (scratch).  Same results as the two calls, bit for bit.
extern "C" int sw_dec_rollout_bwd_dfuse(const float* obsv, const float* pred4, const float* d_w, const float* targets, int t_idx,
                                        const float* z, float g_label, float g_code, float* loss_part, float* dpred4,
                                        const float* enc_w, const float* dec_w, const float* gsave, int B, int To, int Tp,
                                        float* gdelta, float* dhT, float* dcT, float* dS_pool, void* stream) {
  if (!obsv || !pred4 || !d_w || !targets || !z || t_idx < 0 || !dpred4 || !enc_w || !dec_w || !gsave || !gdelta || !dhT || !dcT ||
      B < 0 || To < 2 || Tp < 1)
    return SW_EARG;
  if (Tp > 64) return SW_ESHAPE;
  if (B == 0) return SW_OK;
  int lds = head_lds_b(Tp, head_lds(Tp, 2 * 16 * SW_HLD + 1280).total).total * 4;
  if (lds < BwdLds::total * 4) lds = BwdLds::total * 4;
  if (lds > 163840) return SW_ESHAPE;
  static int have = 0;
  if (int rc = set_lds_bwd((const void*)dec_rollout_bwd_kernel<true>, lds, have)) return rc;
  DecDiscFuse df;
  df.obsv = obsv; df.pred_hat = pred4; df.d_w = d_w; df.dimg = sw_disc_images_for(d_w, Tp).img;
  df.gl = DiscLoss{targets, z, t_idx, t_idx, g_label, g_code, 1, loss_part};
  df.dpred = dpred4;
  const int tiles = (B + SW_TILE - 1) / SW_TILE;
  SW_LAUNCH(dec_rollout_bwd_kernel<true>, dim3(tiles), dim3(SW_THREADS), lds, (hipStream_t)stream, dpred4, enc_w, dec_w, gsave, B, To,
            Tp, gdelta, dhT, dcT, dS_pool, (const float*)nullptr, (float*)nullptr, (const float*)nullptr, 0LL,
            sw_gen_images_for(enc_w, dec_w), df);
  SW_CHECK_LAUNCH("dec_rollout_bwd_kernel");
  return SW_OK;
}

extern "C" int sw_dec_rollout_bwd(const float* dpred4, const float* enc_w, const float* dec_w,
                                  const float* gsave, int B, int To, int Tp, float* gdelta, float* dhT,
                                  float* dcT, float* dS_pool, void* stream) {
  return sw_dec_rollout_bwd_aux(dpred4, enc_w, dec_w, gsave, B, To, Tp, gdelta, dhT, dcT, dS_pool, nullptr, nullptr,
                                nullptr, 0, stream);
}
