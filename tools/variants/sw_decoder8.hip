// sw_decoder8.hip - the decode loop of predict() (train.py:415-432) on 8-agent tiles (sw_n8.h): the narrow
// counterparts of dec_rollout_fwd_kernel / dec_rollout_bwd_kernel of sw_decoder.hip.  Same inputs, same saved rows
// (time-major layouts of sw_common.h), results equal up to summation order.
//
// A workgroup = 8 agents = two groups of 4 (ag), 4 waves: wave w -> agent group ag = w & 1, half uh = w >> 1.
// Per decode step:
//   layer 1   a1 = lrelu(W1h h + u)        160 rows, K = 64   wave (ag, uh): rows [64 uh, 64 uh + 64) and the 16 rows
//                                                              [128 + 16 uh, ..) on 16 live lanes
//   layer 2   a2 = lrelu(W2 a1 + b2)        80 rows, K = 160  wave (ag, 0): rows [0, 64); wave (ag, 1): rows [64, 80)
//   layers 3+4 (composed, sw_decoder.hip)   2 rows, K = 80    every wave for its own agents: 5 instructions with the 16
//                                                              blocks working on 16 different k (cbsz = 0), summed over
//                                                              the blocks with DPP row rotations and two lane swaps
//   LSTM cell on (p, v)                                        sw_n8.h
// Decoder weights stay in LDS (one ROW per lane: strides == 4 (mod 8) words make the row-wise 16-byte reads conflict
// free), W_hh in registers; activations travel between the layers through small LDS tiles written in the order the
// broadcast A operand reads them (sw8_pos*), one barrier per layer.
#include "../../include/socialways_hip.h"
#include "sw_lstm_dev.h"
#include "sw_n8.h"

namespace {
constexpr int LD160 = sw_ld(160);   // 164: weight images, row-per-lane 16-byte reads are conflict free (stride == 4 * odd)
constexpr int HL = 208;             // h tile row: the same 64 values in three operand orders (H4 | H3 | H1), == 16 (mod 64)
constexpr int A1L = 336;            // a1 tile row: T2 order (160) | natural order (160)
constexpr int A2L = 144;            // a2 tile row: 16 blocks x 8 (5 used)
constexpr int UL = 176;             // prologue: u = fc1.0 [S|z] part + bias, [8 agents][160]

// positions of element k inside the tiles (the order each consumer's A operand reads contiguously)
__device__ __forceinline__ int h4_pos(int k) { return ((k & 15) << 2) | (k >> 4); }                    // k = 16 v + blk
__device__ __forceinline__ int h3_pos(int k) { return 64 + ((k & 7) << 3) + (k >> 3); }                 // k = 8 v + a
__device__ __forceinline__ int h1_pos(int k) { return 128 + ((k >> 4) << 4) + ((k & 1) << 3) + ((k & 15) >> 1); }   // k = 16 kp + 2 v + a
__device__ __forceinline__ int a1_pos2(int k) { const int kh = k >= 80, q = k - 80 * kh; return 80 * kh + (q & 3) * 20 + (q >> 2); }   // k = 80 kh + 4 v + a
__device__ __forceinline__ int pos80(int k) { return (k & 15) * 8 + (k >> 4); }                         // k = 16 v + b

struct F8 {   // forward LDS carve (floats)
  static constexpr int img = 0;                            // prologue: one weight image at a time, <= [160][164]
  static constexpr int wx = img + 160 * LD160;             // [256][4] | [256]
  static constexpr int htile = wx + 1280;                  // [2][8][HL]
  static constexpr int a1t = htile + 2 * SW8_TILE * HL;    // [8][A1L]      (prologue: u [8][UL], W43 [2][80] | b43[2])
  static constexpr int a2t = a1t + SW8_TILE * A1L;         // [8][A2L]
  static constexpr int xs = a2t + SW8_TILE * A2L;          // [4 waves][4 agents][4] + dummy words
  static constexpr int w1r = xs + 4 * 16 + 64;             // [32][68]   fc1.0 rows 128..159 (h part): the 8-row remainder jobs
  static constexpr int w2r = w1r + 32 * 68;                // [16][164]  fc1.2 rows 64..79: the 4-row remainder jobs
  static constexpr int total = w2r + 16 * LD160;
};
static_assert(SW8_TILE * UL + 162 <= SW8_TILE * A1L, "prologue alias of the a1 tile");
static_assert(256 * SW8_WLD <= 160 * LD160, "LSTM weight images fit the image area");
static_assert(F8::total * 4 <= 163840, "LDS budget");

// coalesced global -> LDS image of a row-major [R][C] matrix (C % 4 == 0) with row stride ld, NV float4 per thread
template <int NV>
__device__ __forceinline__ void img_load(f32x4 (&v)[NV], const float* __restrict__ src, int n4) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int f = threadIdx.x + SW_THREADS * j;
    v[j] = f < n4 ? ld4(src + 4 * f) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <int NV>
__device__ __forceinline__ void img_store(const f32x4 (&v)[NV], float* dst, int n4, int c4, int ld) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int f = threadIdx.x + SW_THREADS * j, row = f / c4;
    if (f < n4) st4(dst + row * ld + 4 * (f - row * c4), v[j]);
  }
}
// NR floats of a weight row (LDS image) -> registers
template <int NR>
__device__ __forceinline__ void row_regs(float (&w)[NR], const float* row) {
  static_assert(NR % 4 == 0, "");
#pragma unroll
  for (int j = 0; j < NR / 4; ++j) {
    const f32x4 q = ld4(row + 4 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) w[4 * j + e] = q[e];
  }
}
// NV activation registers of the lane (contiguous in the tile)
template <int NV>
__device__ __forceinline__ void act_regs(float (&x)[NV], const float* p) {
  static_assert(NV % 4 == 0, "");
#pragma unroll
  for (int j = 0; j < NV / 4; ++j) {
    const f32x4 q = ld4(p + 4 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[4 * j + e] = q[e];
  }
}
// products with the K dimension laid out as k = NB v + a (a = abid < NB = blocks per broadcast group, cbsz = log2 NB):
// acc += sum_{v, a} A(x[v], block a of the lane's group) * w[NB v + a]; two accumulators alternate
template <int CBSZ, int NV>
__device__ __forceinline__ f32x4 mm_k(f32x4 acc, const float (&x)[NV], const float (&w)[NV << CBSZ]) {
  constexpr int NB = 1 << CBSZ;
  f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    if constexpr (NB == 1) {
      if (v & 1) acc1 = sw_m4<0, 0>(x[v], w[v], acc1);
      else acc = sw_m4<0, 0>(x[v], w[v], acc);
    } else {
      sw_static_for<0, NB / 2>([&](auto ic) {
        constexpr int a = 2 * decltype(ic)::value;
        acc = sw_m4<CBSZ, a>(x[v], w[NB * v + a], acc);
        acc1 = sw_m4<CBSZ, a + 1>(x[v], w[NB * v + a + 1], acc1);
      });
    }
  }
  return acc + acc1;
}
// the 64-row, one-agent-group form used in the prologue: weight row from LDS, k = 16 v + b
template <int NV>
__device__ __forceinline__ f32x4 mm_row_lds(f32x4 acc, const float* wrow, const float (&xv)[NV]) {
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    float w[16];
    row_regs<16>(w, wrow + 16 * v);
    sw_static_for<0, 16>([&](auto ic) {
      constexpr int b = decltype(ic)::value;
      acc = sw_m4<4, b>(xv[v], w[b], acc);
    });
  }
  return acc;
}
}  // namespace

// SAVE: the rows the backward pass needs go to gsave; ADE: displacement-error sums against gt.  Template parameters,
// and the last decode step is peeled, so that the step loop holds no conditional memory operation (see
// dec_rollout_bwd_kernel: the compiler otherwise waits for every outstanding store where a load is first used).
//
// Every weight lives in REGISTERS for the whole kernel (one wave per SIMD: 512 registers per lane).  On 4-agent
// instructions a weight fetched from LDS would feed 4 agents instead of 16: at the MFMA rate the four waves would
// stream 128 B/clk of weights - the LDS limit - so the first version (weights in LDS, like the 16-agent kernel) ran
// LDS-bound at 66.9 us vs 61.5 with only layer 1 in registers.  The register budget is what forces balanced jobs:
// every layer's rows x agent groups x K are cut so that each lane holds (and each wave issues) exactly its share -
//   layer 1 (160 x 64):  64 + 16 = 80      rows [32 w, +32) x 2 agent groups (cbsz 3)  +  rows [128 + 8 w, +8) x 2
//                                           groups x 4 k-parts (cbsz 1, parts summed over lane bits 4, 5)
//   layer 2 (80 x 160):  80 + 20 = 100     rows [16 w, +16) x 2 groups x 2 k-halves (cbsz 2, bit 5)  +  rows
//                                           [64 + 4 w, +4) x 2 groups x 8 k-parts (cbsz 0, bits 3, 4, 5)
//   layers 3+4 (2 x 80):   5               2 rows x 16 k-parts per wave for its own agent group
//   LSTM (256 x 68):     136               sw_n8.h
template <bool SAVE, bool ADE>
__global__ __launch_bounds__(SW_THREADS) void dec_rollout_fwd8_kernel(
    const float* __restrict__ obsv, int To, const float* __restrict__ z, const float* __restrict__ S_pool,
    const float* __restrict__ hT, const float* __restrict__ cT, const float* __restrict__ enc_w,
    const float* __restrict__ dec_w, int B, int Tp, float* __restrict__ pred4, float* __restrict__ h_end,
    float* __restrict__ c_end, float* __restrict__ gsave, const float* __restrict__ gt, float inv_ss,
    float* __restrict__ ade_part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* img = smem + F8::img;
  float* wx_lds = smem + F8::wx;
  float* bx_lds = wx_lds + 1024;
  float* htile = smem + F8::htile;
  float* a1t = smem + F8::a1t;
  float* a2t = smem + F8::a2t;
  float* ubuf = a1t;                              // prologue alias
  float* w43_lds = a1t + SW8_TILE * UL;           // prologue alias: [2][80] | b43[2]
  const Lstm8Lane L;
  const int lane = sw_lane(), wave = sw_wave();
  float* xs = smem + F8::xs + 16 * wave;          // this wave's (p, v) exchange tile [4 agents][4]
  float* xdummy = smem + F8::xs + 64 + lane;      // where the lanes without a live (p, v) entry write
  const int tile0 = blockIdx.x * SW8_TILE;
  const int a0 = tile0 + 4 * L.ag;
  const int bA = min(a0 + L.ai, B - 1);
  int bD[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bD[r] = min(a0 + r, B - 1);
  const GSave gs = gsave_layout(B, To, Tp);
  // the lane's rows / agent groups in the decoder layers
  const int r1m = 32 * wave + (lane & 31), g1m = lane >> 5;                       // layer 1 main
  const int r1r = 128 + 8 * wave + (lane & 7), g1r = (lane >> 3) & 1, kp1 = lane >> 4;   // layer 1 remainder (k-part kp1)
  const int r2m = 16 * wave + (lane & 15), g2m = (lane >> 4) & 1, kh2 = lane >> 5;       // layer 2 main (k-half kh2)
  const int r2r = 64 + 4 * wave + (lane & 3), g2r = (lane >> 2) & 1, kp2 = lane >> 3;    // layer 2 remainder (k-part kp2)

  // ---- prologue: weight images pass through LDS one at a time (coalesced global reads), rows go to registers ------
  Lstm8W W;
  // (the two small remainder jobs read their 16 / 20 weights from LDS every step: the MFMA operand registers that
  // do not fit the 256 accumulation registers compete with everything else for the 256 ordinary ones)
  float w1m[64], w2m[80];
  float* w1r_img = smem + F8::w1r;
  float* w2r_img = smem + F8::w2r;
  f32x4 um, ur;
  {
    f32x4 va[16];
    sw8_stage256_load(va, enc_w + swp::ENC_WIH);
    sw8_stage256_store(va, img);
    sw8_stage256_load(va, enc_w + swp::ENC_WHH);        // in flight under the composition
    sw_barrier();
    lstm8_prep_rows_lds(enc_w + swp::ENC_EMB_W, enc_w + swp::ENC_EMB_B, img, enc_w + swp::ENC_BIH, enc_w + swp::ENC_BHH,
                        wx_lds, bx_lds);
    sw_barrier();
    sw8_stage256_store(va, img);
  }
  f32x4 v1[25];
  img_load<25>(v1, dec_w + swp::DEC_W1, 6400);            // fc1.0.weight (160 x 160), in flight
  sw_barrier();
  lstm8_load_whh(W, img, L, SW8_WLD);
  lstm8_load_wx(W, wx_lds, bx_lds, L);
  sw_barrier();
  img_store<25>(v1, img, 6400, 40, LD160);
  f32x4 v2[13];
  img_load<13>(v2, dec_w + swp::DEC_W2, 3200);            // fc1.2.weight (80 x 160), in flight
  // [S | z] of the lane's agent as broadcast A operand (k = 16 v + blk) for u
  float szv[6];
#pragma unroll
  for (int v = 0; v < 4; ++v) szv[v] = S_pool ? S_pool[(size_t)bA * 64 + 16 * v + L.blk] : 0.f;
#pragma unroll
  for (int v = 0; v < 2; ++v) szv[4 + v] = z[(size_t)bA * 32 + 16 * v + L.blk];
  // composed fc4 . fc3 (2 x 80) + bias (see dec_rollout_fwd_kernel): one output per thread
  if (threadIdx.x < 160) {
    const int cc = threadIdx.x / 80, k = threadIdx.x - cc * 80;
    const float* w4 = dec_w + swp::DEC_W4 + cc * 40;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int m = 0; m < 40; m += 4) {
      s0 = fmaf(w4[m], dec_w[swp::DEC_W3 + m * 80 + k], s0);
      s1 = fmaf(w4[m + 1], dec_w[swp::DEC_W3 + (m + 1) * 80 + k], s1);
      s2 = fmaf(w4[m + 2], dec_w[swp::DEC_W3 + (m + 2) * 80 + k], s2);
      s3 = fmaf(w4[m + 3], dec_w[swp::DEC_W3 + (m + 3) * 80 + k], s3);
    }
    w43_lds[cc * 80 + k] = (s0 + s1) + (s2 + s3);
  } else if (threadIdx.x < 162) {
    const int cc = threadIdx.x - 160;
    float v = dec_w[swp::DEC_B4 + cc];
#pragma unroll
    for (int m = 0; m < 40; ++m) v = fmaf(dec_w[swp::DEC_W4 + cc * 40 + m], dec_w[swp::DEC_B3 + m], v);
    w43_lds[160 + cc] = v;
  }
  // initial state; running position (lane parity = x / y component) of the wave's 4 agents
  f32x4 c, h, pos;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    h[r] = hT[(size_t)bD[r] * 64 + L.u];
    c[r] = cT[(size_t)bD[r] * 64 + L.u];
    pos[r] = obsv[((size_t)bD[r] * To + To - 1) * 2 + (lane & 1)];
  }
  sw_barrier();   // fc1.0 image and W43 complete
  row_regs<64>(w1m, img + r1m * LD160);
#pragma unroll
  for (int j = 0; j < 2; ++j) {   // rows 128..159, columns 0..63 of the image -> their own [32][68] tile
    const int f = threadIdx.x + SW_THREADS * j, row = f >> 4, c4 = f & 15;
    st4(w1r_img + row * 68 + 4 * c4, ld4(img + (128 + row) * LD160 + 4 * c4));
  }
  float w43[5];
#pragma unroll
  for (int v = 0; v < 5; ++v) w43[v] = w43_lds[(lane & 1) * 80 + 16 * v + L.blk];
  const float b43 = w43_lds[160 + (lane & 1)];
  {   // u = W1[:, 64:160] [S; z] + b1 for all 160 rows x 8 agents (64-row one-group jobs), redistributed through LDS
    const int ra = 64 * L.uh + lane, rb = 128 + 16 * L.uh + (lane & 15);
    const float ba = dec_w[swp::DEC_B1 + ra], bb = dec_w[swp::DEC_B1 + rb];
    f32x4 ua = {ba, ba, ba, ba}, ub = {bb, bb, bb, bb};
    ua = mm_row_lds<6>(ua, img + ra * LD160 + 64, szv);
    ub = mm_row_lds<6>(ub, img + rb * LD160 + 64, szv);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ubuf[(4 * L.ag + r) * UL + ra] = ua[r];
      ubuf[(4 * L.ag + r) * UL + rb] = ub[r];
    }
  }
  sw_barrier();   // every wave is done with the fc1.0 image; u complete
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    um[r] = ubuf[(4 * g1m + r) * UL + r1m];
    ur[r] = ubuf[(4 * g1r + r) * UL + r1r];
  }
  img_store<13>(v2, img, 3200, 40, LD160);
  const float b2m = dec_w[swp::DEC_B2 + r2m], b2r = dec_w[swp::DEC_B2 + r2r];
  sw_barrier();   // fc1.2 image complete (and every lane has its u)
  row_regs<80>(w2m, img + r2m * LD160 + 80 * kh2);
#pragma unroll
  for (int j = 0; j < 3; ++j) {   // rows 64..79 of the image -> their own [16][164] tile
    const int f = threadIdx.x + SW_THREADS * j, row = f / 40, c4 = f - row * 40;
    if (f < 640) st4(w2r_img + row * LD160 + 4 * c4, ld4(img + (64 + row) * LD160 + 4 * c4));
  }
  {   // h_0 into the tile (three operand orders)
    const int r0 = L.up ? 2 : 0;
    float* p = htile + (4 * L.ag + r0) * HL;
    const float x0 = L.up ? h[2] : h[0], x1 = L.up ? h[3] : h[1];
    p[h4_pos(L.u)] = x0; p[h3_pos(L.u)] = x0; p[h1_pos(L.u)] = x0;
    p[HL + h4_pos(L.u)] = x1; p[HL + h3_pos(L.u)] = x1; p[HL + h1_pos(L.u)] = x1;
  }
  sw_barrier();

  float e_sum = 0.f, e_last = 0.f, e_sq = 0.f;   // displacement-error sums of the wave's 4 agents (train.py:546-551)
  const int o_gate0 = (L.up ? 64 : 0) + L.u, o_gate1 = (L.up ? 192 : 128) + L.u, o_state = (L.up ? 320 : 256) + L.u;
  // A-operand rows of the lane (blk = lane >> 2, i = lane & 3) in each tile
  const int blk = L.blk, ai = L.ai;
  const float* h4p = htile + (4 * L.ag + ai) * HL + 4 * blk;
  const float* h3p = htile + (4 * (blk >> 3) + ai) * HL + 64 + 8 * (blk & 7);
  const float* h1p = htile + (4 * ((blk >> 1) & 1) + ai) * HL + 128 + 16 * (blk >> 2) + 8 * (blk & 1);
  const float* a12p = a1t + (4 * ((blk >> 2) & 1) + ai) * A1L + 80 * (blk >> 3) + 20 * (blk & 3);
  const float* a10p = a1t + (4 * (blk & 1) + ai) * A1L + 160 + 20 * (blk >> 1);
  const float* a2p = a2t + (4 * L.ag + ai) * A2L + 8 * blk;
  const float* w1rp = w1r_img + (r1r - 128) * 68 + 16 * kp1;
  const float* w2rp = w2r_img + (r2r - 64) * LD160 + 20 * kp2;
  // agents the lane's layer results belong to
  const int t1m = tile0 + 4 * g1m, t1r = tile0 + 4 * g1r, t2m = tile0 + 4 * g2m, t2r = tile0 + 4 * g2r;
  int cur = 0;
  auto step = [&](int i, auto last) {
    f32x4 gti;   // ground truth of step i (component lane & 1) for the error sums: in flight under the layers
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int qa = a0;
      asm volatile("" : "+v"(qa));
      gti[r] = ADE ? gt[(unsigned)((min(qa + r, B - 1) * Tp + i) * 2 + (lane & 1))] : 0.f;
    }
    const int hoff = cur * SW8_TILE * HL;
    // ---- layer 1: a1 = lrelu(W1h h + u) --------------------------------------------------------------------------
    {
      f32x4 ym, yr;
      {
        float x3[8];
        act_regs<8>(x3, h3p + hoff);
        ym = mm_k<3, 8>(um, x3, w1m);
      }
      {
        float x1[8], w1r[16];
        act_regs<8>(x1, h1p + hoff);
        row_regs<16>(w1r, w1rp);
        yr = mm_k<1, 8>(f32x4{0.f, 0.f, 0.f, 0.f}, x1, w1r);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ym[r] = sw_lrelu(ym[r]);
        yr[r] = sw_lrelu(sw8_sum4(yr[r], lane) + ur[r]);      // every k-part lane ends up with the full sum
        float* t1 = a1t + (4 * g1m + r) * A1L;
        t1[a1_pos2(r1m)] = ym[r];
        t1[160 + r1m] = ym[r];
        float* t2 = a1t + (4 * g1r + r) * A1L;
        t2[a1_pos2(r1r)] = yr[r];                                // the 4 k-part lanes write the same value
        t2[160 + r1r] = yr[r];
        if constexpr (SAVE) {   // uniform row base + 32-bit lane offset: no per-stream 64-bit address registers
          float* sb = gsave + gs.a1 + (size_t)i * B * 160;
          int qm = t1m, qr = t1r;                      // opaque per use: offsets are recomputed where they are needed
          asm volatile("" : "+v"(qm), "+v"(qr));      // instead of 36 loop-invariant address registers staying live
          sb[(unsigned)(min(qm + r, B - 1) * 160 + r1m)] = ym[r];
          sb[(unsigned)(min(qr + r, B - 1) * 160 + r1r)] = yr[r];
        }
      }
    }
    sw_barrier();
    // ---- layer 2: a2 = lrelu(W2 a1 + b2) -------------------------------------------------------------------------
    {
      f32x4 ym, yr;
      {
        float x2[20];
        act_regs<20>(x2, a12p);
        ym = mm_k<2, 20>(f32x4{0.f, 0.f, 0.f, 0.f}, x2, w2m);
      }
      {
        float x0[20], w2r[20];
        act_regs<20>(x0, a10p);
        row_regs<20>(w2r, w2rp);
        yr = mm_k<0, 20>(f32x4{0.f, 0.f, 0.f, 0.f}, x0, w2r);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ym[r] = sw_lrelu(sw8_sum2(ym[r], lane) + b2m);
        yr[r] = sw_lrelu(sw8_sum8(yr[r], lane) + b2r);
        a2t[(4 * g2m + r) * A2L + pos80(r2m)] = ym[r];
        a2t[(4 * g2r + r) * A2L + pos80(r2r)] = yr[r];
        if constexpr (SAVE) {
          float* sb = gsave + gs.a2 + (size_t)i * B * 80;
          int qm = t2m, qr = t2r;
          asm volatile("" : "+v"(qm), "+v"(qr));
          sb[(unsigned)(min(qm + r, B - 1) * 80 + r2m)] = ym[r];
          sb[(unsigned)(min(qr + r, B - 1) * 80 + r2r)] = yr[r];
        }
      }
    }
    sw_barrier();
    // ---- layers 3 + 4 composed, position update, the re-fed encoder step (train.py:422-430) ----------------------
    {
      f32x4 vq;
      {
        const f32x4 q0 = ld4(a2p);
        const float q1 = a2p[4];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = sw_m4<0, 0>(q0[0], w43[0], acc);
        acc = sw_m4<0, 0>(q0[1], w43[1], acc);
        acc = sw_m4<0, 0>(q0[2], w43[2], acc);
        acc = sw_m4<0, 0>(q0[3], w43[3], acc);
        acc = sw_m4<0, 0>(q1, w43[4], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) vq[r] = sw8_sum16(acc[r], lane) + b43;   // every lane: v_{x|y by lane parity} of agent r
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) pos[r] += vq[r];
      // (p, v) of the 4 agents -> this wave's exchange tile: lanes 0, 1 of block 0 write, the others hit a dummy word
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* dp = lane < 2 ? xs + 4 * r + lane : xdummy;
        float* dv = lane < 2 ? xs + 4 * r + 2 + lane : xdummy;
        *dp = pos[r];
        *dv = vq[r];
      }
      // outputs: every lane holds a valid copy (component = lane parity) and stores it
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int qa = a0;
        asm volatile("" : "+v"(qa));
        const int bq = min(qa + r, B - 1);
        const unsigned po = (unsigned)((bq * Tp + i) * 4 + (lane & 1));
        pred4[po] = pos[r];
        pred4[po + 2] = vq[r];
        if constexpr (SAVE && !decltype(last)::value) {
          float* sb = gsave + gs.x4s + (size_t)(To + i) * B * 4;
          const unsigned so = (unsigned)(bq * 4 + (lane & 1));
          sb[so] = pos[r];
          sb[so + 2] = vq[r];
        }
      }
      if constexpr (ADE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = (pos[r] - gti[r]) * inv_ss;
          const float d2 = d * d;
          const float q = d2 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d2), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
          const float e = sqrtf(q);
          const float live = (a0 + r) < B ? 1.0f : 0.0f;
          e_sum += live * e;
          e_sq += live * q;
          if constexpr (decltype(last)::value) e_last += live * e;
        }
      }
      if (!decltype(last)::value || h_end) {   // the step after the last decode is dead compute (train.py:430)
        const float xv = xs[4 * ai + (blk & 3)];
        f32x4 g0, g1;
        lstm8_cell(W, L, xv, h4p + hoff, g0, g1, c, h);
        {
          const int r0 = L.up ? 2 : 0;
          float* p = htile + (cur ^ 1) * SW8_TILE * HL + (4 * L.ag + r0) * HL;
          const float x0 = L.up ? h[2] : h[0], x1 = L.up ? h[3] : h[1];
          p[h4_pos(L.u)] = x0; p[h3_pos(L.u)] = x0; p[h1_pos(L.u)] = x0;
          p[HL + h4_pos(L.u)] = x1; p[HL + h3_pos(L.u)] = x1; p[HL + h1_pos(L.u)] = x1;
        }
        if constexpr (SAVE && !decltype(last)::value) {
          float* trow = gsave + gs.act + (size_t)(To + i) * B * 384;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int qa = a0;
            asm volatile("" : "+v"(qa));
            const unsigned ro = (unsigned)(min(qa + r, B - 1) * 384);
            trow[ro + o_gate0] = g0[r];
            trow[ro + o_gate1] = g1[r];
            trow[ro + o_state] = L.up ? h[r] : c[r];
          }
        }
        cur ^= 1;
      }
      sw_barrier();
    }
  };
  for (int i = 0; i < Tp - 1; ++i) step(i, std::false_type{});
  step(Tp - 1, std::true_type{});
  if (h_end) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (L.up) {
        if (c_end) c_end[(size_t)bD[r] * 64 + L.u] = c[r];
      } else {
        h_end[(size_t)bD[r] * 64 + L.u] = h[r];
      }
    }
  }
  if constexpr (ADE) {   // one partial triple per 8-agent tile: the two agent-group waves with uh == 0 meet in LDS
    sw_barrier();
    if (L.uh == 0 && lane == 0) {
      xs[0] = e_sum;
      xs[1] = e_last;
      xs[2] = e_sq;
    }
    sw_barrier();
    if (threadIdx.x == 0) {
      const float* x0 = smem + F8::xs;
      const float* x1 = smem + F8::xs + 16;
      ade_part[(size_t)blockIdx.x * 3 + 0] = (x0[0] + x1[0]) / (float)Tp;
      ade_part[(size_t)blockIdx.x * 3 + 1] = x0[1] + x1[1];
      ade_part[(size_t)blockIdx.x * 3 + 2] = x0[2] + x1[2];
    }
  }
}

static int set_lds8(const void* fn, int bytes) {
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    sw_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
    return SW_EHIP;
  }
  return SW_OK;
}

// called by sw_dec_rollout_fwd_aux when the batch runs on 8-agent tiles; ade_part then holds one triple per 8 agents
int sw_dec_rollout_fwd8_launch(const float* obsv, int To, const float* z, const float* S_pool, const float* hT,
                               const float* cT, const float* enc_w, const float* dec_w, int B, int Tp, float* pred4,
                               float* h_end, float* c_end, float* gsave, const float* gt, float inv_ss, float* ade_part,
                               hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    if (int rc = set_lds8((const void*)dec_rollout_fwd8_kernel<false, false>, F8::total * 4)) return rc;
    if (int rc = set_lds8((const void*)dec_rollout_fwd8_kernel<false, true>, F8::total * 4)) return rc;
    if (int rc = set_lds8((const void*)dec_rollout_fwd8_kernel<true, false>, F8::total * 4)) return rc;
    if (int rc = set_lds8((const void*)dec_rollout_fwd8_kernel<true, true>, F8::total * 4)) return rc;
    attr = true;
  }
  const int tiles = (B + SW8_TILE - 1) / SW8_TILE;
#define SW_DEC8(SV, AD)                                                                                              \
  SW_LAUNCH((dec_rollout_fwd8_kernel<SV, AD>), dim3(tiles), dim3(SW_THREADS), F8::total * 4, stream, obsv, To, z, \
                     S_pool, hT, cT, enc_w, dec_w, B, Tp, pred4, h_end, c_end, gsave, gt, inv_ss, ade_part)
  if (gsave) {
    if (ade_part) SW_DEC8(true, true);
    else SW_DEC8(true, false);
  } else {
    if (ade_part) SW_DEC8(false, true);
    else SW_DEC8(false, false);
  }
#undef SW_DEC8
  SW_CHECK_LAUNCH("dec_rollout_fwd8_kernel");
  return SW_OK;
}
