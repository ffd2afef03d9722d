// sw_common.h - device-side building blocks shared by the Social Ways gfx950 kernels.
//
// Everything GEMM-shaped on the hot path is a chain of small fp32 contractions (LSTM 64x256,
// decoder 160/80/40, pair MLP 32/64/64).  They all run on v_mfma_f32_16x16x4_f32 (exact fp32,
// bit-for-bit an fmaf chain) with ONE tiling convention:
//
//     D[16 output units][16 agents] += W[16 units][K] * X[K][16 agents]
//
//   A operand  = weights, row-major W[unit][k]          (lane: unit ln = lane&15, k-group lg = lane>>4)
//   B operand  = activations, row-major X[agent][k]     (lane: agent ln,          k-group lg)
//   C/D        = lane holds units 4*lg+r (r=0..3) of agent ln   -> one float4 of a row-major
//                [agent][unit] array, so every store/LDS write is a 16-byte access.
//
// The K order is permuted so that each lane's operands are float4s: k-step (j, r) makes lane
// group lg contribute k = 16*j + 4*lg + r.  A lane therefore reads W[unit][16j+4lg .. +3] and
// X[agent][16j+4lg .. +3] as one 16-byte load each and issues 4 MFMAs on them.  A permutation of
// the summation order is all that changes numerically.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define SW_TILE 16       // agents per workgroup tile
#define SW_THREADS 256   // 4 waves: one per SIMD

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding
// global load / store of the wave (s_waitcnt vmcnt(0)): the serial kernels keep save-stores and
// prefetch-loads in flight across their per-layer barriers, and nothing they exchange between waves
// goes through global memory, so only lgkmcnt is waited for.
__device__ __forceinline__ void sw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int sw_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ int sw_wave() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// Gate non-linearities.  SW_FAST_ACT=1 (default): v_exp_f32 / v_rcp_f32 forms (each ~1 ulp of its own
// result; |abs err| of sigmoid/tanh <= ~2e-7), 5-7 VALU ops instead of the ~40-op ocml expf/tanhf -
// the LSTM cell is latency-bound on exactly this chain.  SW_FAST_ACT=0 builds the ocml versions.
#ifndef SW_FAST_ACT
#define SW_FAST_ACT 1
#endif
#if SW_FAST_ACT
__device__ __forceinline__ float sw_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float sw_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + sw_exp(-x)); }
__device__ __forceinline__ float sw_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + sw_exp(2.0f * x)); }
#else
__device__ __forceinline__ float sw_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float sw_tanh(float x) { return tanhf(x); }
#endif
__device__ __forceinline__ float sw_lrelu(float x) { return x > 0.0f ? x : 0.2f * x; }
__device__ __forceinline__ float sw_lrelu_grad(float a, float g) { return a > 0.0f ? g : 0.2f * g; }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// the saved-row / delta-row stores of the time loops (a name of their own: tools/patches/timing_experiments.patch drops them)
__device__ __forceinline__ void st4g(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// acc += W[row][16j+4lg+r] * X[agent][16j+4lg+r], j < KJ.  `wrow` / `xrow` already point at
// column 4*lg of the lane's weight row / activation row.  Works for LDS and global pointers
// (the compiler keeps the address space when the caller is inlined).
//
// All operand loads of a tile are issued before its first MFMA (one LDS round trip per tile, not
// one per k-step: the compiler otherwise waits lgkmcnt(0) in front of every 4 MFMAs), and the
// products alternate between two accumulators: a 16x16x4 f32 MFMA issues every 32 cycles but a
// dependent one only every 40.
template <int KJ>
__device__ __forceinline__ f32x4 tile_mm(const float* wrow, const float* xrow, f32x4 acc) {
  f32x4 a[KJ], b[KJ];
#pragma unroll
  for (int j = 0; j < KJ; ++j) {
    a[j] = ld4(wrow + 16 * j);
    b[j] = ld4(xrow + 16 * j);
  }
  f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < KJ; ++j) {
    acc = SW_MFMA(a[j][0], b[j][0], acc);
    acc1 = SW_MFMA(a[j][1], b[j][1], acc1);
    acc = SW_MFMA(a[j][2], b[j][2], acc);
    acc1 = SW_MFMA(a[j][3], b[j][3], acc1);
  }
  return acc + acc1;
}

// Same with the weight operands held in registers (w[j] = float4 of the lane's row).
template <int KJ>
__device__ __forceinline__ f32x4 tile_mm_reg(const f32x4* w, const float* xrow, f32x4 acc) {
  f32x4 b[KJ];
#pragma unroll
  for (int j = 0; j < KJ; ++j) b[j] = ld4(xrow + 16 * j);
  f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < KJ; ++j) {
    acc = SW_MFMA(w[j][0], b[j][0], acc);
    acc1 = SW_MFMA(w[j][1], b[j][1], acc1);
    acc = SW_MFMA(w[j][2], b[j][2], acc);
    acc1 = SW_MFMA(w[j][3], b[j][3], acc1);
  }
  return acc + acc1;
}

// Stage a row-major [M][K] weight matrix (global, row stride src_ld) into LDS as [Mp][ld] with zero
// padding up to Mp rows / ld columns (ld = roundup(K,16)+4; ld, K, src_ld multiples of 4).  float4
// granularity, 8 independent loads in flight per thread: the whole 115 KB decoder image is a handful
// of L2 round trips instead of one per element.
__device__ __forceinline__ void stage_w(float* dst, int ld, int Mp, const float* src, int src_ld,
                                        int M, int K) {
  const int ldq = ld >> 2, n4 = Mp * ldq;
  for (int base = 0; base < n4; base += SW_THREADS * 16) {
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int f = base + threadIdx.x + SW_THREADS * u;
      int r = f / ldq, c = (f - r * ldq) * 4;
      v[u] = (f < n4 && r < M && c < K) ? ld4(src + (size_t)r * src_ld + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int f = base + threadIdx.x + SW_THREADS * u;
      if (f < n4) st4(dst + (size_t)f * 4, v[u]);
    }
  }
}
// Transposed: dst[c][r] = src[r][c]; dst is [Kp][ld] (rows = source columns), zero padded.  Source
// rows are read as float4s (coalesced), scattered to LDS as scalars.  The caller must have the
// destination zero-filled by stage_zero() and a barrier in between.
__device__ __forceinline__ void stage_zero(float* dst, int nfloats) {
  for (int i = threadIdx.x * 4; i < nfloats; i += SW_THREADS * 4) st4(dst + i, f32x4{0.f, 0.f, 0.f, 0.f});
}
__device__ __forceinline__ void stage_wT(float* dst, int ld, int Kp, const float* src, int src_ld,
                                         int M, int K) {
  const int k4 = K >> 2, n4 = M * k4;  // float4s of the live source block [M][K]
  for (int base = 0; base < n4; base += SW_THREADS * 16) {
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int f = base + threadIdx.x + SW_THREADS * u;
      int r = f / k4, c = (f - r * k4) * 4;
      v[u] = f < n4 ? ld4(src + (size_t)r * src_ld + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int f = base + threadIdx.x + SW_THREADS * u;
      int r = f / k4, c = (f - r * k4) * 4;
      if (f < n4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(size_t)(c + e) * ld + r] = v[u][e];
      }
    }
  }
}

// Split forms of stage_w / stage_wT: a prologue issues the loads of ALL its matrices first (NV float4 registers
// per matrix and thread, NV = ceil(#float4 / 256)) and stores them afterwards - one L2 round trip for the whole
// prologue instead of one per matrix (each staging call used to wait for its own loads before its LDS stores).
template <int NV>
__device__ __forceinline__ void stage_w_load(f32x4 (&v)[NV], int ld, int Mp, const float* src, int src_ld, int M, int K) {
  const int ldq = ld >> 2, n4 = Mp * ldq;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    int f = threadIdx.x + SW_THREADS * u;
    int r = f / ldq, c = (f - r * ldq) * 4;
    v[u] = (f < n4 && r < M && c < K) ? ld4(src + (size_t)r * src_ld + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <int NV>
__device__ __forceinline__ void stage_w_store(const f32x4 (&v)[NV], float* dst, int ld, int Mp) {
  const int n4 = Mp * (ld >> 2);
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    int f = threadIdx.x + SW_THREADS * u;
    if (f < n4) st4(dst + (size_t)f * 4, v[u]);
  }
}
template <int NV>
__device__ __forceinline__ void stage_wT_load(f32x4 (&v)[NV], const float* src, int src_ld, int M, int K) {
  const int k4 = K >> 2, n4 = M * k4;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    int f = threadIdx.x + SW_THREADS * u;
    int r = f / k4, c = (f - r * k4) * 4;
    v[u] = f < n4 ? ld4(src + (size_t)r * src_ld + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <int NV>
__device__ __forceinline__ void stage_wT_store(const f32x4 (&v)[NV], float* dst, int ld, int M, int K) {
  const int k4 = K >> 2, n4 = M * k4;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    int f = threadIdx.x + SW_THREADS * u;
    int r = f / k4, c = (f - r * k4) * 4;
    if (f < n4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[(size_t)(c + e) * ld + r] = v[u][e];
    }
  }
}

// LDS leading dimension for a K-wide operand.
__host__ __device__ constexpr int sw_ld(int K) { return ((K + 15) / 16) * 16 + 4; }

// ---------------------------------------------------------------------------------------------
// Packed weight layouts (state_dict order, each tensor on a 4-float boundary).
// ---------------------------------------------------------------------------------------------
namespace swp {
// EncoderLstm (train.py:245-269)
constexpr int ENC_EMB_W = 0;                    // embed.weight (64,4)
constexpr int ENC_EMB_B = 256;                  // embed.bias (64)
constexpr int ENC_WIH = 320;                    // lstm.weight_ih_l0 (256,64)
constexpr int ENC_WHH = 320 + 16384;            // lstm.weight_hh_l0 (256,64)
constexpr int ENC_BIH = ENC_WHH + 16384;        // lstm.bias_ih_l0 (256)
constexpr int ENC_BHH = ENC_BIH + 256;          // lstm.bias_hh_l0 (256)
constexpr int ENC_N = ENC_BHH + 256;            // 33600
// EmbedSocialFeatures (train.py:178-189)
constexpr int EMB_W0 = 0, EMB_B0 = 96, EMB_W1 = 128, EMB_B1 = 128 + 2048, EMB_W2 = EMB_B1 + 64,
              EMB_B2 = EMB_W2 + 4096, EMB_N = EMB_B2 + 64;  // 6400
// AttentionPooling (train.py:153-175)
constexpr int ATT_W = 0, ATT_B = 4096, ATT_N = 4160;
// DecoderFC(160) (train.py:320-335)
constexpr int DEC_W1 = 0, DEC_B1 = 25600, DEC_W2 = 25760, DEC_B2 = DEC_W2 + 12800, DEC_W3 = DEC_B2 + 80,
              DEC_B3 = DEC_W3 + 3200, DEC_W4 = DEC_B3 + 40, DEC_B4 = DEC_W4 + 80, DEC_N = DEC_B4 + 2;  // 41962
// Discriminator (train.py:272-309); offsets depend on Tp through pred_encoder.0.weight (32,4Tp)
struct Disc {
  int wih, whh, bih, bhh, of0w, of0b, of1w, of1b, pe0w, pe0b, pe1w, pe1b, cl0w, cl0b, cl1w, cl1b, la0w,
      la0b, la1w, la1b, n;
};
__host__ __device__ constexpr int al4(int x) { return (x + 3) & ~3; }
__host__ __device__ constexpr Disc disc(int Tp) {
  Disc d{};
  int o = 0;
  d.wih = o; o = al4(o + 256 * 4);
  d.whh = o; o = al4(o + 256 * 64);
  d.bih = o; o = al4(o + 256);
  d.bhh = o; o = al4(o + 256);
  d.of0w = o; o = al4(o + 32 * 64);
  d.of0b = o; o = al4(o + 32);
  d.of1w = o; o = al4(o + 32 * 32);
  d.of1b = o; o = al4(o + 32);
  d.pe0w = o; o = al4(o + 32 * 4 * Tp);
  d.pe0b = o; o = al4(o + 32);
  d.pe1w = o; o = al4(o + 32 * 32);
  d.pe1b = o; o = al4(o + 32);
  d.cl0w = o; o = al4(o + 32 * 64);
  d.cl0b = o; o = al4(o + 32);
  d.cl1w = o; o = al4(o + 32);
  d.cl1b = o; o = al4(o + 1);
  d.la0w = o; o = al4(o + 32 * 64);
  d.la0b = o; o = al4(o + 32);
  d.la1w = o; o = al4(o + 64);
  d.la1b = o; o = o + 2;
  d.n = o;
  return d;
}
}  // namespace swp

// ---------------------------------------------------------------------------------------------
// Time-major save / delta layouts of the generator (opaque to callers; sizes via the ABI).
//   T_all = To + Tp - 1 LSTM steps (the LSTM step after the last decode is dead compute).
// gsave : act  [T_all][B][384]  i f g o | c | h
//         x4s  [T_all][B][4]
//         a1   [Tp][B][160]   a2 [Tp][B][80]   a3 [Tp][B][40]
// gdelta: dgates [T_all][B][256], dz1 [Tp][B][160], dz2 [Tp][B][80], da3 [Tp][B][40],
//         dv [Tp][B][4] (cols 0,1 used), du [B][160]
// ---------------------------------------------------------------------------------------------
struct GSave {
  size_t act, x4s, a1, a2, a3, total;
};
__host__ __device__ inline GSave gsave_layout(int B, int To, int Tp) {
  GSave g;
  size_t Ta = (size_t)(To + Tp - 1), b = (size_t)B;
  g.act = 0;
  g.x4s = g.act + Ta * b * 384;
  g.a1 = g.x4s + Ta * b * 4;
  g.a2 = g.a1 + (size_t)Tp * b * 160;
  g.a3 = g.a2 + (size_t)Tp * b * 80;
  g.total = g.a3 + (size_t)Tp * b * 40;
  return g;
}
struct GDelta {
  size_t dgates, dz1, dz2, da3, dv, du, total;
};
__host__ __device__ inline GDelta gdelta_layout(int B, int To, int Tp) {
  GDelta g;
  size_t Ta = (size_t)(To + Tp - 1), b = (size_t)B;
  g.dgates = 0;
  g.dz1 = g.dgates + Ta * b * 256;
  g.dz2 = g.dz1 + (size_t)Tp * b * 160;
  g.da3 = g.dz2 + (size_t)Tp * b * 80;
  g.dv = g.da3 + (size_t)Tp * b * 40;
  g.du = g.dv + (size_t)Tp * b * 4;
  g.total = g.du + b * 160;
  return g;
}

// ---------------------------------------------------------------------------------------------
// Derived images of the generator's weights (sw_gen_images, sw_misc.hip): what every workgroup of the encoder / decode
// launches used to derive for itself in its prologue - the composed input matrix W_ih W_embed (256 uncoalesced row reads
// + 80 K MACs per workgroup), fc4 . fc3, and the TRANSPOSED, zero-padded decoder matrices of the backward pass (scalar
// LDS scatter with 8-way bank conflicts) - computed ONCE per step by a few workgroups of the staging launch.  The
// prologues then load them with coalesced 16-byte loads.  Layout in floats:
// ---------------------------------------------------------------------------------------------
namespace swimg {
constexpr int WX = 0;                    // [256][4]   W_ih W_embed
constexpr int BX = 1024;                 // [256]      W_ih b_embed + b_ih + b_hh
constexpr int W43 = 1280;                // [2][80] | b43[2]   fc4 . fc3, fc4 b3 + b4   (176 floats reserved)
// SNAPSHOT of the raw weights the composed maps are made of, as they were when this step started: the kernel that
// back-propagates through the compositions AND applies the generator's Adam step (sw_gen_wgrad_adam) reads these
// while it overwrites the live weights
constexpr int RAW_WIH = W43 + 176;       // [256][64]  encoder LSTM weight_ih
constexpr int RAW_WE = RAW_WIH + 16384;  // [64][4]    embed weight
constexpr int RAW_BE = RAW_WE + 256;     // [64]       embed bias
constexpr int RAW_W3 = RAW_BE + 64;      // [40][80]   fc3 weight
constexpr int RAW_B3 = RAW_W3 + 3200;    // [40]       fc3 bias
constexpr int RAW_W4 = RAW_B3 + 40;      // [2][40]    fc4 weight
constexpr int RAW_END = RAW_W4 + 80;
// MFMA A-OPERAND images of a matrix M [rows][K]: float4 q of lane l of k-step j of 16-row tile t =
// M[16 t + (l & 15)][16 j + 4 (l >> 4) .. + 3] at OP_x + ((t KJ + j) 64 + l) 4 - a wave's operand load is 1 KB of
// consecutive memory instead of 16 rows x 4 pieces (64 cache-line accesses per instruction).  The backward kernels
// use the TRANSPOSED matrices (M = W^T).
constexpr int OP_WHH = RAW_END;          // encoder LSTM weight_hh [256][64]: 16 tiles, KJ 4 (tile = 4 gate + wave)
constexpr int OP_W1H = OP_WHH + 16384;   // fc1.0.weight[:, 0:64]:   10 tiles, KJ 4
constexpr int OP_W1SZ = OP_W1H + 10240;  // fc1.0.weight[:, 64:160]: 10 tiles, KJ 6
constexpr int OP_W2 = OP_W1SZ + 15360;   // fc1.2.weight [80][160]:   5 tiles, KJ 10
constexpr int OP_WHHT = OP_W2 + 12800;   // weight_hh^T [64][256]:    4 tiles, KJ 16
constexpr int OP_W2T = OP_WHHT + 16384;  // fc1.2.weight^T [160][80]: 10 tiles, KJ 5
constexpr int OP_W1HT = OP_W2T + 12800;  // fc1.0.weight[:, 0:64]^T [64][160]: 4 tiles, KJ 10
// the social block's weights (feature embedder fc.2 / fc.4, attention W), when registered with the images
constexpr int OP_E1 = OP_W1HT + 10240;   // embedder fc.2.weight [64][32]:   4 tiles, KJ 2
constexpr int OP_E2 = OP_E1 + 2048;      // embedder fc.4.weight [64][64]:   4 tiles, KJ 4
constexpr int OP_E1T = OP_E2 + 4096;     // fc.2.weight^T [32][64]:          2 tiles, KJ 4
constexpr int OP_E2T = OP_E1T + 2048;    // fc.4.weight^T [64][64]:          4 tiles, KJ 4
constexpr int OP_ATT_T = OP_E2T + 4096;  // attention weight^T [64][64]:     4 tiles, KJ 4
// W_hh once more for the 8-WAVE encoder pilot (enc_lstm_fwd8_kernel, round 5): wave w owns units 8w .. 8w+7 of all four
// gates as TWO row tiles whose row m = 4 g + r is gate 2 tile + (r >> 1) of unit 8 w + 2 g + (r & 1), so that a lane's result
// registers hold i, f (tile 0) and g, o (tile 1) of the same two units: [wave 8][tile 2][j 4][lane 64] float4
constexpr int OP_WHH8 = OP_ATT_T + 4096;
constexpr int N = OP_WHH8 + 16384;
}  // namespace swimg
// the images registered for (enc_w, dec_w) by the current step, or null (sw_gen_images)
const float* sw_gen_images_for(const float* enc_w, const float* dec_w);
// ... whose social part (swimg::OP_E*, OP_ATT_T) was derived from these embedder / attention weights, or null
const float* sw_soc_images_for(const float* emb_w, const float* att_w);

// ---------------------------------------------------------------------------------------------
// Derived images of the DISCRIMINATOR's weights (sw_disc_images, sw_disc.hip).  D's weights change three times per
// training step (two Adam updates, the Linear-only restore of train.py:541-542), so unlike the generator's images they
// are not re-derived by a launch: a per-parameter TABLE maps every float of the packed D buffer to its (<= 2) places in
// the image buffer; the staging launch of a step scatters the whole buffer once, and the thread that applies D's Adam
// update to an element (wgrad_reduce, sw_adam_packed) also stores the new value to its image places.
//   OP_WHH   A-operand image of weight_hh [256][64] (forward: 16 row tiles, KJ 4)         - as swimg::OP_WHH
//   OP_WHHT  A-operand image of weight_hh^T [64][256] (BPTT: 4 row tiles, KJ 16)          - as swimg::OP_WHHT
//   HEADT    the eight TRANSPOSED, zero-padded head matrices exactly as disc_bwd / the fused generator-phase pass keep
//            them in LDS (HeadLdsB, sw_disc.hip: of0T | of1T | pe0T | pe1T | cl0T | la0T | cl1T | la1T), so that their
//            prologues are one contiguous float4 copy instead of eight row gathers + scalar LDS scatters with 8-way
//            bank conflicts.  Padding floats are zero from the allocation on and never written.
// ---------------------------------------------------------------------------------------------
namespace swdimg {
constexpr int OP_WHH = 0;
constexpr int OP_WHHT = 16384;
constexpr int HEADT = 32768;
}  // namespace swdimg
struct DiscImages {
  const float* img = nullptr;   // null: not registered (row-per-lane / gather prologues)
  const int* tab = nullptr;     // [n][2] image offsets per packed float (-1: none)
};
// the images registered for these packed D weights at this horizon (sw_disc_images / sw_stage_step_img), or {null, null}
DiscImages sw_disc_images_for(const float* d_w, int Tp);
void sw_disc_images_register(const float* d_w, const float* img, const int* tab, int Tp);   // all null / 0: drop
// scatter of the packed D weights into their image places, work split over nblk workgroups of 256 threads
__device__ __forceinline__ void disc_images_scatter(const float* __restrict__ d_w, float* __restrict__ img,
                                                    const int* __restrict__ tab, int n, int blk, int nblk) {
  for (int i = blk * 256 + (int)threadIdx.x; i < n; i += nblk * 256) {
    const int2 t = reinterpret_cast<const int2*>(tab)[i];
    const float w = d_w[i];
    if (t.x >= 0) img[t.x] = w;
    if (t.y >= 0) img[t.y] = w;
  }
}

// Every kernel launch of the library goes through SW_LAUNCH: with sw_kernel_timing(1) on (bench.py's roofline leg,
// tools/) each launch is bracketed by two HIP events on ITS stream and sw_kernel_timing_read() reports calls / total
// time per kernel; off (always inside graph capture) it is hipLaunchKernelGGL and nothing else.
void sw_ktime_begin(const char* name, hipStream_t st);
void sw_ktime_end(hipStream_t st);
extern bool g_sw_ktime_on;
#define SW_LAUNCH(kernel, grid, block, lds, stream, ...)                           \
  do {                                                                              \
    if (g_sw_ktime_on) sw_ktime_begin(#kernel, (stream));                           \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);              \
    if (g_sw_ktime_on) sw_ktime_end((stream));                                      \
  } while (0)

// host-side error plumbing ---------------------------------------------------------------------
void sw_set_error(const char* what, hipError_t e);
#define SW_CHECK_LAUNCH(name)                         \
  do {                                                \
    hipError_t _e = hipGetLastError();                \
    if (_e != hipSuccess) {                           \
      sw_set_error(name, _e);                         \
      return SW_EHIP;                                 \
    }                                                 \
  } while (0)
