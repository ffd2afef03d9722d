// sw_social.hip - the pairwise social block of predict() (reference train.py:408-411):
//   SocialFeatures (train.py:208-241)  ->  EmbedSocialFeatures (train.py:178-189)
//   ->  AttentionPooling (train.py:153-175), fused and BLOCK-DIAGONAL.
//
// The reference builds dense (B,B,3) / (B,B,64) tensors over all agents of the packed batch and
// then reads only the in-scene blocks; here one workgroup owns one scene and only
// its n^2 ordered pairs are ever formed.  EmbedSocialFeatures ends in a LINEAR layer (f_ij = W3 h2_ij + b3,
// train.py:183-188) and AttentionPooling uses f_ij only inside sigma_ij = <f_ij, Wh_j> (train.py:166-170), so
//     sigma_ij = <h2_ij, v_j> + c_j,   v_j = W3^T Wh_j,  c_j = <b3, Wh_j>        (per AGENT j: scene_wh_to_v)
// and the 64 -> 64 layer is never run per pair, forward or backward.  Per 16-pair tile (one wave):
//   features (VALU, per lane)  ->  3->32 ReLU as two K = 4 matrix instructions on (f0, f1, f2, 1) x (w0, w1, w2, bias)
//   ->  32->64 ReLU on the matrix cores, the output registers of one layer being the B operands of the
//   next (no LDS round trip: the K order of sw_common.h is chosen so that C/D layout == B layout)
//   ->  score = <h2_ij, v_j> + c_j by a 4-lane shuffle reduction (pair_block_v).
// Backward: dh2_ij = relu'(h2_ij) dsigma_ij v_j; dW3 = sum_j Wh_j Q_j^T, db3 = sum_j Wh_j sd_j, dWh_j = W3 Q_j + b3 sd_j with
// Q_j = sum_i dsigma_ij h2_ij, sd_j = sum_i dsigma_ij (pair_block_dw3).  The stand-alone module API (sw_embed_features) still
// returns f_ij.  Pair embeddings never touch HBM in the forward pass.
#include "../../include/socialways_hip.h"
#include "sw_common.h"
#include "sw_wgrad.h"

#define SW_AMAX 64  // max agents per scene handled by one workgroup (attn row stride)

#ifdef SW_PHASE_STAMPS
__device__ long long sw_soc_stamps[8];
#define SW_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); long long _t = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) sw_soc_stamps[k] += _t - _tprev; _tprev = _t; __builtin_amdgcn_sched_barrier(0); } while (0)
#define SW_STAMP_PARAM , long long& _tprev
#define SW_STAMP_ARG , _tprev
extern "C" int sw_debug_soc_stamps(long long* out, int reset) {
  if (reset) { long long z[8] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(sw_soc_stamps), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sw_soc_stamps), 8 * sizeof(long long));
}
#else
#define SW_STAMP(k)
#define SW_STAMP_PARAM
#define SW_STAMP_ARG
#endif
namespace {
struct PairW {            // per-lane register-resident pair-MLP weights
  f32x4 w1[4][2];         // fc.2.weight[16mt + ln][16j + 4lg ..]   (64 x 32)
  f32x4 w2[4][4];         // fc.4.weight[16mt'+ ln][16mt + 4lg ..]  (64 x 64)
};
__device__ __forceinline__ void load_pair_w(PairW& W, const float* emb_w, int ln, int lg) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) W.w1[mt][j] = ld4(emb_w + swp::EMB_W1 + (16 * mt + ln) * 32 + 16 * j + 4 * lg);
#pragma unroll
    for (int k = 0; k < 4; ++k) W.w2[mt][k] = ld4(emb_w + swp::EMB_W2 + (16 * mt + ln) * 64 + 16 * k + 4 * lg);
  }
}

// fc.2 alone (the kernels that never form f_ij = fc.4(..) explicitly, see "the attention never needs f_ij" below)
struct PairW1 {
  f32x4 w1[4][2];         // fc.2.weight[16mt + ln][16j + 4lg ..]   (64 x 32)
};
__device__ __forceinline__ void load_pair_w1(PairW1& W, const float* emb_w, int ln, int lg) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) W.w1[mt][j] = ld4(emb_w + swp::EMB_W1 + (16 * mt + ln) * 32 + 16 * j + 4 * lg);
  }
}
__device__ __forceinline__ void load_pair_w1_img(PairW1& W, const float* __restrict__ img, int lane) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) W.w1[mt][j] = ld4(img + swimg::OP_E1 + ((mt * 2 + j) * 64 + lane) * 4);
  }
}

// [dist, bearing, dca] of the ordered pair (i, j): dp = p_i - p_j, dv = v_i - v_j (train.py:232-234);
// eps placement as train.py:212,225.  The diagonal gives (0, 0, 0).
__device__ __forceinline__ void pair_feat(f32x4 si, f32x4 sj, float& f0, float& f1, float& f2) {
  float dpx = si[0] - sj[0], dpy = si[1] - sj[1];
  float dvx = si[2] - sj[2], dvy = si[3] - sj[3];
  float l2 = sqrtf(dpx * dpx + dpy * dpy);
  float vn = sqrtf(si[2] * si[2] + si[3] * si[3]);
  f0 = l2;
  f1 = (dpx * si[2] + dpy * si[3]) / (l2 * vn + 1e-6f);
  float ttca = -((dpx * dvx + dpy * dvy) / (dvx * dvx + dvy * dvy + 1e-6f));
  float cx = dpx + ttca * dvx, cy = dpy + ttca * dvy;
  f2 = sqrtf(cx * cx + cy * cy);
}

// layer 1 (3->32, ReLU) produced in B-operand layout: h1[j][r] = unit 16j + 4lg + r
__device__ __forceinline__ void pair_l1(const float* w0b, int lg, float f0, float f1, float f2, f32x4 h1[2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 w = ld4(w0b + (16 * j + 4 * lg + r) * 4);  // (w0, w1, w2, bias)
      float v = fmaf(w[2], f2, fmaf(w[1], f1, fmaf(w[0], f0, w[3])));
      h1[j][r] = fmaxf(v, 0.f);
    }
  }
}
// layer 1 on the matrix cores: K = 4 = (f0, f1, f2, 1) against (w0, w1, w2, bias) - 2 MFMAs instead of 8 LDS operand
// reads + 32 FMAs per lane; the result tile IS the B-operand layout above.  wa[t] = w0b[16t + ln][lg] (loop invariant).
__device__ __forceinline__ void pair_l1_load(const float* w0b, int ln, int lg, float wa[2]) {
  wa[0] = w0b[ln * 4 + lg];
  wa[1] = w0b[(16 + ln) * 4 + lg];
}
__device__ __forceinline__ void pair_l1m(const float wa[2], int lg, float f0, float f1, float f2, f32x4 h1[2]) {
  const float bv = lg == 0 ? f0 : (lg == 1 ? f1 : (lg == 2 ? f2 : 1.0f));
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x4 acc = SW_MFMA(wa[t], bv, (f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
    for (int r = 0; r < 4; ++r) h1[t][r] = fmaxf(acc[r], 0.f);
  }
}
// layers 2, 3 on the matrix cores; pre-activations of layer 2 are returned post-ReLU in h2
__device__ __forceinline__ void pair_l23(const PairW& W, const float* b1, const float* b2, int lg,
                                         const f32x4 h1[2], f32x4 h2[4], f32x4 f[4]) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    f32x4 acc = ld4(b1 + 16 * mt + 4 * lg);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = SW_MFMA(W.w1[mt][j][r], h1[j][r], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) h2[mt][r] = fmaxf(acc[r], 0.f);
  }
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) f[mo] = ld4(b2 + 16 * mo + 4 * lg);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) f[mo] = SW_MFMA(W.w2[mo][k][r], h2[k][r], f[mo]);
    }
  }
}

// layer 2 alone: h2 = relu(fc.2 h1 + b1), C layout
__device__ __forceinline__ void pair_l2(const PairW1& W, const float* b1, int lg, const f32x4 h1[2], f32x4 h2[4]) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    f32x4 acc = ld4(b1 + 16 * mt + 4 * lg);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = SW_MFMA(W.w1[mt][j][r], h1[j][r], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) h2[mt][r] = fmaxf(acc[r], 0.f);
  }
}

// LDS carve shared by forward and backward (floats), sized by the largest scene of the launch rounded
// up to 16 agents (a16): 8-agent scenes need 9 KB, 64-agent scenes 127 KB - small scenes leave the CU's
// LDS to co-resident workgroups.
struct SocL {
  int a16, sa;  // agents capacity, row stride of the n x n score matrices
  int hs, wh, x4, sig, w0b, b12, fwd_total, ds, dsg, dwh, bwd_total;
};
__host__ __device__ inline SocL soc_lds(int a16) {
  SocL L;
  L.a16 = a16;
  L.sa = a16;
  L.hs = 0;                         // [a16][68]  h of the scene
  L.wh = L.hs + a16 * 68;           // [a16][68]  W h + b
  L.x4 = L.wh + a16 * 68;           // [a16][4]
  L.sig = L.x4 + a16 * 4;           // [a16][a16] scores -> attention weights
  L.w0b = L.sig + a16 * a16;        // [32][4]    fc.0 weight|bias
  L.b12 = L.w0b + 128;              // fc.2.bias[64] | fc.4.bias[64]
  L.fwd_total = L.b12 + 128;
  L.ds = L.fwd_total;               // [a16][68]  dS of the scene          (backward only)
  L.dsg = L.ds + a16 * 68;          // [a16][a16] dsigma
  L.dwh = L.dsg + a16 * a16;        // [a16][68]  dWh
  L.bwd_total = L.dwh + a16 * 68;
  return L;
}

// h rows of the scene into LDS (+ zero rows up to a multiple of 16) and Wh = W h + b (train.py:161)
__device__ __forceinline__ void scene_load_h_wh(float* smem, const SocL& Ls, const float* h, const float* att_w, int s0, int n) {
  float* hs = smem + Ls.hs;
  float* wh = smem + Ls.wh;
  // the attention weights are requested first: they arrive together with the h rows (behind the barrier below they
  // were a global round trip of their own)
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  f32x4 wr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wr[j] = ld4(att_w + swp::ATT_W + (16 * wave + ln) * 64 + 16 * j + 4 * lg);
  f32x4 bias = ld4(att_w + swp::ATT_B + 16 * wave + 4 * lg);
  for (int i = threadIdx.x; i < n * 16; i += blockDim.x) {
    int a = i >> 4, q = i & 15;
    st4(&hs[a * 68 + 4 * q], ld4(h + (size_t)(s0 + a) * 64 + 4 * q));
  }
  int npad = ((n + 15) & ~15);
  for (int i = threadIdx.x; i < (npad - n) * 16; i += blockDim.x) {
    int a = n + (i >> 4), q = i & 15;
    st4(&hs[a * 68 + 4 * q], f32x4{0.f, 0.f, 0.f, 0.f});
  }
  sw_barrier();
  for (int at = 0; at < npad / 16; ++at) {
    f32x4 acc = tile_mm_reg<4>(wr, &hs[(16 * at + ln) * 68 + 4 * lg], bias);
    st4(&wh[(16 * at + ln) * 68 + 16 * wave + 4 * lg], acc);
  }
  sw_barrier();
}
// fc.0 weight|bias and fc.2 / fc.4 biases into LDS
__device__ __forceinline__ void stage_pair_consts(float* smem, const SocL& Ls, const float* emb_w) {
  float* w0b = smem + Ls.w0b;
  float* b12 = smem + Ls.b12;
  if (threadIdx.x < 32) {
    int k = threadIdx.x;
    f32x4 v = {emb_w[swp::EMB_W0 + k * 3], emb_w[swp::EMB_W0 + k * 3 + 1], emb_w[swp::EMB_W0 + k * 3 + 2],
               emb_w[swp::EMB_B0 + k]};
    st4(&w0b[k * 4], v);
  }
  if (threadIdx.x >= 64 && threadIdx.x < 192) {
    int k = threadIdx.x - 64;
    b12[k] = k < 64 ? emb_w[swp::EMB_B1 + k] : emb_w[swp::EMB_B2 + k - 64];
  }
}
__device__ __forceinline__ void scene_prologue(float* smem, const SocL& Ls, const float* obsv, int To, const float* h,
                                               const float* emb_w, const float* att_w, int s0, int n) {
  float* x4 = smem + Ls.x4;
  for (int a = threadIdx.x; a < n; a += blockDim.x) {
    const float* p = obsv + ((size_t)(s0 + a) * To + To - 2) * 2;  // last two observed points
    f32x4 v = {p[2], p[3], p[2] - p[0], p[3] - p[1]};
    st4(&x4[a * 4], v);
  }
  stage_pair_consts(smem, Ls, emb_w);
  scene_load_h_wh(smem, Ls, h, att_w, s0, n);
}
// ---- the attention never needs f_ij explicitly ---------------------------------------------------------------------
// f_ij = fc.4(h2_ij) = W3 h2_ij + b3 enters the model only through sigma_ij = <f_ij, Wh_j> (train.py:166-170), and
//      <W3 h2_ij + b3, Wh_j> = <h2_ij, v_j> + c_j,     v_j = W3^T Wh_j,  c_j = <b3, Wh_j>
// - one 64 x 64 product per AGENT instead of one per PAIR (64 of the 98 MFMAs of a pair tile's forward).  Backward the same
// identity gives dWh_j = sum_i dsigma_ij f_ij = W3 Q_j + b3 sd_j with the Q_j / sd_j of pair_block_dw3.
// Turns the scene's Wh rows (LDS [a16][68]) into v_j IN PLACE, c_j in column 64 of row j.
__device__ __forceinline__ void scene_wh_to_v(float* smem, const SocL& Ls, const float* emb_w, int n) {
  float* wh = smem + Ls.wh;
  const float* b3 = smem + Ls.b12 + 64;
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  f32x4 w3t[4];     // A operand: W3[m = 16j + 4lg + r][k = 16 wave + ln] (wave w owns the units k = 16w .. 16w+15 of v)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) w3t[j][r] = emb_w[swp::EMB_W2 + (16 * j + 4 * lg + r) * 64 + 16 * wave + ln];
  }
  const int nt = (n + 15) >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int at = 0; at < 4; ++at) {
    acc[at] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (at < nt) acc[at] = tile_mm_reg<4>(w3t, &wh[(16 * at + ln) * 68 + 4 * lg], acc[at]);
  }
  float c = 0.f;
  if ((int)threadIdx.x < n) {
    for (int u = 0; u < 64; u += 4) {
      const f32x4 x = ld4(&wh[threadIdx.x * 68 + u]), y = ld4(&b3[u]);
      c = fmaf(x[0], y[0], c); c = fmaf(x[1], y[1], c); c = fmaf(x[2], y[2], c); c = fmaf(x[3], y[3], c);
    }
  }
  sw_barrier();     // every wave has read the Wh rows
#pragma unroll
  for (int at = 0; at < 4; ++at)
    if (at < nt) st4(&wh[(16 * at + ln) * 68 + 16 * wave + 4 * lg], acc[at]);
  if ((int)threadIdx.x < n) wh[threadIdx.x * 68 + 64] = c;
  sw_barrier();
}
// softmax over the scene for every agent i (train.py:172, one wave per row) then
// S_i = sum_j a_ij h_j (train.py:173: pools the raw hidden states)
__device__ __forceinline__ void scene_softmax_pool(float* smem, const SocL& Ls, int s0, int n, float* S_out, float* attn) {
  float* hs = smem + Ls.hs;
  float* sig = smem + Ls.sig;
  const int sa = Ls.sa;
  const int lane = sw_lane(), wave = sw_wave();
  for (int i = wave; i < n; i += 4) {
    float v = lane < n ? sig[i * sa + lane] : -INFINITY;
    float m = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float e = lane < n ? expf(v - m) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    float a = e / sum;
    if (lane < n) {
      sig[i * sa + lane] = a;
      if (attn) attn[(size_t)(s0 + i) * SW_AMAX + lane] = a;
    }
  }
  sw_barrier();
  for (int e = threadIdx.x; e < n * 64; e += blockDim.x) {
    int i = e >> 6, u = e & 63;
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc = fmaf(sig[i * sa + j], hs[j * 68 + u], acc);
    S_out[(size_t)(s0 + i) * 64 + u] = acc;
  }
}
}  // namespace

__global__ __launch_bounds__(SW_THREADS) void social_pool_fwd_kernel(
    const float* __restrict__ obsv, int To, const float* __restrict__ h, const int* __restrict__ scene_off,
    const float* __restrict__ emb_w, const float* __restrict__ att_w, float* __restrict__ S_out,
    float* __restrict__ attn, int a16, int S, const float* __restrict__ aux_src, float* __restrict__ aux_dst, long long aux_n,
    const float* __restrict__ simg) {
  // The FIRST workgroups of the grid only copy aux_src -> aux_dst (dense crowds: the second half of the step's z comes
  // out of its pinned host slot here - the decode launch behind is the first consumer; see enc_lstm_fwd_kernel)
  const int extra = (int)gridDim.x - S;
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
  const int scene = (int)blockIdx.x - extra;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SocL Ls = soc_lds(a16);
  const int sa = Ls.sa;
  float* wh = smem + Ls.wh;
  float* x4 = smem + Ls.x4;
  float* sig = smem + Ls.sig;
  const float* w0b = smem + Ls.w0b;
  const float* b12 = smem + Ls.b12;
  const int s0 = scene_off[scene], n = scene_off[scene + 1] - s0;
  if (n <= 0 || n > SW_AMAX) return;   // scenes above SW_AMAX agents go through the row-block kernels below
  if (n == 1) {  // train.py:165: single-agent scenes keep S = 0
    if (threadIdx.x < 16) st4(S_out + (size_t)s0 * 64 + 4 * threadIdx.x, f32x4{0.f, 0.f, 0.f, 0.f});
    if (attn && threadIdx.x == 0) attn[(size_t)s0 * SW_AMAX] = 0.f;
    return;
  }
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  PairW1 W;    // requested first: the weights arrive under the scene prologue
  if (simg) load_pair_w1_img(W, simg, lane);
  else load_pair_w1(W, emb_w, ln, lg);
  scene_prologue(smem, Ls, obsv, To, h, emb_w, att_w, s0, n);
  scene_wh_to_v(smem, Ls, emb_w, n);     // `wh` rows now hold v_j | c_j: sigma_ij = <h2_ij, v_j> + c_j, fc.4 is never run
  float w0a[2];
  pair_l1_load(w0b, ln, lg, w0a);
  const int P = n * n;
  for (int pt = wave; pt * 16 < P; pt += 4) {
    int p = min(pt * 16 + ln, P - 1);
    int i = p / n, j = p - i * n;
    float f0, f1, f2;
    pair_feat(ld4(&x4[i * 4]), ld4(&x4[j * 4]), f0, f1, f2);
    f32x4 h1[2], h2[4];
    pair_l1m(w0a, lg, f0, f1, f2, h1);
    pair_l2(W, b12, lg, h1, h2);
    float part = 0.f;
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) {
      f32x4 w = ld4(&wh[j * 68 + 16 * mo + 4 * lg]);
#pragma unroll
      for (int r = 0; r < 4; ++r) part = fmaf(h2[mo][r], w[r], part);
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (lg == 0 && pt * 16 + ln < P) sig[i * sa + j] = (i == j) ? -1000.0f : part + wh[j * 68 + 64];  // train.py:170
  }
  sw_barrier();
  scene_softmax_pool(smem, Ls, s0, n, S_out, attn);
}

// ---------------------------------------------------------------------------------------------
// Backward.  Inputs carry no gradient (tracks are data); gradients flow to h (through the pooling
// and through W h), to W/b of the attention and to the pair MLP.  The pair MLP forward is
// recomputed per 16-pair tile up to h2 (f_ij is never formed: "the attention never needs f_ij" above,
// "the rank-1 structure" below) and its WEIGHT GRADIENTS ARE ACCUMULATED IN REGISTERS across all tiles
// (and scenes) a wave processes: per tile dh2 and h1 are transposed through a small per-wave LDS scratch
// into "k = pair" MFMA layout and 64 MFMAs update dW2 / produce dh1; per j block 64 more update dW3 from
// Wh and Q.  Writing per-pair rows for a deferred GEMM instead cost 324 floats per pair - 2.7 GB per step
// at 512 x 64-agent scenes.  One partial (6400 floats) per workgroup goes to the wgrad workspace and is
// reduced in fixed order by wgrad_reduce_kernel.  No per-pair data reaches HBM.
// ---------------------------------------------------------------------------------------------
#define SW_SOC_W2LD 68   // LDS row strides of the row-major fc.4 / fc.2 weight images (forward of the row-block kernel)
#define SW_SOC_W1LD 36
#define SW_SOC_WTLD 68   // row stride of the transposed images fc.4.weight^T [64][68], fc.2.weight^T [32][68] (backward)
#define SW_SOC_WT (96 * SW_SOC_WTLD)
#define SW_SOC_TLD 20    // row stride of a 16x16 transposition tile
#define SW_SOC_SCR (4 * 16 * SW_SOC_TLD)   // per-wave scratch: 4 tiles
#define SW_SOC_FUSE_MIN_PAIRS 256   // mean pairs per scene from which the in-register weight gradients pay (>= 4 tiles per wave)
#define SW_SOC_PART (64 * 65 + 64 * 33 + 32 * 4)   // floats per workgroup partial: dW3|db3, dW2|db2, dW1|db1

struct SocPart {   // where workgroup g leaves its partial (wgrad workspace, [g][N][Kc] per problem)
  float *p3, *p2, *p1;
};

// V (C layout: lane holds V[unit 4lg+r][pair ln]) -> T[r] = V[unit ln][pair 4lg+r] through a per-wave tile
__device__ __forceinline__ void tr_put(float* tile, f32x4 v, int ln, int lg) {
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[(4 * lg + r) * SW_SOC_TLD + ln] = v[r];
}
__device__ __forceinline__ f32x4 tr_get(const float* tile, int ln, int lg) { return ld4(tile + ln * SW_SOC_TLD + 4 * lg); }
// The LDS unit executes one wave's DS instructions in order, so a wave re-reading what its own lanes just
// wrote needs no hardware wait - only the compiler must not move the accesses across each other.  (A real
// fence would also drain the wave's outstanding GLOBAL stores - the f rows - at every transposition.)
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }

// In-register gradients of the pair MLP (C layout: [out unit 4lg+r][in unit ln] per 16x16 tile) + lane partials
struct PairGrad {
  f32x4 acc3[4][4], acc2[4][2];   // fc.4.weight (64x64), fc.2.weight (64x32)
  float b3s[4], b2s[4], b1s[2];   // bias sums: unit 16t + ln, partial over (lg, r)
  float w0s[2][3];                // fc.0.weight[unit 16jt + ln][c], partial over (lg, r)
};
__device__ __forceinline__ void pair_grad_zero(PairGrad& G) {
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) G.acc3[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    G.acc2[a][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    G.acc2[a][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    G.b3s[a] = 0.f;
    G.b2s[a] = 0.f;
  }
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    G.b1s[jt] = 0.f;
    G.w0s[jt][0] = G.w0s[jt][1] = G.w0s[jt][2] = 0.f;
  }
}

// fc.4.weight^T | fc.2.weight^T into LDS (once per workgroup)
__device__ __forceinline__ void stage_pair_wt(float* w2t, float* w1t, const float* emb_w) {
  for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {
    const int r = i >> 4, q = i & 15;
    const f32x4 v = ld4(emb_w + swp::EMB_W2 + r * 64 + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) w2t[(4 * q + e) * SW_SOC_WTLD + r] = v[e];
    if (q < 8) {
      const f32x4 u = ld4(emb_w + swp::EMB_W1 + r * 32 + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) w1t[(4 * q + e) * SW_SOC_WTLD + r] = u[e];
    }
  }
}

// Second half of a tile's backward: given dh2 = d(loss)/d(pre-activation of fc.2) (C layout, ReLU' applied, exact zeros for
// invalid pairs): dW2 += dh2 h1^T, dh1 = (W1^T dh2) relu'(h1), dW1 / db1 += dh1 [feat | 1].
__device__ __forceinline__ void pair_tile_bwd_tail(PairGrad& G, float* scr, const float* w1t, const f32x4 h1[2],
                                                   const f32x4 dh2[4], float f0, float f1, float f2, int ln,
                                                   int lg SW_STAMP_PARAM) {
  f32x4 ta[4], tb[4];
  // dW2 += dh2 h1^T
#pragma unroll
  for (int t = 0; t < 4; ++t) tr_put(scr + t * 16 * SW_SOC_TLD, dh2[t], ln, lg);
  wave_lds_fence();
#pragma unroll
  for (int t = 0; t < 4; ++t) ta[t] = tr_get(scr + t * 16 * SW_SOC_TLD, ln, lg);
  wave_lds_fence();
  tr_put(scr, h1[0], ln, lg);
  tr_put(scr + 16 * SW_SOC_TLD, h1[1], ln, lg);
  wave_lds_fence();
  tb[0] = tr_get(scr, ln, lg);
  tb[1] = tr_get(scr + 16 * SW_SOC_TLD, ln, lg);
  wave_lds_fence();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      G.acc2[mt][0] = SW_MFMA(ta[mt][r], tb[0][r], G.acc2[mt][0]);
      G.acc2[mt][1] = SW_MFMA(ta[mt][r], tb[1][r], G.acc2[mt][1]);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) G.b2s[t] += (ta[t][0] + ta[t][1]) + (ta[t][2] + ta[t][3]);
  // dh1 = (W1^T dh2) * relu'(h1)
  f32x4 dh1[2];
  dh1[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  dh1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const f32x4 wa = ld4(w1t + ln * SW_SOC_WTLD + 16 * mt + 4 * lg);
    const f32x4 wb = ld4(w1t + (16 + ln) * SW_SOC_WTLD + 16 * mt + 4 * lg);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dh1[0] = SW_MFMA(wa[r], dh2[mt][r], dh1[0]);
      dh1[1] = SW_MFMA(wb[r], dh2[mt][r], dh1[1]);
    }
  }
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dh1[jt][r] = h1[jt][r] > 0.f ? dh1[jt][r] : 0.f;
  }
  SW_STAMP(4);
  // dW1 += dh1 feat^T, db1 += dh1 (VALU): transposed dh1 against the features of pairs 4lg + r
  tr_put(scr, dh1[0], ln, lg);
  tr_put(scr + 16 * SW_SOC_TLD, dh1[1], ln, lg);
  if (lg == 0) st4(scr + 2 * 16 * SW_SOC_TLD + 4 * ln, f32x4{f0, f1, f2, 0.f});   // feat[pair ln][0..2]
  wave_lds_fence();
  ta[0] = tr_get(scr, ln, lg);
  ta[1] = tr_get(scr + 16 * SW_SOC_TLD, ln, lg);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const f32x4 ft = ld4(scr + 2 * 16 * SW_SOC_TLD + 4 * (4 * lg + r));
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      G.w0s[jt][0] = fmaf(ta[jt][r], ft[0], G.w0s[jt][0]);
      G.w0s[jt][1] = fmaf(ta[jt][r], ft[1], G.w0s[jt][1]);
      G.w0s[jt][2] = fmaf(ta[jt][r], ft[2], G.w0s[jt][2]);
      G.b1s[jt] += ta[jt][r];
    }
  }
  wave_lds_fence();
}

// ---- the rank-1 structure of the attention's gradient -------------------------------------------------------------
// sigma_ij = <f_ij, Wh_j>  =>  dz3_ij = d(loss)/d(f_ij) = dsigma_ij Wh_j: for the 16 pairs (i, j = 16 jb + ln) of a tile the
// 64-vector is the SAME per column j for every i, only its scale dsigma_ij changes.  Two of the four matrix products of a
// tile's backward therefore factor over the i loop of a j block:
//   dh2_ij = relu'(h2_ij) . (W3^T dz3_ij) = relu'(h2_ij) . dsigma_ij v_j,   v_j = W3^T Wh_j   (once per block: pair_block_v)
//   dW3 = sum_ij dz3_ij h2_ij^T = sum_j Wh_j Q_j^T,   Q_j = sum_i dsigma_ij h2_ij             (VALU per tile, then once per
//   db3 = sum_ij dz3_ij = sum_j Wh_j sd_j,            sd_j = sum_i dsigma_ij                    block: pair_block_dw3)
// 128 of a tile's 288 MFMAs and the two largest transpositions become 32 vector FMAs.  (Re-associated sums: the results
// differ from the per-pair order in the last bits, like every split-K order in this library.)
__device__ __forceinline__ void pair_block_v(const float* w2t, const f32x4 whj[4], f32x4 v[4], int ln, int lg) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) v[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) {
    f32x4 wt[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) wt[mt] = ld4(w2t + (16 * mt + ln) * SW_SOC_WTLD + 16 * mo + 4 * lg);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) v[mt] = SW_MFMA(wt[mt][r], whj[mo][r], v[mt]);
    }
  }
}
__device__ __forceinline__ void pair_block_dw3(PairGrad& G, float* scr, const f32x4 whj[4], const f32x4 Q[4], float sd, int ln,
                                               int lg) {
  f32x4 ta[4], tb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) tr_put(scr + t * 16 * SW_SOC_TLD, whj[t], ln, lg);
  wave_lds_fence();
#pragma unroll
  for (int t = 0; t < 4; ++t) ta[t] = tr_get(scr + t * 16 * SW_SOC_TLD, ln, lg);
  wave_lds_fence();
#pragma unroll
  for (int t = 0; t < 4; ++t) tr_put(scr + t * 16 * SW_SOC_TLD, Q[t], ln, lg);
  wave_lds_fence();
#pragma unroll
  for (int t = 0; t < 4; ++t) tb[t] = tr_get(scr + t * 16 * SW_SOC_TLD, ln, lg);
  wave_lds_fence();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) G.acc3[mo][kt] = SW_MFMA(ta[mo][r], tb[kt][r], G.acc3[mo][kt]);
    }
  }
  // ta[t][r] = Wh[unit 16t + ln][agent 4lg + r of the block]; sd of that agent sits in lane 4lg + r (any lane group)
  float sdT[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) sdT[r] = __shfl(sd, 4 * lg + r);
#pragma unroll
  for (int t = 0; t < 4; ++t)
    G.b3s[t] += fmaf(ta[t][0], sdT[0], ta[t][1] * sdT[1]) + fmaf(ta[t][2], sdT[2], ta[t][3] * sdT[3]);
}

// Workgroup partial = sum of its 4 waves' PairGrad in a fixed order through LDS (`red`: SW_SOC_PART floats,
// everything else in LDS is dead), then to the wgrad workspace slices [64][65] | [64][33] | [32][4].
__device__ __forceinline__ void pair_grad_store(PairGrad& G, float* red, const SocPart& part, int slice, int wave, int ln,
                                                int lg) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {   // finish the lane-partial sums over the 4 lane groups (fixed shuffle tree)
    G.b3s[t] += __shfl_xor(G.b3s[t], 16); G.b3s[t] += __shfl_xor(G.b3s[t], 32);
    G.b2s[t] += __shfl_xor(G.b2s[t], 16); G.b2s[t] += __shfl_xor(G.b2s[t], 32);
  }
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    G.b1s[jt] += __shfl_xor(G.b1s[jt], 16); G.b1s[jt] += __shfl_xor(G.b1s[jt], 32);
#pragma unroll
    for (int c = 0; c < 3; ++c) { G.w0s[jt][c] += __shfl_xor(G.w0s[jt][c], 16); G.w0s[jt][c] += __shfl_xor(G.w0s[jt][c], 32); }
  }
  __syncthreads();
  float* r3 = red, *r2 = red + 64 * 65, *r1 = r2 + 64 * 33;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nrow = 16 * mo + 4 * lg + r;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            float* q = r3 + nrow * 65 + 16 * kt + ln;
            *q = (w == 0 ? 0.f : *q) + G.acc3[mo][kt][r];
          }
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            float* q = r2 + nrow * 33 + 16 * jt + ln;
            *q = (w == 0 ? 0.f : *q) + G.acc2[mo][jt][r];
          }
        }
      }
      if (lg == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float* q3 = r3 + (16 * t + ln) * 65 + 64;
          float* q2 = r2 + (16 * t + ln) * 33 + 32;
          *q3 = (w == 0 ? 0.f : *q3) + G.b3s[t];
          *q2 = (w == 0 ? 0.f : *q2) + G.b2s[t];
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          float* q = r1 + (16 * jt + ln) * 4;
          q[0] = (w == 0 ? 0.f : q[0]) + G.w0s[jt][0];
          q[1] = (w == 0 ? 0.f : q[1]) + G.w0s[jt][1];
          q[2] = (w == 0 ? 0.f : q[2]) + G.w0s[jt][2];
          q[3] = (w == 0 ? 0.f : q[3]) + G.b1s[jt];
        }
      }
    }
    __syncthreads();
  }
  float* o3 = part.p3 + (size_t)slice * 64 * 65;
  float* o2 = part.p2 + (size_t)slice * 64 * 33;
  float* o1 = part.p1 + (size_t)slice * 32 * 4;
  for (int e = threadIdx.x; e < 64 * 65; e += blockDim.x) o3[e] = r3[e];
  for (int e = threadIdx.x; e < 64 * 33; e += blockDim.x) o2[e] = r2[e];
  for (int e = threadIdx.x; e < 32 * 4; e += blockDim.x) o1[e] = r1[e];
}

__global__ __launch_bounds__(SW_THREADS) void social_pool_bwd_kernel(
    const float* __restrict__ obsv, int To, const float* __restrict__ h, const int* __restrict__ scene_off,
    int S, const float* __restrict__ emb_w, const float* __restrict__ att_w, const float* __restrict__ attn,
    const float* __restrict__ dS, float* __restrict__ dh, float* __restrict__ dwh_rows, SocPart part, int a16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SocL Ls = soc_lds(a16);
  const int sa = Ls.sa;
  float* hs = smem + Ls.hs;
  float* wh = smem + Ls.wh;
  float* x4 = smem + Ls.x4;
  float* sig = smem + Ls.sig;  // attention weights a_ij
  const float* w0b = smem + Ls.w0b;
  const float* b12 = smem + Ls.b12;
  float* dsl = smem + Ls.ds;
  float* dsg = smem + Ls.dsg;
  float* dwh = smem + Ls.dwh;
  float* w2t = smem + Ls.bwd_total;                   // fc.4.weight^T [64][68]
  float* w1t = w2t + 64 * SW_SOC_WTLD;                // fc.2.weight^T [32][68]
  float* scr_all = w2t + SW_SOC_WT;                   // [4 waves][4 tiles][16][20]
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  float* scr = scr_all + wave * SW_SOC_SCR;

  // ---- once per workgroup: weights ------------------------------------------------------------
  stage_pair_wt(w2t, w1t, emb_w);
  PairW1 W;
  load_pair_w1(W, emb_w, ln, lg);
  // fc.4 itself is only needed for dWh_j = W3 Q_j + b3 sd_j, once per j block: wave w forms the units 16w .. 16w+15
  f32x4 w2own[4];     // A operand: W3[m = 16 wave + ln][k = 16kt + 4lg + r]
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) w2own[kt] = ld4(emb_w + swp::EMB_W2 + (16 * wave + ln) * 64 + 16 * kt + 4 * lg);
  const f32x4 b3own = ld4(emb_w + swp::EMB_B2 + 16 * wave + 4 * lg);
  PairGrad G;
  pair_grad_zero(G);
#ifdef SW_PHASE_STAMPS
  long long _tprev = clock64();
#endif

  for (int sc = blockIdx.x; sc < S; sc += gridDim.x) {
    const int s0 = scene_off[sc], n = scene_off[sc + 1] - s0;
    if (n <= 0 || n > SW_AMAX) continue;
    if (n == 1) {  // S = 0 constant: no gradient anywhere; dWh row is zero
      if (threadIdx.x < 16) st4(dwh_rows + (size_t)s0 * 64 + 4 * threadIdx.x, f32x4{0.f, 0.f, 0.f, 0.f});
      continue;
    }
    __syncthreads();   // previous scene's LDS is done with
    scene_prologue(smem, Ls, obsv, To, h, emb_w, att_w, s0, n);
    for (int i = threadIdx.x; i < n * 16; i += blockDim.x) {
      int a = i >> 4, q = i & 15;
      st4(&dsl[a * 68 + 4 * q], ld4(dS + (size_t)(s0 + a) * 64 + 4 * q));
    }
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
      int i = e / n, j = e - i * n;
      sig[i * sa + j] = attn[(size_t)(s0 + i) * SW_AMAX + j];
    }
    sw_barrier();
    SW_STAMP(0);
    // da_ij = <dS_i, h_j>;  dsigma_ij = a_ij (da_ij - sum_j' a_ij' da_ij')   (softmax backward)
    for (int i = wave; i < n; i += 4) {
      float da = 0.f, a = 0.f;
      if (lane < n) {
        a = sig[i * sa + lane];
        for (int u = 0; u < 64; u += 4) {
          f32x4 x = ld4(&dsl[i * 68 + u]), y = ld4(&hs[lane * 68 + u]);
          da = fmaf(x[0], y[0], da); da = fmaf(x[1], y[1], da); da = fmaf(x[2], y[2], da); da = fmaf(x[3], y[3], da);
        }
      }
      float t = a * da;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
      if (lane < n) dsg[i * sa + lane] = a * (da - t);
    }
    sw_barrier();
    SW_STAMP(1);
    // ---- pair tiles: recompute the MLP, back-propagate, accumulate the weight gradients ----------
    // A tile = 16 pairs (i, j = 16 jb + ln) of ONE i: lane column ln is agent j of the block in every tile, so
    //   dWh_j = sum_i dsigma_ij f_ij   accumulates in registers in the layout the MLP leaves f in (no pair
    // rows in HBM, no second pass), and x_j / Wh_j are loop invariants.  Per block: 4 wave partials (i = wave,
    // wave + 4, ..) summed in a fixed order through the waves' scratch tiles.
    const int nJB = (n + 15) >> 4;
    float w0a[2];
    pair_l1_load(w0b, ln, lg, w0a);
    for (int jb = 0; jb < nJB; ++jb) {
      const int j = 16 * jb + ln;
      const bool valid = j < n;
      const int jc = min(j, n - 1);
      const f32x4 xj = ld4(&x4[jc * 4]);
      f32x4 whj[4];
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) whj[mo] = ld4(&wh[jc * 68 + 16 * mo + 4 * lg]);
      // per block: v_j = W3^T Wh_j; per tile dh2 = relu'(h2) dsigma v_j on the VALU, Q_j / sd_j accumulate for dW3 / db3
      f32x4 vj[4], Q[4];
      float sd = 0.f;
      pair_block_v(w2t, whj, vj, ln, lg);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) Q[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int i = wave; i < n; i += 4) {
        float f0, f1, f2;
        pair_feat(ld4(&x4[i * 4]), xj, f0, f1, f2);
        f32x4 h1[2], h2[4];
        pair_l1m(w0a, lg, f0, f1, f2, h1);
        pair_l2(W, b12, lg, h1, h2);
        const float dsv = valid ? dsg[i * sa + jc] : 0.f;   // invalid lanes contribute exact zeros everywhere below
        f32x4 dh2[4];
        sd += dsv;
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            Q[mo][r] = fmaf(dsv, h2[mo][r], Q[mo][r]);
            dh2[mo][r] = h2[mo][r] > 0.f ? dsv * vj[mo][r] : 0.f;
          }
        }
        SW_STAMP(2);
        SW_STAMP(3);
        pair_tile_bwd_tail(G, scr, w1t, h1, dh2, f0, f1, f2, ln, lg SW_STAMP_ARG);
        SW_STAMP(5);
      }
      {
        f32x4 whr[4];     // re-read: keeps 16 registers out of the tile loop
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) whr[mo] = ld4(&wh[jc * 68 + 16 * mo + 4 * lg]);
        pair_block_dw3(G, scr, whr, Q, sd, ln, lg);
      }
      wave_lds_fence();
      // dWh_j = sum_i dsigma_ij f_ij = W3 Q_j + b3 sd_j with Q_j, sd_j summed over the 4 waves (fixed order) through their
      // scratch tiles; wave w forms the output units 16w .. 16w+15 of the block's 16 agents (16 MFMAs)
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) st4(scr + ln * 68 + 16 * mo + 4 * lg, Q[mo]);
      if (lg == 0) scr[ln * 68 + 64] = sd;
      __syncthreads();
      {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const float* q = scr_all + ln * 68 + 16 * kt + 4 * lg;
          const f32x4 bq = (ld4(q) + ld4(q + SW_SOC_SCR)) + (ld4(q + 2 * SW_SOC_SCR) + ld4(q + 3 * SW_SOC_SCR));
          acc = SW_MFMA(w2own[kt][0], bq[0], acc);
          acc1 = SW_MFMA(w2own[kt][1], bq[1], acc1);
          acc = SW_MFMA(w2own[kt][2], bq[2], acc);
          acc1 = SW_MFMA(w2own[kt][3], bq[3], acc1);
        }
        const float* qs = scr_all + ln * 68 + 64;
        const float sds = (qs[0] + qs[SW_SOC_SCR]) + (qs[2 * SW_SOC_SCR] + qs[3 * SW_SOC_SCR]);
        f32x4 out = acc + acc1;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[r] = fmaf(b3own[r], sds, out[r]);
        const int j2 = 16 * jb + ln;
        if (j2 < n) {
          st4(&dwh[j2 * 68 + 16 * wave + 4 * lg], out);
          st4(dwh_rows + (size_t)(s0 + j2) * 64 + 16 * wave + 4 * lg, out);
        }
      }
      __syncthreads();   // the scratch tiles go back to the transpositions of the next block
      SW_STAMP(6);
    }
    sw_barrier();
    // dh_j += sum_i a_ij dS_i  +  W^T dWh_j  on the matrix cores: D[unit u][agent j], wave w owns units 16w .. 16w+15;
    // first product over k = i (rows beyond n masked to exact zeros), second over k = the 64 units of dWh
    {
      const int npad = (n + 15) & ~15;
      f32x4 wT[4];   // A operand of the second product: W[k = 16kt + 4lg + r][u = 16 wave + ln]
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) wT[kt][r] = att_w[swp::ATT_W + (16 * kt + 4 * lg + r) * 64 + 16 * wave + ln];
      }
      for (int at = 0; at < npad / 16; ++at) {
        f32x4 acc = tile_mm_reg<4>(wT, &dwh[(16 * at + ln) * 68 + 4 * lg], f32x4{0.f, 0.f, 0.f, 0.f});
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < npad / 16; ++kt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * kt + 4 * lg + r;
            const float av = i < n ? dsl[i * 68 + 16 * wave + ln] : 0.f;
            const float bv = i < n ? sig[i * sa + 16 * at + ln] : 0.f;
            acc2 = SW_MFMA(av, bv, acc2);
          }
        }
        const int j = 16 * at + ln;
        if (j < n) {
          float* q = dh + (size_t)(s0 + j) * 64 + 16 * wave + 4 * lg;
          st4(q, ld4(q) + (acc + acc2));
        }
      }
    }
    SW_STAMP(7);
  }
  // ---- epilogue: this workgroup's partial = sum of its 4 waves, in a fixed order through LDS ------
  pair_grad_store(G, w2t, part, blockIdx.x, wave, ln, lg);
}

// ---- variant for SMALL scenes: per-pair rows + deferred GEMM --------------------------------------
// With a handful of pair tiles per scene (8-agent scenes: one tile per wave) the in-register
// accumulation above cannot amortise its fixed costs (weight staging, the 4-wave partial reduction, a
// 25 KB partial per scene); there the backward leaves per-pair rows (h2, dh2 [64], h1, dh1 [32], feat [4] = 196
// floats per pair) and per-AGENT rows (Wh_j, Q_j [64], sd_j) for the grouped weight-gradient GEMM of sw_wgrad.hip:
// dW3 = sum_j Wh_j Q_j^T and db3 = sum_j Wh_j sd_j run over B rows, not over P (pair_block_dw3 has the algebra).
struct PairRows {
  float *h2, *dh2, *h1, *dh1, *feat;
};
__host__ __device__ inline PairRows pair_rows(float* base, long long P) {
  PairRows r;
  r.h2 = base;
  r.dh2 = r.h2 + 64 * P;
  r.h1 = r.dh2 + 64 * P;
  r.dh1 = r.h1 + 32 * P;
  r.feat = r.dh1 + 32 * P;
  return r;
}
#define SW_PAIR_ROW_FLOATS 196
#define SW_AGENT_ROW_FLOATS 200      // per agent in the pair workspace: dWh | Wh | Q [64 each] | sd [4] | pad [4]

__global__ __launch_bounds__(SW_THREADS) void social_pool_bwd_rows_kernel(
    const float* __restrict__ obsv, int To, const float* __restrict__ h, const int* __restrict__ scene_off,
    const long long* __restrict__ pair_off, const float* __restrict__ emb_w, const float* __restrict__ att_w,
    const float* __restrict__ attn, const float* __restrict__ dS, float* __restrict__ dh,
    float* __restrict__ dwh_rows, float* __restrict__ wh_rows, float* __restrict__ q_rows, float* __restrict__ sd_rows,
    PairRows pr, int a16, const float* __restrict__ simg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SocL Ls = soc_lds(a16);
  const int sa = Ls.sa;
  float* hs = smem + Ls.hs;
  float* wh = smem + Ls.wh;
  float* x4 = smem + Ls.x4;
  float* sig = smem + Ls.sig;  // attention weights a_ij
  const float* w0b = smem + Ls.w0b;
  const float* b12 = smem + Ls.b12;
  float* dsl = smem + Ls.ds;
  float* dsg = smem + Ls.dsg;
  float* dwh = smem + Ls.dwh;
  float* vv = dwh;             // v_j = W3^T Wh_j [a16][68] during the pair loop (dWh is formed after it)
  float* qq = hs;              // Q_j | sd_j [a16][68] behind the pair loop (h is last read by the softmax backward)
  const int s0 = scene_off[blockIdx.x], n = scene_off[blockIdx.x + 1] - s0;
  if (n <= 0 || n > SW_AMAX) return;
  if (n == 1) {  // S = 0 constant: no gradient anywhere; the agent's rows are zero
    if (threadIdx.x < 16) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      st4(dwh_rows + (size_t)s0 * 64 + 4 * threadIdx.x, z);
      st4(wh_rows + (size_t)s0 * 64 + 4 * threadIdx.x, z);
      st4(q_rows + (size_t)s0 * 64 + 4 * threadIdx.x, z);
      if (threadIdx.x == 0) st4(sd_rows + (size_t)s0 * 4, z);
    }
    return;
  }
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const long long p0 = pair_off[blockIdx.x];
#ifdef SW_PHASE_STAMPS
  long long _tprev = clock64();
#endif
  // the pair-MLP weights (registers, L2 latency) are requested first: they arrive under the scene prologue
  PairW1 W;
  f32x4 w1T[2][4];  // fc.2.weight^T: [jt][mt][r] = W1[16mt + 4lg + r][16jt + ln]
  f32x4 wT[4];      // attention W^T for the dh rows: W[k = 16kt + 4lg + r][u = 16 wave + ln]
  f32x4 w3t[4];     // fc.4.weight^T rows of this wave's units (v_j): W3[m = 16mo + 4lg + r][k = 16 wave + ln]
  f32x4 w2own[4];   // fc.4.weight rows of this wave's units (dWh_j): W3[m = 16 wave + ln][k = 16kt + 4lg + r]
  if (simg) {       // operand-layout images of the step: every load instruction reads 1 KB of consecutive memory
    load_pair_w1_img(W, simg, lane);
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) {
      w1T[0][mo] = ld4(simg + swimg::OP_E1T + ((0 * 4 + mo) * 64 + lane) * 4);
      w1T[1][mo] = ld4(simg + swimg::OP_E1T + ((1 * 4 + mo) * 64 + lane) * 4);
      w3t[mo] = ld4(simg + swimg::OP_E2T + ((wave * 4 + mo) * 64 + lane) * 4);
      w2own[mo] = ld4(simg + swimg::OP_E2 + ((wave * 4 + mo) * 64 + lane) * 4);
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) wT[kt] = ld4(simg + swimg::OP_ATT_T + ((wave * 4 + kt) * 64 + lane) * 4);
  } else {
    load_pair_w1(W, emb_w, ln, lg);
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* row1 = emb_w + swp::EMB_W1 + (16 * mo + 4 * lg + r) * 32 + ln;
        w1T[0][mo][r] = row1[0];
        w1T[1][mo][r] = row1[16];
        w3t[mo][r] = emb_w[swp::EMB_W2 + (16 * mo + 4 * lg + r) * 64 + 16 * wave + ln];
      }
      w2own[mo] = ld4(emb_w + swp::EMB_W2 + (16 * wave + ln) * 64 + 16 * mo + 4 * lg);
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) wT[kt][r] = att_w[swp::ATT_W + (16 * kt + 4 * lg + r) * 64 + 16 * wave + ln];
    }
  }
  const f32x4 b3own = ld4(emb_w + swp::EMB_B2 + 16 * wave + 4 * lg);
  scene_prologue(smem, Ls, obsv, To, h, emb_w, att_w, s0, n);
  for (int i = threadIdx.x; i < n * 16; i += blockDim.x) {
    int a = i >> 4, q = i & 15;
    st4(&dsl[a * 68 + 4 * q], ld4(dS + (size_t)(s0 + a) * 64 + 4 * q));
    st4(wh_rows + (size_t)(s0 + a) * 64 + 4 * q, ld4(&wh[a * 68 + 4 * q]));     // Wh rows for dW3 / db3 (deferred GEMM)
  }
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    int i = e / n, j = e - i * n;
    sig[i * sa + j] = attn[(size_t)(s0 + i) * SW_AMAX + j];
  }
  const int nt = (n + 15) >> 4;
  for (int at = 0; at < nt; ++at) {     // v_j = W3^T Wh_j: wave w the units 16w .. 16w+15
    const f32x4 acc = tile_mm_reg<4>(w3t, &wh[(16 * at + ln) * 68 + 4 * lg], f32x4{0.f, 0.f, 0.f, 0.f});
    st4(&vv[(16 * at + ln) * 68 + 16 * wave + 4 * lg], acc);
  }
  sw_barrier();
  SW_STAMP(0);
  // da_ij = <dS_i, h_j>;  dsigma_ij = a_ij (da_ij - sum_j' a_ij' da_ij')   (softmax backward)
  for (int i = wave; i < n; i += 4) {
    float da = 0.f, a = 0.f;
    if (lane < n) {
      a = sig[i * sa + lane];
      for (int u = 0; u < 64; u += 4) {
        f32x4 x = ld4(&dsl[i * 68 + u]), y = ld4(&hs[lane * 68 + u]);
        da = fmaf(x[0], y[0], da); da = fmaf(x[1], y[1], da); da = fmaf(x[2], y[2], da); da = fmaf(x[3], y[3], da);
      }
    }
    float t = a * da;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane < n) dsg[i * sa + lane] = a * (da - t);
  }
  sw_barrier();
  SW_STAMP(1);
  // ---- pair tiles: recompute the MLP up to h2, back-propagate, leave rows for the deferred GEMMs --------
  float w0a[2];
  pair_l1_load(w0b, ln, lg, w0a);
  const int P = n * n;
  for (int pt = wave; pt * 16 < P; pt += 4) {
    const bool valid = pt * 16 + ln < P;
    int p = min(pt * 16 + ln, P - 1);
    int i = p / n, j = p - i * n;
    float f0, f1, f2;
    pair_feat(ld4(&x4[i * 4]), ld4(&x4[j * 4]), f0, f1, f2);
    f32x4 h1[2], h2[4];
    pair_l1m(w0a, lg, f0, f1, f2, h1);
    pair_l2(W, b12, lg, h1, h2);
    const float dsv = valid ? dsg[i * sa + j] : 0.f;
    f32x4 dh2[4];     // relu'(h2) . dsigma_ij v_j
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const f32x4 v = ld4(&vv[j * 68 + 16 * mt + 4 * lg]);
#pragma unroll
      for (int r = 0; r < 4; ++r) dh2[mt][r] = h2[mt][r] > 0.f ? dsv * v[r] : 0.f;
    }
    f32x4 dh1[2];
    dh1[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    dh1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dh1[0] = SW_MFMA(w1T[0][mt][r], dh2[mt][r], dh1[0]);
        dh1[1] = SW_MFMA(w1T[1][mt][r], dh2[mt][r], dh1[1]);
      }
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dh1[jt][r] = h1[jt][r] > 0.f ? dh1[jt][r] : 0.f;
    }
    if (valid) {
      const size_t row = (size_t)(p0 + p);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        st4(pr.h2 + row * 64 + 16 * mt + 4 * lg, h2[mt]);
        st4(pr.dh2 + row * 64 + 16 * mt + 4 * lg, dh2[mt]);
      }
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        st4(pr.h1 + row * 32 + 16 * jt + 4 * lg, h1[jt]);
        st4(pr.dh1 + row * 32 + 16 * jt + 4 * lg, dh1[jt]);
      }
      if (lg == 0) st4(pr.feat + row * 4, f32x4{f0, f1, f2, 0.f});
    }
  }
  SW_STAMP(2);
  __syncthreads();  // pair rows of this scene are visible to the whole workgroup (same CU)
  SW_STAMP(3);
  // Q_j = sum_i dsigma_ij h2_ij, sd_j = sum_i dsigma_ij (reads the h2 rows back)
  for (int e = threadIdx.x; e < n * 64; e += blockDim.x) {
    int j = e >> 6, u = e & 63;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc = fmaf(dsg[i * sa + j], pr.h2[(size_t)(p0 + i * n + j) * 64 + u], acc);
    qq[j * 68 + u] = acc;
    q_rows[(size_t)(s0 + j) * 64 + u] = acc;
  }
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += dsg[i * sa + j];
    qq[j * 68 + 64] = acc;
    st4(sd_rows + (size_t)(s0 + j) * 4, f32x4{acc, 0.f, 0.f, 0.f});
  }
  sw_barrier();
  // dWh_j = sum_i dsigma_ij f_ij = W3 Q_j + b3 sd_j: wave w the units 16w .. 16w+15
  for (int at = 0; at < nt; ++at) {
    f32x4 out = tile_mm_reg<4>(w2own, &qq[(16 * at + ln) * 68 + 4 * lg], f32x4{0.f, 0.f, 0.f, 0.f});
    const int j = 16 * at + ln;
    const float sds = qq[min(j, n - 1) * 68 + 64];
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = fmaf(b3own[r], sds, out[r]);
    if (j < n) {
      st4(&dwh[j * 68 + 16 * wave + 4 * lg], out);
      st4(dwh_rows + (size_t)(s0 + j) * 64 + 16 * wave + 4 * lg, out);
    }
  }
  sw_barrier();
  SW_STAMP(6);
  // dh_j += sum_i a_ij dS_i  +  W^T dWh_j  on the matrix cores (see social_pool_bwd_kernel)
  {
    const int npad = (n + 15) & ~15;
    for (int at = 0; at < npad / 16; ++at) {
      f32x4 acc = tile_mm_reg<4>(wT, &dwh[(16 * at + ln) * 68 + 4 * lg], f32x4{0.f, 0.f, 0.f, 0.f});
      f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
      for (int kt = 0; kt < npad / 16; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * kt + 4 * lg + r;
          const float av = i < n ? dsl[i * 68 + 16 * wave + ln] : 0.f;
          const float bv = i < n ? sig[i * sa + 16 * at + ln] : 0.f;
          acc2 = SW_MFMA(av, bv, acc2);
        }
      }
      const int j = 16 * at + ln;
      if (j < n) {
        float* q = dh + (size_t)(s0 + j) * 64 + 16 * wave + 4 * lg;
        st4(q, ld4(q) + (acc + acc2));
      }
    }
  }
  SW_STAMP(7);
}

// ---------------------------------------------------------------------------------------------
// Scenes with more than SW_AMAX agents: ROW-BLOCK kernels.  The scene no longer fits one workgroup's LDS
// (scores are n x n), so the work unit is a block of 16 query agents i of one scene: each of the 4 waves
// owns 4 rows and walks the scene's agents j in tiles of 16 pairs (i fixed, 16 consecutive j) with an
// ONLINE softmax (running max m, normaliser l, weighted sum of h_j) - no n x n array exists anywhere.
// W h_j + b is precomputed per agent (social_wh_kernel).  The backward recomputes sigma_ij and uses the saved
// (m_i, l_i).  Same mathematics as AttentionPooling.forward (train.py:153-175) for any scene size.
// big_blocks: int32 [NB][8] = { scene, i0, partial row base of this block, partial row base of the scene,
//             blocks in the scene, block index in the scene, 0, 0 }.
// ---------------------------------------------------------------------------------------------
#define SW_BIG_REC 8
#define SW_BIG_PROW 132   // floats per partial row of the row-block backward: Q_j [64] | sum_i a_ij dS_i [64] | sd_j, pad [4]
__device__ __forceinline__ f32x4 agent_x4(const float* obsv, int To, int a) {
  const float* p = obsv + ((size_t)a * To + To - 2) * 2;  // last two observed points -> (p, v)
  return f32x4{p[2], p[3], p[2] - p[0], p[3] - p[1]};
}

// Wh = W h + b for the 16 agents of a block (one MFMA tile per wave: units 16w..), and what the attention needs of fc.4:
// v_j = W3^T Wh_j, c_j = <b3, Wh_j> (scene_wh_to_v has the algebra).  wh / vv [B][64], cc [B].
#define SW_BIG_WH_FLOATS 132      // per agent in wh_ws: Wh [64] | v [64] | c [1] + pad (three arrays of B rows)
__global__ __launch_bounds__(SW_THREADS) void social_wh_kernel(const float* __restrict__ h, const int* __restrict__ scene_off,
                                                               const int* __restrict__ blocks, const float* __restrict__ att_w,
                                                               const float* __restrict__ emb_w, float* __restrict__ wh,
                                                               float* __restrict__ vv, float* __restrict__ cc) {
  __shared__ __attribute__((aligned(16))) float tile[16][68];
  const int* rec = blocks + (size_t)blockIdx.x * SW_BIG_REC;
  const int s0 = scene_off[rec[0]], n = scene_off[rec[0] + 1] - s0, i0 = rec[1];
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int a = s0 + min(i0 + ln, n - 1);
  f32x4 wr[4], w3t[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    wr[j] = ld4(att_w + swp::ATT_W + (16 * wave + ln) * 64 + 16 * j + 4 * lg);
#pragma unroll
    for (int r = 0; r < 4; ++r) w3t[j][r] = emb_w[swp::EMB_W2 + (16 * j + 4 * lg + r) * 64 + 16 * wave + ln];
  }
  f32x4 acc = ld4(att_w + swp::ATT_B + 16 * wave + 4 * lg);
  acc = tile_mm_reg<4>(wr, h + (size_t)a * 64 + 4 * lg, acc);
  if (i0 + ln < n) st4(wh + (size_t)a * 64 + 16 * wave + 4 * lg, acc);
  st4(&tile[ln][16 * wave + 4 * lg], acc);
  __syncthreads();
  const f32x4 v = tile_mm_reg<4>(w3t, &tile[ln][4 * lg], f32x4{0.f, 0.f, 0.f, 0.f});
  if (i0 + ln < n) st4(vv + (size_t)a * 64 + 16 * wave + 4 * lg, v);
  if (threadIdx.x < 16 && i0 + (int)threadIdx.x < n) {
    float c = 0.f;
    for (int u = 0; u < 64; ++u) c = fmaf(emb_w[swp::EMB_B2 + u], tile[threadIdx.x][u], c);
    cc[s0 + i0 + threadIdx.x] = c;
  }
}

__global__ __launch_bounds__(SW_THREADS) void social_big_fwd_kernel(
    const float* __restrict__ obsv, int To, const float* __restrict__ h, const float* __restrict__ vv,
    const float* __restrict__ cc, const int* __restrict__ scene_off, const int* __restrict__ blocks,
    const float* __restrict__ emb_w, float* __restrict__ S_out, float* __restrict__ ml) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SocL Ls = soc_lds(16);
  const float* w0b = smem + Ls.w0b;
  const float* b12 = smem + Ls.b12;
  stage_pair_consts(smem, Ls, emb_w);
  const int* rec = blocks + (size_t)blockIdx.x * SW_BIG_REC;
  const int s0 = scene_off[rec[0]], n = scene_off[rec[0] + 1] - s0, i0 = rec[1];
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  PairW1 W;
  load_pair_w1(W, emb_w, ln, lg);
  sw_barrier();
  for (int q = 0; q < 4; ++q) {
    const int i = i0 + 4 * wave + q;
    if (i >= n) break;
    const f32x4 xi = agent_x4(obsv, To, s0 + i);
    float m = -INFINITY, l = 0.f, acc = 0.f;   // acc: unit `lane` of sum_j e^(sigma_ij - m) h_j
    for (int j0 = 0; j0 < n; j0 += 16) {
      const int j = j0 + ln;
      const bool jv = j < n;
      const int jc = min(j, n - 1);
      float f0, f1, f2;
      pair_feat(xi, agent_x4(obsv, To, s0 + jc), f0, f1, f2);
      f32x4 h1[2], h2[4];
      pair_l1(w0b, lg, f0, f1, f2, h1);
      pair_l2(W, b12, lg, h1, h2);
      float part = 0.f;     // sigma_ij = <h2_ij, v_j> + c_j (fc.4 is never run per pair)
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) {
        f32x4 w = ld4(vv + (size_t)(s0 + jc) * 64 + 16 * mo + 4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) part = fmaf(h2[mo][r], w[r], part);
      }
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      part += cc[s0 + jc];
      const float sg = !jv ? -INFINITY : (jc == i ? -1000.0f : part);   // train.py:170
      float tmax = sg;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
      const float mn = fmaxf(m, tmax);
      const float keep = expf(m - mn);              // 0 on the first tile (m = -inf)
      const float pj = jv ? expf(sg - mn) : 0.f;
      float tsum = pj;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) tsum += __shfl_xor(tsum, o);
      l = fmaf(l, keep, tsum);
      acc *= keep;
      const int jn = min(16, n - j0);
      for (int jj = 0; jj < jn; ++jj) acc = fmaf(__shfl(pj, jj), h[(size_t)(s0 + j0 + jj) * 64 + lane], acc);
      m = mn;
    }
    S_out[(size_t)(s0 + i) * 64 + lane] = acc / l;
    if (ml && lane == 0) {
      ml[(size_t)(s0 + i) * 2] = m;
      ml[(size_t)(s0 + i) * 2 + 1] = l;
    }
  }
}

// layer 2 alone with fc.2.weight read from its row-major LDS image
__device__ __forceinline__ void pair_l2_lds(const float* w1s, const float* b1, int ln, int lg, const f32x4 h1[2], f32x4 h2[4]) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    f32x4 acc = ld4(b1 + 16 * mt + 4 * lg);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 w = ld4(w1s + (16 * mt + ln) * SW_SOC_W1LD + 16 * j + 4 * lg);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = SW_MFMA(w[r], h1[j][r], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) h2[mt][r] = fmaxf(acc[r], 0.f);
  }
}
// Backward of one 16-query-agent block of a large scene.  For every j tile (outer loop) the wave visits its
// <= 4 rows i (inner loop): recompute f_ij and sigma_ij, a_ij = e^(sigma_ij - m_i) / l_i,
// dsigma_ij = a_ij (<dS_i, h_j> - <dS_i, S_i>), back-propagate the pair MLP and accumulate its weight gradients
// in registers (pair_tile_bwd).  The two sums over i that belong to agent j,
//   dWh_j = sum_i dsigma_ij f_ij      and      sum_i a_ij dS_i   (part of dh_j),
// are lane-local over the wave's rows (pair lane = j), summed over the 4 waves through LDS and left as one
// partial row (128 floats) per (block, j); social_big_finish_kernel adds the blocks of a scene.
__global__ __launch_bounds__(SW_THREADS) void social_big_bwd_kernel(
    const float* __restrict__ obsv, int To, const float* __restrict__ h, const float* __restrict__ wh,
    const float* __restrict__ vv, const float* __restrict__ cc, const int* __restrict__ scene_off,
    const int* __restrict__ blocks, const float* __restrict__ emb_w, const float* __restrict__ S_pool,
    const float* __restrict__ ml, const float* __restrict__ dS, float* __restrict__ part_rows, SocPart part, int slice0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SocL Ls = soc_lds(16);
  const float* w0b = smem + Ls.w0b;
  const float* b12 = smem + Ls.b12;
  float* w2s = smem + Ls.fwd_total;                   // fc.4.weight [64][68]
  float* w1s = w2s + 64 * SW_SOC_W2LD;                // fc.2.weight [64][36]
  float* scr_all = w1s + 64 * SW_SOC_W1LD;            // [4 waves][4 tiles][16][20]
  float* jred = scr_all + 4 * SW_SOC_SCR;             // [16 j][SW_BIG_PROW]: Q | sum a dS | sd of the current j tile
  float* w2t = jred + 16 * SW_BIG_PROW;               // fc.4.weight^T [64][68] | fc.2.weight^T [32][68]
  float* w1t = w2t + 64 * SW_SOC_WTLD;
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  float* scr = scr_all + wave * SW_SOC_SCR;
  stage_pair_consts(smem, Ls, emb_w);
  for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {
    int r = i >> 4, q = i & 15;
    st4(&w2s[r * SW_SOC_W2LD + 4 * q], ld4(emb_w + swp::EMB_W2 + r * 64 + 4 * q));
    if (q < 8) st4(&w1s[r * SW_SOC_W1LD + 4 * q], ld4(emb_w + swp::EMB_W1 + r * 32 + 4 * q));
  }
  stage_pair_wt(w2t, w1t, emb_w);
#ifdef SW_PHASE_STAMPS
  long long _tprev = clock64();
#endif
  const int* rec = blocks + (size_t)blockIdx.x * SW_BIG_REC;
  const int s0 = scene_off[rec[0]], n = scene_off[rec[0] + 1] - s0, i0 = rec[1], prow0 = rec[2];
  PairGrad G;
  pair_grad_zero(G);
  // the wave's rows: position/velocity, softmax statistics, D_i = <dS_i, S_i>
  const int nrows = max(0, min(4, n - (i0 + 4 * wave)));
  f32x4 xi[4];
  float mi[4], rli[4], Di[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int a = s0 + min(i0 + 4 * wave + q, n - 1);
    xi[q] = agent_x4(obsv, To, a);
    mi[q] = ml[(size_t)a * 2];
    rli[q] = 1.0f / ml[(size_t)a * 2 + 1];
    float d = dS[(size_t)a * 64 + lane] * S_pool[(size_t)a * 64 + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    Di[q] = d;
  }
  sw_barrier();
  for (int j0 = 0; j0 < n; j0 += 16) {
    const int j = j0 + ln;
    const bool jv = j < n;
    const int jc = min(j, n - 1);
    const f32x4 xj = agent_x4(obsv, To, s0 + jc);
    f32x4 vj[4], hj[4], Q[4], dhj_acc[4];     // Q_j = sum_i dsigma_ij h2_ij (-> dW3, dWh_j), see pair_block_dw3
    float sd = 0.f;
    const float cj = cc[s0 + jc];
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) {
      vj[mo] = ld4(vv + (size_t)(s0 + jc) * 64 + 16 * mo + 4 * lg);
      hj[mo] = ld4(h + (size_t)(s0 + jc) * 64 + 16 * mo + 4 * lg);
      Q[mo] = f32x4{0.f, 0.f, 0.f, 0.f};
      dhj_acc[mo] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int q = 0; q < nrows; ++q) {
      const int i = i0 + 4 * wave + q;
      const f32x4 xq = q == 0 ? xi[0] : (q == 1 ? xi[1] : (q == 2 ? xi[2] : xi[3]));
      const float mq = q == 0 ? mi[0] : (q == 1 ? mi[1] : (q == 2 ? mi[2] : mi[3]));
      const float rlq = q == 0 ? rli[0] : (q == 1 ? rli[1] : (q == 2 ? rli[2] : rli[3]));
      const float Dq = q == 0 ? Di[0] : (q == 1 ? Di[1] : (q == 2 ? Di[2] : Di[3]));
      float f0, f1, f2;
      pair_feat(xq, xj, f0, f1, f2);
      f32x4 h1[2], h2[4];
      pair_l1(w0b, lg, f0, f1, f2, h1);
      pair_l2_lds(w1s, b12, ln, lg, h1, h2);
      float part_s = 0.f, da = 0.f;
      f32x4 dsi[4];
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) {
        dsi[mo] = ld4(dS + (size_t)(s0 + i) * 64 + 16 * mo + 4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          part_s = fmaf(h2[mo][r], vj[mo][r], part_s);
          da = fmaf(dsi[mo][r], hj[mo][r], da);
        }
      }
      part_s += __shfl_xor(part_s, 16);
      part_s += __shfl_xor(part_s, 32);
      da += __shfl_xor(da, 16);
      da += __shfl_xor(da, 32);
      const float sg = jc == i ? -1000.0f : part_s + cj;                  // train.py:170
      const float a = jv ? expf(sg - mq) * rlq : 0.f;
      const float dsv = a * (da - Dq);                                     // softmax backward; 0 for invalid lanes
      f32x4 dh2[4];
      sd += dsv;
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Q[mo][r] = fmaf(dsv, h2[mo][r], Q[mo][r]);
          dh2[mo][r] = h2[mo][r] > 0.f ? dsv * vj[mo][r] : 0.f;
          dhj_acc[mo][r] = fmaf(a, dsi[mo][r], dhj_acc[mo][r]);
        }
      }
      pair_tile_bwd_tail(G, scr, w1t, h1, dh2, f0, f1, f2, ln, lg SW_STAMP_ARG);
    }
    {     // dW3 += Wh Q^T, db3 += Wh sd over the block's 16 agents (this wave's rows)
      f32x4 whj[4];
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) whj[mo] = ld4(wh + (size_t)(s0 + jc) * 64 + 16 * mo + 4 * lg);
      pair_block_dw3(G, scr, whj, Q, sd, ln, lg);
      wave_lds_fence();
    }
    // sum the 4 waves' j-tile partials in a fixed order, then one partial row per agent j of the tile
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) {
          float* a0 = jred + ln * SW_BIG_PROW + 16 * mo + 4 * lg;
          st4(a0, w == 0 ? Q[mo] : ld4(a0) + Q[mo]);
          st4(a0 + 64, w == 0 ? dhj_acc[mo] : ld4(a0 + 64) + dhj_acc[mo]);
        }
        if (lg == 0) {
          float* a1 = jred + ln * SW_BIG_PROW + 128;
          *a1 = w == 0 ? sd : *a1 + sd;
        }
      }
      sw_barrier();
    }
    const int jn = min(16, n - j0);
    for (int e = threadIdx.x; e < jn * 33; e += blockDim.x) {
      const int jr = e / 33, c4 = e - jr * 33;     // 132 floats per row: Q [64] | sum a dS [64] | sd, pad [4]
      st4(part_rows + ((size_t)prow0 + j0 + jr) * SW_BIG_PROW + 4 * c4, ld4(jred + jr * SW_BIG_PROW + 4 * c4));
    }
    sw_barrier();
  }
  pair_grad_store(G, w2s, part, slice0 + blockIdx.x, wave, ln, lg);
}

// Per 16 agents j of a large scene: dWh_j = sum over the scene's blocks, dh_j += sum_blocks (sum_i a_ij dS_i) + W^T dWh_j
__global__ __launch_bounds__(SW_THREADS) void social_big_finish_kernel(const int* __restrict__ scene_off,
                                                                       const int* __restrict__ blocks,
                                                                       const float* __restrict__ att_w,
                                                                       const float* __restrict__ emb_w,
                                                                       const float* __restrict__ part_rows,
                                                                       float* __restrict__ dh, float* __restrict__ dwh_rows) {
  __shared__ float dwh[16][68], qs[16][68];
  const int* rec = blocks + (size_t)blockIdx.x * SW_BIG_REC;
  const int s0 = scene_off[rec[0]], n = scene_off[rec[0] + 1] - s0, j0 = rec[1], srow0 = rec[3], nblk = rec[4];
  const int jn = min(16, n - j0);
  float dhj[4] = {0.f, 0.f, 0.f, 0.f};
  for (int q = 0, e = threadIdx.x; q < 4; ++q, e += 256) {   // 16 x 64 elements, 4 per thread: Q_j, sum a dS over the blocks
    const int jr = e >> 6, u = e & 63;
    float sw = 0.f, sh = 0.f, ss = 0.f;
    if (jr < jn) {
      for (int k = 0; k < nblk; ++k) {
        const float* row = part_rows + ((size_t)srow0 + (size_t)k * n + j0 + jr) * SW_BIG_PROW;
        sw += row[u];
        sh += row[64 + u];
        if (u == 0) ss += row[128];
      }
    }
    qs[jr][u] = sw;
    if (u == 0) qs[jr][64] = ss;
    dhj[q] = sh;
  }
  __syncthreads();
  for (int q = 0, e = threadIdx.x; q < 4; ++q, e += 256) {   // dWh_j = W3 Q_j + b3 sd_j (pair_block_dw3 has the algebra)
    const int jr = e >> 6, u = e & 63;
    float acc = emb_w[swp::EMB_B2 + u] * qs[jr][64];
    const float* w3 = emb_w + swp::EMB_W2 + u * 64;
    for (int k = 0; k < 64; ++k) acc = fmaf(w3[k], qs[jr][k], acc);
    dwh[jr][u] = acc;
    if (jr < jn) dwh_rows[(size_t)(s0 + j0 + jr) * 64 + u] = acc;
  }
  __syncthreads();
  for (int q = 0, e = threadIdx.x; q < 4; ++q, e += 256) {
    const int jr = e >> 6, u = e & 63;
    if (jr >= jn) continue;
    float acc = dhj[q];
    const float* wc = att_w + swp::ATT_W + u;
    for (int k = 0; k < 64; ++k) acc = fmaf(wc[k * 64], dwh[jr][k], acc);
    dh[(size_t)(s0 + j0 + jr) * 64 + u] += acc;
  }
}

// dense SocialFeatures for the module-level API (train.py:229-241): feat[i][j][0..2]
__global__ void social_features_kernel(const float* __restrict__ x4, int B, float* __restrict__ feat) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)B * B) return;
  int i = e / B, j = e - (size_t)i * B;
  float f0, f1, f2;
  pair_feat(ld4(x4 + (size_t)i * 4), ld4(x4 + (size_t)j * 4), f0, f1, f2);
  feat[e * 3 + 0] = f0;
  feat[e * 3 + 1] = f1;
  feat[e * 3 + 2] = f2;
}


// SocialFeatures on the ordered IN-SCENE pairs only (the block-diagonal of train.py:229-241), as rows for the generic-width
// path (sw_generic.hip): feat [P][4] = (dist, bearing, dca, 0) at row pair_off[s] + i_local n + j_local; single-agent
// scenes own no rows.  x4_last [B][4] = (p, v) of the last observed step.
__global__ __launch_bounds__(256) void pair_features_kernel(const float* __restrict__ x4, const int* __restrict__ scene_off,
                                                            const long long* __restrict__ pair_off, int S,
                                                            float* __restrict__ feat) {
  const int s = blockIdx.x;
  const int s0 = scene_off[s], n = scene_off[s + 1] - s0;
  if (n <= 1) return;
  const long long p0 = pair_off[s];
  for (int e = threadIdx.x; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    float f0, f1, f2;
    pair_feat(ld4(x4 + (size_t)(s0 + i) * 4), ld4(x4 + (size_t)(s0 + j) * 4), f0, f1, f2);
    st4(feat + (size_t)(p0 + e) * 4, f32x4{f0, f1, f2, 0.f});
  }
}
extern "C" int sw_pair_features(const float* x4_last, const int* scene_off, const long long* pair_off, int S, float* feat,
                                void* stream) {
  if (!x4_last || !scene_off || !pair_off || !feat || S < 1) return SW_EARG;
  SW_LAUNCH(pair_features_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, x4_last, scene_off, pair_off, S, feat);
  SW_CHECK_LAUNCH("pair_features_kernel");
  return SW_OK;
}

static int set_lds(const void* fn, int bytes) {
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    sw_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
    return SW_EHIP;
  }
  return SW_OK;
}

// ---- module-level API helpers (dense layouts of the reference, small batches) -----------------
// AttentionPooling.forward on a dense (B,B,64) embedding tensor: only in-scene blocks are read.
__global__ __launch_bounds__(SW_THREADS) void attention_pool_dense_kernel(
    const float* __restrict__ f, const float* __restrict__ h, const int* __restrict__ scene_off, int B,
    const float* __restrict__ att_w, float* __restrict__ S_out, int a16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SocL Ls = soc_lds(a16);
  const int sa = Ls.sa;
  float* wh = smem + Ls.wh;
  float* sig = smem + Ls.sig;
  const int s0 = scene_off[blockIdx.x], n = scene_off[blockIdx.x + 1] - s0;
  if (n <= 0) return;
  if (n == 1) {
    if (threadIdx.x < 16) st4(S_out + (size_t)s0 * 64 + 4 * threadIdx.x, f32x4{0.f, 0.f, 0.f, 0.f});
    return;
  }
  scene_load_h_wh(smem, Ls, h, att_w, s0, n);
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    int i = e / n, j = e - i * n;
    const float* fr = f + ((size_t)(s0 + i) * B + (s0 + j)) * 64;
    float acc = 0.f;
    for (int u = 0; u < 64; u += 4) {
      f32x4 x = ld4(fr + u), y = ld4(&wh[j * 68 + u]);
      acc = fmaf(x[0], y[0], acc); acc = fmaf(x[1], y[1], acc); acc = fmaf(x[2], y[2], acc); acc = fmaf(x[3], y[3], acc);
    }
    sig[i * sa + j] = (i == j) ? -1000.0f : acc;
  }
  sw_barrier();
  scene_softmax_pool(smem, Ls, s0, n, S_out, nullptr);
}

// EmbedSocialFeatures.forward on R rows of 3 features (one wave per 16 rows).
__global__ __launch_bounds__(SW_THREADS) void embed_features_kernel(const float* __restrict__ feat, long long R,
                                                                    const float* __restrict__ emb_w,
                                                                    float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SocL Ls = soc_lds(16);
  const float* w0b = smem + Ls.w0b;
  const float* b12 = smem + Ls.b12;
  stage_pair_consts(smem, Ls, emb_w);
  sw_barrier();
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  PairW W;
  load_pair_w(W, emb_w, ln, lg);
  long long r0 = ((long long)blockIdx.x * 4 + wave) * 16;
  if (r0 >= R) return;
  long long row = r0 + ln < R ? r0 + ln : R - 1;
  float f0 = feat[row * 3], f1 = feat[row * 3 + 1], f2 = feat[row * 3 + 2];
  f32x4 h1[2], h2[4], f[4];
  pair_l1(w0b, lg, f0, f1, f2, h1);
  pair_l23(W, b12, b12 + 64, lg, h1, h2, f);
  if (r0 + ln < R) {
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) st4(out + row * 64 + 16 * mo + 4 * lg, f[mo]);
  }
}

// Backward of EmbedSocialFeatures.forward on R rows (one wave per 16 rows): the MLP is recomputed on the matrix cores,
// dout is back-propagated to the hidden layers; rows = h2 [R][64] | dh2 [R][64] | h1 [R][32] | dh1 [R][32] | feat4 [R][4]
// are what the weight-gradient GEMMs contract over (sw_linear_wgrad: dout x h2, dh2 x h1, dh1 x feat4); dfeat [R][3]
// (optional) = fc.0.weight^T dh1.
__global__ __launch_bounds__(SW_THREADS) void embed_features_bwd_kernel(const float* __restrict__ feat, long long R,
                                                                        const float* __restrict__ emb_w,
                                                                        const float* __restrict__ dout,
                                                                        float* __restrict__ rows, float* __restrict__ dfeat) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SocL Ls = soc_lds(16);
  const float* w0b = smem + Ls.w0b;
  const float* b12 = smem + Ls.b12;
  stage_pair_consts(smem, Ls, emb_w);
  sw_barrier();
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  PairW W;
  load_pair_w(W, emb_w, ln, lg);
  f32x4 w2T[4][4], w1T[2][4];   // fc.4.weight^T, fc.2.weight^T as A operands (see social_pool_bwd_rows_kernel)
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* row2 = emb_w + swp::EMB_W2 + (16 * mo + 4 * lg + r) * 64 + ln;
      const float* row1 = emb_w + swp::EMB_W1 + (16 * mo + 4 * lg + r) * 32 + ln;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) w2T[mt][mo][r] = row2[16 * mt];
      w1T[0][mo][r] = row1[0];
      w1T[1][mo][r] = row1[16];
    }
  }
  const long long r0 = ((long long)blockIdx.x * 4 + wave) * 16;
  if (r0 >= R) return;
  const bool valid = r0 + ln < R;
  const long long row = valid ? r0 + ln : R - 1;
  const float f0 = feat[row * 3], f1 = feat[row * 3 + 1], f2 = feat[row * 3 + 2];
  f32x4 h1[2], h2[4], fo[4];
  pair_l1(w0b, lg, f0, f1, f2, h1);
  pair_l23(W, b12, b12 + 64, lg, h1, h2, fo);
  f32x4 dz3[4], dh2[4], dh1[2];
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) dz3[mo] = ld4(dout + row * 64 + 16 * mo + 4 * lg);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) dh2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mo = 0; mo < 4; ++mo)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) dh2[mt] = SW_MFMA(w2T[mt][mo][r], dz3[mo][r], dh2[mt]);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) dh2[mt][r] = h2[mt][r] > 0.f ? dh2[mt][r] : 0.f;
  dh1[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  dh1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dh1[0] = SW_MFMA(w1T[0][mt][r], dh2[mt][r], dh1[0]);
      dh1[1] = SW_MFMA(w1T[1][mt][r], dh2[mt][r], dh1[1]);
    }
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) dh1[jt][r] = h1[jt][r] > 0.f ? dh1[jt][r] : 0.f;
  // fc.0.weight^T dh1: this lane's 8 hidden units, then the four lane groups of the row
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 w = ld4(w0b + (16 * jt + 4 * lg + r) * 4);
      g0 = fmaf(w[0], dh1[jt][r], g0);
      g1 = fmaf(w[1], dh1[jt][r], g1);
      g2 = fmaf(w[2], dh1[jt][r], g2);
    }
  g0 += __shfl_xor(g0, 16); g0 += __shfl_xor(g0, 32);
  g1 += __shfl_xor(g1, 16); g1 += __shfl_xor(g1, 32);
  g2 += __shfl_xor(g2, 16); g2 += __shfl_xor(g2, 32);
  if (valid) {
    float* ph2 = rows;
    float* pdh2 = ph2 + 64 * R;
    float* ph1 = pdh2 + 64 * R;
    float* pdh1 = ph1 + 32 * R;
    float* pf4 = pdh1 + 32 * R;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      st4(ph2 + row * 64 + 16 * mt + 4 * lg, h2[mt]);
      st4(pdh2 + row * 64 + 16 * mt + 4 * lg, dh2[mt]);
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      st4(ph1 + row * 32 + 16 * jt + 4 * lg, h1[jt]);
      st4(pdh1 + row * 32 + 16 * jt + 4 * lg, dh1[jt]);
    }
    if (lg == 0) {
      st4(pf4 + row * 4, f32x4{f0, f1, f2, 0.f});
      if (dfeat) {
        dfeat[row * 3] = g0;
        dfeat[row * 3 + 1] = g1;
        dfeat[row * 3 + 2] = g2;
      }
    }
  }
}
extern "C" int sw_embed_features_bwd(const float* feat, long long R, const float* emb_w, const float* dout, float* rows,
                                     float* dfeat, void* stream) {
  if (!feat || !emb_w || !dout || !rows || R < 0) return SW_EARG;
  if (R == 0) return SW_OK;
  static bool attr = false;
  if (!attr) {
    if (int rc = set_lds((const void*)embed_features_bwd_kernel, soc_lds(16).fwd_total * 4)) return rc;
    attr = true;
  }
  SW_LAUNCH(embed_features_bwd_kernel, dim3((unsigned)((R + 63) / 64)), dim3(SW_THREADS), soc_lds(16).fwd_total * 4,
                     (hipStream_t)stream, feat, R, emb_w, dout, rows, dfeat);
  SW_CHECK_LAUNCH("embed_features_bwd_kernel");
  return SW_OK;
}

extern "C" int sw_attention_pool_dense(const float* f, const float* h, const int* scene_off, int S, int B,
                                       const float* att_w, float* S_out, void* stream) {
  if (!f || !h || !scene_off || !att_w || !S_out || S < 0 || B < 0) return SW_EARG;
  if (S == 0 || B == 0) return SW_OK;
  static bool attr = false;
  if (!attr) {
    if (int rc = set_lds((const void*)attention_pool_dense_kernel, soc_lds(SW_AMAX).fwd_total * 4)) return rc;
    attr = true;
  }
  SW_LAUNCH(attention_pool_dense_kernel, dim3(S), dim3(SW_THREADS), soc_lds(SW_AMAX).fwd_total * 4,
                     (hipStream_t)stream, f, h, scene_off, B, att_w, S_out, SW_AMAX);
  SW_CHECK_LAUNCH("attention_pool_dense_kernel");
  return SW_OK;
}

extern "C" int sw_embed_features(const float* feat, long long R, const float* emb_w, float* out, void* stream) {
  if (!feat || !emb_w || !out || R < 0) return SW_EARG;
  if (R == 0) return SW_OK;
  static bool attr = false;
  if (!attr) {
    if (int rc = set_lds((const void*)embed_features_kernel, soc_lds(16).fwd_total * 4)) return rc;
    attr = true;
  }
  SW_LAUNCH(embed_features_kernel, dim3((unsigned)((R + 63) / 64)), dim3(SW_THREADS), soc_lds(16).fwd_total * 4,
                     (hipStream_t)stream, feat, R, emb_w, out);
  SW_CHECK_LAUNCH("embed_features_kernel");
  return SW_OK;
}

extern "C" int sw_social_features(const float* x4_last, int B, float* feat, void* stream) {
  if (!x4_last || !feat || B < 0) return SW_EARG;
  if (B == 0) return SW_OK;
  size_t n = (size_t)B * B;
  SW_LAUNCH(social_features_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x4_last, B, feat);
  SW_CHECK_LAUNCH("social_features_kernel");
  return SW_OK;
}

extern "C" int sw_social_pool_fwd_aux(const float* obsv, int To, const float* h, const int* scene_off, int S, int B,
                                      int Amax, const float* emb_w, const float* att_w, float* S_out, float* attn,
                                      const int* big_blocks, int NB, float* wh_ws, float* ml, const float* aux_src,
                                      float* aux_dst, long long aux_n, void* stream) {
  if (!obsv || !h || !scene_off || !emb_w || !att_w || !S_out || S < 0 || B < 0 || To < 2 || NB < 0) return SW_EARG;
  if (aux_n < 0 || (aux_n & 3) || (aux_n > 0 && (!aux_src || !aux_dst))) return SW_EARG;
  if (NB > 0 && (!big_blocks || !wh_ws)) return SW_EARG;
  if (Amax > SW_AMAX) return SW_ESHAPE;      // Amax = largest scene handled by the one-workgroup-per-scene kernel
  if (S == 0 || B == 0) return SW_OK;
  static bool attr = false;
  if (!attr) {
    if (int rc = set_lds((const void*)social_pool_fwd_kernel, soc_lds(SW_AMAX).fwd_total * 4)) return rc;
    attr = true;
  }
  const int a16 = Amax < 16 ? 16 : ((Amax + 15) & ~15);
  int extra = aux_n > 0 ? (int)((aux_n / 4 + SW_THREADS - 1) / SW_THREADS) : 0;
  if (extra > 64) extra = 64;
  SW_LAUNCH(social_pool_fwd_kernel, dim3(S + extra), dim3(SW_THREADS), soc_lds(a16).fwd_total * 4, (hipStream_t)stream,
                     obsv, To, h, scene_off, emb_w, att_w, S_out, attn, a16, S, aux_src, aux_dst, aux_n,
                     sw_soc_images_for(emb_w, att_w));
  SW_CHECK_LAUNCH("social_pool_fwd_kernel");
  if (NB > 0) {   // scenes above SW_AMAX agents
    // wh_ws: Wh [B][64] | v [B][64] | c [B] (+ pad): SW_BIG_WH_FLOATS per agent
    SW_LAUNCH(social_wh_kernel, dim3(NB), dim3(SW_THREADS), 0, (hipStream_t)stream, h, scene_off, big_blocks, att_w,
                       emb_w, wh_ws, wh_ws + (size_t)B * 64, wh_ws + (size_t)B * 128);
    SW_CHECK_LAUNCH("social_wh_kernel");
    SW_LAUNCH(social_big_fwd_kernel, dim3(NB), dim3(SW_THREADS), soc_lds(16).fwd_total * 4, (hipStream_t)stream,
                       obsv, To, h, wh_ws + (size_t)B * 64, wh_ws + (size_t)B * 128, scene_off, big_blocks, emb_w, S_out, ml);
    SW_CHECK_LAUNCH("social_big_fwd_kernel");
  }
  return SW_OK;
}

extern "C" int sw_social_pool_fwd(const float* obsv, int To, const float* h, const int* scene_off, int S, int B,
                                  int Amax, const float* emb_w, const float* att_w, float* S_out, float* attn,
                                  const int* big_blocks, int NB, float* wh_ws, float* ml, void* stream) {
  return sw_social_pool_fwd_aux(obsv, To, h, scene_off, S, B, Amax, emb_w, att_w, S_out, attn, big_blocks, NB, wh_ws, ml,
                                nullptr, nullptr, 0, stream);
}

extern "C" int sw_social_pool_bwd(const float* obsv, int To, const float* h, const int* scene_off,
                                  const long long* pair_off, int S, int B, int Amax, long long P,
                                  const float* emb_w, const float* att_w, const float* attn, const float* dS,
                                  float* dh, float* d_emb_w, float* d_att_w, float* pair_ws, float* wgrad_ws,
                                  const int* big_blocks, int NB, const float* wh_ws, const float* ml,
                                  const float* S_pool, float* big_part_ws, sw_wgrad_batch* defer, void* stream) {
  if (!obsv || !h || !scene_off || !pair_off || !emb_w || !att_w || !attn || !dS || !dh || !d_emb_w || !d_att_w ||
      !pair_ws || !wgrad_ws || S < 0 || B < 0 || P < 0 || To < 2 || NB < 0)
    return SW_EARG;
  if (NB > 0 && (!big_blocks || !wh_ws || !ml || !S_pool || !big_part_ws)) return SW_EARG;
  if (Amax > SW_AMAX) return SW_ESHAPE;
  if (S == 0 || B == 0) return SW_OK;
  hipStream_t st = (hipStream_t)stream;
  const int a16 = Amax < 16 ? 16 : ((Amax + 15) & ~15);
  const int extra = SW_SOC_WT + 4 * SW_SOC_SCR;       // scene kernel: transposed weight images + scratch
  const int lds = (soc_lds(a16).bwd_total + extra) * 4;
  const int lds_big = (soc_lds(16).fwd_total + 64 * SW_SOC_W2LD + 64 * SW_SOC_W1LD + 4 * SW_SOC_SCR + 16 * SW_BIG_PROW + SW_SOC_WT) * 4;
  static_assert(SW_SOC_WT + 4 * SW_SOC_SCR >= SW_SOC_PART, "epilogue staging area");
  static_assert(64 * SW_SOC_W2LD + 64 * SW_SOC_W1LD + 4 * SW_SOC_SCR >= SW_SOC_PART, "epilogue staging area (row-block kernel)");
  static bool attr = false;
  if (!attr) {
    if (int rc = set_lds((const void*)social_pool_bwd_kernel, (soc_lds(SW_AMAX).bwd_total + extra) * 4)) return rc;
    if (int rc = set_lds((const void*)social_big_bwd_kernel, lds_big)) return rc;
    attr = true;
  }
  // pair_ws: [B][64] dWh rows, then the pair rows
  float* dwh_rows = pair_ws;
  if (NB == 0 && P < (long long)S * SW_SOC_FUSE_MIN_PAIRS) {   // small scenes only: per-pair rows + deferred GEMM
    static bool attr2 = false;
    if (!attr2) {
      if (int rc = set_lds((const void*)social_pool_bwd_rows_kernel, soc_lds(SW_AMAX).bwd_total * 4)) return rc;
      attr2 = true;
    }
    float* wh_rows = pair_ws + (size_t)B * 64;
    float* q_rows = pair_ws + (size_t)B * 128;
    float* sd_rows = pair_ws + (size_t)B * 192;
    PairRows pr = pair_rows(pair_ws + (size_t)B * SW_AGENT_ROW_FLOATS, P);
    SW_LAUNCH(social_pool_bwd_rows_kernel, dim3(S), dim3(SW_THREADS), soc_lds(a16).bwd_total * 4, st, obsv, To, h,
                       scene_off, pair_off, emb_w, att_w, attn, dS, dh, dwh_rows, wh_rows, q_rows, sd_rows, pr, a16,
                       sw_soc_images_for(emb_w, att_w));
    SW_CHECK_LAUNCH("social_pool_bwd_rows_kernel");
    WgBatch wr_local;
    WgBatch& wr = defer ? *wg_pending(defer) : wr_local;
    wr = WgBatch();
    int rc_r = 0;
    rc_r |= wg_add(wr, dwh_rows, 64, h, 64, B, 64, 64, d_att_w + swp::ATT_W, 64, d_att_w + swp::ATT_B, nullptr, 0);
    if (P > 0) {
      // dW3 = sum_j Wh_j Q_j^T, db3 = sum_j Wh_j sd_j: B rows (the bias as a one-column problem: its "act" is sd)
      rc_r |= wg_add(wr, wh_rows, 64, q_rows, 64, B, 64, 64, d_emb_w + swp::EMB_W2, 64, nullptr, nullptr, 0);
      rc_r |= wg_add(wr, wh_rows, 64, sd_rows, 4, B, 64, 1, d_emb_w + swp::EMB_B2, 1, nullptr, nullptr, 0);
      rc_r |= wg_add(wr, pr.dh2, 64, pr.h1, 32, (int)P, 64, 32, d_emb_w + swp::EMB_W1, 32, d_emb_w + swp::EMB_B1, nullptr, 0);
      rc_r |= wg_add(wr, pr.dh1, 32, pr.feat, 4, (int)P, 32, 3, d_emb_w + swp::EMB_W0, 3, d_emb_w + swp::EMB_B0, nullptr, 0);
    }
    if (rc_r) return SW_ESHAPE;
    if (defer) return SW_OK;   // launched together with the caller's later problems (sw_gen_wgrad)
    return wg_launch(wr, wgrad_ws, st);
  }
  // in-register weight gradients: one partial slice per workgroup of the scene kernel (G) and of the row-block
  // kernel (NB), reduced together
  const int G = S < 1024 ? S : 1024;   // workgroups: each walks scenes g, g+G, .. and leaves ONE weight-gradient partial
  WgBatch wb_local;
  WgBatch& wb = defer ? *wg_pending(defer) : wb_local;
  wb = WgBatch();
  int rc_add = 0;
  rc_add |= wg_add(wb, dwh_rows, 64, h, 64, B, 64, 64, d_att_w + swp::ATT_W, 64, d_att_w + swp::ATT_B, nullptr, 0);
  const int i3 = wb.np, i2 = wb.np + 1, i1 = wb.np + 2;
  rc_add |= wg_add_pre(wb, 64, 64, d_emb_w + swp::EMB_W2, 64, d_emb_w + swp::EMB_B2, G + NB);
  rc_add |= wg_add_pre(wb, 64, 32, d_emb_w + swp::EMB_W1, 32, d_emb_w + swp::EMB_B1, G + NB);
  rc_add |= wg_add_pre(wb, 32, 3, d_emb_w + swp::EMB_W0, 3, d_emb_w + swp::EMB_B0, G + NB);
  if (rc_add) return SW_ESHAPE;
  SocPart part{wgrad_ws + wb.p[i3].ws_off, wgrad_ws + wb.p[i2].ws_off, wgrad_ws + wb.p[i1].ws_off};
  if (NB > 0) {   // scenes above SW_AMAX agents: fills dh / dwh_rows of their agents
    SW_LAUNCH(social_big_bwd_kernel, dim3(NB), dim3(SW_THREADS), lds_big, st, obsv, To, h, wh_ws,
                       wh_ws + (size_t)B * 64, wh_ws + (size_t)B * 128, scene_off, big_blocks, emb_w, S_pool, ml, dS,
                       big_part_ws, part, G);
    SW_CHECK_LAUNCH("social_big_bwd_kernel");
    SW_LAUNCH(social_big_finish_kernel, dim3(NB), dim3(SW_THREADS), 0, st, scene_off, big_blocks, att_w, emb_w,
                       big_part_ws, dh, dwh_rows);
    SW_CHECK_LAUNCH("social_big_finish_kernel");
  }
  SW_LAUNCH(social_pool_bwd_kernel, dim3(G), dim3(SW_THREADS), lds, st, obsv, To, h, scene_off, S, emb_w,
                     att_w, attn, dS, dh, dwh_rows, part, a16);
  SW_CHECK_LAUNCH("social_pool_bwd_kernel");
  if (defer) return SW_OK;
  return wg_launch(wb, wgrad_ws, st);
}
