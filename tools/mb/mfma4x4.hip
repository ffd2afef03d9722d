// Micro-benchmark for v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per instruction, K = 1):
//   1. operand layout incl. the A-broadcast controls (cbsz / abid), checked against a host model;
//   2. issue cadence with 1..4 independent accumulator chains;
//   3. a mock LSTM step on an 8-agent tile (wave = 4 agents x 32 units x 4 gates, x / h as the BROADCAST A operand,
//      W_hh rows as the B operand, 136 instructions per wave and step) with cycle stamps - to compare with
//      tools/mb/lstm_phases.hip (16-agent tile on v_mfma_f32_16x16x4_f32: 68 instructions of 32 cycles).
//   hipcc --offload-arch=gfx950 -O3 -I socialways_amd/csrc tools/mb/mfma4x4.hip -o tools/mb/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define M4(a, b, c, cbsz, abid) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), (cbsz), (abid), 0)

template <int CBSZ, int ABID>
__global__ void layout_k(float* out) {
  const int lane = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = M4((float)(lane + 1), 100.0f * (lane + 1), c, CBSZ, ABID);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
template <int CBSZ, int ABID>
int check_layout(float* dout) {
  hipLaunchKernelGGL((layout_k<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, dout);
  float h[256];
  (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int r = 0; r < 4; ++r) {
      const int b = lane >> 2, j = lane & 3;
      const int grp = 1 << CBSZ;
      const int ab = CBSZ ? (b / grp) * grp + ABID : b;    // block whose A rows this block uses
      const float want = (float)(4 * ab + r + 1) * 100.0f * (4 * b + j + 1);   // D_b[i=r][j] = A_ab[r] * B_b[j]
      if (h[lane * 4 + r] != want) ++bad;
    }
  printf("layout cbsz=%d abid=%d: %d mismatches vs {A row i <- lane 4b+i, B col j <- lane 4b+j, D[i][j] -> vgpr i of lane 4b+j}\n",
         CBSZ, ABID, bad);
  return bad;
}

template <int C>
__global__ __launch_bounds__(256) void chain_k(float* out, long long* cyc, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = M4(x, y, a0, 4, 3);
      if (C > 1) a1 = M4(x, y, a1, 4, 5);
      if (C > 2) a2 = M4(x, y, a2, 4, 7);
      if (C > 3) a3 = M4(x, y, a3, 4, 9);
    }
  }
  long long t1 = clock64();
  f32x4 s = a0 + a1 + a2 + a3;
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int C>
void run_chain(float* out, long long* cyc) {
  int iters = 4000;
  hipLaunchKernelGGL(chain_k<C>, dim3(256), dim3(256), 0, 0, out, cyc, 100);
  hipLaunchKernelGGL(chain_k<C>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long h;
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("4x4x1: %d chains: %.1f cycles per MFMA\n", C, h / (double)(iters * 8 * C));
}

// ---- mock LSTM step ---------------------------------------------------------------------------------------------
// workgroup = 8 agents (2 groups of 4), 4 waves: wave w -> agent group ag = w & 1, unit half uh = w >> 1.
// lane l: lower half (l < 32) rows of gates i (pass 0) and g (pass 1) of unit 32 uh + l; upper half gates f / o of
// unit 32 uh + l - 32.  B operand = W row of the lane, A operand = h / x of the 4 agents, broadcast from block abid.
#define STAMP(v, dep) do { asm volatile("s_nop 0" :: "v"(dep)); __builtin_amdgcn_sched_barrier(0); v = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x)); }
#define HLD 68
template <int K0>
__device__ __forceinline__ void mm16(f32x4& a0, f32x4& a1, const float* w0, const float* w1, float hv) {
  // 16 k-steps: block b of hv holds k = K0' + b for the lane's agent
#define ST(b) a0 = M4(hv, w0[K0 + b], a0, 4, b); a1 = M4(hv, w1[K0 + b], a1, 4, b);
  ST(0) ST(1) ST(2) ST(3) ST(4) ST(5) ST(6) ST(7) ST(8) ST(9) ST(10) ST(11) ST(12) ST(13) ST(14) ST(15)
#undef ST
}
__global__ __launch_bounds__(256) void lstm8_k(const float* __restrict__ x, const float* __restrict__ whh,
                                               const float* __restrict__ wx, const float* __restrict__ bias, int B, int T,
                                               float* __restrict__ act, float* __restrict__ hT, long long* cyc, int save) {
  __shared__ __attribute__((aligned(16))) float hbuf[2][8 * HLD];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ag = wave & 1, uh = wave >> 1;
  const int up = lane >> 5;                    // 0: gates i / g, 1: gates f / o
  const int unit = 32 * uh + (lane & 31);
  const int row0 = (up ? 1 : 0) * 64 + unit;   // pass 0: i or f
  const int row1 = (up ? 3 : 2) * 64 + unit;   // pass 1: g or o
  const int a0g = blockIdx.x * 8 + 4 * ag;     // first agent of this wave's group
  // W rows in registers: w0[k], w1[k]
  float w0[64], w1[64], wx0[4], wx1[4];
#pragma unroll
  for (int k = 0; k < 64; ++k) { w0[k] = whh[row0 * 64 + k]; w1[k] = whh[row1 * 64 + k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { wx0[k] = wx[row0 * 4 + k]; wx1[k] = wx[row1 * 4 + k]; }
  const float b0 = bias[row0], b1 = bias[row1];
  const float sc = up ? 1.0f : 2.0f;           // pass 1: tanh(x) = 2 sigmoid(2x) - 1 on the lower half, sigmoid on the upper
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  // A-operand lane mapping: lane (blk = lane >> 2, i = lane & 3) holds h[agent i][16 v + blk], v = 0..3
  const int ai = lane & 3, blk = lane >> 2;
  for (int i = threadIdx.x; i < 2 * 8 * HLD; i += 256) (&hbuf[0][0])[i] = 0.f;
  __syncthreads();
  long long d_mm = 0, d_act = 0, d_st = 0, d_bar = 0, t0, t1, t2, t3, t4;
  for (int t = 0; t < T; ++t) {
    const int bq = min(a0g + ai, B - 1);
    const float xv = x[((size_t)bq * T + t) * 4 + (blk & 3)];     // block kk (0..3) holds component kk
    STAMP(t0, xv);
    const float* hrow = &hbuf[t & 1][(4 * ag + ai) * HLD + blk];
    float hv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) hv[v] = hrow[16 * v];
    f32x4 acc0 = {b0, b0, b0, b0}, acc1 = {b1, b1, b1, b1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      switch (k) {   // abid must be an immediate
        case 0: acc0 = M4(xv, wx0[0], acc0, 4, 0); acc1 = M4(xv, wx1[0], acc1, 4, 0); break;
        case 1: acc0 = M4(xv, wx0[1], acc0, 4, 1); acc1 = M4(xv, wx1[1], acc1, 4, 1); break;
        case 2: acc0 = M4(xv, wx0[2], acc0, 4, 2); acc1 = M4(xv, wx1[2], acc1, 4, 2); break;
        default: acc0 = M4(xv, wx0[3], acc0, 4, 3); acc1 = M4(xv, wx1[3], acc1, 4, 3); break;
      }
    }
    mm16<0>(acc0, acc1, w0, w1, hv[0]);
    mm16<16>(acc0, acc1, w0, w1, hv[1]);
    mm16<32>(acc0, acc1, w0, w1, hv[2]);
    mm16<48>(acc0, acc1, w0, w1, hv[3]);
    STAMP(t1, acc1[3]);
    // activations: acc0 -> sigmoid (i | f); acc1 -> tanh (g, lower) | sigmoid (o, upper)
    f32x4 g0, g1, ig, hn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      g0[r] = fsig(acc0[r]);
      g1[r] = fmaf(sc, fsig(sc * acc1[r]), 1.0f - sc);
      ig[r] = g0[r] * g1[r];
    }
    // upper half fetches i*g of its unit from the lower half
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float o = __shfl_xor(ig[r], 32);
      const float cn = fmaf(g0[r], c[r], o);            // f c + i g (upper half meaningful)
      c[r] = cn;
      hn[r] = g1[r] * fmaf(2.0f, fsig(2.0f * cn), -1.0f);
    }
    STAMP(t2, hn[3]);
    if (up) {
#pragma unroll
      for (int r = 0; r < 4; ++r) hbuf[(t + 1) & 1][(4 * ag + r) * HLD + unit] = hn[r];
    }
    if (save) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int bb = min(a0g + r, B - 1);
        float* row = act + ((size_t)t * B + bb) * 384;
        row[(up ? 64 : 0) + unit] = g0[r];
        row[(up ? 192 : 128) + unit] = g1[r];
        if (up) { row[256 + unit] = c[r]; row[320 + unit] = hn[r]; }
      }
    }
    STAMP(t3, hn[0]);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    STAMP(t4, hn[1]);
    d_mm += t1 - t0; d_act += t2 - t1; d_st += t3 - t2; d_bar += t4 - t3;
  }
  if (up) {
#pragma unroll
    for (int r = 0; r < 4; ++r) hT[(size_t)min(a0g + r, B - 1) * 64 + unit] = hbuf[T & 1][(4 * ag + r) * HLD + unit];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = d_mm; cyc[1] = d_act; cyc[2] = d_st; cyc[3] = d_bar; }
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 64);
  check_layout<0, 0>(out);
  check_layout<4, 0>(out);
  check_layout<4, 5>(out);
  check_layout<2, 1>(out);
  run_chain<1>(out, cyc); run_chain<2>(out, cyc); run_chain<3>(out, cyc); run_chain<4>(out, cyc);

  const int B = 2048, T = 64;
  float *x, *whh, *wx, *bias, *act, *hT;
  (void)hipMalloc(&x, (size_t)B * T * 16); (void)hipMalloc(&whh, 256 * 64 * 4); (void)hipMalloc(&wx, 256 * 4 * 4);
  (void)hipMalloc(&bias, 256 * 4); (void)hipMalloc(&act, (size_t)T * B * 384 * 4); (void)hipMalloc(&hT, B * 64 * 4);
  std::vector<float> w(256 * 64), wxs(256 * 4), bs(256), xs((size_t)B * T * 4);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : w) v = 0.25f * rnd();
  for (auto& v : wxs) v = 0.5f * rnd();
  for (auto& v : bs) v = 0.2f * rnd();
  for (auto& v : xs) v = rnd();
  (void)hipMemcpy(whh, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(wx, wxs.data(), wxs.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(bias, bs.data(), bs.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(x, xs.data(), xs.size() * 4, hipMemcpyHostToDevice);
  for (int save : {0, 1}) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(lstm8_k, dim3(B / 8), dim3(256), 0, 0, x, whh, wx, bias, B, T, act, hT, cyc, save);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(lstm8_k, dim3(B / 8), dim3(256), 0, 0, x, whh, wx, bias, B, T, act, hT, cyc, save);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("lstm8 save=%d: %.2f us/step total (256 WGs x 8 agents); cycles per step: MFMA %lld, activations %lld, stores issue %lld, barrier wait %lld\n",
           save, ms * 1e3 / T, h[0] / T, h[1] / T, h[2] / T, h[3] / T);
  }
  // numerical check of hT for a few agents against a host LSTM
  std::vector<float> hTd((size_t)B * 64);
  (void)hipMemcpy(hTd.data(), hT, hTd.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int b : {0, 5, 1027, 2047}) {
    std::vector<double> h(64, 0.0), c(64, 0.0);
    for (int t = 0; t < T; ++t) {
      std::vector<double> g(256);
      for (int r = 0; r < 256; ++r) {
        double a = bs[r];
        for (int k = 0; k < 4; ++k) a += wxs[r * 4 + k] * xs[((size_t)b * T + t) * 4 + k];
        for (int k = 0; k < 64; ++k) a += w[r * 64 + k] * h[k];
        g[r] = a;
      }
      for (int u = 0; u < 64; ++u) {
        double i = 1 / (1 + exp(-g[u])), f = 1 / (1 + exp(-g[64 + u])), gg = tanh(g[128 + u]), o = 1 / (1 + exp(-g[192 + u]));
        c[u] = f * c[u] + i * gg;
        h[u] = o * tanh(c[u]);
      }
    }
    for (int u = 0; u < 64; ++u) maxerr = fmax(maxerr, fabs(h[u] - hTd[(size_t)b * 64 + u]));
  }
  printf("lstm8 h_T max abs err vs host fp64 LSTM: %.3g\n", maxerr);
  return 0;
}
