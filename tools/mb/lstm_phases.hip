// Where does one LSTM time step of enc_lstm_fwd go?  Same device code (sw_lstm_dev.h), with s_memtime stamps
// between the phases of a step; wave 0 of workgroup 0 reports cycles per phase averaged over the steps.
//   hipcc --offload-arch=gfx950 -O3 -I socialways_amd/csrc tools/mb/lstm_phases.hip -o tools/mb/lstm_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "sw_lstm_dev.h"
void sw_set_error(const char*, hipError_t) {}
#define STAMP(v, dep) do { asm volatile("s_nop 0" :: "v"(dep)); __builtin_amdgcn_sched_barrier(0); v = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
__global__ __launch_bounds__(SW_THREADS) void k(const float* __restrict__ x, const float* __restrict__ whh, int B, int T,
                                                float* __restrict__ act, float* __restrict__ hT, long long* cyc, int save) {
  __shared__ __attribute__((aligned(16))) float hbuf[2][SW_TILE * SW_HLD];
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  const int u0 = wave * 16, a0 = blockIdx.x * SW_TILE, b = min(a0 + ln, B - 1);
  LstmW W;
  lstm_load_whh(W, whh, u0, ln, lg);
  for (int g = 0; g < 4; ++g) { W.wx[g] = 0.01f * (g + 1); W.bias[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};
  st4(&hbuf[0][ln * SW_HLD + u0 + 4 * lg], h);
  sw_barrier();
  f32x4 pgate[4], pc = c, ph = h;
  for (int g = 0; g < 4; ++g) pgate[g] = c;
  long long d_mm = 0, d_act = 0, d_st = 0, d_bar = 0, t0, t1, t2, t3, t4;
  for (int t = 0; t < T; ++t) {
    float xb = x[((size_t)b * T + t) * 4 + lg];
    STAMP(t0, xb);
    // MFMA part of lstm_cell
    const float* hrow = &hbuf[t & 1][ln * SW_HLD + 4 * lg];
    f32x4 acc[4], bb[4];
    for (int j = 0; j < 4; ++j) bb[j] = ld4(hrow + 16 * j);
    for (int g = 0; g < 4; ++g) acc[g] = SW_MFMA(W.wx[g], xb, W.bias[g]);
    for (int j = 0; j < 4; ++j) {
      for (int r = 0; r < 4; ++r)
        for (int g = 0; g < 4; ++g) acc[g] = SW_MFMA(W.whh[g][j][r], bb[j][r], acc[g]);
      if (save == 2 && j == 0 && t > 0) {   // previous step's rows go out under this step's MFMAs
        float* row = act + ((size_t)(t - 1) * B + b) * 384 + u0 + 4 * lg;
        for (int g = 0; g < 4; ++g) st4(row + g * 64, pgate[g]);
        st4(row + 256, pc);
        st4(row + 320, ph);
      }
    }
    STAMP(t1, acc[3][3]);
    f32x4 gate[4];
    for (int r = 0; r < 4; ++r) {
      float i = sw_sigmoid(acc[0][r]), f = sw_sigmoid(acc[1][r]), g = sw_tanh(acc[2][r]), o = sw_sigmoid(acc[3][r]);
      float cn = fmaf(f, c[r], i * g);
      gate[0][r] = i; gate[1][r] = f; gate[2][r] = g; gate[3][r] = o;
      c[r] = cn;
      h[r] = o * sw_tanh(cn);
    }
    STAMP(t2, h[3]);
    st4(&hbuf[(t + 1) & 1][ln * SW_HLD + u0 + 4 * lg], h);
    if (save == 2) { for (int g = 0; g < 4; ++g) pgate[g] = gate[g]; pc = c; ph = h; }
    if (save == 1) {
      float* row = act + ((size_t)t * B + b) * 384 + u0 + 4 * lg;
      for (int g = 0; g < 4; ++g) st4(row + g * 64, gate[g]);
      st4(row + 256, c);
      st4(row + 320, h);
    }
    STAMP(t3, h[0]);
    sw_barrier();
    STAMP(t4, h[1]);
    d_mm += t1 - t0; d_act += t2 - t1; d_st += t3 - t2; d_bar += t4 - t3;
  }
  st4(hT + (size_t)b * 64 + u0 + 4 * lg, h);
  if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = d_mm; cyc[1] = d_act; cyc[2] = d_st; cyc[3] = d_bar; }
}
int main() {
  const int B = 2048, T = 64;
  float *x, *whh, *act, *hT; long long* cyc;
  (void)hipMalloc(&x, (size_t)B * T * 16); (void)hipMalloc(&whh, 256 * 64 * 4); (void)hipMalloc(&act, (size_t)T * B * 384 * 4);
  (void)hipMalloc(&hT, B * 64 * 4); (void)hipMalloc(&cyc, 64);
  std::vector<float> w(256 * 64, 0.01f), xs((size_t)B * T * 4, 0.1f);
  (void)hipMemcpy(whh, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(x, xs.data(), xs.size() * 4, hipMemcpyHostToDevice);
  for (int save : {0, 1, 2}) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(B / 16), dim3(256), 0, 0, x, whh, B, T, act, hT, cyc, save);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(B / 16), dim3(256), 0, 0, x, whh, B, T, act, hT, cyc, save);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("save=%d: %.2f us/step total; cycles per step: h-read+MFMA %lld, activations %lld, LDS+global stores issue %lld, barrier wait %lld\n",
           save, ms * 1e3 / T, h[0] / T, h[1] / T, h[2] / T, h[3] / T);
  }
  return 0;
}
