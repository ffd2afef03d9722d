// sw_misc.hip - ABI plumbing (errors, packed-weight layout, workspace sizes), the generator's
// deferred weight-gradient batch, the LSGAN/InfoGAN loss kernel and the ADE/FDE reduction.
#include "../../include/socialways_hip.h"
#include "sw_common.h"
#include "sw_lstm_dev.h"
#include <stdlib.h>
#include "sw_wgrad.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[256] = "";
void sw_set_error(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}
extern "C" const char* sw_last_error(void) { return g_err; }

// ---- per-kernel timing with HIP events on the launch stream (SW_LAUNCH, sw_common.h) ----------------------------------
#include <string>
#include <vector>
#include <map>
bool g_sw_ktime_on = false;
#include <mutex>
namespace {
struct KtRec { const char* name; hipEvent_t e0, e1; bool ended; };
std::vector<KtRec> g_kt;           // shared by the host threads that launch: every access under g_kt_mu
std::mutex g_kt_mu;
thread_local long t_kt_open = -1;  // the record this thread's sw_ktime_begin opened (-1: none - e.g. event creation failed)
}
void sw_ktime_begin(const char* name, hipStream_t st) {
  t_kt_open = -1;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
  KtRec r{name, nullptr, nullptr, false};
  if (hipEventCreate(&r.e0) != hipSuccess) { (void)hipGetLastError(); return; }
  if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(r.e0); return; }
  (void)hipEventRecord(r.e0, st);
  std::lock_guard<std::mutex> lk(g_kt_mu);
  g_kt.push_back(r);
  t_kt_open = (long)g_kt.size() - 1;
}
void sw_ktime_end(hipStream_t st) {
  const long i = t_kt_open;
  t_kt_open = -1;
  if (i < 0) return;
  std::lock_guard<std::mutex> lk(g_kt_mu);
  if (i >= (long)g_kt.size() || g_kt[i].ended) return;      // (the list was reset in between)
  (void)hipEventRecord(g_kt[i].e1, st);
  g_kt[i].ended = true;
}
extern "C" int sw_kernel_timing(int on) {
  std::lock_guard<std::mutex> lk(g_kt_mu);
  for (auto& r : g_kt) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_kt.clear();
  g_sw_ktime_on = on != 0;
  return SW_OK;
}
// "kernel calls total_us\n" per kernel name (template arguments stripped) into buf; returns the bytes needed (<= cap: complete)
extern "C" int sw_kernel_timing_read(char* buf, int cap) {
  if (hipDeviceSynchronize() != hipSuccess) return SW_EHIP;
  std::map<std::string, std::pair<long, double>> agg;
  std::lock_guard<std::mutex> lk(g_kt_mu);
  for (auto& r : g_kt) {
    if (!r.ended) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) { (void)hipGetLastError(); continue; }
    std::string n(r.name);
    size_t a = n.find_first_not_of("( ");
    n = n.substr(a == std::string::npos ? 0 : a);
    n = n.substr(0, n.find_first_of("<)"));
    auto& e = agg[n];
    e.first += 1;
    e.second += 1e3 * ms;
  }
  std::string out;
  char line[256];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s %ld %.3f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    out += line;
  }
  if (buf && cap > 0) {
    const size_t k = out.size() < (size_t)cap - 1 ? out.size() : (size_t)cap - 1;
    memcpy(buf, out.data(), k);
    buf[k] = 0;
  }
  return (int)out.size() + 1;
}
// a kernel that keeps the stream busy for ~`us` microseconds: queued in front of a timed sequence, it lets the host run
// ahead so that the sequence's launches reach the GPU back to back (event timings then hold no host gaps)
__global__ void spin_kernel(long long cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
}
__global__ void nop_kernel() {}
extern "C" int sw_debug_spin(double us, void* stream) {
  if (us < 0 || us > 2e6) return SW_EARG;
  if (us == 0.0) {      // calibration of event timings: a kernel that does nothing, under a name of its own
    SW_LAUNCH(nop_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
    SW_CHECK_LAUNCH("nop_kernel");
    return SW_OK;
  }
  SW_LAUNCH(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)(us * 100.0));   // wall_clock64: 100 MHz
  SW_CHECK_LAUNCH("spin_kernel");
  return SW_OK;
}

// ---- Adam over a packed buffer (train.py:379-385; torch's fused Adam restated operation by operation: sw_wgrad.h) for the
//      paths where no gradient-finishing kernel can carry the update: data-parallel ranks all-reduce the packed gradient
//      buffer between the backward pass and the optimizer step.  Keeps registered discriminator images current. --------
__global__ __launch_bounds__(256) void adam_packed_kernel(WgAdam ad, long long n) {
  __shared__ float bc[2];
  if (threadIdx.x == 0) wg_adam_bc_compute(ad.step, ad.beta1, ad.beta2, bc[0], bc[1]);
  __syncthreads();
  const float bc1 = bc[0], bc2s = bc[1];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float g = ad.g0[i];
    wg_adam_fin(ad, wg_adam_pre(ad, ad.g0 + i), bc1, bc2s, g);
  }
}
extern "C" int sw_adam_packed(float* w, const float* g, float* m, float* v, long long n, const float* step, double lr,
                              double beta1, double beta2, double eps, int disc_Tp, void* stream) {
  if (!w || !g || !m || !v || !step || n < 1) return SW_EARG;
  WgAdam ad;
  ad.w = w; ad.m = m; ad.v = v; ad.g0 = g; ad.step = step; ad.n = (size_t)n;
  ad.lr = lr; ad.beta1 = beta1; ad.beta2 = beta2; ad.eps = eps;
  if (disc_Tp > 0) {
    const DiscImages di = sw_disc_images_for(w, disc_Tp);
    ad.img = const_cast<float*>(di.img);
    ad.tab = di.tab;
  }
  int blocks = (int)((n + 255) / 256);
  if (blocks > 512) blocks = 512;
  SW_LAUNCH(adam_packed_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ad, n);
  SW_CHECK_LAUNCH("adam_packed_kernel");
  return SW_OK;
}
extern "C" int sw_version(void) { return 3; }

size_t sw_dsave_floats(int B, int To, int Tp, int nb);
size_t sw_ddelta_floats(int B, int To, int Tp, int nb);

// ---- packed-weight layout (state_dict order) ---------------------------------------------------
static int group_offsets(int grp, int Tp, int* off, int* count) {
  using namespace swp;
  switch (grp) {
    case SW_GRP_ENC: {
      const int o[] = {ENC_EMB_W, ENC_EMB_B, ENC_WIH, ENC_WHH, ENC_BIH, ENC_BHH};
      memcpy(off, o, sizeof(o));
      *count = ENC_N;
      return 6;
    }
    case SW_GRP_EMB: {
      const int o[] = {EMB_W0, EMB_B0, EMB_W1, EMB_B1, EMB_W2, EMB_B2};
      memcpy(off, o, sizeof(o));
      *count = EMB_N;
      return 6;
    }
    case SW_GRP_ATT: {
      const int o[] = {ATT_W, ATT_B};
      memcpy(off, o, sizeof(o));
      *count = ATT_N;
      return 2;
    }
    case SW_GRP_DEC: {
      const int o[] = {DEC_W1, DEC_B1, DEC_W2, DEC_B2, DEC_W3, DEC_B3, DEC_W4, DEC_B4};
      memcpy(off, o, sizeof(o));
      *count = DEC_N;
      return 8;
    }
    case SW_GRP_DISC: {
      if (Tp < 1) return -1;
      Disc d = disc(Tp);
      const int o[] = {d.wih,  d.whh,  d.bih,  d.bhh,  d.of0w, d.of0b, d.of1w, d.of1b, d.pe0w, d.pe0b,
                       d.pe1w, d.pe1b, d.cl0w, d.cl0b, d.cl1w, d.cl1b, d.la0w, d.la0b, d.la1w, d.la1b};
      memcpy(off, o, sizeof(o));
      *count = d.n;
      return 20;
    }
  }
  return -1;
}
extern "C" int sw_param_tensors(int grp) {
  int off[32], n;
  return group_offsets(grp, 1, off, &n);
}
extern "C" int sw_param_count(int grp, int Tp) {
  int off[32], n = -1;
  if (group_offsets(grp, Tp, off, &n) < 0) return SW_EARG;
  return n;
}
extern "C" int sw_param_offset(int grp, int idx, int Tp) {
  int off[32], n;
  int k = group_offsets(grp, Tp, off, &n);
  if (k < 0 || idx < 0 || idx >= k) return SW_EARG;
  return off[idx];
}

extern "C" size_t sw_workspace_floats(int ws_id, int B, int To, int Tp, int nb, long long P) {
  if (B < 0 || To < 1 || Tp < 1) return 0;
  switch (ws_id) {
    case SW_WS_GSAVE: return gsave_layout(B, To, Tp).total;
    case SW_WS_GDELTA: return gdelta_layout(B, To, Tp).total;
    case SW_WS_DSAVE: return sw_dsave_floats(B, To, Tp, nb < 1 ? 1 : nb);
    case SW_WS_DDELTA: return sw_ddelta_floats(B, To, Tp, nb < 1 ? 1 : nb);
    case SW_WS_WGRAD: return SW_WG_WS_FLOATS;
    case SW_WS_PAIRS: return (size_t)B * 200 + (size_t)(P < 0 ? 0 : P) * 196;  // per-agent rows (dWh | Wh | Q | sd) + pair rows of the small-scene path (sw_social.hip)
  }
  return 0;
}

// ---- encoder: gradients of the composed input matrix back to embed / W_ih --------------------
//   Wx = Wih We, bx = Wih be + bih + bhh   (sw_lstm_dev.h)
// Blocks >= 128 (when dec_w != null): the decoder's fc3 / fc4 run as ONE composed 2 x 80 map in the kernels
// (v = W43 a2 + b43, sw_decoder.hip); their four gradients follow from M = dv^T a2 (2 x 80) and s = sum dv (2):
//   dW3 = W4^T M,  db3 = W4^T s,  dW4 = M W3^T + s b3^T,  db4 = s        (exactly autograd's values, reassociated)
#define SW_DEC_COMPOSE_BLOCKS 13   // 3322 outputs
// With `ad.w` set the thread that forms a gradient element also applies the generator's Adam update to its weight
// (wg_adam1); the compositions then read the weights from `snap` - the step-start snapshot inside the image buffer
// (swimg::RAW_*) - because other workgroups of this launch are overwriting the live ones.
__device__ __forceinline__ void compose_bwd_block(const int blk, const float* __restrict__ enc_w,
                                                  const float* __restrict__ dWx, const float* __restrict__ dbx,
                                                  float* __restrict__ d_enc_w, const float* __restrict__ dec_w,
                                                  const float* __restrict__ Ms, float* __restrict__ d_dec_w,
                                                  const WgAdam& ad, const float* __restrict__ snap) {
  using namespace swp;
  float bc1 = 1.f, bc2s = 1.f;
  if (ad.w) wg_adam_bc(ad, bc1, bc2s);
  auto put = [&](float* dst, float g) {
    *dst = g;
    if (ad.w) wg_adam1(ad, bc1, bc2s, dst, g);
  };
  if (blk >= 128) {
    const int o = (blk - 128) * 256 + threadIdx.x;
    const float* M = Ms;            // [2][80]
    const float* sv = Ms + 160;     // [2]
    const float* W3 = snap ? snap + swimg::RAW_W3 : dec_w + DEC_W3;
    const float* W4 = snap ? snap + swimg::RAW_W4 : dec_w + DEC_W4;
    const float* b3 = snap ? snap + swimg::RAW_B3 : dec_w + DEC_B3;
    if (o < 3200) {                 // dW3[m][k]
      const int m = o / 80, k = o - m * 80;
      put(d_dec_w + DEC_W3 + o, fmaf(W4[m], M[k], W4[40 + m] * M[80 + k]));
    } else if (o < 3240) {          // db3[m]
      const int m = o - 3200;
      put(d_dec_w + DEC_B3 + m, fmaf(W4[m], sv[0], W4[40 + m] * sv[1]));
    } else if (o < 3320) {          // dW4[c][m]
      const int c = (o - 3240) / 40, m = (o - 3240) - c * 40;
      float v = sv[c] * b3[m];
      for (int k = 0; k < 80; ++k) v = fmaf(M[c * 80 + k], W3[m * 80 + k], v);
      put(d_dec_w + DEC_W4 + c * 40 + m, v);
    } else if (o < 3322) {          // db4[c]
      put(d_dec_w + DEC_B4 + (o - 3320), sv[o - 3320]);
    }
    return;
  }
  const float* We = snap ? snap + swimg::RAW_WE : enc_w + ENC_EMB_W;
  const float* be = snap ? snap + swimg::RAW_BE : enc_w + ENC_EMB_B;
  const float* Wih = snap ? snap + swimg::RAW_WIH : enc_w + ENC_WIH;
  const int t = threadIdx.x;
  if (blk < 64) {  // dWih[row][e] = sum_c dWx[row][c] We[e][c] + dbx[row] be[e]; 256 elements per block
    int i = blk * 256 + t;
    int row = i >> 6, e = i & 63;
    f32x4 g = ld4(dWx + row * 4), w = ld4(We + e * 4);
    float v = dbx[row] * be[e];
    v = fmaf(g[0], w[0], v); v = fmaf(g[1], w[1], v); v = fmaf(g[2], w[2], v); v = fmaf(g[3], w[3], v);
    put(d_enc_w + ENC_WIH + i, v);
    if (blk == 0) {
      put(d_enc_w + ENC_BIH + t, dbx[t]);
      put(d_enc_w + ENC_BHH + t, dbx[t]);
    }
    return;
  }
  // blocks 64..127: one embed unit e each; thread = gate row; dWe[e][c] = sum_row Wih[row][e] dWx[row][c],
  // dbe[e] = sum_row Wih[row][e] dbx[row]
  __shared__ float red[5][256];
  const int e = blk - 64;
  float w = Wih[t * 64 + e];
  f32x4 g = ld4(dWx + t * 4);
  red[0][t] = w * g[0]; red[1][t] = w * g[1]; red[2][t] = w * g[2]; red[3][t] = w * g[3]; red[4][t] = w * dbx[t];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
#pragma unroll
      for (int c = 0; c < 5; ++c) red[c][t] += red[c][t + o];
    }
    __syncthreads();
  }
  if (t < 4) put(d_enc_w + ENC_EMB_W + e * 4 + t, red[t][0]);
  if (t == 4) put(d_enc_w + ENC_EMB_B + e, red[4][0]);
}

__global__ __launch_bounds__(256) void enc_compose_bwd_kernel(const float* __restrict__ enc_w,
                                                              const float* __restrict__ dWx,
                                                              const float* __restrict__ dbx,
                                                              float* __restrict__ d_enc_w,
                                                              const float* __restrict__ dec_w,
                                                              const float* __restrict__ Ms,
                                                              float* __restrict__ d_dec_w, WgAdam ad,
                                                              const float* __restrict__ snap) {
  compose_bwd_block((int)blockIdx.x, enc_w, dWx, dbx, d_enc_w, dec_w, Ms, d_dec_w, ad, snap);
}
// one half on its own (blocks block0 ..): the stand-alone module backward passes below
__global__ __launch_bounds__(256) void compose_bwd_part_kernel(const float* __restrict__ enc_w, const float* __restrict__ dWx,
                                                               const float* __restrict__ dbx, float* __restrict__ d_enc_w,
                                                               const float* __restrict__ dec_w, const float* __restrict__ Ms,
                                                               float* __restrict__ d_dec_w, int block0) {
  compose_bwd_block((int)blockIdx.x + block0, enc_w, dWx, dbx, d_enc_w, dec_w, Ms, d_dec_w, WgAdam(), nullptr);
}

// part 0: everything.  part 1: what is available right after dec_rollout_bwd (all decoder problems +
// the LSTM rows of the decode phase, t >= To).  part 2: the LSTM rows of the observation phase
// (after enc_lstm_bwd), accumulated on top of part 1, then the composed-input-matrix back-propagation.
// Parts 1 and 2 may run on different streams (different partial workspaces `wgrad_ws`); `tmp` holds
// the 256x4 + 256 composed-matrix gradient between them.
static int gen_wgrad_impl(const float* enc_w, const float* dec_w, const float* gsave, const float* gdelta, const float* z,
                          const float* S_pool, int B, int To, int Tp, float* d_enc_w, float* d_dec_w, int part,
                          float* wgrad_ws, float* tmp, sw_wgrad_batch* pending, void* stream, WgAdam ad,
                          const float* snap) {
  if (!enc_w || !dec_w || !gsave || !gdelta || !z || !S_pool || !d_enc_w || !d_dec_w || !wgrad_ws || !tmp || B < 1 || To < 2 ||
      Tp < 1 || part < 0 || part > 2)
    return SW_EARG;
  using namespace swp;
  const GSave gs = gsave_layout(B, To, Tp);
  const GDelta gd = gdelta_layout(B, To, Tp);
  const int Ta = To + Tp - 1;
  float* dWx = tmp;
  float* dbx = tmp + 1024;
  float* dM = tmp + 1280;   // [2][80] dv^T a2, then [2] sum dv
  hipStream_t st = (hipStream_t)stream;
  WgBatch wb;
  if (pending) {   // problems another module left for this launch (sw_social_pool_bwd with `defer`)
    wb = *wg_pending(pending);
    *wg_pending(pending) = WgBatch();
  }
  int rc_add = 0;
  // EncoderLstm: W_hh against h_{t-1} (rows t >= 1), composed input matrix against x4 (all rows)
  const int t_lo = part == 1 ? To : 0, t_hi = part == 2 ? To : Ta;   // LSTM rows [t_lo, t_hi)
  const int acc = part == 2 ? 1 : 0;
  // W_hh against h_{t-1} (rows t >= 1) and the composed input matrix against x4 (all rows) share the dgates rows:
  // one problem with a tail segment, dgates fetched once.  Logical row r = (t - t_lo) B + b.
  if (t_hi > t_lo)
    rc_add |= wg_add_tail(wb, gdelta + gd.dgates + (size_t)t_lo * B * 256, 256,
                          gsave + gs.act + ((ptrdiff_t)t_lo - 1) * B * 384 + 320, 384, (t_hi - t_lo) * B, 256, 64,
                          d_enc_w + ENC_WHH, 64, gsave + gs.x4s + (size_t)t_lo * B * 4, 4, 4, dWx, 4,
                          t_lo < 1 ? B : 0 /*t = 0 has no h_{t-1}: zero initial state (train.py:399-400)*/, dbx, nullptr, acc);
  if (part != 2) {
    // DecoderFC: fc1.0 split in its h / S / z column blocks; h of decode step i is LSTM row To-1+i
    rc_add |= wg_add(wb, gdelta + gd.dz1, 160, gsave + gs.act + (size_t)(To - 1) * B * 384 + 320, 384, Tp * B, 160, 64,
                     d_dec_w + DEC_W1, 160, nullptr, nullptr, 0);
    rc_add |= wg_add(wb, gdelta + gd.du, 160, S_pool, 64, B, 160, 64, d_dec_w + DEC_W1 + 64, 160, nullptr, nullptr, 0);
    rc_add |= wg_add(wb, gdelta + gd.du, 160, z, 32, B, 160, 32, d_dec_w + DEC_W1 + 128, 160, d_dec_w + DEC_B1, nullptr, 0);
    rc_add |= wg_add(wb, gdelta + gd.dz2, 80, gsave + gs.a1, 160, Tp * B, 80, 160, d_dec_w + DEC_W2, 160, d_dec_w + DEC_B2,
                     nullptr, 0);
    // fc3 / fc4 (one composed map in the kernels): M = dv^T a2 and s = sum dv; enc_compose_bwd_kernel derives
    // dW3, db3, dW4, db4 from them
    rc_add |= wg_add(wb, gdelta + gd.dv, 4, gsave + gs.a2, 80, Tp * B, 2, 80, dM, 80, dM + 160, nullptr, 0);
  }
  if (rc_add) return SW_ESHAPE;
  if (int rc = ad.w ? wg_launch_adam(wb, wgrad_ws, ad, st) : wg_launch(wb, wgrad_ws, st)) return rc;
  if (part != 1) {
    SW_LAUNCH(enc_compose_bwd_kernel, dim3(128 + SW_DEC_COMPOSE_BLOCKS), dim3(256), 0, st, enc_w, dWx, dbx, d_enc_w,
                       dec_w, dM, d_dec_w, ad, snap);
    SW_CHECK_LAUNCH("enc_compose_bwd_kernel");
  }
  return SW_OK;
}
extern "C" int sw_gen_wgrad(const float* enc_w, const float* dec_w, const float* gsave, const float* gdelta, const float* z,
                            const float* S_pool, int B, int To, int Tp, float* d_enc_w, float* d_dec_w, int part,
                            float* wgrad_ws, float* tmp, sw_wgrad_batch* pending, void* stream) {
  return gen_wgrad_impl(enc_w, dec_w, gsave, gdelta, z, S_pool, B, To, Tp, d_enc_w, d_dec_w, part, wgrad_ws, tmp, pending,
                        stream, WgAdam(), nullptr);
}
// sw_gen_wgrad(part 0) that also applies the generator's Adam update: every generator parameter's gradient is finished
// either by the reduction of the grouped GEMM or by the composition kernel behind it, and the thread that finishes an
// element updates exp_avg / exp_avg_sq / the weight in the packed buffers (adam_w / adam_m / adam_v, laid out like the
// packed gradient buffer adam_g of adam_n floats that d_enc_w, d_dec_w and the pending problems' outputs point into).
// Needs the weight images of this step (sw_gen_images / sw_stage_step_img): the composition reads their snapshot.
extern "C" int sw_gen_wgrad_adam(const float* enc_w, const float* dec_w, const float* gsave, const float* gdelta,
                                 const float* z, const float* S_pool, int B, int To, int Tp, float* d_enc_w, float* d_dec_w,
                                 float* wgrad_ws, float* tmp, sw_wgrad_batch* pending, float* adam_w, float* adam_m,
                                 float* adam_v, const float* adam_g, long long adam_n, const float* adam_step, double lr,
                                 double beta1, double beta2, double eps, void* stream) {
  if (!adam_w || !adam_m || !adam_v || !adam_g || !adam_step || adam_n < 1 || !enc_w || !dec_w || !d_enc_w || !d_dec_w)
    return SW_EARG;
  // weights and gradients must share one layout
  if (enc_w - adam_w != d_enc_w - adam_g || dec_w - adam_w != d_dec_w - adam_g || enc_w < adam_w ||
      dec_w + swp::DEC_N > adam_w + adam_n || enc_w + swp::ENC_N > adam_w + adam_n || dec_w < adam_w)
    return SW_EARG;
  const float* img = sw_gen_images_for(enc_w, dec_w);
  if (!img) return SW_EARG;
  WgAdam ad;
  ad.w = adam_w; ad.m = adam_m; ad.v = adam_v; ad.g0 = adam_g; ad.step = adam_step; ad.n = (size_t)adam_n;
  ad.lr = lr; ad.beta1 = beta1; ad.beta2 = beta2; ad.eps = eps;
  return gen_wgrad_impl(enc_w, dec_w, gsave, gdelta, z, S_pool, B, To, Tp, d_enc_w, d_dec_w, 0, wgrad_ws, tmp, pending, stream,
                        ad, img);
}

// ---- weight gradients of the STAND-ALONE EncoderLstm / DecoderFC modules (model.py: _EncFn, _DecFn) ----------------
// EncoderLstm over T steps from state (h0, c0): act [T][B][384] / x4s [T][B][4] as sw_enc_lstm_fwd left them, dgates
// [T][B][256] from sw_enc_lstm_bwd; h0 NULL = zero initial state.  Writes every gradient of the packed encoder buffer.
extern "C" int sw_enc_lstm_wgrad(const float* enc_w, const float* act, const float* x4s, const float* h0, const float* dgates,
                                 int B, int T, float* d_enc_w, float* wgrad_ws, float* tmp, void* stream) {
  if (!enc_w || !act || !x4s || !dgates || !d_enc_w || !wgrad_ws || !tmp || B < 1 || T < 1) return SW_EARG;
  using namespace swp;
  hipStream_t st = (hipStream_t)stream;
  float* dWx = tmp;
  float* dbx = tmp + 1024;
  WgBatch wb;
  // W_hh against h_{t-1} (rows t >= 1; row t - 1 of `act`) and the composed input matrix against x4 (all rows)
  if (wg_add_tail(wb, dgates, 256, act - (ptrdiff_t)B * 384 + 320, 384, T * B, 256, 64, d_enc_w + ENC_WHH, 64, x4s, 4, 4, dWx, 4,
                  B, dbx, nullptr, 0))
    return SW_ESHAPE;
  if (int rc = wg_launch(wb, wgrad_ws, st)) return rc;
  if (h0) {   // ... and the first step against the given initial state
    WgBatch w0;
    if (wg_add(w0, dgates, 256, h0, 64, B, 256, 64, d_enc_w + ENC_WHH, 64, nullptr, nullptr, 1)) return SW_ESHAPE;
    if (int rc = wg_launch(w0, wgrad_ws, st)) return rc;
  }
  SW_LAUNCH(compose_bwd_part_kernel, dim3(128), dim3(256), 0, st, enc_w, dWx, dbx, d_enc_w, nullptr, nullptr, nullptr, 0);
  SW_CHECK_LAUNCH("compose_bwd_part_kernel");
  return SW_OK;
}
// DecoderFC on one batch (a Tp = 1 rollout: gsave / gdelta of sw_dec_rollout_fwd / _bwd with To, Tp = 1) with its
// inputs h, s (NULL = zeros), z given explicitly.  Writes every gradient of the packed decoder buffer.
extern "C" int sw_dec_fc_wgrad(const float* dec_w, const float* gsave, const float* gdelta, const float* h, const float* s,
                               const float* z, int B, int To, float* d_dec_w, float* wgrad_ws, float* tmp, void* stream) {
  if (!dec_w || !gsave || !gdelta || !h || !z || !d_dec_w || !wgrad_ws || !tmp || B < 1 || To < 2) return SW_EARG;
  using namespace swp;
  hipStream_t st = (hipStream_t)stream;
  const GSave gs = gsave_layout(B, To, 1);
  const GDelta gd = gdelta_layout(B, To, 1);
  float* dM = tmp + 1280;
  WgBatch wb;
  int rc = 0;
  rc |= wg_add(wb, gdelta + gd.dz1, 160, h, 64, B, 160, 64, d_dec_w + DEC_W1, 160, nullptr, nullptr, 0);
  if (s) rc |= wg_add(wb, gdelta + gd.dz1, 160, s, 64, B, 160, 64, d_dec_w + DEC_W1 + 64, 160, nullptr, nullptr, 0);
  rc |= wg_add(wb, gdelta + gd.dz1, 160, z, 32, B, 160, 32, d_dec_w + DEC_W1 + 128, 160, d_dec_w + DEC_B1, nullptr, 0);
  rc |= wg_add(wb, gdelta + gd.dz2, 80, gsave + gs.a1, 160, B, 80, 160, d_dec_w + DEC_W2, 160, d_dec_w + DEC_B2, nullptr, 0);
  rc |= wg_add(wb, gdelta + gd.dv, 4, gsave + gs.a2, 80, B, 2, 80, dM, 80, dM + 160, nullptr, 0);
  if (rc) return SW_ESHAPE;
  if (int r2 = wg_launch(wb, wgrad_ws, st)) return r2;
  if (!s) {     // no pooled social vector: that block of fc1.0.weight has no gradient
    if (hipMemset2DAsync(d_dec_w + DEC_W1 + 64, 160 * sizeof(float), 0, 64 * sizeof(float), 160, st) != hipSuccess) return SW_EHIP;
  }
  SW_LAUNCH(compose_bwd_part_kernel, dim3(SW_DEC_COMPOSE_BLOCKS), dim3(256), 0, st, nullptr, nullptr, nullptr, nullptr,
                     dec_w, dM, d_dec_w, 128);
  SW_CHECK_LAUNCH("compose_bwd_part_kernel");
  return SW_OK;
}

// ---- losses (train.py:484-494, 512-523) ---------------------------------------------------------
// block-wide sum of 3 values with 1024 threads: wave shuffles, then 16 partials through LDS
__device__ __forceinline__ void block_sum3(float& a, float& b, float& c, float (*red)[16]) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
    c += __shfl_xor(c, o);
  }
  int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) { red[0][w] = a; red[1][w] = b; red[2][w] = c; }
  __syncthreads();
  if (w == 0) {
    a = l < 16 ? red[0][l] : 0.f;
    b = l < 16 ? red[1][l] : 0.f;
    c = l < 16 ? red[2][l] : 0.f;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      a += __shfl_xor(a, o);
      b += __shfl_xor(b, o);
      c += __shfl_xor(c, o);
    }
  }
}

__global__ __launch_bounds__(1024) void gan_loss_kernel(
    const float* __restrict__ label_a, const float* __restrict__ targets, int ia, const float* __restrict__ code_a,
    const float* __restrict__ z, const float* __restrict__ label_b, int ib, int B, float g_label, float g_code,
    float* __restrict__ out, float* __restrict__ dlabel_a, float* __restrict__ dcode_a, float* __restrict__ dlabel_b,
    float* __restrict__ dcode_b) {
  __shared__ float red[3][16];
  const float t_a = targets[ia], t_b = targets[ib];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    float e = label_a[b] - t_a;
    s0 = fmaf(e, e, s0);
    if (dlabel_a) dlabel_a[b] = 2.0f * e * g_label;
    float c0 = code_a[(size_t)b * 2] - z[(size_t)b * SW_Z], c1 = code_a[(size_t)b * 2 + 1] - z[(size_t)b * SW_Z + 1];
    s1 += c0 * c0 + c1 * c1;
    if (dcode_a) {
      dcode_a[(size_t)b * 2] = 2.0f * c0 * g_code;
      dcode_a[(size_t)b * 2 + 1] = 2.0f * c1 * g_code;
    }
    if (label_b) {
      float f = label_b[b] - t_b;
      s2 = fmaf(f, f, s2);
      if (dlabel_b) dlabel_b[b] = 2.0f * f * g_label;
    }
    if (dcode_b) {
      dcode_b[(size_t)b * 2] = 0.f;
      dcode_b[(size_t)b * 2 + 1] = 0.f;
    }
  }
  block_sum3(s0, s1, s2, red);
  if (threadIdx.x == 0) { out[3 * blockIdx.x] = s0; out[3 * blockIdx.x + 1] = s1; out[3 * blockIdx.x + 2] = s2; }
}

// second stage of the large-batch reductions: G per-block partial triples -> out[3], fixed order
__global__ __launch_bounds__(64) void sum3_kernel(const float* __restrict__ part, int G, float s_a, float* __restrict__ out) {
  float a = 0.f, b = 0.f, c = 0.f;
  for (int g = threadIdx.x; g < G; g += 64) { a += part[3 * g]; b += part[3 * g + 1]; c += part[3 * g + 2]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
    c += __shfl_xor(c, o);
  }
  if (threadIdx.x == 0) { out[0] = a * s_a; out[1] = b; out[2] = c; }
}
// one 1024-thread block up to SW_RED_SINGLE items (launch-latency bound there), else up to 64 blocks
#define SW_RED_SINGLE 32768
static inline int red_blocks(long long n, const float* scratch) {
  if (!scratch || n <= SW_RED_SINGLE) return 1;
  long long g = (n + 8191) / 8192;
  return (int)(g > SW_RED_BLOCKS ? SW_RED_BLOCKS : g);
}

extern "C" int sw_gan_loss(const float* label_a, const float* targets, int ia, const float* code_a, const float* z,
                           const float* label_b, int ib, int B, float g_label, float g_code, float* out_sums,
                           float* dlabel_a, float* dcode_a, float* dlabel_b, float* dcode_b, float* scratch,
                           void* stream) {
  if (!label_a || !targets || !code_a || !z || !out_sums || B < 1 || ia < 0 || ib < 0) return SW_EARG;
  const int G = red_blocks((long long)B * 4, scratch);
  SW_LAUNCH(gan_loss_kernel, dim3(G), dim3(1024), 0, (hipStream_t)stream, label_a, targets, ia, code_a, z,
                     label_b, ib, B, g_label, g_code, G > 1 ? scratch : out_sums, dlabel_a, dcode_a, dlabel_b, dcode_b);
  SW_CHECK_LAUNCH("gan_loss_kernel");
  if (G > 1) {
    SW_LAUNCH(sum3_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, G, 1.0f, out_sums);
    SW_CHECK_LAUNCH("sum3_kernel");
  }
  return SW_OK;
}

// ---- ADE/FDE partial sums (train.py:546-551) ----------------------------------------------------
__global__ __launch_bounds__(1024) void ade_fde_kernel(const float* __restrict__ pred4, const float* __restrict__ gt,
                                                       int B, int Tp, float inv_ss, float* __restrict__ out,
                                                       int single) {
  __shared__ float red[3][16];
  float sa = 0.f, sf = 0.f, sl = 0.f;
  const int n = B * Tp;
  for (int base = blockIdx.x * 1024 * 8; base < n; base += gridDim.x * 1024 * 8) {  // 8 independent loads in flight per thread
    f32x4 p[8];
    float2 g[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int i = base + threadIdx.x + 1024 * u;
      bool ok = i < n;
      p[u] = ok ? ld4(pred4 + (size_t)i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      g[u] = ok ? *reinterpret_cast<const float2*>(gt + (size_t)i * 2) : float2{0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int i = base + threadIdx.x + 1024 * u;
      float dx = (p[u][0] - g[u].x) * inv_ss, dy = (p[u][1] - g[u].y) * inv_ss;
      float q = dx * dx + dy * dy;
      float e = sqrtf(q);
      sa += e;
      sl += q;
      if (i < n && i % Tp == Tp - 1) sf += e;
    }
  }
  block_sum3(sa, sf, sl, red);
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x] = single ? sa / (float)Tp : sa;
    out[3 * blockIdx.x + 1] = sf;
    out[3 * blockIdx.x + 2] = sl;  // sum of squared (scaled) displacement errors, for the L2 term
  }
}

extern "C" int sw_ade_fde(const float* pred4, const float* gt, int B, int Tp, float inv_ss, float* out, float* scratch,
                          void* stream) {
  if (!pred4 || !gt || !out || B < 1 || Tp < 1) return SW_EARG;
  const int G = red_blocks((long long)B * Tp, scratch);
  SW_LAUNCH(ade_fde_kernel, dim3(G), dim3(1024), 0, (hipStream_t)stream, pred4, gt, B, Tp, inv_ss,
                     G > 1 ? scratch : out, G == 1 ? 1 : 0);
  SW_CHECK_LAUNCH("ade_fde_kernel");
  if (G > 1) {
    SW_LAUNCH(sum3_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, G, 1.0f / (float)Tp, out);
    SW_CHECK_LAUNCH("sum3_kernel");
  }
  return SW_OK;
}

// ---- optional L2 term of the generator loss (train.py:512, 525-526; variety as written :527-536) ----
//   dpred4[b][t][0:2] += scale * (p_hat - p)   for rows b in [row0, row1)
__global__ __launch_bounds__(256) void l2_grad_kernel(const float* __restrict__ pred4, const float* __restrict__ gt,
                                                       int Tp, int row0, int row1, float scale,
                                                       float* __restrict__ dpred4) {
  const long long n = (long long)(row1 - row0) * Tp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const size_t e = (size_t)row0 * Tp + i;
    f32x4 p = ld4(pred4 + e * 4), d = ld4(dpred4 + e * 4);
    float2 g = *reinterpret_cast<const float2*>(gt + e * 2);
    d[0] = fmaf(scale, p[0] - g.x, d[0]);
    d[1] = fmaf(scale, p[1] - g.y, d[1]);
    st4(dpred4 + e * 4, d);
  }
}
extern "C" int sw_l2_grad(const float* pred4, const float* gt, int B, int Tp, int row0, int row1, float scale,
                          float* dpred4, void* stream) {
  if (!pred4 || !gt || !dpred4 || B < 1 || Tp < 1 || row0 < 0 || row1 > B || row0 > row1) return SW_EARG;
  if (row0 == row1) return SW_OK;
  long long n = (long long)(row1 - row0) * Tp;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  SW_LAUNCH(l2_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred4, gt, Tp, row0, row1, scale,
                     dpred4);
  SW_CHECK_LAUNCH("l2_grad_kernel");
  return SW_OK;
}

// ---- variety loss with its intended semantics (train.py:527-536 fixed; Social-GAN's best-of-K L2) ----
//   K rollouts of the same batch with independent z are folded into one batch of K*B rows (copy k = rows
//   [k*B, (k+1)*B)).  Per agent b: l2_k = mean_{t,c} (p_hat_k[b][t][c] - p[b][t][c])^2, k* = argmin_k l2_k (ties: the
//   smallest k), variety = mean_b l2_{k*}; the gradient reaches copy k* only:
//       dpred4[k* B + b][t][0:2] += scale * (p_hat - p),   scale = loss_l2_w / (B_global * Tp).
//   One wave per agent, lane = sample k (K <= 64), lexicographic (l2, k) minimum through a shuffle tree.
__global__ __launch_bounds__(256) void variety_grad_kernel(const float* __restrict__ predK, const float* __restrict__ gt,
                                                            int K, int B, int Tp, float scale, float* __restrict__ dpredK,
                                                            int* __restrict__ kmin_out, float* __restrict__ l2min_out) {
  const int lane = sw_lane(), b = blockIdx.x * 4 + sw_wave();
  if (b >= B) return;
  float l2 = 3.0e38f;
  if (lane < K) {
    float s = 0.f;
    const float* ph = predK + ((size_t)lane * B + b) * Tp * 4;
    for (int t = 0; t < Tp; ++t) {
      const f32x4 p = ld4(ph + t * 4);
      const float2 g = *reinterpret_cast<const float2*>(gt + ((size_t)b * Tp + t) * 2);
      const float dx = p[0] - g.x, dy = p[1] - g.y;
      s = fmaf(dx, dx, fmaf(dy, dy, s));
    }
    l2 = s / (float)(2 * Tp);
  }
  int km = lane;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ol = __shfl_xor(l2, o);
    const int ok = __shfl_xor(km, o);
    if (ol < l2 || (ol == l2 && ok < km)) {
      l2 = ol;
      km = ok;
    }
  }
  if (lane == 0) {
    if (kmin_out) kmin_out[b] = km;
    if (l2min_out) l2min_out[b] = l2;
  }
  for (int t = lane; t < Tp; t += 64) {
    const size_t e = ((size_t)km * B + b) * Tp + t;
    f32x4 p = ld4(predK + e * 4), d = ld4(dpredK + e * 4);
    const float2 g = *reinterpret_cast<const float2*>(gt + ((size_t)b * Tp + t) * 2);
    d[0] = fmaf(scale, p[0] - g.x, d[0]);
    d[1] = fmaf(scale, p[1] - g.y, d[1]);
    st4(dpredK + e * 4, d);
  }
}
extern "C" int sw_variety_grad(const float* predK, const float* gt, int K, int B, int Tp, float scale, float* dpredK,
                               int* kmin, float* l2min, void* stream) {
  if (!predK || !gt || !dpredK || B < 1 || Tp < 1 || K < 1) return SW_EARG;
  if (K > 64) return SW_ESHAPE;
  SW_LAUNCH(variety_grad_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, predK, gt, K, B, Tp, scale,
                     dpredK, kmin, l2min);
  SW_CHECK_LAUNCH("variety_grad_kernel");
  return SW_OK;
}

// ---- toy statistics: pairwise mean displacement between sample sets (calc_statistics.py:28-32, 56-60) ----
//   D[k][i][j] = mean_{t >= t0} || a[i][k][t] - b[j][k][t] ||
__global__ __launch_bounds__(256) void traj_dist_kernel(const float* __restrict__ a, const float* __restrict__ b, int Na,
                                                         int Nb, int nPed, int T, int t0, float* __restrict__ D) {
  const long long n = (long long)nPed * Na * Nb;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int j = (int)(e % Nb), i = (int)((e / Nb) % Na), k = (int)(e / ((long long)Nb * Na));
    const float2* pa = reinterpret_cast<const float2*>(a) + ((size_t)i * nPed + k) * T;
    const float2* pb = reinterpret_cast<const float2*>(b) + ((size_t)j * nPed + k) * T;
    float s = 0.f;
    for (int t = t0; t < T; ++t) {
      float2 x = pa[t], y = pb[t];
      float dx = x.x - y.x, dy = x.y - y.y;
      s += sqrtf(dx * dx + dy * dy);
    }
    D[e] = s / (float)(T - t0);
  }
}
extern "C" int sw_traj_dist(const float* a, const float* b, int Na, int Nb, int nPed, int T, int t0, float* D,
                            void* stream) {
  if (!a || !b || !D || Na < 1 || Nb < 1 || nPed < 1 || T < 1 || t0 < 0 || t0 >= T) return SW_EARG;
  long long n = (long long)nPed * Na * Nb;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  SW_LAUNCH(traj_dist_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, Na, Nb, nPed, T, t0, D);
  SW_CHECK_LAUNCH("traj_dist_kernel");
  return SW_OK;
}

// ---- derived weight images of the generator (swimg, sw_common.h) ---------------------------------------------------
#define SW_IMG_BLOCKS 64
__device__ __forceinline__ void gen_images_block(const float* __restrict__ enc_w, const float* __restrict__ dec_w,
                                                 const float* __restrict__ emb_w, const float* __restrict__ att_w,
                                                 float* __restrict__ img, int blk) {
  using namespace swp;
  if (blk == 0) {          // composed input matrix: the very code the kernels ran per workgroup (bit-identical values)
    lstm_prep_rows(enc_w + ENC_EMB_W, enc_w + ENC_EMB_B, enc_w + ENC_WIH, enc_w + ENC_BIH, enc_w + ENC_BHH, true,
                   img + swimg::WX, img + swimg::BX);
    return;
  }
  if (blk == 1) {          // fc4 . fc3: same partial-sum order as the kernels' own fallback code
    const int t = threadIdx.x;
    if (t < 160) {
      const int c = t / 80, k = t - c * 80;
      const float* w4 = dec_w + DEC_W4 + c * 40;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll
      for (int m = 0; m < 40; m += 4) {
        v0 = fmaf(w4[m], dec_w[DEC_W3 + m * 80 + k], v0);
        v1 = fmaf(w4[m + 1], dec_w[DEC_W3 + (m + 1) * 80 + k], v1);
        v2 = fmaf(w4[m + 2], dec_w[DEC_W3 + (m + 2) * 80 + k], v2);
        v3 = fmaf(w4[m + 3], dec_w[DEC_W3 + (m + 3) * 80 + k], v3);
      }
      img[swimg::W43 + c * 80 + k] = (v0 + v1) + (v2 + v3);
    } else if (t < 162) {
      const int c = t - 160;
      float v = dec_w[DEC_B4 + c];
#pragma unroll
      for (int m = 0; m < 40; m += 4) {   // the order dec_rollout_fwd uses (bit-identical b43)
        const f32x4 w = ld4(dec_w + DEC_W4 + c * 40 + m), bb = ld4(dec_w + DEC_B3 + m);
        v = fmaf(w[0], bb[0], fmaf(w[1], bb[1], fmaf(w[2], bb[2], fmaf(w[3], bb[3], v))));
      }
      img[swimg::W43 + 160 + c] = v;
    }
    return;
  }
  const int nth = (SW_IMG_BLOCKS - 2) * 256, tid = (blk - 2) * 256 + threadIdx.x;
  // snapshot of the raw weights behind the compositions (swimg::RAW_*), float4 granularity
  for (int i = tid; i < (swimg::RAW_END - swimg::RAW_WIH) / 4; i += nth) {
    const int o = swimg::RAW_WIH + 4 * i;
    const float* src = o < swimg::RAW_WE   ? enc_w + ENC_WIH + (o - swimg::RAW_WIH)
                       : o < swimg::RAW_BE ? enc_w + ENC_EMB_W + (o - swimg::RAW_WE)
                       : o < swimg::RAW_W3 ? enc_w + ENC_EMB_B + (o - swimg::RAW_BE)
                       : o < swimg::RAW_B3 ? dec_w + DEC_W3 + (o - swimg::RAW_W3)
                       : o < swimg::RAW_W4 ? dec_w + DEC_B3 + (o - swimg::RAW_B3)
                                           : dec_w + DEC_W4 + (o - swimg::RAW_W4);
    st4(img + o, ld4(src));
  }
  // MFMA A-operand images (swimg::OP_*) of W[rows][c0 + K] and of transposes M = W^T (M[m][k] = W[k][m])
  auto op_image = [&](int dst0, const float* W, int ldw, int c0, int KJ, int ntile) {
    for (int f = tid; f < ntile * KJ * 64; f += nth) {
      const int t = f / (KJ * 64), rem = f - t * (KJ * 64), j = rem >> 6, l = rem & 63;
      st4(img + dst0 + 4 * (size_t)f, ld4(W + (size_t)(16 * t + (l & 15)) * ldw + c0 + 16 * j + 4 * (l >> 4)));
    }
  };
  auto op_image_T = [&](int dst0, const float* W, int ldw, int KJ, int ntile) {
    for (int f = tid; f < ntile * KJ * 64; f += nth) {
      const int t = f / (KJ * 64), rem = f - t * (KJ * 64), j = rem >> 6, l = rem & 63;
      const float* col = W + (size_t)(16 * j + 4 * (l >> 4)) * ldw + 16 * t + (l & 15);
      st4(img + dst0 + 4 * (size_t)f, f32x4{col[0], col[ldw], col[2 * ldw], col[3 * ldw]});
    }
  };
  op_image(swimg::OP_WHH, enc_w + ENC_WHH, 64, 0, 4, 16);
  op_image(swimg::OP_W1H, dec_w + DEC_W1, 160, 0, 4, 10);
  op_image(swimg::OP_W1SZ, dec_w + DEC_W1, 160, 64, 6, 10);
  op_image(swimg::OP_W2, dec_w + DEC_W2, 160, 0, 10, 5);
  op_image_T(swimg::OP_WHHT, enc_w + ENC_WHH, 64, 16, 4);
  op_image_T(swimg::OP_W2T, dec_w + DEC_W2, 160, 5, 10);
  op_image_T(swimg::OP_W1HT, dec_w + DEC_W1, 160, 10, 4);
  for (int f = tid; f < 8 * 2 * 4 * 64; f += nth) {      // OP_WHH8 (sw_common.h): the 8-wave row assignment of W_hh
    const int l = f & 63, j = (f >> 6) & 3, tile = (f >> 8) & 1, w = f >> 9, m = l & 15;
    const int row = (2 * tile + ((m & 3) >> 1)) * 64 + 8 * w + 2 * (m >> 2) + (m & 1);
    st4(img + swimg::OP_WHH8 + 4 * (size_t)f, ld4(enc_w + ENC_WHH + (size_t)row * 64 + 16 * j + 4 * (l >> 4)));
  }
  if (emb_w) {
    op_image(swimg::OP_E1, emb_w + EMB_W1, 32, 0, 2, 4);
    op_image(swimg::OP_E2, emb_w + EMB_W2, 64, 0, 4, 4);
    op_image_T(swimg::OP_E1T, emb_w + EMB_W1, 32, 4, 2);
    op_image_T(swimg::OP_E2T, emb_w + EMB_W2, 64, 4, 4);
    op_image_T(swimg::OP_ATT_T, att_w + ATT_W, 64, 4, 4);
  }
}
__global__ __launch_bounds__(256) void gen_images_kernel(const float* __restrict__ enc_w, const float* __restrict__ dec_w,
                                                          const float* __restrict__ emb_w, const float* __restrict__ att_w,
                                                          float* __restrict__ img) {
  gen_images_block(enc_w, dec_w, emb_w, att_w, img, blockIdx.x);
}
// The registration is PER HOST THREAD (a step registers, launches and drops on one thread): trainers stepping on different
// threads do not see or clear each other's images.
static thread_local const float *g_img_enc = nullptr, *g_img_dec = nullptr, *g_img_emb = nullptr, *g_img_att = nullptr,
                                *g_img = nullptr;
const float* sw_soc_images_for(const float* emb_w, const float* att_w) {
  return (g_img && emb_w && emb_w == g_img_emb && att_w == g_img_att) ? g_img : nullptr;
}
const float* sw_gen_images_for(const float* enc_w, const float* dec_w) {   // dec_w null: the encoder part alone
  return (g_img && enc_w == g_img_enc && (!dec_w || dec_w == g_img_dec)) ? g_img : nullptr;
}
extern "C" int sw_gen_image_floats(void) { return swimg::N; }
static void gen_images_register(const float* enc_w, const float* dec_w, const float* emb_w, const float* att_w,
                                const float* img) {
  g_img_enc = enc_w; g_img_dec = dec_w; g_img_emb = emb_w; g_img_att = att_w; g_img = img;
}
extern "C" int sw_gen_images(const float* enc_w, const float* dec_w, const float* emb_w, const float* att_w, float* img,
                             void* stream) {
  if (!img) {                       // unregister: the weights are about to change (or have changed)
    gen_images_register(nullptr, nullptr, nullptr, nullptr, nullptr);
    return SW_OK;
  }
  if (!enc_w || !dec_w || ((emb_w != nullptr) != (att_w != nullptr))) return SW_EARG;
  SW_LAUNCH(gen_images_kernel, dim3(SW_IMG_BLOCKS), dim3(256), 0, (hipStream_t)stream, enc_w, dec_w, emb_w, att_w, img);
  SW_CHECK_LAUNCH("gen_images_kernel");
  gen_images_register(enc_w, dec_w, emb_w, att_w, img);
  return SW_OK;
}

// ---- one-kernel input staging of a hipGraph-replayed training step ---------------------------------
// `slot` is a host-pinned (device-mapped) buffer the host fills before every replay, 4-byte words:
//   [0,1] device pointer of obsv (B,To,2)   [2,3] device pointer of pred (B,Tp,2)
//   [4] zeros_val  [5] ones_val (train.py:471-472)   [6] D updates applied so far  [7] G updates applied so far
//   [8 ..] z (B*32, train.py:473) - or, z_device = 1, [8,9] = device pointer of a z the caller already holds in HBM
// The kernel is a node of the captured graph with FIXED arguments; what changes per step travels through
// the slot.  It copies the tracks into the graph's static buffers, forms the real future as (p, v) rows
// (get_traj_4d, train.py:135-137) and pulls the scalars + z over PCIe.
__global__ __launch_bounds__(256) void stage_step_kernel(const float* __restrict__ slot, int B, int To, int Tp,
                                                          float* __restrict__ obsv_dst, float* __restrict__ pred_dst,
                                                          float* __restrict__ pred4_dst, float* __restrict__ targets_dst,
                                                          float* __restrict__ z_dst, float* __restrict__ steps_dst,
                                                          int n_d_updates, const float* __restrict__ enc_w,
                                                          const float* __restrict__ dec_w, const float* __restrict__ emb_w,
                                                          const float* __restrict__ att_w, float* __restrict__ img,
                                                          int img_blocks, const float* __restrict__ d_w,
                                                          float* __restrict__ d_img, const int* __restrict__ d_tab, int d_n,
                                                          int dimg_blocks, int z_device) {
  // the last img_blocks workgroups derive the generator's weight images of this step (sw_gen_images), the dimg_blocks
  // in front of them scatter the discriminator's weights into theirs (sw_disc_images)
  if ((int)blockIdx.x >= (int)gridDim.x - img_blocks) {
    gen_images_block(enc_w, dec_w, emb_w, att_w, img, (int)blockIdx.x - ((int)gridDim.x - img_blocks));
    return;
  }
  if ((int)blockIdx.x >= (int)gridDim.x - img_blocks - dimg_blocks) {
    disc_images_scatter(d_w, d_img, d_tab, d_n, (int)blockIdx.x - ((int)gridDim.x - img_blocks - dimg_blocks), dimg_blocks);
    return;
  }
  const unsigned long long* ptrs = reinterpret_cast<const unsigned long long*>(slot);
  const float* obsv = reinterpret_cast<const float*>(ptrs[0]);
  const float* pred = reinterpret_cast<const float*>(ptrs[1]);
  const int gid = blockIdx.x * 256 + threadIdx.x, gsz = ((int)gridDim.x - img_blocks - dimg_blocks) * 256;
  if (z_dst) {
    const float* zs = z_device ? reinterpret_cast<const float*>(ptrs[4]) : slot + 8;
    for (int i = gid; i < B * SW_Z / 4; i += gsz) st4(z_dst + 4 * (size_t)i, ld4(zs + 4 * (size_t)i));
  }
  if (gid < 2) targets_dst[gid] = slot[4 + gid];
  // 1-based Adam step indices of this training step's updates: D update u -> [u], the G update -> [n_d_updates]
  if (steps_dst && gid <= n_d_updates) steps_dst[gid] = gid < n_d_updates ? slot[6] + 1.0f + (float)gid : slot[7] + 1.0f;
  for (int i = gid; i < B * To; i += gsz)
    *reinterpret_cast<float2*>(obsv_dst + 2 * (size_t)i) = *reinterpret_cast<const float2*>(obsv + 2 * (size_t)i);
  for (int k = gid; k < B * Tp; k += gsz) {
    const int b = k / Tp, t = k - b * Tp;
    const float2 p = *reinterpret_cast<const float2*>(pred + 2 * (size_t)k);
    const float2 q = *reinterpret_cast<const float2*>(t == 0 ? obsv + ((size_t)b * To + To - 1) * 2 : pred + 2 * (size_t)k - 2);
    *reinterpret_cast<float2*>(pred_dst + 2 * (size_t)k) = p;
    st4(pred4_dst + 4 * (size_t)k, f32x4{p.x, p.y, p.x - q.x, p.y - q.y});
  }
}
#define SW_DIMG_BLOCKS 16
extern "C" int sw_stage_step_zdev(const float* slot, int B, int To, int Tp, float* obsv_dst, float* pred_dst,
                                  float* pred4_dst, float* targets_dst, float* z_dst, float* steps_dst, int n_d_updates,
                                  const float* enc_w, const float* dec_w, const float* emb_w, const float* att_w, float* img,
                                  const float* d_w, float* d_img, const int* d_tab, int z_device, void* stream) {
  if (z_device && !z_dst) return SW_EARG;
  if (!slot || !obsv_dst || !pred_dst || !pred4_dst || !targets_dst || B < 1 || To < 2 || Tp < 1 ||
      n_d_updates < 0 || n_d_updates > 254)
    return SW_EARG;
  if (img && (!enc_w || !dec_w || ((emb_w != nullptr) != (att_w != nullptr)))) return SW_EARG;
  if (d_img && (!d_w || !d_tab || Tp > 64)) return SW_EARG;
  int n = z_dst ? B * SW_Z / 4 : B * (To > Tp ? To : Tp);
  int blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  const int ib = img ? SW_IMG_BLOCKS : 0, db = d_img ? SW_DIMG_BLOCKS : 0;
  SW_LAUNCH(stage_step_kernel, dim3(blocks + ib + db), dim3(256), 0, (hipStream_t)stream, slot, B, To, Tp, obsv_dst,
                     pred_dst, pred4_dst, targets_dst, z_dst, steps_dst, n_d_updates, enc_w, dec_w, emb_w, att_w, img, ib,
                     d_w, d_img, d_tab, d_img ? swp::disc(Tp).n : 0, db, z_device ? 1 : 0);
  SW_CHECK_LAUNCH("stage_step_kernel");
  if (img) gen_images_register(enc_w, dec_w, emb_w, att_w, img);
  if (d_img) sw_disc_images_register(d_w, d_img, d_tab, Tp);
  return SW_OK;
}
extern "C" int sw_stage_step_img(const float* slot, int B, int To, int Tp, float* obsv_dst, float* pred_dst,
                                 float* pred4_dst, float* targets_dst, float* z_dst, float* steps_dst, int n_d_updates,
                                 const float* enc_w, const float* dec_w, const float* emb_w, const float* att_w, float* img,
                                 const float* d_w, float* d_img, const int* d_tab, void* stream) {
  return sw_stage_step_zdev(slot, B, To, Tp, obsv_dst, pred_dst, pred4_dst, targets_dst, z_dst, steps_dst, n_d_updates, enc_w,
                            dec_w, emb_w, att_w, img, d_w, d_img, d_tab, 0, stream);
}
extern "C" int sw_stage_step(const float* slot, int B, int To, int Tp, float* obsv_dst, float* pred_dst,
                             float* pred4_dst, float* targets_dst, float* z_dst, float* steps_dst, int n_d_updates,
                             void* stream) {
  return sw_stage_step_img(slot, B, To, Tp, obsv_dst, pred_dst, pred4_dst, targets_dst, z_dst, steps_dst, n_d_updates, nullptr,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
