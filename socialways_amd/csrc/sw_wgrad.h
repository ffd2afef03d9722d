// sw_wgrad.h - grouped split-K weight-gradient GEMM (see sw_wgrad.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define SW_WG_MAXP 24
#define SW_WG_MAXSPLIT 256
// upper bound of the partial workspace one batch may need (floats): the largest batch (the
// generator's) has < 64K output elements
#define SW_WG_WS_FLOATS ((size_t)SW_WG_MAXSPLIT * 65536)

struct WgProblem {      // one column block (<= 64 act columns, optional trailing ones column) of a problem
  const float* delta;   // [R][ldd], N columns used
  const float* act;     // [R][lda], K columns used (already offset to the block's first column)
  float* dW;            // [N][ldw], already offset to the block's first column
  float* db;            // [N] or null: receives the ones column
  float* db2;           // second copy of the bias gradient (LSTM b_ih / b_hh) or null
  size_t ws_off;
  int ldd, lda, ldw, R, N, K, ones, accumulate;
  int nbn, nbk, nsplit, job0, out0;
  int pre;              // > 0: the `pre` slice partials are produced elsewhere (fused kernels); only reduced here
  // optional TAIL segment: K2 (< 16) more act columns from a second array, placed in the k tile after the K
  // (multiple of 16) columns of `act`, in front of the ones column - the LSTM's x_t next to h_{t-1}, so that the
  // dgates rows are fetched once for W_hh, W_ih and the biases.  Rows r < row0 have no `act` operand (t = 0).
  const float* act2;
  float* dW2;
  int lda2, ldw2, K2, row0;
};
struct WgBatch {
  // compact search keys first: a workgroup finds its problem by scanning these (kernel-argument memory is
  // fetched by dependent scalar loads: one or two cache lines here instead of one per problem descriptor)
  int job0s[SW_WG_MAXP], out0s[SW_WG_MAXP];
  WgProblem p[SW_WG_MAXP];
  int np = 0, total_jobs = 0, total_out = 0;
  size_t top_reserved = 0;   // floats handed out from the TOP of the workspace to precomputed-partial problems
};

// ---- Adam fused into the reduction: the thread that produces the final value of a gradient element also applies the
// update to its weight (torch's fused Adam, aten/src/ATen/native/cuda/fused_adam_utils.cuh, restated operation by
// operation incl. its float / double mix).  Valid when every parameter of the optimizer is an output of this very
// reduction (the discriminator pass) and no all-reduce sits between gradient and update (single process).
struct WgAdam {
  float* w = nullptr;          // packed weights; m, v: packed Adam moments; all indexed like the packed gradient buffer g0
  float* m = nullptr;
  float* v = nullptr;
  const float* g0 = nullptr;
  const float* step = nullptr; // device scalar: 1-based index of this update (float, like torch's state step)
  double lr = 0, beta1 = 0, beta2 = 0, eps = 0;
  size_t n = ~(size_t)0;       // floats in the packed buffers: gradient outputs outside [g0, g0 + n) are scratch, not parameters
  const float* bc = nullptr;   // {bias_correction1, sqrt(bias_correction2)} of this update if some earlier kernel of the
                               // sequence has computed them (wg_launch_adam: the GEMM launch), else every wave does
  float* img = nullptr;        // derived weight images kept current by the update (swdimg, sw_common.h): the new value of
  const int* tab = nullptr;    // element i also goes to img[tab[2i]], img[tab[2i+1]] (entries < 0: none)
};
// bias corrections of the update whose 1-based index sits in *step (torch: 1 - beta^step in double, then float)
__device__ __forceinline__ void wg_adam_bc_compute(const float* step, double beta1, double beta2, float& bc1, float& bc2s) {
  const float st = *step;
  bc1 = (float)(1 - pow(beta1, (double)st));
  bc2s = (float)sqrt(1 - pow(beta2, (double)st));
}
__device__ __forceinline__ void wg_adam_bc(const WgAdam& A, float& bc1, float& bc2s) {
  // two double-precision pow() cost a kernel of this size 2-3 us (measured: wgrad_reduce 7.2 -> 10.2 us): they are
  // computed ONCE, by one thread of the GEMM launch in front, wherever there is one
  if (A.bc) {
    bc1 = A.bc[0];
    bc2s = A.bc[1];
    return;
  }
  wg_adam_bc_compute(A.step, A.beta1, A.beta2, bc1, bc2s);
}
// The update in two halves so that a caller can issue the three state loads BEFORE the work that produces the
// gradient (their round trip then hides under it): wg_adam_pre() loads, wg_adam_fin() computes and stores.
struct WgAdamPre {
  size_t i;
  float m, v, w;
  int2 t;      // image places of the element (WgAdam::tab), fetched with the state
};
__device__ __forceinline__ WgAdamPre wg_adam_pre(const WgAdam& A, const float* gptr) {
  WgAdamPre s;
  s.i = (size_t)(gptr - A.g0);
  s.t = int2{-1, -1};
  if (s.i >= A.n) {          // scratch output, not a parameter
    s.i = ~(size_t)0;
    s.m = s.v = s.w = 0.f;
    return s;
  }
  s.m = A.m[s.i];
  s.v = A.v[s.i];
  s.w = A.w[s.i];
  if (A.img) s.t = reinterpret_cast<const int2*>(A.tab)[s.i];
  return s;
}
__device__ __forceinline__ void wg_adam_fin(const WgAdam& A, const WgAdamPre& s, float bc1, float bc2s, float grad) {
  if (s.i == ~(size_t)0) return;
  // double arithmetic, rounded on assignment.  exp_avg as a lerp, m + (1 - beta1) (g - m): of the candidate forms this
  // is the one that agrees with torch._fused_adam_ of this build on 99.8 % of random inputs bit for bit, exp_avg_sq
  // below on 100 % (tools/dbg/adam_probe.py); the remaining last-bit differences are fused-multiply-add placement
  const float m = (float)((double)s.m + (1 - A.beta1) * ((double)grad - (double)s.m));
  const float v = (float)(A.beta2 * s.v + (1 - A.beta2) * grad * grad);
  const float step_size = (float)(A.lr / bc1);
  const float denom = (float)((sqrtf(v) / bc2s) + A.eps);
  A.m[s.i] = m;
  A.v[s.i] = v;
  const float wn = s.w - step_size * m / denom;
  A.w[s.i] = wn;
  if (A.img) {
    if (s.t.x >= 0) A.img[s.t.x] = wn;
    if (s.t.y >= 0) A.img[s.t.y] = wn;
  }
}
__device__ __forceinline__ void wg_adam1(const WgAdam& A, float bc1, float bc2s, const float* gptr, float grad) {
  wg_adam_fin(A, wg_adam_pre(A, gptr), bc1, bc2s, grad);
}
int wg_reduce_launch_adam(WgBatch& b, float* ws, const WgAdam& ad, hipStream_t stream);
int wg_reduce_launch(WgBatch& b, float* ws, hipStream_t stream);
// GEMM + reduction; `ad` comes back with `bc` set when the GEMM launch computed the bias corrections (kernels that
// follow in the same stream - the composition back-propagation - may use them)
int wg_launch_adam(WgBatch& b, float* ws, WgAdam& ad, hipStream_t stream);

int wg_add(WgBatch& b, const float* delta, int ldd, const float* act, int lda, int R, int N, int K, float* dW,
           int ldw, float* db, float* db2, int accumulate);
// same with a tail segment (see WgProblem): K % 16 == 0, K <= 64, K2 + (db ? 1 : 0) <= 16
int wg_add_tail(WgBatch& b, const float* delta, int ldd, const float* act, int lda, int R, int N, int K, float* dW,
                int ldw, const float* act2, int lda2, int K2, float* dW2, int ldw2, int row0, float* db, float* db2,
                int accumulate);
double wg_total_work(const WgBatch& b);
size_t wg_finalize(WgBatch& b);
int wg_launch(WgBatch& b, float* ws, hipStream_t stream);
// A problem whose per-slice partials [nslices][N][K+1] (bias in column K) another kernel writes at ws + ws_off;
// ws_off is fixed here (allocated from the top of the workspace), the launch then only reduces it.
int wg_add_pre(WgBatch& b, int N, int K, float* dW, int ldw, float* db, int nslices);
int wg_launch_finalized(WgBatch& b, float* ws, hipStream_t stream);
struct sw_wgrad_batch;
WgBatch* wg_pending(sw_wgrad_batch* h);   // the batch inside a handle (null -> null)
