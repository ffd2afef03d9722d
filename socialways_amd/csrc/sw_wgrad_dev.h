// sw_wgrad_dev.h - device code of the grouped split-K weight-gradient GEMM (see sw_wgrad.hip): the per-job body of
// wgrad_partial_kernel.
#pragma once
#include <type_traits>
#include "sw_common.h"
#include "sw_wgrad.h"
#ifndef SW_WG_DEPTH
#define SW_WG_DEPTH 4
#endif
#ifndef SW_WG_DEPTH5
#define SW_WG_DEPTH5 2
#endif
#ifndef SW_WG_DEPTH_TAIL
#define SW_WG_DEPTH_TAIL SW_WG_DEPTH
#endif

#define SW_WG_RLD 69   // LDS row stride of a wave's 64 x (<= 69) block: 64 act columns + tail segment + ones

// NV consecutive floats as ONE vector load (dword / dwordx2 / dwordx3 / dwordx4)
template <int NV>
__device__ __forceinline__ void wg_ldv(float (&v)[NV], const float* p) {
  if constexpr (NV == 4) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(p);
    v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
  } else if constexpr (NV == 3) {
    struct __attribute__((packed, aligned(4))) P3 { float a, b, c; };
    const P3 q = *reinterpret_cast<const P3*>(p);
    v[0] = q.a; v[1] = q.b; v[2] = q.c;
  } else if constexpr (NV == 2) {
    const float2 q = *reinterpret_cast<const float2*>(p);
    v[0] = q.x; v[1] = q.y;
  } else {
    v[0] = p[0];
  }
}

// Tiles per lane vector: the smallest t with 16 t >= n and n % t == 0 (every lane's vector is then wholly live or wholly
// dead: no partial lane), t in {1, 2, 4} for delta columns, {1, 2, 3, 4} for act columns; 0 = no such t (the host
// rejects the problem: SW_ESHAPE - cannot happen for the layer widths of this model, all multiples of 4 or <= 16)
__host__ __device__ inline int wg_tiles(int n, bool allow3) {
  if (n <= 16) return 1;
  for (int t = (n + 15) >> 4; t <= 4; ++t)
    if ((t != 3 || allow3) && n % t == 0) return t;
  return 0;
}

// One wave = one job: a 64 x (<= 69) output block of one column block of one problem over one row slice, accumulated
// in NA x (KR + XT) MFMA tiles.  MFMA K dimension = rows (4 per instruction: lane group lg = row r0 + lg).
//
// Operand loads are VECTOR loads: lane ln fetches the NA consecutive delta columns n0 + NA ln .. and the KR consecutive
// act columns KR ln .. of its row - a 16-lane group reads one contiguous 64 NA / 64 KR-byte piece of the row - and
// feeds component i to MFMA tile i.  Tile i therefore covers the columns {NA m + i}: a permutation of the output rows /
// columns among the tiles, undone where the block is written out (`mine`).  (Round 1 loaded one dword per tile and
// lane: 9 load instructions of 64-byte pieces per 20 MFMAs; the waves waited 58 % of their cycles.)  The summation
// order of every output element - its rows, 4 per MFMA, in slice order - is unchanged: bit-identical results.
//   NA = min(4, tiles of 16 delta columns in this block), KR = tiles of real act columns (1..4),
//   XT = 1: one more k-tile holding the tail segment (act2, K2 columns) and / or the ones column at lanes ln = 0..
// Branch-free streaming body: loads are unconditional from clamped addresses and masked by 0/1 factors; the pipeline
// registers hold RAW loaded values (arithmetic attached to a load would sit in front of the loop's back edge and
// drain the pipeline once per DEPTH groups).
template <int NA, int KR, int K2, int ONES, int DSCALE = 1>
__device__ __forceinline__ void wg_run(const float* __restrict__ dbase, const float* __restrict__ abase, int ldd, int lda,
                                       int rbeg, int rend, int rmax, int acol, int bcol, int lg, int ln,
                                       float* __restrict__ mine,
                                       const float* __restrict__ abase2, int lda2, int row0, int xoff) {
  constexpr int KT = KR, XC = K2 + ONES;
  f32x4 acc[NA][KT];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[i][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // The tail segment (K2 = 4 columns of act2: the LSTM's x_t) and the ones column (bias gradient) do NOT get an MFMA
  // tile of their own - it would carry 5 (or 1) useful columns of 16, a fifth of the LSTM problem's matrix work -:
  // the lane that holds delta[row][n] multiplies it with the row's 4 x values / adds it up on the VALU, which runs
  // beside the matrix pipe; the per-row-group partial sums meet in two lane shuffles at the end of the slice.
  float xacc[NA][XC > 0 ? XC : 1];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
#pragma unroll
    for (int c = 0; c < (XC > 0 ? XC : 1); ++c) xacc[i][c] = 0.f;
  }
  // 4-row groups in flight (swept on the GPU at 8 waves per CU)
  constexpr int DEPTH = (K2 > 0 ? SW_WG_DEPTH_TAIL : SW_WG_DEPTH) * DSCALE;
  float a[DEPTH][NA], b[DEPTH][KT];
  f32x4 xq[DEPTH];
  auto load = [&](int r0, float (&av)[NA], float (&bv)[KT], f32x4& xv) {
    const int rc = min(r0 + lg, rmax);
    // 32-bit element offsets (the host rejects a problem whose rows x stride reach 2^31): one multiply-add per load
    // instead of a 64-bit multiply-add + shift-add
    wg_ldv<NA>(av, dbase + (unsigned)(rc * ldd + acol));
    if constexpr (K2 > 0) {
      wg_ldv<KR>(bv, abase + (unsigned)(max(rc, row0) * lda + bcol));   // rows below row0 have no `act` operand
      xv = ld4(abase2 + (unsigned)(rc * lda2));                          // the row's tail columns (same address for 16 lanes)
    } else {
      wg_ldv<KR>(bv, abase + (unsigned)(rc * lda + bcol));              // (row0 = 0 without a tail segment)
    }
  };
#pragma unroll
  for (int q = 0; q < DEPTH - 1; ++q) load(rbeg + 4 * q, a[q], b[q], xq[q]);
  // Row masks only (rows >= rend belong to the next slice; rows below row0 have no act operand - a zero delta row
  // already kills its products).  There are no COLUMN masks: a lane beyond the block's live delta / act columns
  // (clamped to the last live ones) only feeds output rows / columns that are never written out - an MFMA output
  // element mixes nothing but its own row and column.  (Peeling the unmasked interior into a branch of its own was
  // measured 1.8 x slower: memory operations under a branch cost the exact vmcnt bookkeeping, see DESIGN.md.)
  for (int r = rbeg; r < rend; r += 4 * DEPTH) {
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) {
      load(r + 4 * (q + DEPTH - 1), a[(q + DEPTH - 1) % DEPTH], b[(q + DEPTH - 1) % DEPTH], xq[(q + DEPTH - 1) % DEPTH]);
      asm volatile("" ::: "memory");   // the loads are issued HERE (DEPTH - 1 groups ahead), not sunk to their uses
#pragma unroll
      for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(a[q][i]));   // ... and group q is first touched here
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) asm volatile("" : "+v"(b[q][kt]));
      if constexpr (K2 > 0) asm volatile("" : "+v"(xq[q]));
      const int rr = r + 4 * q + lg;
      const float rs = rr < rend ? 1.0f : 0.0f;
      float av[NA], bv[KT];
#pragma unroll
      for (int i = 0; i < NA; ++i) av[i] = a[q][i] * rs;
      if constexpr (K2 > 0) {       // only problems with a tail segment have rows without an `act` operand (row0 > 0)
        const float rs0 = rr >= row0 ? 1.0f : 0.0f;
#pragma unroll
        for (int kt = 0; kt < KR; ++kt) bv[kt] = b[q][kt] * rs0;
      } else {
#pragma unroll
        for (int kt = 0; kt < KR; ++kt) bv[kt] = b[q][kt];
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          acc[i][kt] = SW_MFMA(av[i], bv[kt], acc[i][kt]);
        }
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        if constexpr (K2 > 0) {
#pragma unroll
          for (int c = 0; c < K2; ++c) xacc[i][c] = fmaf(av[i], xq[q][c], xacc[i][c]);
        }
        if constexpr (ONES) xacc[i][K2] += av[i];
      }
    }
  }
  // tile (i, kt) element (m = 4 lg + r, n = ln)  =  output row NA m + i, act column KR n + kt
#pragma unroll
  for (int i = 0; i < NA; ++i) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = NA * (4 * lg + r) + i;
        if (KR * ln + kt < xoff) mine[orow * SW_WG_RLD + KR * ln + kt] = acc[i][kt][r];      // xoff = K: live columns only
      }
    }
  }
  // extra columns xoff + c of output row NA ln + i: sum of the four row groups (lanes ln, ln + 16, ln + 32, ln + 48)
  if constexpr (XC > 0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
#pragma unroll
      for (int c = 0; c < XC; ++c) {
        float v = xacc[i][c];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lg == 0) mine[(NA * ln + i) * SW_WG_RLD + xoff + c] = v;
      }
    }
  }
}

// One workgroup-job = 4 waves = 4 consecutive row slices of one output block (summed through LDS at the end); every
// wave streams independently (no LDS, no barriers in the loop).  `red` = 4 x 64 x SW_WG_RLD floats of LDS.
#define SW_WG_RED_FLOATS (4 * 64 * SW_WG_RLD)
template <int DSCALE = 1>
__device__ __forceinline__ void wg_job(const WgBatch& batch, float* __restrict__ ws, int job, float* red) {
  const int lane = sw_lane(), wave = sw_wave(), ln = lane & 15, lg = lane >> 4;
  int p = 0;
#pragma unroll 1
  while (p + 1 < batch.np && job >= batch.job0s[p + 1]) ++p;
  const WgProblem& P = batch.p[p];
  const int j = job - P.job0;
  const int NB = (P.N + 63) >> 6;       // 64-row output blocks
  const int sg = j / NB, nb = j - sg * NB;
  const int s = sg * 4 + wave;          // this wave's row slice (may be empty)
  const int N = P.N, K = P.K, Kc = P.K + P.K2 + P.ones;
  const int n0 = nb * 64;
  const int Nb = min(64, N - n0);                       // live delta columns of this block
  const int NA = wg_tiles(Nb, false);                   // delta tiles
  const int KR = K > 0 ? wg_tiles(K, true) : 1;         // real act tiles (a block with only the ones column: one masked tile)
  const int nsub = P.nsplit * 4;
  const int rows_per = (((P.R + nsub - 1) / nsub) + 3) & ~3;
  const int rbeg = min(P.R, s * rows_per);
  const int rend = min(P.R, rbeg + rows_per);
  // the lane's NA delta columns n0 + NA ln + i and KR act columns KR ln + kt; bases clamped into the row, dead
  // components masked
  const int acol = n0 + max(0, min(NA * ln, Nb - NA));
  const int bcol = max(0, min(KR * ln, K - KR));
  float* mine = red + wave * 64 * SW_WG_RLD;
  // extra columns on the VALU: a tail segment of exactly 4 columns (K2; only next to K = 64) and / or the ones column
#define WG_CASE(na, kr, k2, on)                                                                                     \
  case ((na * 8 + kr) * 2 + (k2 ? 1 : 0)) * 2 + on:                                                                 \
    wg_run<na, kr, k2, on, DSCALE>(P.delta, P.act, P.ldd, P.lda, rbeg, rend, P.R - 1, acol, bcol, lg, ln,              \
                                   mine, P.act2 ? P.act2 : P.delta, P.act2 ? P.lda2 : P.ldd, P.row0, K);            \
    break;
#define WG_CASES(na)                                                                                                \
  WG_CASE(na, 1, 0, 0) WG_CASE(na, 2, 0, 0) WG_CASE(na, 3, 0, 0) WG_CASE(na, 4, 0, 0)                               \
  WG_CASE(na, 1, 0, 1) WG_CASE(na, 2, 0, 1) WG_CASE(na, 3, 0, 1) WG_CASE(na, 4, 0, 1)                               \
  WG_CASE(na, 4, 4, 0) WG_CASE(na, 4, 4, 1)
  switch (((NA * 8 + KR) * 2 + (P.K2 ? 1 : 0)) * 2 + (P.ones ? 1 : 0)) {
    WG_CASES(1) WG_CASES(2) WG_CASES(4)
  }
#undef WG_CASES
#undef WG_CASE
  sw_barrier();
  // one partial per workgroup (4 row slices summed): ws[ws_off + (sg*N + n)*Kc + k]
  float* out = ws + P.ws_off + (size_t)sg * N * Kc;
  const int rows = min(64, N - n0), cols = min(SW_WG_RLD, Kc);
  // element e = rr * cols + cc walks the block row-major; (rr, cc) advance incrementally (one division per thread)
  const int dq = SW_THREADS / cols, dr = SW_THREADS - dq * cols;
  int rr = threadIdx.x / cols, cc = threadIdx.x - rr * cols;
  const float* r0 = red, *r1 = red + 64 * SW_WG_RLD, *r2 = red + 2 * 64 * SW_WG_RLD, *r3 = red + 3 * 64 * SW_WG_RLD;
  for (int e = threadIdx.x; e < rows * cols; e += SW_THREADS) {
    const int o = rr * SW_WG_RLD + cc;
    const float v = (r0[o] + r1[o]) + (r2[o] + r3[o]);
    out[(size_t)(n0 + rr) * Kc + cc] = v;
    cc += dr;
    rr += dq;
    if (cc >= cols) {
      cc -= cols;
      ++rr;
    }
  }
}
