// Device building blocks of the fused sparse evaluation (sgpr_fused.h): staging, row-lane distances, the 64 x 64 chain, image moves.
// Kernel-free: included by sf_cell.hip, sf_pass1.hip and sf_pass2.hip.
#pragma once
#include <type_traits>

#include "chain64.h"
#include "gprx_common.h"
#include "kfun.h"
#include "sgpr_fused.h"
#include "sgpr_small_ops.h"

namespace gprx {

// ---- staging --------------------------------------------------------------------------------------------------------
// 64 points x 16 dimensions [k0, k0 + 16) of `pts` (row-major, d per point), scaled by the lengthscales, into dst[64][SF_DKP].
// Points >= nvalid and dimensions >= d are staged as zeros (they add exact zeros to every sum).  FORM as kmat.h: 0 = x * (1 / l), the
// kernel build's difference form; 1 = x / l, gpflow's literal arithmetic.
template <int FORM, int NT = 256>
__device__ __forceinline__ void sf_stage(const double* __restrict__ pts, int base, int nvalid, int d, int k0, const double* __restrict__ ls,
                                         double* __restrict__ dst, int tid) {
  constexpr int REPS = NB * SF_DK / NT, PSTEP = NT / SF_DK;  // values per thread, points per repetition
  const int kk = tid & 15;
  const int kc = min(k0 + kk, d - 1);
  const bool live = k0 + kk < d;
  const double l = ls[kc];
  double raw[REPS];
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int pt = (tid >> 4) + PSTEP * rep;
    raw[rep] = pts[(int64_t)min(base + pt, nvalid - 1) * d + kc];  // (unconditional loads from clamped indices, masks afterwards)
  }
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int pt = (tid >> 4) + PSTEP * rep;
    double v;
    if constexpr (FORM == 0) {
      const double inv = 1.0 / l;
      v = raw[rep] * inv;
    } else {
      v = raw[rep] / l;
    }
    dst[pt * SF_DKP + kk] = (live && base + pt < nvalid) ? v : 0.0;
  }
}

// The same in two halves, so that the loads of the NEXT tile's points travel while the current tile is computed: sf_stage_fetch issues
// them (raw values in registers), sf_stage_put scales and stores.  8 waves: 2 values per thread.
template <int NT>
__device__ __forceinline__ void sf_stage_fetch(const double* __restrict__ pts, int base, int nvalid, int d, int k0, double (&raw)[NB * SF_DK / NT],
                                               int tid) {
  constexpr int REPS = NB * SF_DK / NT, PSTEP = NT / SF_DK;
  const int kc = min(k0 + (tid & 15), d - 1);
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) raw[rep] = pts[(int64_t)min(base + (tid >> 4) + PSTEP * rep, nvalid - 1) * d + kc];
}
template <int FORM, int NT>
__device__ __forceinline__ void sf_stage_put(const double (&raw)[NB * SF_DK / NT], int base, int nvalid, int d, int k0, const double* __restrict__ ls,
                                             double* __restrict__ dst, int tid) {
  constexpr int REPS = NB * SF_DK / NT, PSTEP = NT / SF_DK;
  const int kk = tid & 15;
  const bool live = k0 + kk < d;
  const double l = ls[min(k0 + kk, d - 1)];
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int pt = (tid >> 4) + PSTEP * rep;
    double v;
    if constexpr (FORM == 0) {
      const double inv = 1.0 / l;
      v = raw[rep] * inv;
    } else {
      v = raw[rep] / l;
    }
    dst[pt * SF_DKP + kk] = (live && base + pt < nvalid) ? v : 0.0;
  }
}

// Squared scaled distances between this lane's row point and the 16 column points of its wave, one staged chunk of dimensions (dk live
// ones; the staged zeros beyond them add exact zeros).  Loop order: dimension pairs outside, columns inside -- the row point's coordinates
// pass through two registers instead of living in sixteen.
// FORM 0: r2 += sum_k (z_k - x_k)^2 as an fma chain in k order (kmat.h's difference form, same bits).
// FORM 1: the three parts of gpflow's square_distance: na, nb (rounded squares added in k order) and the dot product (fma chain).
// The 16-byte LDS reads of one pair of dimensions: this lane's row point and the NC column points of its wave (wave-uniform addresses:
// broadcast reads).
template <int NC>
__device__ __forceinline__ void sf_load_pair(const double* __restrict__ sRow, const double* __restrict__ sCol, int lane, int wave, int kk, d2& zv,
                                             d2 (&xv)[NC]) {
  zv = *reinterpret_cast<const d2*>(sRow + lane * SF_DKP + kk);
#pragma unroll
  for (int jj = 0; jj < NC; ++jj) xv[jj] = *reinterpret_cast<const d2*>(sCol + (wave * NC + jj) * SF_DKP + kk);
}
template <int FORM, int NC>
__device__ __forceinline__ void sf_r2_pair(const d2& zv, const d2 (&xv)[NC], double (&r2)[NC], double& na, double (&nb)[NC]) {
  if constexpr (FORM != 0) {
#pragma clang fp contract(off)
    const double s0 = zv.x * zv.x;
    na = na + s0;
    const double s1 = zv.y * zv.y;
    na = na + s1;
  }
#pragma unroll
  for (int jj = 0; jj < NC; ++jj) {
    if constexpr (FORM == 0) {
      const double d0 = zv.x - xv[jj].x, d1 = zv.y - xv[jj].y;
      r2[jj] = __builtin_fma(d0, d0, r2[jj]);
      r2[jj] = __builtin_fma(d1, d1, r2[jj]);
    } else {
      {
#pragma clang fp contract(off)
        const double s0 = xv[jj].x * xv[jj].x;
        nb[jj] = nb[jj] + s0;
        const double s1 = xv[jj].y * xv[jj].y;
        nb[jj] = nb[jj] + s1;
      }
      r2[jj] = __builtin_fma(zv.x, xv[jj].x, r2[jj]);
      r2[jj] = __builtin_fma(zv.y, xv[jj].y, r2[jj]);
    }
  }
}

// NP > 0: the first NP pairs of dimensions unconditionally (staged zeros beyond d add exact zeros), software-pipelined by hand: the reads
// of pair p + 1 are issued before the arithmetic of pair p, and scheduling barriers keep them there -- left alone the compiler sinks
// every read to its use and waits out its LDS latency each time (48 serialised round trips per tile: 4 k clocks of a 14 k tile).
template <int FORM, int NC = 16, int NP = 0>
__device__ __forceinline__ void sf_r2_chunk(const double* __restrict__ sRow, const double* __restrict__ sCol, int lane, int wave, int dk,
                                            double (&r2)[NC], double& na, double (&nb)[NC]) {
  if constexpr (NP > 0) {
    d2 za, zb, xa[NC], xb[NC];
    sf_load_pair<NC>(sRow, sCol, lane, wave, 0, za, xa);
#pragma unroll
    for (int pp = 0; pp < NP; pp += 2) {
      if (pp + 1 < NP) sf_load_pair<NC>(sRow, sCol, lane, wave, 2 * (pp + 1), zb, xb);
      __builtin_amdgcn_sched_barrier(0);
      sf_r2_pair<FORM, NC>(za, xa, r2, na, nb);
      __builtin_amdgcn_sched_barrier(0);
      if (pp + 2 < NP) sf_load_pair<NC>(sRow, sCol, lane, wave, 2 * (pp + 2), za, xa);
      __builtin_amdgcn_sched_barrier(0);
      if (pp + 1 < NP) sf_r2_pair<FORM, NC>(zb, xb, r2, na, nb);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int kk = 0; kk < SF_DK; kk += 2) {
      if (kk < dk) {  // (a wave-uniform guard, not a break: the loop unrolls completely and every accumulator index stays static)
        d2 zv, xv[NC];
        sf_load_pair<NC>(sRow, sCol, lane, wave, kk, zv, xv);
        sf_r2_pair<FORM, NC>(zv, xv, r2, na, nb);
      }
    }
  }
}

// sum_c src[c * stride] for c = 0 .. count - 1, added in that order; the loads go out eight at a time (a rolled loop waits for every
// load before it issues the next: 17 dependent round trips to memory another CU wrote cost ~14 us in the first version of sf_final)
__device__ __forceinline__ double sf_sum_chunks(const double* __restrict__ src, int64_t stride, int count) {
  double acc = 0.0;
  int c = 0;
  for (; c + 8 <= count; c += 8) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[(int64_t)(c + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += t[u];
  }
  if (c < count) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[(int64_t)min(c + u, count - 1) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (c + u < count) acc += t[u];
  }
  return acc;
}

// the same for NE outputs of one thread at once (offsets off[u]; clamped duplicates for outputs a thread does not have): NE x 8 loads in
// flight per batch.  Each output's additions are exactly sf_sum_chunks's.
template <int NE>
__device__ __forceinline__ void sf_sum_chunks_n(const double* __restrict__ src, const int (&off)[NE], int64_t stride, int count, double (&acc)[NE]) {
#pragma unroll
  for (int u = 0; u < NE; ++u) acc[u] = 0.0;
  int c = 0;
  for (; c + 8 <= count; c += 8) {
    double t[NE][8];
#pragma unroll
    for (int u = 0; u < NE; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) t[u][k] = src[(int64_t)(c + k) * stride + off[u]];
#pragma unroll
    for (int u = 0; u < NE; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[u] += t[u][k];
  }
  if (c < count) {
    double t[NE][8];
#pragma unroll
    for (int u = 0; u < NE; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) t[u][k] = src[(int64_t)min(c + k, count - 1) * stride + off[u]];
#pragma unroll
    for (int u = 0; u < NE; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (c + k < count) acc[u] += t[u][k];
  }
}

// ---- the 64 x 64 chain: Cholesky factor and its inverse in one workgroup ------------------------------------------------------
// sImg: the symmetric matrix (lower triangle read), row stride ld.  On return acc[0] holds L, acc[1] holds L^-T, both in the MFMA C
// layout of tile_ops.h chain_step (wave w: rows 16 w .. 16 w + 15).  Returns the 1-based failing pivot or 0.
__device__ __forceinline__ int sf_chain(const double* __restrict__ sImg, int ld, double* __restrict__ sIn, double* __restrict__ sX, d4 (&acc)[2][4],
                                        int tid, unsigned long long* stamps = nullptr) {
#define SF_CHAIN_STAMP(i_) \
  if (stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) stamps[8 + (i_)] = __builtin_amdgcn_s_memtime();
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * wave + g + 4 * q, col = 16 * kt + r;
      const double v = sImg[row * ld + col];
      acc[0][kt][q] = col <= row ? v : 0.0;
      acc[1][kt][q] = col == row ? 1.0 : 0.0;
    }
  ChainCtx c;
  c.sIn = sIn;
  c.sX = sX;
  c.tid = tid;
  c.wave = wave;
  c.g = g;
  c.r = r;
  c.bad = 0;
  SF_CHAIN_STAMP(0)
  chain_step<0>(acc, c);
  SF_CHAIN_STAMP(1)
  chain_step<1>(acc, c);
  SF_CHAIN_STAMP(2)
  chain_step<2>(acc, c);
  SF_CHAIN_STAMP(3)
  chain_step<3>(acc, c);
  SF_CHAIN_STAMP(4)
  chain_step<4>(acc, c);
  SF_CHAIN_STAMP(5)
  chain_step<5>(acc, c);
  SF_CHAIN_STAMP(6)
  chain_step<6>(acc, c);
  SF_CHAIN_STAMP(7)
  chain_step<7>(acc, c);
  SF_CHAIN_STAMP(8)
#undef SF_CHAIN_STAMP
  return c.bad;
}

// 64 x 64 LDS image (row stride ld) -> row-major global block, 16 bytes per lane and instruction
__device__ __forceinline__ void sf_image_out(const double* __restrict__ sImg, int ld, double* __restrict__ dst, int tid) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int q = tid + 256 * e;  // 2048 chunks of 16 bytes: row = q >> 5, chunk = q & 31
    *reinterpret_cast<d2*>(dst + (q >> 5) * NB + 2 * (q & 31)) = *reinterpret_cast<const d2*>(sImg + (q >> 5) * ld + 2 * (q & 31));
  }
}
__device__ __forceinline__ void sf_image_in(const double* __restrict__ src, double* __restrict__ sImg, int ld, int tid) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int q = tid + 256 * e;
    *reinterpret_cast<d2*>(sImg + (q >> 5) * ld + 2 * (q & 31)) = *reinterpret_cast<const d2*>(src + (q >> 5) * NB + 2 * (q & 31));
  }
}

}  // namespace gprx
