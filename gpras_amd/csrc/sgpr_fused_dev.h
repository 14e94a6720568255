// Device building blocks of the fused sparse evaluation (sgpr_fused.h): staging, row-lane distances, the 64 x 64 chain, image moves.
// Kernel-free: included by sf_cell.hip, sf_pass1.hip and sf_pass2.hip.
#pragma once
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
template <int FORM>
__device__ __forceinline__ void sf_stage(const double* __restrict__ pts, int base, int nvalid, int d, int k0, const double* __restrict__ ls,
                                         double* __restrict__ dst, int tid) {
  const int kk = tid & 15;
  const int kc = min(k0 + kk, d - 1);
  const bool live = k0 + kk < d;
  const double l = ls[kc];
  double raw[4];
#pragma unroll
  for (int rep = 0; rep < 4; ++rep) {
    const int pt = (tid >> 4) + 16 * rep;
    raw[rep] = pts[(int64_t)min(base + pt, nvalid - 1) * d + kc];  // (unconditional loads from clamped indices, masks afterwards)
  }
#pragma unroll
  for (int rep = 0; rep < 4; ++rep) {
    const int pt = (tid >> 4) + 16 * rep;
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
template <int FORM>
__device__ __forceinline__ void sf_r2_chunk(const double* __restrict__ sRow, const double* __restrict__ sCol, int lane, int wave, int dk,
                                            double (&r2)[16], double& na, double (&nb)[16]) {
#pragma unroll
  for (int kk = 0; kk < SF_DK; kk += 2) {
    if (kk < dk) {  // (a wave-uniform guard, not a break: the loop unrolls completely and every accumulator index stays static)
    const d2 zv = *reinterpret_cast<const d2*>(sRow + lane * SF_DKP + kk);
    if constexpr (FORM != 0) {
#pragma clang fp contract(off)
      const double s0 = zv.x * zv.x;
      na = na + s0;
      const double s1 = zv.y * zv.y;
      na = na + s1;
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const d2 xv = *reinterpret_cast<const d2*>(sCol + (wave * 16 + jj) * SF_DKP + kk);  // (wave-uniform address: an LDS broadcast read)
      if constexpr (FORM == 0) {
        const double d0 = zv.x - xv.x, d1 = zv.y - xv.y;
        r2[jj] = __builtin_fma(d0, d0, r2[jj]);
        r2[jj] = __builtin_fma(d1, d1, r2[jj]);
      } else {
        {
#pragma clang fp contract(off)
          const double s0 = xv.x * xv.x;
          nb[jj] = nb[jj] + s0;
          const double s1 = xv.y * xv.y;
          nb[jj] = nb[jj] + s1;
        }
        r2[jj] = __builtin_fma(zv.x, xv.x, r2[jj]);
        r2[jj] = __builtin_fma(zv.y, xv.y, r2[jj]);
      }
    }
    }
  }
}

// ---- the 64 x 64 chain: Cholesky factor and its inverse in one workgroup ------------------------------------------------------
// sImg: the symmetric matrix (lower triangle read), row stride ld.  On return acc[0] holds L, acc[1] holds L^-T, both in the MFMA C
// layout of tile_ops.h chain_step (wave w: rows 16 w .. 16 w + 15).  Returns the 1-based failing pivot or 0.
__device__ __forceinline__ int sf_chain(const double* __restrict__ sImg, int ld, double* __restrict__ sIn, double* __restrict__ sX, d4 (&acc)[2][4],
                                        int tid) {
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
  chain_step<0>(acc, c);
  chain_step<1>(acc, c);
  chain_step<2>(acc, c);
  chain_step<3>(acc, c);
  chain_step<4>(acc, c);
  chain_step<5>(acc, c);
  chain_step<6>(acc, c);
  chain_step<7>(acc, c);
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
