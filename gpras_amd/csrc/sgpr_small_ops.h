// 64 x 64 building blocks of the sparse model's M x M algebra on LDS images (device inline functions only: no kernels, so any translation
// unit may include this).  Split out of sgpr.h in round 5 for the fused evaluation (sf_*.hip).
#pragma once
#include "gprx_common.h"

namespace gprx {

// W  = Qinv - Sinv - m m^T                         (weights of dELBO/dKuf, before the 1/s)
// GQ = (2 Qinv - Sinv - T - m m^T) / 2            (dELBO/dKuu;  T = Linv^T B Linv)
// (one rounding per operation, no contraction: two kernels evaluate these and must agree bit for bit)
__device__ __forceinline__ void sgpr_combine(double q, double s, double t, double mi, double mj, double& w, double& gq) {
#pragma clang fp contract(off)
  const double mm = mi * mj;
  w = q - s - mm;
  gq = 0.5 * (2.0 * q - s - t - mm);
}

// ---- M <= 64: the M x M algebra between the second factorisation and the contractions, one workgroup per cell ----
// 64 x 64 product in gemm_f64's operation order (stages of 16 along k; instruction j of a stage takes k = k0 + 4 g + j from
// lane group g; accumulators start at zero; alpha = 1, beta = 0), so the values equal those of launch_gemm bit for bit.
// TA: op(A)[i][k] = A[k][i], else A[i][k];  op(B)[k][j] = B[k][j].  Operands in LDS with row stride SM_LD.
constexpr int SM_LD = NB + 1;
template <bool TA>
__device__ __forceinline__ void mm64(const double* __restrict__ sA, const double* __restrict__ sB, d4 (&acc)[2][2], int wm, int wn, int g, int r) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k0 = 0; k0 < NB; k0 += 16) {
    double fa[2][4], fb[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int row = wm * 32 + a * 16 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + 4 * g + j;
        fa[a][j] = TA ? sA[k * SM_LD + row] : sA[row * SM_LD + k];
      }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = wn * 32 + b * 16 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[b][j] = sB[(k0 + 4 * g + j) * SM_LD + col];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
  }
}
__device__ __forceinline__ void mm64_store(const d4 (&acc)[2][2], double* __restrict__ sC, int wm, int wn, int g, int r) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) sC[(wm * 32 + a * 16 + g + 4 * q) * SM_LD + wn * 32 + b * 16 + r] = acc[a][b][q];
}
__device__ __forceinline__ void load64(const double* __restrict__ src, double* __restrict__ dst, int tid) {
  for (int e = tid; e < NB * NB; e += 256) dst[(e >> 6) * SM_LD + (e & 63)] = src[e];
}
// x = invD^T b for one 64 x 64 block, trsv_bwd_step's order: four groups of 16 rows, partial sums added in order
__device__ __forceinline__ void trsv_t64(const double* __restrict__ sInv, double* __restrict__ sb, double (*part)[NB], int tid) {
  const int t = tid & 63, grp = tid >> 6;
  double s = 0.0;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int mm = grp * 16 + m;
    s = __builtin_fma(sInv[mm * SM_LD + t], sb[mm], s);
  }
  part[grp][t] = s;
  __syncthreads();
  if (tid < NB) sb[tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
  __syncthreads();
}

}  // namespace gprx
