// Fused sparse evaluation (sgpr_fused.h), launch 2, for ONE kernel id: compiled with -DSF_KID=k (gpras_amd/_build.py).
#include "sgpr_fused_dev.h"

#ifndef SF_KID
#error "compile with -DSF_KID=0..4"
#endif
#define SF_CAT2(a, b) a##b
#define SF_CAT(a, b) SF_CAT2(a, b)

namespace gprx {

// ---- launch 2: per chunk of 256 columns, S_c = A' A'^T and u_c = A' y with A' = L^-1 Kuf ---------------------------------
// NKC == 1: d <= 16, Z is staged once per workgroup and this lane's coordinates stay in registers; NKC == 0: any d <= 64, restaged per
// tile and chunk of dimensions.
template <int KID, int FORM, int NKC>
__global__ __launch_bounds__(256, 2) void sf_pass1_kernel(SfParams p) {
  __shared__ __attribute__((aligned(16))) double sPA[NB * SF_LD];  // the tile of Kuf, then A' in its place
  __shared__ __attribute__((aligned(16))) double sZ[NB * SF_DKP];
  __shared__ __attribute__((aligned(16))) double sXc[NB * SF_DKP];
  __shared__ __attribute__((aligned(16))) double sTab[64];
  __shared__ double sY[NB];
  __shared__ double sU[4][NB];
  const int cell = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, r = lane & 15;
  const double* par = p.cpar + (int64_t)cell * CELL_PAR;
  const double* ls = par + CELL_PAR_LS;
  const double variance = par[0];
  const int unit = (int)par[2];
  double* A = p.arena + (int64_t)cell * p.ss;
  const double* zp = A + p.oZ;
  const double* yp = p.Y + (int64_t)unit * p.np;
  exp_tab_fill(sTab);
  // L^-1 as MFMA A-operand fragments: rows wm 32 + a 16 + r, k = 16 ks + 4 g + j
  double fl[2][4][4];
  {
    const double* Li = A + p.oLinv;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const double* src = Li + (wm * 32 + a * 16 + r) * NB + ks * 16 + 4 * g;
        const d2 lo = *reinterpret_cast<const d2*>(src), hi = *reinterpret_cast<const d2*>(src + 2);
        fl[a][ks][0] = lo.x; fl[a][ks][1] = lo.y; fl[a][ks][2] = hi.x; fl[a][ks][3] = hi.y;
      }
  }
  d4 accS[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) accS[a][b] = d4{0.0, 0.0, 0.0, 0.0};
  double uacc = 0.0;
  if constexpr (NKC == 1) sf_stage<FORM>(zp, 0, p.m, p.d, 0, ls, sZ, tid);  // (published by the first tile's staging barrier)
  const int ntiles = min(SF_TILES, (p.np - chunk * SF_CHUNK) / NB);
  for (int t = 0; t < ntiles; ++t) {
    const int j0 = chunk * SF_CHUNK + t * NB;
    double r2[16], nb[16], na = 0.0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) r2[jj] = nb[jj] = 0.0;
    for (int k0 = 0; k0 < p.d; k0 += SF_DK) {
      __syncthreads();  // the previous tile's (chunk's) readers of sXc / sY / sPA are done
      if constexpr (NKC != 1) sf_stage<FORM>(zp, 0, p.m, p.d, k0, ls, sZ, tid);
      sf_stage<FORM>(p.X, j0, p.n, p.d, k0, ls, sXc, tid);
      if (k0 == 0 && tid < NB) sY[tid] = yp[j0 + tid];  // (zero beyond n: Y is padded)
      __syncthreads();
      sf_r2_chunk<FORM>(sZ, sXc, lane, wave, min(SF_DK, p.d - k0), r2, na, nb);
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int col = wave * 16 + jj;
      double rr = r2[jj];
      if constexpr (FORM != 0) rr = expanded_r2(na, nb[jj], rr);
      const double pv = (lane < p.m && j0 + col < p.n) ? variance * corr_g<KID>(rr, sTab) : 0.0;
      sPA[lane * SF_LD + col] = pv;
    }
    __syncthreads();
    // A' = L^-1 P
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double fb[2][4];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[b][j] = sPA[(ks * 16 + 4 * g + j) * SF_LD + wn * 32 + b * 16 + r];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fl[a][ks][j], fb[b][j], acc[a][b], 0, 0, 0);
    }
    __syncthreads();  // every wave has read P
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) sPA[(wm * 32 + a * 16 + g + 4 * q) * SF_LD + wn * 32 + b * 16 + r] = acc[a][b][q];
    __syncthreads();
    // S += A' A'^T (the upper-right quadrant is the transpose of the lower-left one, bit for bit: its wave skips the products)
    if (!(wm == 0 && wn == 1)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        double fa[2][4], fb[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const double* src = sPA + (wm * 32 + a * 16 + r) * SF_LD + ks * 16 + 4 * g;
          const d2 lo = *reinterpret_cast<const d2*>(src), hi = *reinterpret_cast<const d2*>(src + 2);
          fa[a][0] = lo.x; fa[a][1] = lo.y; fa[a][2] = hi.x; fa[a][3] = hi.y;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const double* src = sPA + (wn * 32 + b * 16 + r) * SF_LD + ks * 16 + 4 * g;
          const d2 lo = *reinterpret_cast<const d2*>(src), hi = *reinterpret_cast<const d2*>(src + 2);
          fb[b][0] = lo.x; fb[b][1] = lo.y; fb[b][2] = hi.x; fb[b][3] = hi.y;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) accS[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a][j], fb[b][j], accS[a][b], 0, 0, 0);
      }
    }
    // u += A' y: this lane's row against its wave's 16 columns
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) uacc = __builtin_fma(sPA[lane * SF_LD + wave * 16 + jj], sY[wave * 16 + jj], uacc);
  }
  // slab of this chunk: S quadrants (0,0), (1,0), (1,1) row-major 64 x 64 (the (0,1) quadrant is never read), then u
  double* slab = A + p.oSlab + (int64_t)chunk * NB * NB;
  if (!(wm == 0 && wn == 1)) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) slab[(wm * 32 + a * 16 + g + 4 * q) * NB + wn * 32 + b * 16 + r] = accS[a][b][q];
  }
  sU[wave][lane] = uacc;
  __syncthreads();
  if (tid < NB) A[p.oU + (int64_t)chunk * NB + tid] = ((sU[0][tid] + sU[1][tid]) + sU[2][tid]) + sU[3][tid];
}


hipError_t SF_CAT(sf_launch_pass1_kid, SF_KID)(hipStream_t st, int form, const SfParams& p, int cells) {
  const dim3 grid(p.nchunks, cells), block(256);
  const bool one = p.d <= SF_DK;
  if (form) {
    if (one) hipLaunchKernelGGL((sf_pass1_kernel<SF_KID, 1, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((sf_pass1_kernel<SF_KID, 1, 0>), grid, block, 0, st, p);
  } else {
    if (one) hipLaunchKernelGGL((sf_pass1_kernel<SF_KID, 0, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((sf_pass1_kernel<SF_KID, 0, 0>), grid, block, 0, st, p);
  }
  return hipGetLastError();
}

}  // namespace gprx
