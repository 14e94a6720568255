// Fused sparse evaluation (sgpr_fused.h), launch 2, for ONE kernel id: compiled with -DSF_KID=k (gpras_amd/_build.py).
#include "sgpr_fused_dev.h"

#ifndef SF_KID
#error "compile with -DSF_KID=0..4"
#endif
#define SF_CAT2(a, b) a##b
#define SF_CAT(a, b) SF_CAT2(a, b)

namespace gprx {

// ---- launch 2: per chunk of 256 columns, S_c = A' A'^T and u_c = A' y with A' = L^-1 Kuf ---------------------------------
// 512 threads = 8 waves, two per SIMD: one wave alone issues fp64 vector instructions at half the SIMD's rate (measured: the four-wave
// first version spent 3.8 k clocks on the distances of a tile and 3.4 k on its exponentials, twice what the instruction count asks for).
// Row-lane layout: lane = inducing point, wave w = columns 8 w .. 8 w + 7 of the tile.  MFMA layout: wave w owns row block wm(w) (16 rows)
// and the two column blocks 2 wn, 2 wn + 1.  Both products use the structure: L^-1 is lower triangular (row block wm needs k blocks
// 0 .. wm only: 160 of 256 MFMAs; the row blocks are dealt so that the two waves of a SIMD get 0 + 3 or 1 + 2), and S is symmetric (the
// ten 16 x 16 tiles on or below the diagonal: waves 0 .. 7 take tiles 0 .. 7, waves 0 and 1 also 8 and 9).
// NP = 6 / 8: d <= 12 / 16, Z is staged once per workgroup and the distance loops run over NP pairs of dimensions unconditionally;
// NP == 0: any d <= 64, restaged per tile and chunk of 16 dimensions.
constexpr int SF_NT = 512;
constexpr int SF_NC = 8;  // columns per lane in the row-lane layout
__device__ __forceinline__ int sf_row_block(int wave) { return (wave < 4) ? (wave & 1) : 3 - (wave & 1); }  // 0 1 0 1 3 2 3 2
// the ten lower tiles of S in row-major order: tile t -> (row block, column block)
__device__ __forceinline__ void sf_lower_tile(int t, int& rb, int& cb) {
  rb = t < 1 ? 0 : (t < 3 ? 1 : (t < 6 ? 2 : 3));
  cb = t - rb * (rb + 1) / 2;
}

template <int KID, int FORM, int NP>
__global__ __launch_bounds__(SF_NT) void sf_pass1_kernel(SfParams p) {
  __shared__ __attribute__((aligned(16))) double sPA[NB * SF_LD];  // the tile of Kuf, then A' in its place
  __shared__ __attribute__((aligned(16))) double sZ[NB * SF_DKP];
  __shared__ __attribute__((aligned(16))) double sXc[NB * SF_DKP];
  __shared__ __attribute__((aligned(16))) double sTab[64];
  __shared__ double sY[NB];
  __shared__ double sU[8][NB];
  const int cell = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = sf_row_block(wave), wn = (wave >> 1) & 1, g = lane >> 4, r = lane & 15;
  if (p.active != nullptr && p.active[cell] == 0) return;  // (uniform over the workgroup)
  const double* par = p.cpar + (int64_t)cell * CELL_PAR;
  const double* ls = par + CELL_PAR_LS;
  const double variance = par[0];
  const int unit = (int)par[2];
  double* A = p.arena + (int64_t)cell * p.ss;
  const double* zp = A + p.oZ;
  const double* yp = p.Y + (int64_t)unit * p.np;
  SF_STAMP(p, 32, 0)
  exp_tab_fill(sTab);
  const int ntiles = min(SF_TILES, (p.np - chunk * SF_CHUNK) / NB);
  // the first tile's points are requested before anything else
  double raw[NB * SF_DK / SF_NT];
  sf_stage_fetch<SF_NT>(p.X, chunk * SF_CHUNK, p.n, p.d, 0, raw, tid);
  // L^-1 as MFMA A-operand fragments: rows 16 wm + r, k = 16 ks + 4 g + j (k blocks above the row block are zero: never used)
  double fl[4][4];
  {
    const double* Li = A + p.oLinv + (wm * 16 + r) * NB + 4 * g;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const d2 lo = *reinterpret_cast<const d2*>(Li + ks * 16), hi = *reinterpret_cast<const d2*>(Li + ks * 16 + 2);
      fl[ks][0] = lo.x; fl[ks][1] = lo.y; fl[ks][2] = hi.x; fl[ks][3] = hi.y;
    }
  }
  // this wave's tiles of S
  int rb0, cb0, rb1 = 0, cb1 = 0;
  sf_lower_tile(wave, rb0, cb0);
  const bool two = wave < 2;
  if (two) sf_lower_tile(8 + wave, rb1, cb1);
  d4 accS0 = d4{0.0, 0.0, 0.0, 0.0}, accS1 = d4{0.0, 0.0, 0.0, 0.0};
  double uacc = 0.0;
  if constexpr (NP > 0) sf_stage<FORM, SF_NT>(zp, 0, p.m, p.d, 0, ls, sZ, tid);  // (published by the first tile's staging barrier)
  SF_STAMP(p, 32, 1)
  for (int t = 0; t < ntiles; ++t) {
    const int j0 = chunk * SF_CHUNK + t * NB;
    double r2[SF_NC], nb[SF_NC], na = 0.0;
#pragma unroll
    for (int jj = 0; jj < SF_NC; ++jj) r2[jj] = nb[jj] = 0.0;
    for (int k0 = 0; k0 < p.d; k0 += SF_DK) {
      __syncthreads();  // the previous tile's (chunk's) readers of sXc / sY / sPA are done
      if constexpr (NP == 0) sf_stage<FORM, SF_NT>(zp, 0, p.m, p.d, k0, ls, sZ, tid);
      if (k0 == 0) {
        sf_stage_put<FORM, SF_NT>(raw, j0, p.n, p.d, 0, ls, sXc, tid);
        if (tid < NB) sY[tid] = yp[j0 + tid];  // (zero beyond n: Y is padded)
      } else {
        sf_stage<FORM, SF_NT>(p.X, j0, p.n, p.d, k0, ls, sXc, tid);
      }
      __syncthreads();
      if (k0 == 0 && t + 1 < ntiles) sf_stage_fetch<SF_NT>(p.X, j0 + NB, p.n, p.d, 0, raw, tid);  // the next tile's points travel from here on
      if (t == 0 && k0 == 0) { SF_STAMP(p, 32, 2) }
      sf_r2_chunk<FORM, SF_NC, NP>(sZ, sXc, lane, wave, min(SF_DK, p.d - k0), r2, na, nb);
    }
    if (t == 0) { SF_STAMP(p, 32, 3) }
#pragma unroll
    for (int jj = 0; jj < SF_NC; ++jj) {
      const int col = wave * SF_NC + jj;
      double rr = r2[jj];
      if constexpr (FORM != 0) rr = expanded_r2(na, nb[jj], rr);
      const double pv = variance * corr_g<KID>(rr, sTab);  // (evaluated for every lane, selected afterwards: no branch around the table read)
      sPA[lane * SF_LD + col] = (lane < p.m && j0 + col < p.n) ? pv : 0.0;
    }
    __syncthreads();
    if (t == 0) { SF_STAMP(p, 32, 4) }
    // A' = L^-1 P: row block wm against the k blocks 0 .. wm
    d4 acc[2];
    acc[0] = acc[1] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks <= wm) {
        double fb[2][4];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[b][j] = sPA[(ks * 16 + 4 * g + j) * SF_LD + wn * 32 + b * 16 + r];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fl[ks][j], fb[b][j], acc[b], 0, 0, 0);
      }
    }
    if (t == 0) { SF_STAMP(p, 32, 5) }
    __syncthreads();  // every wave has read P
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) sPA[(wm * 16 + g + 4 * q) * SF_LD + wn * 32 + b * 16 + r] = acc[b][q];
    __syncthreads();
    if (t == 0) { SF_STAMP(p, 32, 6) }
    // S += A' A'^T on the lower tiles
    {
      const double* pa0 = sPA + (rb0 * 16 + r) * SF_LD + 4 * g;
      const double* pb0 = sPA + (cb0 * 16 + r) * SF_LD + 4 * g;
      const double* pa1 = sPA + (rb1 * 16 + r) * SF_LD + 4 * g;
      const double* pb1 = sPA + (cb1 * 16 + r) * SF_LD + 4 * g;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const d2 alo = *reinterpret_cast<const d2*>(pa0 + ks * 16), ahi = *reinterpret_cast<const d2*>(pa0 + ks * 16 + 2);
        const d2 blo = *reinterpret_cast<const d2*>(pb0 + ks * 16), bhi = *reinterpret_cast<const d2*>(pb0 + ks * 16 + 2);
        accS0 = __builtin_amdgcn_mfma_f64_16x16x4f64(alo.x, blo.x, accS0, 0, 0, 0);
        accS0 = __builtin_amdgcn_mfma_f64_16x16x4f64(alo.y, blo.y, accS0, 0, 0, 0);
        accS0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ahi.x, bhi.x, accS0, 0, 0, 0);
        accS0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ahi.y, bhi.y, accS0, 0, 0, 0);
        if (two) {
          const d2 clo = *reinterpret_cast<const d2*>(pa1 + ks * 16), chi = *reinterpret_cast<const d2*>(pa1 + ks * 16 + 2);
          const d2 dlo = *reinterpret_cast<const d2*>(pb1 + ks * 16), dhi = *reinterpret_cast<const d2*>(pb1 + ks * 16 + 2);
          accS1 = __builtin_amdgcn_mfma_f64_16x16x4f64(clo.x, dlo.x, accS1, 0, 0, 0);
          accS1 = __builtin_amdgcn_mfma_f64_16x16x4f64(clo.y, dlo.y, accS1, 0, 0, 0);
          accS1 = __builtin_amdgcn_mfma_f64_16x16x4f64(chi.x, dhi.x, accS1, 0, 0, 0);
          accS1 = __builtin_amdgcn_mfma_f64_16x16x4f64(chi.y, dhi.y, accS1, 0, 0, 0);
        }
      }
    }
    // u += A' y: this lane's row against its wave's 8 columns
#pragma unroll
    for (int jj = 0; jj < SF_NC; ++jj) uacc = __builtin_fma(sPA[lane * SF_LD + wave * SF_NC + jj], sY[wave * SF_NC + jj], uacc);
    if (t == 0) { SF_STAMP(p, 32, 7) }
  }
  SF_STAMP(p, 32, 8)
  // slab of this chunk: the ten lower 16 x 16 tiles of S at their places in a row-major 64 x 64 block (the rest is never read), then u
  double* slab = A + p.oSlab + (int64_t)chunk * NB * NB;
#pragma unroll
  for (int q = 0; q < 4; ++q) slab[(rb0 * 16 + g + 4 * q) * NB + cb0 * 16 + r] = accS0[q];
  if (two) {
#pragma unroll
    for (int q = 0; q < 4; ++q) slab[(rb1 * 16 + g + 4 * q) * NB + cb1 * 16 + r] = accS1[q];
  }
  sU[wave][lane] = uacc;
  __syncthreads();
  if (tid < NB)
    A[p.oU + (int64_t)chunk * NB + tid] = (((sU[0][tid] + sU[1][tid]) + (sU[2][tid] + sU[3][tid])) + ((sU[4][tid] + sU[5][tid]) + (sU[6][tid] + sU[7][tid])));
  SF_STAMP(p, 32, 9)
}

hipError_t SF_CAT(sf_launch_pass1_kid, SF_KID)(hipStream_t st, int form, const SfParams& p, int cells) {
  const dim3 grid(p.nchunks, cells), block(SF_NT);
  const int np = p.d <= 12 ? 6 : (p.d <= SF_DK ? 8 : 0);
#define SF_P1(F_)                                                                              \
  if (np == 6) hipLaunchKernelGGL((sf_pass1_kernel<SF_KID, F_, 6>), grid, block, 0, st, p);    \
  else if (np == 8) hipLaunchKernelGGL((sf_pass1_kernel<SF_KID, F_, 8>), grid, block, 0, st, p); \
  else hipLaunchKernelGGL((sf_pass1_kernel<SF_KID, F_, 0>), grid, block, 0, st, p);
  if (form) {
    SF_P1(1)
  } else {
    SF_P1(0)
  }
#undef SF_P1
  return hipGetLastError();
}

}  // namespace gprx
