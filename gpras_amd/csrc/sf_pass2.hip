// Fused sparse evaluation (sgpr_fused.h), launch 4, for ONE kernel id: compiled with -DSF_KID=k (gpras_amd/_build.py).
#include "sgpr_fused_dev.h"

#ifndef SF_KID
#error "compile with -DSF_KID=0..4"
#endif
#define SF_CAT2(a, b) a##b
#define SF_CAT(a, b) SF_CAT2(a, b)

namespace gprx {

// ---- launch 4: the contractions -------------------------------------------------------------------------------------------
// Workgroup (chunk < nchunks, cell): the chunk's columns of Kuf.  Workgroup (nchunks, cell): Kuu with weights G_Q.
// Partial block of a workgroup (p2w doubles): [0] sum w g, [1] sum w v h r2 (ISO), [2] sum (y - P^T m)^2, [4 + k] sum w v h ds_k^2,
// [SF_P2_HEAD + i d + k] sum_j w v h ds_k  -- all UNSCALED by the lengthscales; the Kuu workgroup stores its dZ sums doubled
// (G_Q is symmetric: both index positions of z_i contribute).
// Registers: what crosses the MFMA product per element is ONE value (v h; isotropic: also r2) -- sum w g is formed in the MFMA layout
// as sum (W P) o P + sum_j y_j (P^T m)_j, so g dies with the store of P.
template <int KID, int FORM, int ISO, int NKC>
__global__ __launch_bounds__(256, NKC == 1 ? 2 : 1) void sf_pass2_kernel(SfParams p) {
  constexpr int DZN = NKC == 1 ? SF_DK : 4 * SF_DK;  // dimensions this lane accumulates dK/dZ for
  __shared__ __attribute__((aligned(16))) double sPA[NB * SF_LD];  // the tile of Kuf, then W P in its place; at the end the dZ exchange
  __shared__ __attribute__((aligned(16))) double sZ[NB * SF_DKP];
  __shared__ __attribute__((aligned(16))) double sXc[NB * SF_DKP];
  __shared__ double sY[NB], sM[NB];
  __shared__ double sQp[4][NB];
  __shared__ double sRed[4][4];
  static_assert(4 * NB * SF_DK <= NB * SF_LD, "the dZ exchange of one chunk of dimensions fits into the tile image");
  const int cell = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, r = lane & 15;
  const bool isq = chunk == p.nchunks;
  const double* par = p.cpar + (int64_t)cell * CELL_PAR;
  const double* ls = par + CELL_PAR_LS;
  const double variance = par[0], inv_s = par[3];
  const int unit = (int)par[2];
  double* A = p.arena + (int64_t)cell * p.ss;
  const double* zp = A + p.oZ;
  const double* yp = p.Y + (int64_t)unit * p.np;
  const double* colpts = isq ? zp : p.X;
  const int ncolpts = isq ? p.m : p.n;
  double fw[2][4][4];  // W as MFMA A-operand fragments (Kuf workgroups)
  if (!isq) {
    const double* W = A + p.oW;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const double* src = W + (wm * 32 + a * 16 + r) * NB + ks * 16 + 4 * g;
        const d2 lo = *reinterpret_cast<const d2*>(src), hi = *reinterpret_cast<const d2*>(src + 2);
        fw[a][ks][0] = lo.x; fw[a][ks][1] = lo.y; fw[a][ks][2] = hi.x; fw[a][ks][3] = hi.y;
      }
    if (tid < NB) sM[tid] = A[p.oM + tid];
  }
  const double mrow = isq ? 0.0 : A[p.oM + lane];
  double sg = 0.0, siso = 0.0, resid = 0.0;
  double dz[DZN], lsk[ISO ? 1 : DZN];
#pragma unroll
  for (int k = 0; k < DZN; ++k) dz[k] = 0.0;
#pragma unroll
  for (int k = 0; k < (ISO ? 1 : DZN); ++k) lsk[k] = 0.0;
  if constexpr (NKC == 1) sf_stage<FORM>(zp, 0, p.m, p.d, 0, ls, sZ, tid);  // (published by the first tile's staging barrier)
  const int ntiles = isq ? 1 : min(SF_TILES, (p.np - chunk * SF_CHUNK) / NB);
  for (int t = 0; t < ntiles; ++t) {
    const int j0 = isq ? 0 : chunk * SF_CHUNK + t * NB;
    double r2[16], nb[16], na = 0.0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) r2[jj] = nb[jj] = 0.0;
    for (int k0 = 0; k0 < p.d; k0 += SF_DK) {
      __syncthreads();
      if constexpr (NKC != 1) sf_stage<FORM>(zp, 0, p.m, p.d, k0, ls, sZ, tid);
      sf_stage<FORM>(colpts, j0, ncolpts, p.d, k0, ls, sXc, tid);
      if (k0 == 0 && tid < NB) sY[tid] = isq ? 0.0 : yp[j0 + tid];
      __syncthreads();
      sf_r2_chunk<FORM>(sZ, sXc, lane, wave, min(SF_DK, p.d - k0), r2, na, nb);
    }
    // per element: g (Kuf workgroups: into the P tile, then dead), vh = v h, and for the isotropic lengthscale r2 stays
    double wh[16];  // v h now, w v h once the weights are known
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int col = wave * 16 + jj;
      double rr = r2[jj];
      if constexpr (FORM != 0) rr = expanded_r2(na, nb[jj], rr);
      r2[jj] = rr;
      double gv, hv;
      corr_gh<KID>(rr, gv, hv);
      wh[jj] = variance * hv;
      if (!isq) {
        sPA[lane * SF_LD + col] = (lane < p.m && j0 + col < p.n) ? variance * gv : 0.0;
      } else {
        // sum G_Q g directly (no product in front of it)
        const double wq = (lane < p.m && col < p.m) ? A[p.oGQ + lane * NB + col] : 0.0;
        sg = __builtin_fma(wq, gv, sg);
        wh[jj] = wq * wh[jj];
      }
    }
    if (!isq) {
      __syncthreads();
      // W P on MFMA; P^T m by columns
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
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fw[a][ks][j], fb[b][j], acc[a][b], 0, 0, 0);
      }
      {
        const int col = tid & 63, qq = tid >> 6;
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum = __builtin_fma(sPA[(16 * qq + i) * SF_LD + col], sM[16 * qq + i], sum);
        sQp[qq][col] = sum;
      }
      // sum (W P) o P in the MFMA layout (P is zero wherever an element is masked)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) sg = __builtin_fma(acc[a][b][q], sPA[(wm * 32 + a * 16 + g + 4 * q) * SF_LD + wn * 32 + b * 16 + r], sg);
      __syncthreads();  // every wave has read P
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) sPA[(wm * 32 + a * 16 + g + 4 * q) * SF_LD + wn * 32 + b * 16 + r] = acc[a][b][q];
      __syncthreads();
      if (tid < NB) {
        const double qv = ((sQp[0][tid] + sQp[1][tid]) + sQp[2][tid]) + sQp[3][tid];
        const bool live = j0 + tid < p.n;
        const double rv = live ? sY[tid] - qv : 0.0;
        resid = __builtin_fma(rv, rv, resid);
        if (live) sg = __builtin_fma(sY[tid], qv, sg);  // the rank-one part of sum G_P o P: sum_j y_j (P^T m)_j
      }
      // w v h with w = G_P = (W P + m y^T) / s
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int col = wave * 16 + jj;
        double wv = inv_s * sPA[lane * SF_LD + col];
        wv = __builtin_fma(inv_s * mrow, sY[col], wv);
        wh[jj] = (lane < p.m && j0 + col < p.n) ? wv * wh[jj] : 0.0;
      }
    }
    if constexpr (ISO != 0) {
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) siso = __builtin_fma(wh[jj], r2[jj], siso);
    }
    // per-dimension sums: dK/dZ (in this lane: its own inducing point) and, anisotropic, dK/dl_k
    for (int k0 = 0; k0 < p.d; k0 += SF_DK) {
      if constexpr (NKC != 1) {
        if (p.d > SF_DK) {  // more than one chunk of dimensions: the staged coordinates of this chunk again
          __syncthreads();
          sf_stage<FORM>(zp, 0, p.m, p.d, k0, ls, sZ, tid);
          sf_stage<FORM>(colpts, j0, ncolpts, p.d, k0, ls, sXc, tid);
          __syncthreads();
        }
      }
      const int dk = min(SF_DK, p.d - k0);
      const int kc = NKC == 1 ? 0 : k0 / SF_DK;
#pragma unroll
      for (int kcs = 0; kcs < (NKC == 1 ? 1 : 4); ++kcs) {  // (static accumulator indices: one guarded block is taken)
        if (kcs == kc) {
#pragma unroll
        for (int kk = 0; kk < SF_DK; kk += 2) {
          if (kk < dk) {
          const d2 zv = *reinterpret_cast<const d2*>(sZ + lane * SF_DKP + kk);
#pragma unroll
          for (int jj = 0; jj < 16; ++jj) {
            const d2 xv = *reinterpret_cast<const d2*>(sXc + (wave * 16 + jj) * SF_DKP + kk);
            const double d0 = zv.x - xv.x, d1 = zv.y - xv.y;
            if constexpr (ISO != 0) {
              dz[kcs * SF_DK + kk] = __builtin_fma(wh[jj], d0, dz[kcs * SF_DK + kk]);
              dz[kcs * SF_DK + kk + 1] = __builtin_fma(wh[jj], d1, dz[kcs * SF_DK + kk + 1]);
            } else {
              const double t0 = wh[jj] * d0, t1 = wh[jj] * d1;
              dz[kcs * SF_DK + kk] += t0;
              dz[kcs * SF_DK + kk + 1] += t1;
              lsk[kcs * SF_DK + kk] = __builtin_fma(t0, d0, lsk[kcs * SF_DK + kk]);
              lsk[kcs * SF_DK + kk + 1] = __builtin_fma(t1, d1, lsk[kcs * SF_DK + kk + 1]);
            }
          }
          }
        }
        }
      }
    }
  }
  // ---- this workgroup's partial block ----
  double* out = A + p.oP2 + (int64_t)chunk * p.p2w;
  {
    // (Kuf workgroups: sg so far is sum (W P) o P + sum y (P^T m), i.e. s v times sum G_P g)
    const double a = wave_sum(sg), b = wave_sum(siso), c = wave_sum(resid);
    if (lane == 0) {
      sRed[wave][0] = a;
      sRed[wave][1] = b;
      sRed[wave][2] = c;
    }
  }
  __syncthreads();
  if (tid < 3) {
    double v = ((sRed[0][tid] + sRed[1][tid]) + sRed[2][tid]) + sRed[3][tid];
    if (tid == 0 && !isq) v = v * inv_s / variance;
    out[tid] = v;
  }
  double* sEx = sPA;  // [4 waves][64 rows][16]
  for (int kc = 0; kc * SF_DK < p.d; ++kc) {
    __syncthreads();
#pragma unroll
    for (int kcs = 0; kcs < (NKC == 1 ? 1 : 4); ++kcs) {
      if (kcs == kc) {
#pragma unroll
        for (int kk = 0; kk < SF_DK; ++kk) sEx[(wave * NB + lane) * SF_DK + kk] = dz[kcs * SF_DK + kk];
      }
    }
    __syncthreads();
    for (int e = tid; e < NB * SF_DK; e += 256) {
      const int i = e >> 4, kk = e & 15, k = kc * SF_DK + kk;
      if (k < p.d) {
        const double v = ((sEx[(0 * NB + i) * SF_DK + kk] + sEx[(1 * NB + i) * SF_DK + kk]) + sEx[(2 * NB + i) * SF_DK + kk]) + sEx[(3 * NB + i) * SF_DK + kk];
        out[SF_P2_HEAD + i * p.d + k] = isq ? 2.0 * v : v;
      }
    }
    if constexpr (ISO == 0) {
      // dK/dl_k: sum over the 64 rows (lanes) and the 4 waves
      __syncthreads();
#pragma unroll
      for (int kcs = 0; kcs < (NKC == 1 ? 1 : 4); ++kcs) {
        if (kcs == kc) {
#pragma unroll
          for (int kk = 0; kk < SF_DK; ++kk) {
            const double a = wave_sum(lsk[kcs * SF_DK + kk]);
            if (lane == 0) sEx[wave * SF_DK + kk] = a;
          }
        }
      }
      __syncthreads();
      if (tid < SF_DK && kc * SF_DK + tid < p.d)
        out[4 + kc * SF_DK + tid] = ((sEx[tid] + sEx[SF_DK + tid]) + sEx[2 * SF_DK + tid]) + sEx[3 * SF_DK + tid];
    }
  }
}

hipError_t SF_CAT(sf_launch_pass2_kid, SF_KID)(hipStream_t st, int form, int iso, const SfParams& p, int cells) {
  const dim3 grid(p.nchunks + 1, cells), block(256);
  const bool one = p.d <= SF_DK;
  if (form) {
    if (one) hipLaunchKernelGGL((sf_pass2_kernel<SF_KID, 1, 0, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((sf_pass2_kernel<SF_KID, 1, 0, 0>), grid, block, 0, st, p);
  } else if (iso) {
    if (one) hipLaunchKernelGGL((sf_pass2_kernel<SF_KID, 0, 1, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((sf_pass2_kernel<SF_KID, 0, 1, 0>), grid, block, 0, st, p);
  } else {
    if (one) hipLaunchKernelGGL((sf_pass2_kernel<SF_KID, 0, 0, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((sf_pass2_kernel<SF_KID, 0, 0, 0>), grid, block, 0, st, p);
  }
  return hipGetLastError();
}

}  // namespace gprx
