// Fused sparse evaluation (sgpr_fused.h), launch 4, for ONE kernel id: compiled with -DSF_KID=k (gpras_amd/_build.py).
#include "sgpr_fused_dev.h"

#ifndef SF_KID
#error "compile with -DSF_KID=0..4"
#endif
#define SF_CAT2(a, b) a##b
#define SF_CAT(a, b) SF_CAT2(a, b)

namespace gprx {

// ---- launch 4: the contractions -------------------------------------------------------------------------------------------
// Workgroup (chunk, cell): the chunk's 256 columns of Kuf with weights G_P, then ITS SLICE of the Kuu part -- columns
// [chunk qw, (chunk + 1) qw) of Kuu, qw = ceil(64 / nchunks), with weights G_Q: a workgroup of its own for Kuu was a seventeenth
// workgroup per cell, i.e. a second round of the launch at 16 cells (73 against 54 us), and appended whole to one workgroup it made
// that one a fifth longer than the rest (42 against 34 us); sliced it costs every workgroup ~1 us.  The slicing depends on the
// problem's shape only, never on the batch: same bits alone and in a batch.
// Partial block of a workgroup (p2w doubles): [0] sum w g, [1] sum w v h r2 (ISO), [2] sum (y - P^T m)^2, [4 + k] sum w v h ds_k^2,
// [SF_P2_HEAD + i d + k] sum_j w v h ds_k  -- all UNSCALED by the lengthscales, Kuf and Kuu contributions added; the Kuu terms enter
// dZ doubled (G_Q is symmetric: both index positions of z_i contribute).
// 512 threads = 8 waves (two per SIMD, see sf_pass1.hip).  Row-lane layout: lane = inducing point, wave w = columns 8 w .. 8 w + 7;
// MFMA layout: wave w owns row block w >> 1 (16 rows) and the column blocks 2 (w & 1), 2 (w & 1) + 1 of W P.
// Registers: what crosses the MFMA product per element is ONE value (v h; isotropic: also r2) -- sum w g is formed in the MFMA layout
// as sum (W P) o P + sum_j y_j (P^T m)_j, so g dies with the store of P.
constexpr int SF_NT = 512;
constexpr int SF_NC = 8;  // columns per lane in the row-lane layout

// The end of a pass-2 workgroup: its slice of the Kuu part (weights G_Q, both points inducing
// points: their staged coordinates are in sZ), then the partial block.  sEx: LDS scratch of at least 4 * NB * SF_DK doubles that nobody
// reads any more; sRed: [8][4].
template <int KID, int FORM, int ISO, int NP>
__device__ __forceinline__ void sf_pass2_tail(const SfParams& p, double* __restrict__ A, const double* __restrict__ zp, const double* __restrict__ ls,
                                              double variance, double inv_s, int chunk, int tid, double* __restrict__ sZ, double* __restrict__ sEx,
                                              double (*__restrict__ sRed)[4], double sg, double siso, double resid,
                                              double (&dz)[NP > 0 ? 2 * NP : 4 * SF_DK], double (&lsk)[ISO ? 1 : (NP > 0 ? 2 * NP : 4 * SF_DK)]) {
  constexpr int NKC = NP > 0 ? 1 : 0;
  constexpr int DZN = NP > 0 ? 2 * NP : 4 * SF_DK;
  const int lane = tid & 63, wave = tid >> 6;
  // ---- this workgroup's slice of the Kuu part: weights G_Q, both points inducing points (their staged coordinates are in sZ) ----
  double sgq = 0.0;
  {
    const int qw = (NB + p.nchunks - 1) / p.nchunks;
    const int qc0 = chunk * qw, qc1 = min(qc0 + qw, NB);
    if (qc0 < NB) {  // (uniform over the workgroup)
      const int cbase = qc0 + wave * SF_NC;          // this wave's first column of the slice
      const int base = min(cbase, NB - SF_NC);       // (clamped for addressing: slots outside the slice are masked)
      double r2[SF_NC], nb[SF_NC], na = 0.0;
#pragma unroll
      for (int jj = 0; jj < SF_NC; ++jj) r2[jj] = nb[jj] = 0.0;
      if constexpr (NKC == 1) {
        sf_r2_chunk<FORM, SF_NC, 0>(sZ, sZ + base * SF_DKP, lane, 0, p.d, r2, na, nb);
      } else {
        for (int k0 = 0; k0 < p.d; k0 += SF_DK) {
          __syncthreads();
          sf_stage<FORM, SF_NT>(zp, 0, p.m, p.d, k0, ls, sZ, tid);
          __syncthreads();
          sf_r2_chunk<FORM, SF_NC, 0>(sZ, sZ + base * SF_DKP, lane, 0, min(SF_DK, p.d - k0), r2, na, nb);
        }
      }
      double whq[SF_NC];
#pragma unroll
      for (int jj = 0; jj < SF_NC; ++jj) {
        const int col = base + jj;
        const bool live = col >= cbase && col < qc1 && col < p.m && lane < p.m;
        double rr = r2[jj];
        if constexpr (FORM != 0) rr = expanded_r2(na, nb[jj], rr);
        double gv, hv;
        corr_gh<KID>(rr, gv, hv);
        const double wq = live ? A[p.oGQ + lane * NB + min(col, NB - 1)] : 0.0;
        sgq = __builtin_fma(wq, gv, sgq);
        whq[jj] = live ? wq * variance * hv : 0.0;
        if constexpr (ISO != 0) siso = __builtin_fma(whq[jj], rr, siso);
      }
      for (int k0 = 0; k0 < p.d; k0 += SF_DK) {
        if constexpr (NKC != 1) {
          if (p.d > SF_DK) {
            __syncthreads();
            sf_stage<FORM, SF_NT>(zp, 0, p.m, p.d, k0, ls, sZ, tid);
            __syncthreads();
          }
        }
        const int dk = min(SF_DK, p.d - k0);
#pragma unroll
        for (int kcs = 0; kcs < (NKC == 1 ? 1 : 4); ++kcs) {
          if (kcs == (NKC == 1 ? 0 : k0 / SF_DK)) {
#pragma unroll
            for (int kk = 0; kk < (NP > 0 ? 2 * NP : SF_DK); kk += 2) {
              if (kk < dk) {
                d2 zv, xv[SF_NC];
                sf_load_pair<SF_NC>(sZ, sZ + base * SF_DKP, lane, 0, kk, zv, xv);
#pragma unroll
                for (int jj = 0; jj < SF_NC; ++jj) {
                  const double d0 = zv.x - xv[jj].x, d1 = zv.y - xv[jj].y;
                  const double t0 = whq[jj] * d0, t1 = whq[jj] * d1;
                  // (G_Q is symmetric: z_i sits at both index positions -> the dZ terms count twice; the lengthscale sums run over all pairs already)
                  dz[kcs * SF_DK + kk] = __builtin_fma(2.0, t0, dz[kcs * SF_DK + kk]);
                  dz[kcs * SF_DK + kk + 1] = __builtin_fma(2.0, t1, dz[kcs * SF_DK + kk + 1]);
                  if constexpr (ISO == 0) {
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
  }
  // ---- this workgroup's partial block ----
  double* out = A + p.oP2 + (int64_t)chunk * p.p2w;
  {
    // (Kuf workgroups: sg so far is sum (W P) o P + sum y (P^T m), i.e. s v times sum G_P g)
    const double a = wave_sum_dpp(sg), b = wave_sum_dpp(siso), c = wave_sum_dpp(resid), dq = wave_sum_dpp(sgq);
    if (lane == 0) {
      sRed[wave][0] = a;
      sRed[wave][1] = b;
      sRed[wave][2] = c;
      sRed[wave][3] = dq;
    }
  }
  __syncthreads();
  if (tid < 3) {
    double v = ((sRed[0][tid] + sRed[1][tid]) + (sRed[2][tid] + sRed[3][tid])) + ((sRed[4][tid] + sRed[5][tid]) + (sRed[6][tid] + sRed[7][tid]));
    if (tid == 0) {
      const double q = ((sRed[0][3] + sRed[1][3]) + (sRed[2][3] + sRed[3][3])) + ((sRed[4][3] + sRed[5][3]) + (sRed[6][3] + sRed[7][3]));
      v = v * inv_s / variance + q;  // sum G_P g (the accumulated sum is s v times it) + this slice's sum G_Q g
    }
    out[tid] = v;
  }
  // dZ: the eight waves' sums of every (row, dimension), four waves at a time through the tile image
  for (int kc = 0; kc * SF_DK < p.d; ++kc) {
    double part[2] = {0.0, 0.0};  // this thread's two (row, dimension) entries: e = tid, tid + 512
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
      if ((wave >> 2) == half) {
#pragma unroll
        for (int kcs = 0; kcs < (NKC == 1 ? 1 : 4); ++kcs) {
          if (kcs == kc) {
#pragma unroll
            for (int kk = 0; kk < SF_DK; ++kk) sEx[((wave & 3) * NB + lane) * SF_DK + kk] = kcs * SF_DK + kk < DZN ? dz[(kcs * SF_DK + kk) % DZN] : 0.0;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = tid + SF_NT * u, i = e >> 4, kk = e & 15;
        const double v = ((sEx[(0 * NB + i) * SF_DK + kk] + sEx[(1 * NB + i) * SF_DK + kk]) + sEx[(2 * NB + i) * SF_DK + kk]) + sEx[(3 * NB + i) * SF_DK + kk];
        part[u] = half == 0 ? v : part[u] + v;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + SF_NT * u, i = e >> 4, kk = e & 15, k = kc * SF_DK + kk;
      if (k < p.d) out[SF_P2_HEAD + i * p.d + k] = part[u];
    }
    if constexpr (ISO == 0) {
      // dK/dl_k: sum over the 64 rows (lanes) and the 8 waves
      __syncthreads();
#pragma unroll
      for (int kcs = 0; kcs < (NKC == 1 ? 1 : 4); ++kcs) {
        if (kcs == kc) {
#pragma unroll
          for (int kk = 0; kk < SF_DK; ++kk) {
            if (kcs * SF_DK + kk < DZN) {
              const double a = wave_sum_dpp(lsk[(kcs * SF_DK + kk) % DZN]);
              if (lane == 0) sEx[wave * SF_DK + kk] = a;
            }
          }
        }
      }
      __syncthreads();
      if (tid < SF_DK && kc * SF_DK + tid < p.d) {
        double v = 0.0;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) v += sEx[w8 * SF_DK + tid];
        out[4 + kc * SF_DK + tid] = v;
      }
    }
  }
}


template <int KID, int FORM, int ISO, int NP>
__global__ __launch_bounds__(SF_NT) void sf_pass2_kernel(SfParams p) {
  constexpr int NKC = NP > 0 ? 1 : 0;
  constexpr int DZN = NP > 0 ? 2 * NP : 4 * SF_DK;  // dimensions this lane accumulates dK/dZ for
  __shared__ __attribute__((aligned(16))) double sPA[NB * SF_LD];  // the tile of Kuf, then W P in its place; at the end the dZ exchange
  __shared__ __attribute__((aligned(16))) double sZ[NB * SF_DKP];
  __shared__ __attribute__((aligned(16))) double sXc[NB * SF_DKP];
  __shared__ double sY[NB], sM[NB];
  __shared__ double sQp[8][NB];
  __shared__ double sRed[8][4];
  static_assert(4 * NB * SF_DK <= NB * SF_LD, "the dZ exchange of four waves and one chunk of dimensions fits into the tile image");
  const int cell = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, r = lane & 15;
  if (p.active != nullptr && p.active[cell] == 0) return;  // (uniform over the workgroup)
  const double* par = p.cpar + (int64_t)cell * CELL_PAR;
  const double* ls = par + CELL_PAR_LS;
  const double variance = par[0], inv_s = par[3];
  const int unit = (int)par[2];
  double* A = p.arena + (int64_t)cell * p.ss;
  const double* zp = A + p.oZ;
  const double* yp = p.Y + (int64_t)unit * p.np;
  SF_STAMP(p, 96, 0)
  constexpr bool isq = false;  // (the tile loop below only sees columns of Kuf; its Kuu branches are compiled out)
  const double* colpts = p.X;
  const int ncolpts = p.n;
  const int ntiles = min(SF_TILES, (p.np - chunk * SF_CHUNK) / NB);
  double raw[NB * SF_DK / SF_NT];
  sf_stage_fetch<SF_NT>(colpts, isq ? 0 : chunk * SF_CHUNK, ncolpts, p.d, 0, raw, tid);  // the first tile's points, before anything else
  double fw[4][4];  // W as MFMA A-operand fragments: rows 16 wm + r, k = 16 ks + 4 g + j (Kuf workgroups)
  if (!isq) {
    const double* W = A + p.oW + (wm * 16 + r) * NB + 4 * g;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const d2 lo = *reinterpret_cast<const d2*>(W + ks * 16), hi = *reinterpret_cast<const d2*>(W + ks * 16 + 2);
      fw[ks][0] = lo.x; fw[ks][1] = lo.y; fw[ks][2] = hi.x; fw[ks][3] = hi.y;
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
  if constexpr (NKC == 1) sf_stage<FORM, SF_NT>(zp, 0, p.m, p.d, 0, ls, sZ, tid);  // (published by the first tile's staging barrier)
  SF_STAMP(p, 96, 1)
  for (int t = 0; t < ntiles; ++t) {
    const int j0 = isq ? 0 : chunk * SF_CHUNK + t * NB;
    double r2[SF_NC], nb[SF_NC], na = 0.0;
#pragma unroll
    for (int jj = 0; jj < SF_NC; ++jj) r2[jj] = nb[jj] = 0.0;
    for (int k0 = 0; k0 < p.d; k0 += SF_DK) {
      __syncthreads();
      if constexpr (NKC != 1) sf_stage<FORM, SF_NT>(zp, 0, p.m, p.d, k0, ls, sZ, tid);
      if (k0 == 0) {
        sf_stage_put<FORM, SF_NT>(raw, j0, ncolpts, p.d, 0, ls, sXc, tid);
        if (tid < NB) sY[tid] = isq ? 0.0 : yp[j0 + tid];
      } else {
        sf_stage<FORM, SF_NT>(colpts, j0, ncolpts, p.d, k0, ls, sXc, tid);
      }
      __syncthreads();
      if (k0 == 0 && t + 1 < ntiles) sf_stage_fetch<SF_NT>(colpts, j0 + NB, ncolpts, p.d, 0, raw, tid);  // the next tile's points travel from here on
      if (t == 0 && k0 == 0) { SF_STAMP(p, 96, 2) }
      sf_r2_chunk<FORM, SF_NC, NP>(sZ, sXc, lane, wave, min(SF_DK, p.d - k0), r2, na, nb);
    }
    if (t == 0) { SF_STAMP(p, 96, 3) }
    // per element: g (Kuf workgroups: into the P tile, then dead), vh = v h, and for the isotropic lengthscale r2 stays
    double wh[SF_NC];  // v h now, w v h once the weights are known
#pragma unroll
    for (int jj = 0; jj < SF_NC; ++jj) {
      const int col = wave * SF_NC + jj;
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
    if (t == 0) { SF_STAMP(p, 96, 4) }
    if (!isq) {
      __syncthreads();
      // W P on MFMA; P^T m by columns
      d4 acc[2];
      acc[0] = acc[1] = d4{0.0, 0.0, 0.0, 0.0};
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
          for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fw[ks][j], fb[b][j], acc[b], 0, 0, 0);
      }
      {
        const int col = tid & 63, oct = tid >> 6;
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) sum = __builtin_fma(sPA[(8 * oct + i) * SF_LD + col], sM[8 * oct + i], sum);
        sQp[oct][col] = sum;
      }
      // sum (W P) o P in the MFMA layout (P is zero wherever an element is masked)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) sg = __builtin_fma(acc[b][q], sPA[(wm * 16 + g + 4 * q) * SF_LD + wn * 32 + b * 16 + r], sg);
      if (t == 0) { SF_STAMP(p, 96, 5) }
      __syncthreads();  // every wave has read P
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) sPA[(wm * 16 + g + 4 * q) * SF_LD + wn * 32 + b * 16 + r] = acc[b][q];
      __syncthreads();
      if (tid < NB) {
        const double qv = ((sQp[0][tid] + sQp[1][tid]) + (sQp[2][tid] + sQp[3][tid])) + ((sQp[4][tid] + sQp[5][tid]) + (sQp[6][tid] + sQp[7][tid]));
        const bool live = j0 + tid < p.n;
        const double rv = live ? sY[tid] - qv : 0.0;
        resid = __builtin_fma(rv, rv, resid);
        if (live) sg = __builtin_fma(sY[tid], qv, sg);  // the rank-one part of sum G_P o P: sum_j y_j (P^T m)_j
      }
      // w v h with w = G_P = (W P + m y^T) / s
#pragma unroll
      for (int jj = 0; jj < SF_NC; ++jj) {
        const int col = wave * SF_NC + jj;
        double wv = inv_s * sPA[lane * SF_LD + col];
        wv = __builtin_fma(inv_s * mrow, sY[col], wv);
        wh[jj] = (lane < p.m && j0 + col < p.n) ? wv * wh[jj] : 0.0;
      }
    }
    if constexpr (ISO != 0) {
#pragma unroll
      for (int jj = 0; jj < SF_NC; ++jj) siso = __builtin_fma(wh[jj], r2[jj], siso);
    }
    if (t == 0) { SF_STAMP(p, 96, 6) }
    // per-dimension sums: dK/dZ (in this lane: its own inducing point) and, anisotropic, dK/dl_k
    for (int k0 = 0; k0 < p.d; k0 += SF_DK) {
      if constexpr (NKC != 1) {
        if (p.d > SF_DK) {  // more than one chunk of dimensions: the staged coordinates of this chunk again
          __syncthreads();
          sf_stage<FORM, SF_NT>(zp, 0, p.m, p.d, k0, ls, sZ, tid);
          sf_stage<FORM, SF_NT>(colpts, j0, ncolpts, p.d, k0, ls, sXc, tid);
          __syncthreads();
        }
      }
      const int dk = min(SF_DK, p.d - k0);
      const int kc = NKC == 1 ? 0 : k0 / SF_DK;
      // one pair of dimensions: dK/dZ sums (and, anisotropic, dK/dl_k) of this lane against its wave's columns
      auto grad_pair = [&](auto kbase_c, const d2& zv, const d2 (&xv)[SF_NC]) {
        constexpr int KB = decltype(kbase_c)::value;
#pragma unroll
        for (int jj = 0; jj < SF_NC; ++jj) {
          const double d0 = zv.x - xv[jj].x, d1 = zv.y - xv[jj].y;
          if constexpr (ISO != 0) {
            dz[KB] = __builtin_fma(wh[jj], d0, dz[KB]);
            dz[KB + 1] = __builtin_fma(wh[jj], d1, dz[KB + 1]);
          } else {
            const double t0 = wh[jj] * d0, t1 = wh[jj] * d1;
            dz[KB] += t0;
            dz[KB + 1] += t1;
            lsk[KB] = __builtin_fma(t0, d0, lsk[KB]);
            lsk[KB + 1] = __builtin_fma(t1, d1, lsk[KB + 1]);
          }
        }
      };
      if constexpr (NP > 0) {
        // software-pipelined as sf_r2_chunk: the reads of pair p + 1 under the arithmetic of pair p
        d2 za, zb, xa[SF_NC], xb[SF_NC];
        sf_load_pair<SF_NC>(sZ, sXc, lane, wave, 0, za, xa);
        auto step2 = [&](auto pp_c) {
          constexpr int PP = decltype(pp_c)::value;
          if constexpr (PP < NP) {
            if constexpr (PP + 1 < NP) sf_load_pair<SF_NC>(sZ, sXc, lane, wave, 2 * (PP + 1), zb, xb);
            __builtin_amdgcn_sched_barrier(0);
            grad_pair(std::integral_constant<int, 2 * PP>{}, za, xa);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PP + 2 < NP) sf_load_pair<SF_NC>(sZ, sXc, lane, wave, 2 * (PP + 2), za, xa);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PP + 1 < NP) grad_pair(std::integral_constant<int, 2 * (PP + 1)>{}, zb, xb);
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        step2(std::integral_constant<int, 0>{});
        step2(std::integral_constant<int, 2>{});
        step2(std::integral_constant<int, 4>{});
        step2(std::integral_constant<int, 6>{});
      } else {
#pragma unroll
        for (int kcs = 0; kcs < 4; ++kcs) {  // (static accumulator indices: one guarded block is taken)
          if (kcs == kc) {
#pragma unroll
            for (int kk = 0; kk < SF_DK; kk += 2) {
              if (kk < dk) {
                d2 zv, xv[SF_NC];
                sf_load_pair<SF_NC>(sZ, sXc, lane, wave, kk, zv, xv);
                auto call = [&](auto kb) { grad_pair(kb, zv, xv); };
                // (kcs and kk are unrolled constants, but not constant expressions: dispatch on the 32 possible bases)
                const int kbv = kcs * SF_DK + kk;
#define SF_GP(K_) if (kbv == K_) call(std::integral_constant<int, K_>{});
                SF_GP(0) SF_GP(2) SF_GP(4) SF_GP(6) SF_GP(8) SF_GP(10) SF_GP(12) SF_GP(14) SF_GP(16) SF_GP(18) SF_GP(20) SF_GP(22) SF_GP(24) SF_GP(26)
                SF_GP(28) SF_GP(30) SF_GP(32) SF_GP(34) SF_GP(36) SF_GP(38) SF_GP(40) SF_GP(42) SF_GP(44) SF_GP(46) SF_GP(48) SF_GP(50) SF_GP(52)
                SF_GP(54) SF_GP(56) SF_GP(58) SF_GP(60) SF_GP(62)
#undef SF_GP
              }
            }
          }
        }
      }
    }
    if (t == 0) { SF_STAMP(p, 96, 7) }
  }
  sf_pass2_tail<KID, FORM, ISO, NP>(p, A, zp, ls, variance, inv_s, chunk, tid, sZ, sPA, sRed, sg, siso, resid, dz, lsk);
  SF_STAMP(p, 96, 9)
}

hipError_t SF_CAT(sf_launch_pass2_kid, SF_KID)(hipStream_t st, int form, int iso, const SfParams& p, int cells) {
  const dim3 grid(p.nchunks, cells), block(SF_NT);
  const int np = p.d <= 12 ? 6 : (p.d <= SF_DK ? 8 : 0);
#define SF_P2(F_, I_)                                                                                \
  if (np == 6) hipLaunchKernelGGL((sf_pass2_kernel<SF_KID, F_, I_, 6>), grid, block, 0, st, p);      \
  else if (np == 8) hipLaunchKernelGGL((sf_pass2_kernel<SF_KID, F_, I_, 8>), grid, block, 0, st, p); \
  else hipLaunchKernelGGL((sf_pass2_kernel<SF_KID, F_, I_, 0>), grid, block, 0, st, p);
  if (form) {
    SF_P2(1, 0)
  } else if (iso) {
    SF_P2(0, 1)
  } else {
    SF_P2(0, 0)
  }
#undef SF_P2
  return hipGetLastError();
}

}  // namespace gprx
