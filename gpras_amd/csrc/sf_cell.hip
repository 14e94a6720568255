// Fused sparse evaluation (sgpr_fused.h): the one-workgroup-per-cell launches -- prep (Kuu, L, L^-1), mid (B, LB, c, the M x M gradient
// algebra) and final (partials in chunk order -> the pinned result block) -- and the dispatch of the pass launchers by kernel id.
#include "sgpr_fused_dev.h"

namespace gprx {


// ---- launch 1: stage-in, Kuu, its factor and the factor's inverse ------------------------------------------------------------
// par_src: `cells` rows of CELL_PAR doubles (pinned host memory or device memory); z_src: (cells, m, d) inducing inputs or nullptr
// when the cell blocks already hold Z (device-resident optimiser).  The row is copied into the device table that the other launches
// read, the cell's result words are cleared.
template <int KID, int FORM>
__global__ __launch_bounds__(256) void sf_prep_kernel(SfParams p, const double* __restrict__ par_src, const double* __restrict__ z_src,
                                                      double* __restrict__ cpar_dst) {
  __shared__ __attribute__((aligned(16))) double sQ[NB * SF_LD];
  __shared__ __attribute__((aligned(16))) double sZ[NB * SF_DKP];
  __shared__ __attribute__((aligned(16))) double sIn[2 * NB * PSUB];
  __shared__ __attribute__((aligned(16))) double sXb[2 * NB * PSUB];
  __shared__ __attribute__((aligned(16))) double sTab[64];
  __shared__ double sPar[CELL_PAR];
  const int cell = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* A = p.arena + (int64_t)cell * p.ss;
  if (tid < CELL_PAR) {
    const double v = par_src[(int64_t)cell * CELL_PAR + tid];
    sPar[tid] = v;
    cpar_dst[(int64_t)cell * CELL_PAR + tid] = v;
  }
  if (tid < p.cellres_stride) p.cellres[(int64_t)cell * p.cellres_stride + tid] = 0.0;
  const double* zp = A + p.oZ;
  if (z_src) {
    zp = z_src + (int64_t)cell * p.m * p.d;
    for (int e = tid; e < p.m * p.d; e += 256) A[p.oZ + e] = zp[e];
  }
  exp_tab_fill(sTab);
  __syncthreads();
  const double variance = sPar[0];
  const double* ls = sPar + CELL_PAR_LS;
  double r2[16], nb[16], na = 0.0;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) r2[jj] = nb[jj] = 0.0;
  for (int k0 = 0; k0 < p.d; k0 += SF_DK) {
    if (k0 > 0) __syncthreads();
    sf_stage<FORM>(zp, 0, p.m, p.d, k0, ls, sZ, tid);
    __syncthreads();
    sf_r2_chunk<FORM>(sZ, sZ, lane, wave, min(SF_DK, p.d - k0), r2, na, nb);
  }
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const int col = wave * 16 + jj;
    double rr = r2[jj];
    if constexpr (FORM != 0) rr = expanded_r2(na, nb[jj], rr);
    double q;
    if (lane < p.m && col < p.m) {
      q = variance * corr_g<KID>(rr, sTab);
      if (lane == col) q += JITTER;
    } else {
      q = lane == col ? 1.0 : 0.0;  // identity padding
    }
    sQ[lane * SF_LD + col] = q;
  }
  __syncthreads();
  d4 acc[2][4];
  const int bad = sf_chain(sQ, SF_LD, sIn, sXb, acc, tid);
  if (bad != 0 && tid == 0) atomicCAS(reinterpret_cast<int*>(p.cellres + (int64_t)cell * p.cellres_stride + 2), 0, bad);
  // L straight from the accumulators (16 lanes = one 128-byte line); L^-1 = (acc[1])^T through LDS
  const int g = lane >> 4, r = lane & 15;
  __syncthreads();  // (every wave has finished reading sQ)
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * wave + g + 4 * q, col = 16 * kt + r;
      A[p.oL + row * NB + col] = acc[0][kt][q];
      sQ[col * SF_LD + row] = acc[1][kt][q];
    }
  __syncthreads();
  sf_image_out(sQ, SF_LD, A + p.oLinv, tid);
}

// ---- launch 3: one workgroup per cell ------------------------------------------------------------------------------------
// red (8 doubles per cell): [0] sum log diag LB, [1] |c|^2, [2] tr(A A^T) = tr(S) / s, [3] |LB^-1|_F^2
__global__ __launch_bounds__(256) void sf_mid_kernel(SfParams p) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sL = smem;                 // L^-1            (row stride SM_LD, the images mm64 works on)
  double* sB = sL + NB * SM_LD;      // LB^-1, then B
  double* sC = sB + NB * SM_LD;      // chain buffers, then R, then T2
  double* sBf = sC + NB * SM_LD;     // B (symmetric, both triangles)
  double* sb = sBf + NB * SM_LD;     // [64] right-hand side / c / m
  double* part = sb + NB;            // [4][64]
  double* sdiag = part + 4 * NB;     // [64]
  double* srow = sdiag + NB;         // [64]
  const int cell = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, r = lane & 15;
  const double* par = p.cpar + (int64_t)cell * CELL_PAR;
  const double inv_s = par[3];
  double* A = p.arena + (int64_t)cell * p.ss;
  double* red = A + p.oRed;
  // S = sum over the chunks, in chunk order, of the three stored quadrants; B = I + S / s
  {
    const double* slab = A + p.oSlab;
    // quadrant blocks of 32 x 32 = 1024 elements, three of them: 12 per thread, two neighbours per load
    double s[6][2];
#pragma unroll
    for (int e = 0; e < 6; ++e) s[e][0] = s[e][1] = 0.0;
    for (int c = 0; c < p.nchunks; ++c) {
      const double* sc = slab + (int64_t)c * NB * NB;
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const int q = tid + 256 * e;         // 1536 pairs: block = q / 512, inside: row = (q % 512) / 16, pair = q % 16
        const int blk = q >> 9, row = (q & 511) >> 4, pr = q & 15;
        const int i = (blk == 0 ? 0 : 32) + row, j = (blk == 2 ? 32 : 0) + 2 * pr;
        const d2 v = *reinterpret_cast<const d2*>(sc + i * NB + j);
        s[e][0] += v.x;
        s[e][1] += v.y;
      }
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int q = tid + 256 * e;
      const int blk = q >> 9, row = (q & 511) >> 4, pr = q & 15;
      const int i = (blk == 0 ? 0 : 32) + row, j = (blk == 2 ? 32 : 0) + 2 * pr;
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const double sv = s[e][c2];
        const int jj = j + c2;
        const double bv = __builtin_fma(sv, inv_s, i == jj ? 1.0 : 0.0);
        sBf[i * SM_LD + jj] = bv;
        if (blk == 1) sBf[jj * SM_LD + i] = bv;  // mirror of the lower-left quadrant
        if (i == jj) sdiag[i] = sv * inv_s;
      }
    }
    if (tid < NB) {
      double u = 0.0;
      for (int c = 0; c < p.nchunks; ++c) u += A[p.oU + (int64_t)c * NB + tid];
      sb[tid] = u * inv_s;
    }
  }
  load64(A + p.oLinv, sL, tid);
  __syncthreads();
  if (wave == 0) {  // tr(A A^T)
    const double a = wave_sum(sdiag[lane]);
    if (lane == 0) red[2] = a;
  }
  d4 acc[2][4];
  const int bad = sf_chain(sBf, SM_LD, sC, sC + 2 * NB * PSUB, acc, tid);
  if (bad != 0 && tid == 0) atomicCAS(reinterpret_cast<int*>(p.cellres + (int64_t)cell * p.cellres_stride + 2), 0, NB + bad);
  __syncthreads();
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * wave + g + 4 * q, col = 16 * kt + r;
      A[p.oLB + row * NB + col] = acc[0][kt][q];
      if (row == col) sdiag[row] = acc[0][kt][q];
      sB[col * SM_LD + row] = acc[1][kt][q];  // LB^-1 = (acc[1])^T
    }
  __syncthreads();
  // LB^-1 to memory (the predict path reads it), c = LB^-1 (u / s)
  for (int e = tid; e < NB * NB; e += 256) A[p.oLBinv + e] = sB[(e >> 6) * SM_LD + (e & 63)];
  {
    const int row = tid & 63, qq = tid >> 6;
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum = __builtin_fma(sB[row * SM_LD + 16 * qq + i], sb[16 * qq + i], sum);
    part[qq * NB + row] = sum;
  }
  __syncthreads();
  if (tid < NB) {
    const double cv = ((part[tid] + part[NB + tid]) + part[2 * NB + tid]) + part[3 * NB + tid];
    sb[tid] = cv;
    A[p.oLB + NB * NB + tid] = cv;  // row 64 of the LB block: c (sgpr_predict_batch's layout)
  }
  __syncthreads();
  if (wave == 0) {
    const double a = wave_sum(log(sdiag[lane]));
    double cq = sb[lane];
    cq = wave_sum(cq * cq);
    if (lane == 0) {
      red[0] = a;
      red[1] = cq;
    }
  }
  if (!p.want_grad) return;
  // ---- gradient algebra (the sequence of sgpr.h sgpr_small_kernel) ----
  for (int row = wave; row < NB; row += 4) {
    const double v = sB[row * SM_LD + lane];
    const double a = wave_sum(v * v);
    if (lane == 0) srow[row] = a;
  }
  trsv_t64(sB, sb, reinterpret_cast<double(*)[NB]>(part), tid);  // LB^-T c   (its barriers also publish srow)
  if (wave == 0) {
    const double a = wave_sum(srow[lane]);
    if (lane == 0) red[3] = a;
  }
  trsv_t64(sL, sb, reinterpret_cast<double(*)[NB]>(part), tid);  // m = L^-T LB^-T c
  if (tid < NB) A[p.oM + tid] = sb[tid];
  d4 accR[2][2], accS[2][2], accT[2][2], accQ[2][2];
  mm64<false>(sB, sL, accR, wm, wn, g, r);  // R = LB^-1 L^-1
  mm64_store(accR, sC, wm, wn, g, r);
  __syncthreads();
  mm64<true>(sC, sC, accS, wm, wn, g, r);   // Sigma^-1 = R^T R
  __syncthreads();
  mm64<false>(sBf, sL, accR, wm, wn, g, r);  // T2 = B L^-1
  mm64_store(accR, sC, wm, wn, g, r);
  __syncthreads();
  mm64<true>(sL, sC, accT, wm, wn, g, r);   // T1 = L^-T T2
  mm64<true>(sL, sL, accQ, wm, wn, g, r);   // Q^-1 = L^-T L^-1
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = wm * 32 + a * 16 + g + 4 * q, col = wn * 32 + b * 16 + r;
        double w, gq;
        sgpr_combine(accQ[a][b][q], accS[a][b][q], accT[a][b][q], sb[row], sb[col], w, gq);
        A[p.oW + row * NB + col] = w;
        A[p.oGQ + row * NB + col] = gq;
      }
}
constexpr size_t SF_MID_SMEM = sizeof(double) * (4 * NB * SM_LD + NB + 4 * NB + NB + NB);
static_assert(2 * 2 * NB * PSUB <= NB * SM_LD, "the chain's two sub-panel buffers fit into one 64 x 64 image");

// ---- launch 5 (host-driven evaluation): partials in chunk order -> the pinned result block ---------------------------------
// red_host (8 per cell): [0..3] from sf_mid, [4] |y - P^T m|^2.  sums_host (2 width per cell, width = 2 + d): the layout the host tail of
// sgpr_objective_batch has always read: [0] sum G_P g, [2 + k] dELBO/dl_k through Kuf (isotropic: the total in [2]), then the same
// for Kuu.  dz_host: (m, d) dELBO/dZ.
template <int ISO>
__global__ __launch_bounds__(256) void sf_final_kernel(SfParams p, double* __restrict__ res_host, double* __restrict__ red_host,
                                                       double* __restrict__ sums_host, double* __restrict__ dz_host) {
  const int cell = blockIdx.x, tid = threadIdx.x;
  const double* par = p.cpar + (int64_t)cell * CELL_PAR;
  const double* ls = par + CELL_PAR_LS;
  const double* A = p.arena + (int64_t)cell * p.ss;
  const double* P2 = A + p.oP2;
  const int width = 2 + p.d;
  if (tid < p.cellres_stride) res_host[(int64_t)cell * p.cellres_stride + tid] = p.cellres[(int64_t)cell * p.cellres_stride + tid];
  if (tid < 4) red_host[(int64_t)cell * 8 + tid] = A[p.oRed + tid];
  if (!p.want_grad) return;
  if (tid == 4) {
    double s = 0.0;
    for (int c = 0; c < p.nchunks; ++c) s += P2[(int64_t)c * p.p2w + 2];
    red_host[(int64_t)cell * 8 + 4] = s;
  }
  double* sums = sums_host + (int64_t)cell * 2 * width;
  if (tid >= 8 && tid < 8 + width) {  // through Kuf
    const int e = tid - 8;
    double v = 0.0;
    if (e == 0) {
      for (int c = 0; c < p.nchunks; ++c) v += P2[(int64_t)c * p.p2w];
    } else if (e >= 2) {
      const int k = e - 2;
      if (ISO) {
        if (k == 0) {
          for (int c = 0; c < p.nchunks; ++c) v += P2[(int64_t)c * p.p2w + 1];
          v = -v / ls[0];
        }
      } else {
        for (int c = 0; c < p.nchunks; ++c) v += P2[(int64_t)c * p.p2w + 4 + k];
        v = -v / ls[k];
      }
    }
    sums[e] = v;
  }
  if (tid >= 128 && tid < 128 + width) {  // through Kuu
    const int e = tid - 128;
    const double* Pq = P2 + (int64_t)p.nchunks * p.p2w;
    double v = 0.0;
    if (e == 0) {
      v = Pq[0];
    } else if (e >= 2) {
      const int k = e - 2;
      if (ISO) {
        if (k == 0) v = -Pq[1] / ls[0];
      } else {
        v = -Pq[4 + k] / ls[k];
      }
    }
    sums[width + e] = v;
  }
  for (int e = tid; e < p.m * p.d; e += 256) {
    const int k = e % p.d;
    double v = 0.0;
    for (int c = 0; c <= p.nchunks; ++c) v += P2[(int64_t)c * p.p2w + SF_P2_HEAD + e];
    dz_host[(int64_t)cell * p.m * p.d + e] = v / ls[k];
  }
}


#define SF_KID_SWITCH(kid, MACRO) \
  switch (kid) {                  \
    case 0: MACRO(0) break;       \
    case 1: MACRO(1) break;       \
    case 2: MACRO(2) break;       \
    case 3: MACRO(3) break;       \
    case 4: MACRO(4) break;       \
    default: return hipErrorInvalidValue; \
  }

hipError_t sf_launch_prep(hipStream_t st, int kid, int form, const SfParams& p, int cells, const double* par_src, const double* z_src,
                          double* cpar_dst) {
#define SF_CASE(K_)                                                                                               \
  if (form) hipLaunchKernelGGL((sf_prep_kernel<K_, 1>), dim3(cells), dim3(256), 0, st, p, par_src, z_src, cpar_dst); \
  else hipLaunchKernelGGL((sf_prep_kernel<K_, 0>), dim3(cells), dim3(256), 0, st, p, par_src, z_src, cpar_dst);
  SF_KID_SWITCH(kid, SF_CASE)
#undef SF_CASE
  return hipGetLastError();
}

hipError_t sf_launch_mid(hipStream_t st, const SfParams& p, int cells) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sf_mid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SF_MID_SMEM);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(sf_mid_kernel, dim3(cells), dim3(256), SF_MID_SMEM, st, p);
  return hipGetLastError();
}

hipError_t sf_launch_final(hipStream_t st, int iso, const SfParams& p, int cells, double* res_host, double* red_host, double* sums_host,
                           double* dz_host) {
  if (iso) hipLaunchKernelGGL((sf_final_kernel<1>), dim3(cells), dim3(256), 0, st, p, res_host, red_host, sums_host, dz_host);
  else hipLaunchKernelGGL((sf_final_kernel<0>), dim3(cells), dim3(256), 0, st, p, res_host, red_host, sums_host, dz_host);
  return hipGetLastError();
}

hipError_t sf_launch_pass1(hipStream_t st, int kid, int form, const SfParams& p, int cells) {
#define SF_CASE(K_) return sf_launch_pass1_kid##K_(st, form, p, cells);
  SF_KID_SWITCH(kid, SF_CASE)
#undef SF_CASE
}

hipError_t sf_launch_pass2(hipStream_t st, int kid, int form, int iso, const SfParams& p, int cells) {
#define SF_CASE(K_) return sf_launch_pass2_kid##K_(st, form, iso, p, cells);
  SF_KID_SWITCH(kid, SF_CASE)
#undef SF_CASE
}

}  // namespace gprx
