// Fused sparse evaluation (sgpr_fused.h): the one-workgroup-per-cell launches -- prep (Kuu, L, L^-1), mid (B, LB, c, the M x M gradient
// algebra) and final (partials in chunk order -> the pinned result block) -- and the dispatch of the pass launchers by kernel id.
#include "sf_cell_dev.h"

namespace gprx {


// ---- launch 1: stage-in, Kuu, its factor and the factor's inverse ------------------------------------------------------------
// par_src: `cells` rows of CELL_PAR doubles (pinned host memory or device memory); z_src: (cells, m, d) inducing inputs or nullptr
// when the cell blocks already hold Z (device-resident optimiser).  The row is copied into the device table that the other launches
// read, the cell's result words are cleared.
template <int KID, int FORM>
__global__ __launch_bounds__(256) void sf_prep_kernel(SfParams p, const double* __restrict__ par_src, const double* __restrict__ z_src,
                                                      double* __restrict__ cpar_dst, SfAdam ad) {
  SF_PREP_LDS_DECL
  const int cell = blockIdx.x, tid = threadIdx.x;
  double* A = p.arena + (int64_t)cell * p.ss;
  if (p.active != nullptr && p.active[cell] == 0) return;
  SF_STAMP(p, 0, 0)
  // the parameter row and the inducing inputs come over the host link when the evaluation is host-driven: ONE round trip -- both
  // requests go out before either is used; Z stays in LDS (raw, in the image that Kuu will take afterwards) for the staging passes
  double* sZraw = sQ;  // [m][d] <= 64 x 64 doubles
  static_assert(NB * (CELL_PAR - CELL_PAR_LS) <= NB * SF_LD, "the raw inducing inputs fit into the Kuu image");
  const int nz = p.m * p.d;
  const double* zsrc = z_src ? z_src + (int64_t)cell * nz : A + p.oZ;
  double parv = 0.0;
  if (ad.theta != nullptr) {
    // resident optimiser: the row is formed here from the unconstrained variables (decode_theta of gprx.hip on the device: px_math.h
    // gives the host's bits)
    parv = sf_par_from_theta(ad.theta + (int64_t)cell * ad.nt, ad, cell, p.d, tid);
  } else if (tid < CELL_PAR) {
    parv = par_src[(int64_t)cell * CELL_PAR + tid];
  }
  for (int e0 = 0; e0 < nz; e0 += 256 * 8) {
    double zv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) zv[u] = zsrc[min(e0 + 256 * u + tid, nz - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u + tid;
      if (e < nz) {
        sZraw[e] = zv[u];
        if (z_src) A[p.oZ + e] = zv[u];
      }
    }
  }
  if (tid < CELL_PAR) {
    sPar[tid] = parv;
    cpar_dst[(int64_t)cell * CELL_PAR + tid] = parv;
  }
  if (tid < p.cellres_stride) p.cellres[(int64_t)cell * p.cellres_stride + tid] = 0.0;
  sf_prep_compute<KID, FORM>(p, cell, A, sQ, sZ, sIn, sXb, sTab, sPar, tid);
}

// The five products of the gradient algebra for ONE wave (template: everything unrolls).  Tiles (rt << 4 | ct), up to four per wave:
// R = LB^-1 L^-1 is lower triangular (tile (rt, ct): k blocks ct .. rt -- 20 blocks in all, 5 per wave); Sigma^-1 = R^T R, T1 = L^-T T2 and
// Q^-1 = L^-T L^-1 are symmetric (lower tiles, k blocks rt .. 3 -- 5 per wave); T2 = B L^-1: wave w = row block w, k blocks ct .. 3.
// Every wave passes the same three barriers.
template <bool TA, int RT, int CT, int KB0, int KB1>
__device__ __forceinline__ d4 sf_tile_mm(const double* __restrict__ sa, const double* __restrict__ sbm, int g, int r) {
  d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kb = KB0; kb < KB1; ++kb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kb * 16 + 4 * g + j;
      const double av = TA ? sa[k * SM_LD + RT * 16 + r] : sa[(RT * 16 + r) * SM_LD + k];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, sbm[k * SM_LD + CT * 16 + r], acc, 0, 0, 0);
    }
  return acc;
}
template <int RT, int CT>
__device__ __forceinline__ void sf_tile_store(const d4& acc, double* __restrict__ dst, int g, int r) {
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[(RT * 16 + g + 4 * q) * SM_LD + CT * 16 + r] = acc[q];
}
template <int WAVE>
struct SfMidTiles {
  // slot -> code (rt << 4 | ct), 0xff = none
  static constexpr int r_tile(int sl) {
    constexpr int t[4][4] = {{0x30, 0x00, 0xff, 0xff}, {0x31, 0x10, 0xff, 0xff}, {0x20, 0x21, 0xff, 0xff}, {0x32, 0x11, 0x22, 0x33}};
    return t[WAVE][sl];
  }
  static constexpr int s_tile(int sl) {
    constexpr int t[4][4] = {{0x00, 0x30, 0xff, 0xff}, {0x10, 0x20, 0xff, 0xff}, {0x11, 0x21, 0xff, 0xff}, {0x22, 0x31, 0x32, 0x33}};
    return t[WAVE][sl];
  }
};
template <int WAVE, int SL>
__device__ __forceinline__ void sf_mid_r_slot(const double* __restrict__ sB, const double* __restrict__ sL, double* __restrict__ sC, int g, int r) {
  constexpr int code = SfMidTiles<WAVE>::r_tile(SL);
  if constexpr (code != 0xff) {
    constexpr int rt = code >> 4, ct = code & 15;
    sf_tile_store<rt, ct>(sf_tile_mm<false, rt, ct, ct, rt + 1>(sB, sL, g, r), sC, g, r);
  }
}
template <int WAVE, int SL>
__device__ __forceinline__ void sf_mid_s_slot(const double* __restrict__ sC, d4& accS, int g, int r) {
  constexpr int code = SfMidTiles<WAVE>::s_tile(SL);
  if constexpr (code != 0xff) accS = sf_tile_mm<true, (code >> 4), (code & 15), (code >> 4), 4>(sC, sC, g, r);
}
template <int WAVE, int SL>
__device__ __forceinline__ void sf_mid_w_slot(const double* __restrict__ sL, const double* __restrict__ sC, const d4& accS, const double* __restrict__ sb,
                                              double* __restrict__ Wout, double* __restrict__ GQout, int g, int r) {
  constexpr int code = SfMidTiles<WAVE>::s_tile(SL);
  if constexpr (code != 0xff) {
    constexpr int rt = code >> 4, ct = code & 15;
    const d4 accT = sf_tile_mm<true, rt, ct, rt, 4>(sL, sC, g, r);
    const d4 accQ = sf_tile_mm<true, rt, ct, rt, 4>(sL, sL, g, r);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rt * 16 + g + 4 * q, col = ct * 16 + r;
      double w, gq;
      sgpr_combine(accQ[q], accS[q], accT[q], sb[row], sb[col], w, gq);
      Wout[row * NB + col] = w;
      GQout[row * NB + col] = gq;
      if (rt != ct) {  // W and G_Q on both sides of the diagonal
        Wout[col * NB + row] = w;
        GQout[col * NB + row] = gq;
      }
    }
  }
}
template <int WAVE>
__device__ __forceinline__ void sf_mid_products(const double* __restrict__ sL, const double* __restrict__ sB, double* __restrict__ sC,
                                                const double* __restrict__ sBf, const double* __restrict__ sb, double* __restrict__ Wout,
                                                double* __restrict__ GQout, int g, int r) {
  // R = LB^-1 L^-1 (lower) -> sC
  sf_mid_r_slot<WAVE, 0>(sB, sL, sC, g, r);
  sf_mid_r_slot<WAVE, 1>(sB, sL, sC, g, r);
  sf_mid_r_slot<WAVE, 2>(sB, sL, sC, g, r);
  sf_mid_r_slot<WAVE, 3>(sB, sL, sC, g, r);
  __syncthreads();
  // Sigma^-1 = R^T R on the lower tiles (R[k][i] is zero for k < i: k blocks rt .. 3)
  d4 accS[4];
  sf_mid_s_slot<WAVE, 0>(sC, accS[0], g, r);
  sf_mid_s_slot<WAVE, 1>(sC, accS[1], g, r);
  sf_mid_s_slot<WAVE, 2>(sC, accS[2], g, r);
  sf_mid_s_slot<WAVE, 3>(sC, accS[3], g, r);
  __syncthreads();
  // T2 = B L^-1 (all sixteen tiles: wave w = row block w; k blocks ct .. 3) -> sC
  sf_tile_store<WAVE, 0>(sf_tile_mm<false, WAVE, 0, 0, 4>(sBf, sL, g, r), sC, g, r);
  sf_tile_store<WAVE, 1>(sf_tile_mm<false, WAVE, 1, 1, 4>(sBf, sL, g, r), sC, g, r);
  sf_tile_store<WAVE, 2>(sf_tile_mm<false, WAVE, 2, 2, 4>(sBf, sL, g, r), sC, g, r);
  sf_tile_store<WAVE, 3>(sf_tile_mm<false, WAVE, 3, 3, 4>(sBf, sL, g, r), sC, g, r);
  __syncthreads();
  // T1 = L^-T T2 and Q^-1 = L^-T L^-1 on the lower tiles, combined into W and G_Q
  sf_mid_w_slot<WAVE, 0>(sL, sC, accS[0], sb, Wout, GQout, g, r);
  sf_mid_w_slot<WAVE, 1>(sL, sC, accS[1], sb, Wout, GQout, g, r);
  sf_mid_w_slot<WAVE, 2>(sL, sC, accS[2], sb, Wout, GQout, g, r);
  sf_mid_w_slot<WAVE, 3>(sL, sC, accS[3], sb, Wout, GQout, g, r);
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
  if (p.active != nullptr && p.active[cell] == 0) return;
  SF_STAMP(p, 64, 0)
  load64(A + p.oLinv, sL, tid);  // (requested first: the loads travel under the slab sums)
  // S = sum over the chunks, in chunk order, of the three stored quadrants; B = I + S / s
  {
    const double* slab = A + p.oSlab;
    // the ten lower 16 x 16 tiles = 1280 pairs of neighbours: five per thread
    double s[5][2];
#pragma unroll
    for (int e = 0; e < 5; ++e) s[e][0] = s[e][1] = 0.0;
    int soff[5], si[5], sj[5], sdiag_tile[5];
#pragma unroll
    for (int e = 0; e < 5; ++e) {
      const int q = tid + 256 * e;  // tile = q / 128 (row-major over the lower tiles), inside: row = (q % 128) / 8, pair = q % 8
      const int t = q >> 7, row = (q & 127) >> 3, pr = q & 7;
      const int rb = t < 1 ? 0 : (t < 3 ? 1 : (t < 6 ? 2 : 3)), cb = t - rb * (rb + 1) / 2;
      si[e] = rb * 16 + row;
      sj[e] = cb * 16 + 2 * pr;
      sdiag_tile[e] = rb == cb;
      soff[e] = si[e] * NB + sj[e];
    }
    // (eight chunks' loads in flight: a rolled loop over the chunks waits for each chunk's loads -- ~2 us each, the slabs come from other
    // CUs' stores -- before it issues the next ones)
    int c = 0;
    for (; c + 8 <= p.nchunks; c += 8) {
      d2 v[8][5];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 5; ++e) v[u][e] = *reinterpret_cast<const d2*>(slab + (int64_t)(c + u) * NB * NB + soff[e]);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 5; ++e) {
          s[e][0] += v[u][e].x;
          s[e][1] += v[u][e].y;
        }
    }
    for (; c < p.nchunks; ++c) {
#pragma unroll
      for (int e = 0; e < 5; ++e) {
        const d2 v = *reinterpret_cast<const d2*>(slab + (int64_t)c * NB * NB + soff[e]);
        s[e][0] += v.x;
        s[e][1] += v.y;
      }
    }
#pragma unroll
    for (int e = 0; e < 5; ++e) {
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const double sv = s[e][c2];
        const int i = si[e], jj = sj[e] + c2;
        const double bv = __builtin_fma(sv, inv_s, i == jj ? 1.0 : 0.0);
        sBf[i * SM_LD + jj] = bv;
        if (!sdiag_tile[e]) sBf[jj * SM_LD + i] = bv;  // mirror of a tile below the diagonal (diagonal tiles were computed whole)
        if (i == jj) sdiag[i] = sv * inv_s;
      }
    }
    if (tid < NB) sb[tid] = sf_sum_chunks(A + p.oU + tid, NB, p.nchunks) * inv_s;
  }
  __syncthreads();
  SF_STAMP(p, 64, 1)
  if (wave == 0) {  // tr(A A^T)
    const double a = wave_sum_dpp(sdiag[lane]);
    if (lane == 0) red[2] = a;
  }
  d4 acc[2][4];
  const int bad = sf_chain(sBf, SM_LD, sC, sC + 2 * NB * PSUB, acc, tid);
  SF_STAMP(p, 64, 2)
  if (bad != 0 && tid == 0) atomicCAS(reinterpret_cast<int*>(p.cellres + (int64_t)cell * p.cellres_stride + 2), 0, NB + bad);
  __syncthreads();
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * wave + g + 4 * q, col = 16 * kt + r;
      if (p.store_factors) A[p.oLB + row * NB + col] = acc[0][kt][q];
      if (row == col) sdiag[row] = acc[0][kt][q];
      sB[col * SM_LD + row] = acc[1][kt][q];  // LB^-1 = (acc[1])^T
    }
  __syncthreads();
  // LB^-1 to memory (the predict path reads it), c = LB^-1 (u / s)
  if (p.store_factors)
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
    if (p.store_factors) A[p.oLB + NB * NB + tid] = cv;  // row 64 of the LB block: c (sgpr_predict_batch's layout)
  }
  __syncthreads();
  if (wave == 0) {
    const double a = wave_sum_dpp(log(sdiag[lane]));
    double cq = sb[lane];
    cq = wave_sum_dpp(cq * cq);
    if (lane == 0) {
      red[0] = a;
      red[1] = cq;
    }
  }
  SF_STAMP(p, 64, 3)
  if (!p.want_grad) return;
  // ---- gradient algebra (sgpr.h sgpr_small_kernel's quantities) with the structure used: L^-1, LB^-1 and R are lower triangular, Sigma^-1,
  // T1 and Q^-1 symmetric -- 120 MFMAs per wave instead of 320.  Work is dealt in 16 x 16 tiles with their k-block ranges (tables below).
  {
    double sq = 0.0;  // |LB^-1|_F^2: sixteen elements per thread, then the lanes, then the waves
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = tid + 256 * e;
      const double v = sB[(q >> 6) * SM_LD + (q & 63)];
      sq = __builtin_fma(v, v, sq);
    }
    sq = wave_sum_dpp(sq);
    if (lane == 0) srow[wave] = sq;
  }
  trsv_t64(sB, sb, reinterpret_cast<double(*)[NB]>(part), tid);  // LB^-T c   (its barriers also publish srow)
  if (tid == 0) red[3] = (srow[0] + srow[1]) + (srow[2] + srow[3]);
  trsv_t64(sL, sb, reinterpret_cast<double(*)[NB]>(part), tid);  // m = L^-T LB^-T c
  if (tid < NB) A[p.oM + tid] = sb[tid];
  SF_STAMP(p, 64, 4)
  // (each wave runs its own fully unrolled copy: tile indices and k ranges are compile-time constants there, so the compiler hoists the
  // LDS reads and interleaves the MFMAs of independent tiles -- an accumulation chain on ONE tile issues only every ~150 clocks)
  switch (wave) {
    case 0: sf_mid_products<0>(sL, sB, sC, sBf, sb, A + p.oW, A + p.oGQ, g, r); break;
    case 1: sf_mid_products<1>(sL, sB, sC, sBf, sb, A + p.oW, A + p.oGQ, g, r); break;
    case 2: sf_mid_products<2>(sL, sB, sC, sBf, sb, A + p.oW, A + p.oGQ, g, r); break;
    default: sf_mid_products<3>(sL, sB, sC, sBf, sb, A + p.oW, A + p.oGQ, g, r); break;
  }
  SF_STAMP(p, 64, 5)
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
  SF_STAMP(p, 128, 0)
  if (tid < p.cellres_stride) res_host[(int64_t)cell * p.cellres_stride + tid] = p.cellres[(int64_t)cell * p.cellres_stride + tid];
  if (tid < 4) red_host[(int64_t)cell * 8 + tid] = A[p.oRed + tid];
  if (!p.want_grad) return;
  sf_reduce_sums<ISO>(p, P2, ls, tid, red_host + (int64_t)cell * 8 + 4, sums_host + (int64_t)cell * 2 * width);
  {
    const int nz = p.m * p.d;
    for (int e0 = 0; e0 < nz; e0 += 256 * 4) {  // four outputs per thread at once: 32 loads in flight
      int off[4];
      double acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) off[u] = SF_P2_HEAD + min(e0 + 256 * u + tid, nz - 1);
      sf_sum_chunks_n<4>(P2, off, p.p2w, p.nchunks, acc);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + 256 * u + tid;
        if (e < nz) dz_host[(int64_t)cell * nz + e] = acc[u] / ls[e % p.d];
      }
    }
  }
  SF_STAMP(p, 128, 1)
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
                          double* cpar_dst, const SfAdam* adam) {
  SfAdam ad{};
  if (adam) ad = *adam;
#define SF_CASE(K_)                                                                                                   \
  if (form) hipLaunchKernelGGL((sf_prep_kernel<K_, 1>), dim3(cells), dim3(256), 0, st, p, par_src, z_src, cpar_dst, ad); \
  else hipLaunchKernelGGL((sf_prep_kernel<K_, 0>), dim3(cells), dim3(256), 0, st, p, par_src, z_src, cpar_dst, ad);
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
