// Sparse model (SGPR, Titsias bound), M <= 64 inducing points: ONE evaluation of the loss and its gradient in FIVE launches.
//
// Reference hot loop: /root/reference/gpras/gpr.py:147-173 (every Adam step is one SGPR.training_loss + gradient) and the sweeps of
// production/analysis/cross_validation.py:100-126.  Round 4 evaluated this with 21 launches of 5-26 us each (kernel-matrix pair, two
// generic panel factorisations, seven GEMM-shaped launches with one tile per cell, ...): pure launch latency for ~1e8 flops per cell.
// Here the evaluation is cut where the data dependencies are, and nowhere else:
//
//   sf_prep   (cells)            stage-in, Kuu + jitter -> L, L^-1 (one workgroup per cell, the 64 x 64 chain of tile_ops.h)
//   sf_pass1  (chunks, cells)    per 256 columns of Kuf: P tile from x and Z (never stored), A' = L^-1 P on MFMA against the register-
//                                resident L^-1, partial S = A' A'^T and u = A' y      -> one slab per chunk
//   sf_mid    (cells)            S, u summed over the chunks in chunk order; B = I + S / s -> LB, LB^-1, c, the ELBO's reductions, and
//                                the M x M algebra of the gradient (R, Sigma^-1, T, Q^-1, m, W, G_Q)
//   sf_pass2  (chunks + 1, cells) per 256 columns: P and h recomputed, W P on MFMA, G_P = (W P + m y^T) / s contracted with dk/dtheta,
//                                dk/dZ and the residual |y - P^T m|^2; the extra workgroup of a cell does the same with G_Q against Kuu
//   sf_final  (cells)            partials summed in chunk order -> pinned host block (or, in the device-resident Adam loop, the
//                                priors, the chain rule through softplus and the parameter update)
//
// Determinism: a chunk is ALWAYS 256 columns and its tiles are accumulated in column order; chunks are added in chunk order by one
// thread per output.  Nothing depends on the number of cells in a launch or on which workgroup ran where, so a model evaluated alone
// and the same model evaluated inside a batch give the same bits.
//
// Layouts.  MFMA (v_mfma_f64_16x16x4): lane (g = lane >> 4, r = lane & 15) of wave (wm, wn) holds C[wm 32 + a 16 + g + 4 q][wn 32 + b 16 + r]
// (gemm_f64.h's convention: stages of 16 along k, instruction j of a stage takes k = k0 + 4 g + j).  Elementwise work (kernel values,
// their derivatives, the contractions) runs in the ROW-LANE layout: lane = inducing point i, wave w = columns 16 w .. 16 w + 15 of the
// tile, so that everything indexed by (i, dimension) -- dK/dZ above all -- accumulates inside one lane with no cross-lane traffic; the
// two layouts meet in LDS.
#pragma once
#include "gprx_common.h"

namespace gprx {

constexpr int SF_CHUNK = 256;  // columns of Kuf per pass workgroup -- fixed, see "Determinism"
constexpr int SF_TILES = SF_CHUNK / NB;
constexpr int SF_DK = 16;      // dimensions per staging pass
constexpr int SF_DKP = 18;     // LDS row stride of the staged coordinates (144 bytes: 16-byte aligned rows, lanes 36 dwords apart)
constexpr int SF_LD = 68;      // LDS row stride of the 64 x 64 operand images (16-byte aligned rows; 4 rows = 32 banks apart)
constexpr int SF_P2_HEAD = 4 + 64;  // pass-2 partial block: [0] sum w g, [1] sum w v h r2 (isotropic), [2] residual, [3] -, [4 ..) sum w v h ds_k^2, then dZ

struct SfParams {
  const double* X;     // (n, d) training inputs
  const double* Y;     // (units, np) outputs, unit-major, zero padded
  double* arena;       // cell blocks, ss doubles apart
  int64_t ss;
  const double* cpar;  // device cell-parameter table (kmat.h: CELL_PAR doubles per cell)
  int n, np, m, d, nchunks;
  // offsets inside a cell block (doubles)
  int64_t oZ, oL, oLinv, oLB, oLBinv, oW, oGQ, oM, oSlab, oU, oP2, oRed;
  int p2w;             // doubles per pass-2 partial block: SF_P2_HEAD + 64 * d
  double* cellres;     // per cell CELL_RES doubles; [2] carries the pivot status as an int
  int cellres_stride;  // doubles
  int want_grad;
  int store_factors;   // 1: L, LB, LB^-1 and c go to the cell block as well (the predict path reads them); the resident optimiser needs none of them
  const int* active;   // device-resident optimiser: per cell 0 = this cell has stopped (every launch returns at once for it); null: all run
  unsigned long long* stamps;  // development aid: null, or SF_STAMP_WORDS words that workgroup (0, 0) fills with s_memtime at its phase boundaries
};
// Device-resident Adam (gpr.py:147-173 for every cell of a batch): the optimiser's state lives in device memory, a step is the five
// launches with sf_adam_kernel as the fifth, the host looks at the stop flags every few steps only.
struct SfAdam {
  double* theta;      // (cells, nt) unconstrained variables [variance, lengthscales .., noise]; Z lives in the cell blocks (oZ)
  double* mom;        // (cells, nt + m d) first moments  [theta | Z]
  double* vel;        // (cells, nt + m d) second moments
  double* best;       // (cells) best loss so far
  double* loss;       // (cells) loss of the last evaluation
  int* stale;         // (cells) steps without an improvement of more than tol
  int* active;        // (cells) 1 while the cell runs
  int* n_evals;       // (cells) evaluations taken part in
  int* tstep;         // (cells) steps done
  const int* units;   // (cells) output column of each cell
  const double* alpha;  // [0 .. max_iter]: lr sqrt(1 - beta2^t) / (1 - beta1^t), formed on the host (pow)
  const double* yy;   // (units) y.y
  int* error;         // [0]: 0, or 1 + the first cell whose Kuu or B stopped being positive definite
  int nt, nlen, ard, mask, max_iter;
};

constexpr int SF_STAMP_WORDS = 5 * 32;  // prep | pass 1 | mid | pass 2 | final, 32 words each ([31] of each block: s_memrealtime at entry, 100 MHz)
#define SF_STAMP(p_, base_, i_)                                                                                   \
  if ((p_).stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                         \
    (p_).stamps[(base_) + (i_)] = __builtin_amdgcn_s_memtime();                                                   \
    if ((i_) == 0) (p_).stamps[(base_) + 31] = __builtin_amdgcn_s_memrealtime();                                  \
  }

// ---- launchers (defined in sf_cell.hip, sf_pass1.hip, sf_pass2.hip; the pass kernels are compiled once per kernel id, -DSF_KID=k,
// so that the translation units build in parallel) ------------------------------------------------------------------------------
// par_src: `cells` rows of CELL_PAR doubles (pinned host memory or device memory); z_src: (cells, m, d) inducing inputs, or nullptr when
// the cell blocks already hold Z; cpar_dst: the device parameter table the other launches read.
hipError_t sf_launch_prep(hipStream_t st, int kid, int form, const SfParams& p, int cells, const double* par_src, const double* z_src,
                          double* cpar_dst, const SfAdam* adam = nullptr);
// The launch between two steps of the resident Adam loop (sf_adam.hip): partial sums in chunk order, loss, gradient, update, stop rule of
// step t, then -- for cells that keep running -- Kuu, L, L^-1 of the new variables (the prep of step t + 1).  A step is FOUR launches:
// pass 1, mid, pass 2, this one; sf_launch_prep opens the first step.
hipError_t sf_launch_adam_prep(hipStream_t st, int kid, int form, int iso, const SfParams& p, int cells, const SfAdam& adam, double* cpar_dst);
hipError_t sf_launch_pass1(hipStream_t st, int kid, int form, const SfParams& p, int cells);
hipError_t sf_launch_mid(hipStream_t st, const SfParams& p, int cells);
hipError_t sf_launch_pass2(hipStream_t st, int kid, int form, int iso, const SfParams& p, int cells);
hipError_t sf_launch_final(hipStream_t st, int iso, const SfParams& p, int cells, double* res_host, double* red_host, double* sums_host,
                           double* dz_host);
#define SF_DECLARE_PASS(K_)                                                                       \
  hipError_t sf_launch_pass1_kid##K_(hipStream_t st, int form, const SfParams& p, int cells);     \
  hipError_t sf_launch_pass2_kid##K_(hipStream_t st, int form, int iso, const SfParams& p, int cells);
SF_DECLARE_PASS(0)
SF_DECLARE_PASS(1)
SF_DECLARE_PASS(2)
SF_DECLARE_PASS(3)
SF_DECLARE_PASS(4)
#undef SF_DECLARE_PASS

}  // namespace gprx
