// Device code shared by the one-workgroup-per-cell launches of the fused sparse evaluation (sf_cell.hip: prep, mid, final; sf_adam.hip: the
// merged Adam + prep launch of the resident optimiser).  Kernel-free.
#pragma once
#include "sgpr_asm.h"
#include "sgpr_fused_dev.h"

namespace gprx {

// entry `tid` of a cell's parameter row (kfun.h CELL_PAR layout) from the unconstrained variables th[nt]: decode_theta of gprx.hip on
// the device (px_math.h gives the host's bits); zero for the unused entries
__device__ __forceinline__ double sf_par_from_theta(const double* __restrict__ th, const SfAdam& ad, int cell, int d, int tid) {
  double parv = 0.0;
  if (tid == 0) parv = px_softplus(th[0]);
  if (tid == 1) parv = NOISE_LOWER + px_softplus(th[ad.nt - 1]);
  if (tid == 2) parv = (double)ad.units[cell];
  if (tid == 3) parv = 1.0 / (NOISE_LOWER + px_softplus(th[ad.nt - 1]));
  if (tid >= CELL_PAR_LS && tid < CELL_PAR_LS + d) parv = px_softplus(th[1 + (ad.ard ? tid - CELL_PAR_LS : 0)]);
  return parv;
}

// LDS of the prep launch (and of the merged Adam + prep launch): 63.5 KB static
struct SfPrepLds {
  double* sQ;    // [NB * SF_LD]: the raw inducing inputs first, then Kuu, then L^-1
  double* sZ;    // [NB * SF_DKP]
  double* sIn;   // [2 * NB * PSUB]
  double* sXb;   // [2 * NB * PSUB]
  double* sTab;  // [64]
  double* sPar;  // [CELL_PAR]
};
#define SF_PREP_LDS_DECL                                                      \
  __shared__ __attribute__((aligned(16))) double sQ[NB * SF_LD];              \
  __shared__ __attribute__((aligned(16))) double sZ[NB * SF_DKP];             \
  __shared__ __attribute__((aligned(16))) double sIn[2 * NB * PSUB];          \
  __shared__ __attribute__((aligned(16))) double sXb[2 * NB * PSUB];          \
  __shared__ __attribute__((aligned(16))) double sTab[64];                    \
  __shared__ double sPar[CELL_PAR];

// Kuu + jitter from the raw inducing inputs in sQ and the parameter row in sPar (both complete: the caller has not yet synchronised),
// its factor and the factor's inverse -> the cell block.
template <int KID, int FORM>
__device__ __forceinline__ void sf_prep_compute(const SfParams& p, int cell, double* __restrict__ A, double* __restrict__ sQ, double* __restrict__ sZ,
                                                double* __restrict__ sIn, double* __restrict__ sXb, double* __restrict__ sTab,
                                                const double* __restrict__ sPar, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const double* sZraw = sQ;
  const double* zp = sZraw;
  exp_tab_fill(sTab);
  __syncthreads();
  SF_STAMP(p, 0, 1)
  const double variance = sPar[0];
  const double* ls = sPar + CELL_PAR_LS;
  double r2[16], nb[16], na = 0.0;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) r2[jj] = nb[jj] = 0.0;
  for (int k0 = 0; k0 < p.d; k0 += SF_DK) {
    if (k0 > 0) __syncthreads();
    sf_stage<FORM>(zp, 0, p.m, p.d, k0, ls, sZ, tid);
    __syncthreads();
    // (all eight pairs of the chunk, unconditionally: dimensions beyond d are staged as zeros and add exact zeros, and the unguarded
    // loop is the hand-pipelined one -- the guarded loop waits out every LDS read)
    sf_r2_chunk<FORM, 16, 8>(sZ, sZ, lane, wave, SF_DK, r2, na, nb);
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
  SF_STAMP(p, 0, 2)
  d4 acc[2][4];
  const int bad = sf_chain(sQ, SF_LD, sIn, sXb, acc, tid, p.stamps);
  SF_STAMP(p, 0, 3)
  if (bad != 0 && tid == 0) atomicCAS(reinterpret_cast<int*>(p.cellres + (int64_t)cell * p.cellres_stride + 2), 0, bad);
  // L straight from the accumulators (16 lanes = one 128-byte line); L^-1 = (acc[1])^T through LDS
  const int g = lane >> 4, r = lane & 15;
  __syncthreads();  // (every wave has finished reading sQ)
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * wave + g + 4 * q, col = 16 * kt + r;
      if (p.store_factors) A[p.oL + row * NB + col] = acc[0][kt][q];
      sQ[col * SF_LD + row] = acc[1][kt][q];
    }
  __syncthreads();
  sf_image_out(sQ, SF_LD, A + p.oLinv, tid);
  SF_STAMP(p, 0, 4)
}

// the sums of a cell's pass-2 partial blocks, in chunk order: red4 = |y - P^T m|^2, sums[2 width] (thread ranges [8, 8 + width) and
// [128, 128 + width)); shared by the host-driven final launch and the resident optimiser's, so both see the same bits
template <int ISO>
__device__ __forceinline__ void sf_reduce_sums(const SfParams& p, const double* __restrict__ P2, const double* __restrict__ ls, int tid,
                                               double* __restrict__ red4, double* __restrict__ sums) {
  const int width = 2 + p.d;
  if (tid == 4) *red4 = sf_sum_chunks(P2 + 2, p.p2w, p.nchunks);
  if (tid >= 8 && tid < 8 + width) {  // through Kuf
    const int e = tid - 8;
    double v = 0.0;
    if (e == 0) {
      v = sf_sum_chunks(P2, p.p2w, p.nchunks);
    } else if (e >= 2) {
      const int k = e - 2;
      if (ISO) {
        if (k == 0) v = -sf_sum_chunks(P2 + 1, p.p2w, p.nchunks) / ls[0];
      } else {
        v = -sf_sum_chunks(P2 + 4 + k, p.p2w, p.nchunks) / ls[k];
      }
    }
    sums[e] = v;
  }
  if (tid >= 128 && tid < 128 + width) sums[width + (tid - 128)] = 0.0;  // (the Kuu terms are inside the chunks' blocks since the slicing of pass 2)
}

// ---- one step of the resident Adam loop for one cell ---------------------------------------------------------------------------
// What sgpr_objective_batch's host tail, chain_rule, log_prior and gprx_adam_batch's host loop do for one cell and one step, in their
// order and with their arithmetic (sgpr_asm.h): the same variables after every step, bit for bit.  shs [2 width], sred [8]: LDS scratch.
// sTh (nt) and sZnew (m d), when given, receive the variables after the update (LDS: the merged launch goes on to the next step's
// Kuu from them).  Returns through *keep (LDS int, written by thread 255) whether the cell keeps running.  Every thread must call it;
// it contains one __syncthreads().
template <int ISO>
__device__ __forceinline__ void sf_adam_body(const SfParams& p, const SfAdam& ad, int cell, int tid, double* __restrict__ shs, double* __restrict__ sred,
                                             double* __restrict__ sTh, double* __restrict__ sZnew, int* __restrict__ keep) {
  const double* par = p.cpar + (int64_t)cell * CELL_PAR;
  const double* ls = par + CELL_PAR_LS;
  double* A = p.arena + (int64_t)cell * p.ss;
  const double* P2 = A + p.oP2;
  const int width = 2 + p.d, nt = ad.nt, nz = p.m * p.d, gw = nt + nz;
  if (tid < 4) sred[tid] = A[p.oRed + tid];
  sf_reduce_sums<ISO>(p, P2, ls, tid, &sred[4], shs);
  __syncthreads();
  const double variance = par[0], noise = par[1];
  const double nn = (double)p.n;
  const int t = ad.tstep[cell] + 1;
  const double alpha = ad.alpha[t];
  double* th = ad.theta + (int64_t)cell * nt;
  double* mom = ad.mom + (int64_t)cell * gw;
  double* vel = ad.vel + (int64_t)cell * gw;
  if (tid < nt) {
    const int k = tid;
    const double du = sgpr_asm_dparam(k, ad.nlen, ad.ard, p.d, width, nn, NB, variance, noise, sred, shs);
    double u, w = th[k];
    bool trainable;
    if (k == 0) {
      u = variance;
      trainable = (ad.mask & ASM_TRAIN_VARIANCE) != 0;
    } else if (k < nt - 1) {
      u = ls[k - 1];
      trainable = (ad.mask & ASM_TRAIN_LENGTHSCALE) != 0;
    } else {
      u = noise;
      trainable = (ad.mask & ASM_TRAIN_NOISE) != 0;
    }
    const double ge = sgpr_asm_chain(du, u, w, trainable);
    if (trainable) {
      double mo = mom[k], ve = vel[k];
      adam_element(ge, alpha, mo, ve, w);
      mom[k] = mo;
      vel[k] = ve;
      th[k] = w;
    }
    if (sTh) sTh[k] = w;
  }
  const bool train_z = (ad.mask & ASM_TRAIN_Z) != 0;
  if (train_z || sZnew) {
    for (int e0 = 0; e0 < nz; e0 += 256 * 4) {  // four elements per thread at once: their 32 + 12 loads in flight together
      int off[4];
      double acc[4], mo[4], ve[4], x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = min(e0 + 256 * u + tid, nz - 1);
        off[u] = SF_P2_HEAD + e;
        mo[u] = train_z ? mom[nt + e] : 0.0;
        ve[u] = train_z ? vel[nt + e] : 0.0;
        x[u] = A[p.oZ + e];
      }
      if (train_z) sf_sum_chunks_n<4>(P2, off, p.p2w, p.nchunks, acc);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + 256 * u + tid;
        if (e < nz) {
          if (train_z) {
            const double ge = -(acc[u] / ls[e % p.d]);
            adam_element(ge, alpha, mo[u], ve[u], x[u]);
            mom[nt + e] = mo[u];
            vel[nt + e] = ve[u];
            A[p.oZ + e] = x[u];
          }
          if (sZnew) sZnew[e] = x[u];
        }
      }
    }
  }
  if (tid == 255) {  // (a thread with no hyperparameter of its own)
    const double elbo = sgpr_asm_elbo(nn, ad.yy[ad.units[cell]], variance, noise, sred);
    double lp = 0.0;
    {
#pragma clang fp contract(off)
      if (ad.mask & ASM_TRAIN_VARIANCE) lp += px_ln_logpdf(variance);
      if (ad.mask & ASM_TRAIN_LENGTHSCALE)
        for (int k = 0; k < ad.nlen; ++k) lp += px_ln_logpdf(ls[k]);
      if (ad.mask & ASM_TRAIN_NOISE) lp += px_ln_logpdf(noise);
    }
    const double loss = -(elbo + lp);
    double best = ad.best[cell];
    int stale = ad.stale[cell];
    const bool go = adam_keep_running(loss, best, stale) && t < ad.max_iter;
    ad.best[cell] = best;
    ad.stale[cell] = stale;
    ad.loss[cell] = loss;
    ad.n_evals[cell] += 1;
    ad.tstep[cell] = t;
    if (!go) ad.active[cell] = 0;
    if (keep) *keep = go ? 1 : 0;
  }
}

// the evaluation of this cell failed (Kuu or B not positive definite): it counts, nothing is updated, the call ends with GPRX_ENOTPD
__device__ __forceinline__ void sf_adam_failed(const SfAdam& ad, int cell, int tid) {
  if (tid == 0) {
    atomicCAS(ad.error, 0, cell + 1);
    ad.n_evals[cell] += 1;
    ad.active[cell] = 0;
    ad.loss[cell] = __builtin_nan("");
  }
}

}  // namespace gprx
