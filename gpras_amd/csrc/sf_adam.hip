// Fused sparse evaluation (sgpr_fused.h), resident Adam loop: the launch that closes step t and opens step t + 1 of every cell --
// partial sums of pass 2 in chunk order, loss, gradient, Keras's update and the stop rule (sf_adam_body), then, for a cell that keeps
// running, Kuu of the NEW variables, its factor and the factor's inverse (sf_prep_compute): the updated Z and hyperparameters pass
// through LDS, one dependent launch and one trip through memory fewer per step than sf_adam + sf_prep.
#include "sf_cell_dev.h"

namespace gprx {

template <int KID, int FORM, int ISO>
__global__ __launch_bounds__(256) void sf_adam_prep_kernel(SfParams p, SfAdam ad, double* __restrict__ cpar_dst) {
  SF_PREP_LDS_DECL
  const int cell = blockIdx.x, tid = threadIdx.x;
  if (ad.active[cell] == 0) return;
  double* A = p.arena + (int64_t)cell * p.ss;
  int info = 0;
  __builtin_memcpy(&info, p.cellres + (int64_t)cell * p.cellres_stride + 2, sizeof(int));
  if (info != 0) {
    sf_adam_failed(ad, cell, tid);
    return;
  }
  // scratch of the update inside the chain's buffers (free until the factorisation): sums, reductions, the new theta, the go flag
  double* shs = sIn;
  double* sred = sIn + 2 * (2 + CELL_PAR - CELL_PAR_LS);
  double* sTh = sred + 8;
  int* keep = reinterpret_cast<int*>(sTh + (2 + CELL_PAR - CELL_PAR_LS));
  static_assert(2 * (2 + CELL_PAR - CELL_PAR_LS) + 8 + (2 + CELL_PAR - CELL_PAR_LS) + 2 <= 2 * NB * PSUB, "the update's scratch fits into the sub-panel buffer");
  sf_adam_body<ISO>(p, ad, cell, tid, shs, sred, sTh, sQ, keep);
  __syncthreads();
  if (*keep == 0) return;
  const double parv = sf_par_from_theta(sTh, ad, cell, p.d, tid);
  __syncthreads();  // (every thread has read sTh: sPar and the chain's buffers may be written)
  if (tid < CELL_PAR) {
    sPar[tid] = parv;
    cpar_dst[(int64_t)cell * CELL_PAR + tid] = parv;
  }
  if (tid < p.cellres_stride) p.cellres[(int64_t)cell * p.cellres_stride + tid] = 0.0;
  sf_prep_compute<KID, FORM>(p, cell, A, sQ, sZ, sIn, sXb, sTab, sPar, tid);
}

hipError_t sf_launch_adam_prep(hipStream_t st, int kid, int form, int iso, const SfParams& p, int cells, const SfAdam& adam, double* cpar_dst) {
#define SF_AP(K_, F_, I_) hipLaunchKernelGGL((sf_adam_prep_kernel<K_, F_, I_>), dim3(cells), dim3(256), 0, st, p, adam, cpar_dst);
#define SF_CASE(K_)                  \
  case K_:                           \
    if (form) {                      \
      SF_AP(K_, 1, 0)                \
    } else if (iso) {                \
      SF_AP(K_, 0, 1)                \
    } else {                         \
      SF_AP(K_, 0, 0)                \
    }                                \
    break;
  switch (kid) {
    SF_CASE(0)
    SF_CASE(1)
    SF_CASE(2)
    SF_CASE(3)
    SF_CASE(4)
    default: return hipErrorInvalidValue;
  }
#undef SF_CASE
#undef SF_AP
  return hipGetLastError();
}

}  // namespace gprx
