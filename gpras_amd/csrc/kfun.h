// Correlation functions of the five stationary kernels (RBF, Matern-1/2, -3/2, -5/2, gpflow "Exponential"), their derivative factor h,
// and gpflow's expanded squared distance -- device inline functions only (no kernels: any translation unit may include this).  Split out
// of kmat.h in round 5.
#pragma once
#include "gprx_common.h"

namespace gprx {

// cell-parameter table (batched launches): CELL_PAR doubles per cell: [0] variance, [1] noise / diag_add, [2] unit, [3] 1 / noise,
// [CELL_PAR_LS .. CELL_PAR_LS + d) lengthscales
constexpr int CELL_PAR = 72;
constexpr int CELL_PAR_LS = 8;

// the build's exponential: table + degree-5 polynomial (gprx_common.h exp_nonpos_tab; `tab` = the workgroup's LDS copy of 2^(j/64));
// -DGPRX_KMAT_OLDEXP restores the degree-13 Taylor form for A/B measurements
__device__ __forceinline__ double kmat_exp(double x, const double* __restrict__ tab) {
#ifdef GPRX_KMAT_OLDEXP
  return exp_nonpos(x);
#else
  return exp_nonpos_tab(x, tab);
#endif
}

template <int KID>
__device__ __forceinline__ double corr_g(double r2, const double* __restrict__ tab) {
  if constexpr (KID == 0) {
    return kmat_exp(-0.5 * r2, tab);
  } else {
    const double r = sqrt(fmax(r2, R2_FLOOR));
    if constexpr (KID == 1) return kmat_exp(-r, tab);
    if constexpr (KID == 2) {
      const double t = 1.7320508075688772 * r;
      return (1.0 + t) * kmat_exp(-t, tab);
    }
    if constexpr (KID == 3) {
      const double t = 2.23606797749979 * r;
      return (1.0 + t + (5.0 / 3.0) * r * r) * kmat_exp(-t, tab);
    }
    return kmat_exp(-0.5 * r, tab);
  }
}

// g and h = 2 dg/d(r2) = g'(r)/r together (h == 0 where gpflow's max(r2, 1e-36) stops the gradient)
template <int KID>
__device__ __forceinline__ void corr_gh(double r2, double& g, double& h) {
  if constexpr (KID == 0) {
    g = exp_nonpos(-0.5 * r2);
    h = -g;
  } else {
    const bool live = r2 >= R2_FLOOR;
    const double r = sqrt(fmax(r2, R2_FLOOR));
    if constexpr (KID == 1) {
      g = exp_nonpos(-r);
      h = -g / r;
    } else if constexpr (KID == 2) {
      const double t = 1.7320508075688772 * r;
      const double e = exp_nonpos(-t);
      g = (1.0 + t) * e;
      h = -3.0 * e;
    } else if constexpr (KID == 3) {
      const double t = 2.23606797749979 * r;
      const double e = exp_nonpos(-t);
      g = (1.0 + t + (5.0 / 3.0) * r * r) * e;
      h = -(5.0 / 3.0) * (1.0 + t) * e;
    } else {
      g = exp_nonpos(-0.5 * r);
      h = -0.5 * g / r;
    }
    if (!live) h = 0.0;
  }
}

// gpflow's square_distance (utilities/ops.py): Xs = reduce_sum(square(X), -1) -- every square rounded, then added in k
// order --, dist = -2 X X2^T + Xs + X2s.  No fused multiply-add in the norms (a TensorFlow reduce_sum over rounded squares).
__device__ __forceinline__ void sqnorm_accumulate(double& acc, const double (&v)[8]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const double sq = v[kk] * v[kk];
    acc = acc + sq;
  }
}
__device__ __forceinline__ double expanded_r2(double na, double nb, double dot) {
#pragma clang fp contract(off)
  return (na + nb) - 2.0 * dot;
}

}  // namespace gprx
