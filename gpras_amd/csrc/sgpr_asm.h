// The scalar tail of one SGPR evaluation -- ELBO from the device reductions, derivatives w.r.t. the constrained hyperparameters,
// LogNormal priors, the chain rule through softplus -- and one element of Keras's Adam update (gpr.py:147-173), written ONCE for the
// host (sgpr_objective_batch's tail, gprx_adam_batch's host-stepped loop) and the device (sf_adam_kernel, the resident loop): the same
// operations in the same order with one rounding each (px_math.h), so both give the same bits.
// Formulas: oracle/sgpr.py elbo_grads / loss_and_grad (gpflow SGPR.elbo, priors over trainable parameters only).
#pragma once
#include "px_math.h"

namespace gprx {

constexpr int ASM_TRAIN_VARIANCE = 1, ASM_TRAIN_LENGTHSCALE = 2, ASM_TRAIN_NOISE = 4, ASM_TRAIN_Z = 8;  // include/gprx.h GPRX_TRAIN_*

// red: [0] sum log diag LB, [1] |c|^2, [2] tr(A A^T), [3] |LB^-1|_F^2, [4] |y - P^T m|^2
GPRX_HD double sgpr_asm_elbo(double nn, double yy, double v, double s, const double* red) {
#pragma clang fp contract(off)
  return -0.5 * nn * PX_LOG_2PI - red[0] - 0.5 * nn * px_log(s) - 0.5 * (nn * v / s - red[2]) - 0.5 * (yy / s - red[1]);
}

// dELBO / d(constrained parameter k): k = 0 variance, 1 .. nlen lengthscales, 1 + nlen noise.  hs: [through Kuf (width) | through Kuu (width)],
// width = 2 + d: [0] sum G g, [2 + j] lengthscale sums (one shared lengthscale: the entries are added in j order).
GPRX_HD double sgpr_asm_dparam(int k, int nlen, int ard, int d, int width, double nn, int mp, double v, double s, const double* red, const double* hs) {
#pragma clang fp contract(off)
  if (k == 0) return -nn / (2.0 * s) + hs[0] + hs[width];
  if (k <= nlen) {
    if (ard) return hs[2 + (k - 1)] + hs[width + 2 + (k - 1)];
    double acc = 0.0;
    for (int j = 0; j < d; ++j) acc += hs[2 + j] + hs[width + 2 + j];
    return acc;
  }
  const double tr_sinv_pp = s * ((double)mp - red[3]);
  const double tr_qinv_pp = s * red[2];
  return (tr_sinv_pp - tr_qinv_pp + red[4] + nn * v) / (2.0 * s * s) - nn / (2.0 * s);
}

// d loss / d(unconstrained variable) of a positive parameter with value u = softplus(w) (+ the noise floor): priors only on trainable ones
GPRX_HD double sgpr_asm_chain(double delbo_du, double u, double w, bool trainable) {
#pragma clang fp contract(off)
  return trainable ? -(delbo_du + px_ln_dlogpdf(u)) * px_sigmoid(w) : 0.0;
}

// Keras Adam, one element: moments and variable updated in place.  alpha = lr sqrt(1 - beta2^t) / (1 - beta1^t) comes from the caller
// (the host's pow: a table on the device).
constexpr double ADAM_BETA1 = 0.9, ADAM_BETA2 = 0.999, ADAM_EPS = 1e-7, ADAM_LR = 1e-3, ADAM_TOL = 10e-6;
constexpr int ADAM_PATIENCE = 50;
GPRX_HD void adam_element(double ge, double alpha, double& mo, double& ve, double& x) {
#pragma clang fp contract(off)
  mo = ADAM_BETA1 * mo + (1.0 - ADAM_BETA1) * ge;
  ve = ADAM_BETA2 * ve + ((1.0 - ADAM_BETA2) * ge) * ge;
  x = x - (alpha * mo) / (sqrt(ve) + ADAM_EPS);
}
// the early-stop rule of gpr.py:160-171: returns whether the cell keeps running
GPRX_HD bool adam_keep_running(double loss, double& best, int& stale) {
#pragma clang fp contract(off)
  if (((best - loss) / fabs(loss)) > ADAM_TOL) {
    best = loss;
    stale = 0;
    return true;
  }
  return ++stale <= ADAM_PATIENCE;
}

}  // namespace gprx
