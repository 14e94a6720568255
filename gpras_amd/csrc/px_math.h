// Portable double-precision exp / log / softplus / sigmoid: the SAME source on the host and on the device, built from IEEE + - * /, fma,
// rint, frexp and ldexp only, every operation rounded once (no contraction), so that a value computed on the host and the same value
// computed inside a kernel agree bit for bit.  Why: the Adam loop of the sparse model runs resident on the device (sf_adam_kernel) and
// must reproduce the host-stepped loop -- the positive transforms (gpflow positive(): softplus, noise shifted by 1e-6), their
// derivatives and the LogNormal(0, 1) log-priors (gpr.py:298-305) enter every step.  libm (host) and ocml (device) differ in the last
// bit here and there; these do not.  Accuracy: exp <= 1 ulp-ish (degree-13 Taylor on |r| <= ln2 / 2 after a two-term reduction), log
// < 1 ulp (the classical atanh series with the seven published coefficients Lg1..Lg7 of Sun's fdlibm e_log.c, restated).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GPRX_HD __host__ __device__ inline
#else
#define GPRX_HD inline
#endif

namespace gprx {

constexpr double PX_LOG_2PI = 1.8378770664093453;  // log(2 pi)
constexpr double PX_LN2_HI = 6.93147180369123816490e-01, PX_LN2_LO = 1.90821492927058770002e-10;  // hi: 32 trailing zero bits, n * hi exact

GPRX_HD double px_exp(double x) {
#pragma clang fp contract(off)
  if (x > 709.0) x = 709.0;     // (callers stay far inside; overflow is not a case of this library)
  if (x < -745.0) return 0.0;
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = __builtin_fma(n, -PX_LN2_HI, x);
  r = __builtin_fma(n, -PX_LN2_LO, r);
  double p = 1.6059043836821613e-10;  // 1 / 13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);
  p = __builtin_fma(p, r, 2.505210838544172e-08);
  p = __builtin_fma(p, r, 2.755731922398589e-07);
  p = __builtin_fma(p, r, 2.7557319223985893e-06);
  p = __builtin_fma(p, r, 2.48015873015873e-05);
  p = __builtin_fma(p, r, 1.984126984126984e-04);
  p = __builtin_fma(p, r, 1.388888888888889e-03);
  p = __builtin_fma(p, r, 8.333333333333333e-03);
  p = __builtin_fma(p, r, 4.1666666666666664e-02);
  p = __builtin_fma(p, r, 1.6666666666666666e-01);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

// log(x), x > 0 and finite (normal or subnormal)
GPRX_HD double px_log(double x) {
#pragma clang fp contract(off)
  int e = 0;
  double m = frexp(x, &e);  // m in [0.5, 1)
  if (m < 0.70710678118654752440) {
    m = m * 2.0;
    e -= 1;
  }
  const double f = m - 1.0;  // in [sqrt(1/2) - 1, sqrt(2) - 1)
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  return dk * PX_LN2_HI - ((hfsq - (s * (hfsq + R) + dk * PX_LN2_LO)) - f);
}

// log(1 + u) for 0 <= u <= 1 (the arguments softplus produces): log of the rounded sum, corrected for the rounding of the sum
GPRX_HD double px_log1p01(double u) {
#pragma clang fp contract(off)
  const double y = 1.0 + u;
  return px_log(y) - ((y - 1.0) - u) / y;
}

// gpflow positive(): softplus(w) = log(1 + e^w), written so that neither branch overflows
GPRX_HD double px_softplus(double w) {
#pragma clang fp contract(off)
  return w > 0.0 ? w + px_log1p01(px_exp(-w)) : px_log1p01(px_exp(w));
}
// d softplus / dw
GPRX_HD double px_sigmoid(double w) {
#pragma clang fp contract(off)
  if (w >= 0.0) return 1.0 / (1.0 + px_exp(-w));
  const double e = px_exp(w);
  return e / (1.0 + e);
}
// log density of LogNormal(0, 1) at u > 0 and its derivative (the priors of gpr.py:303-305)
GPRX_HD double px_ln_logpdf(double u) {
#pragma clang fp contract(off)
  const double lu = px_log(u);
  return -lu - 0.5 * PX_LOG_2PI - 0.5 * lu * lu;
}
GPRX_HD double px_ln_dlogpdf(double u) {
#pragma clang fp contract(off)
  return -(1.0 + px_log(u)) / u;
}

}  // namespace gprx
