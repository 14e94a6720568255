// Shared definitions for libgprx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gprx {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int NB = 64;  // Cholesky panel width, diagonal-block size, padding granule

constexpr int GEMM_C_LOWER = 1;
constexpr int GEMM_A_LOWER = 2;
constexpr int GEMM_A_UPPER = 4;
constexpr int GEMM_B_LOWER = 8;
constexpr int GEMM_B_UPPER = 16;

constexpr double R2_FLOOR = 1e-36;    // gpflow: r = sqrt(max(r2, 1e-36))
constexpr double NOISE_LOWER = 1e-6;  // gpflow Gaussian likelihood variance lower bound
constexpr double JITTER = 1e-6;       // gpflow default_jitter()

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// wave64 sum (all lanes receive the total)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// 1/sqrt(a) to full f64 precision: hardware estimate + two Newton steps
__device__ __forceinline__ double rsqrt_f64(double a) {
  double y = __builtin_amdgcn_rsq(a);
  double h = 0.5 * a;
  y = y * __builtin_fma(-h * y, y, 1.5);
  y = y * __builtin_fma(-h * y, y, 1.5);
  return y;
}

// exp(x) for x <= 0 (kernel arguments are -r2/2 or -c r): n = rint(x log2 e), r = x - n ln2 (two-term ln2),
// degree-13 Taylor polynomial on |r| <= 0.347 (truncation 4e-18), result scaled by 2^n with v_ldexp_f64.
// The coefficients live in constant memory: they are fetched with scalar loads and enter v_fma_f64 as SGPR
// operands, so the Horner chain needs no per-step constant moves into vector registers.
__constant__ double kExpCoef[16] = {1.6059043836821613e-10, 2.08767569878681e-09,  2.505210838544172e-08, 2.755731922398589e-07,
                                    2.7557319223985893e-06, 2.48015873015873e-05,  1.984126984126984e-04, 1.388888888888889e-03,
                                    8.333333333333333e-03,  4.1666666666666664e-02, 1.6666666666666666e-01, 0.5,
                                    1.0,                    1.0,                   -6.93147180369123816490e-01, -1.90821492927058770002e-10};

__device__ __forceinline__ double exp_nonpos(double x) {
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = __builtin_fma(n, kExpCoef[14], x);
  r = __builtin_fma(n, kExpCoef[15], r);
  double p = kExpCoef[0];
#pragma unroll
  for (int i = 1; i < 14; ++i) p = __builtin_fma(p, r, kExpCoef[i]);
  return __builtin_amdgcn_ldexp(p, (int)n);  // n >= -1075 here: underflows gracefully to 0
}

}  // namespace gprx
