// Shared definitions for libgprx (gfx950 only).
#pragma once
#include <type_traits>
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

// Sum over the 64 lanes of a wave by DPP moves (row shifts inside the rows of 16 lanes, then the two row broadcasts): ~20 vector
// instructions with no LDS traffic; gprx_common.h wave_sum is 12 ds_bpermute round trips (~700 clocks when nothing hides them).  The total
// is formed in lane 63 and read back with v_readlane: every lane of the wave receives it.  Other order of additions than wave_sum.
__device__ __forceinline__ double wave_sum_dpp(double v) {
  auto shifted = [](double x, auto ctrl, auto row_mask) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false);
    return __hiloint2double(hi, lo);
  };
  using std::integral_constant;
  v += shifted(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});  // row_shr:1
  v += shifted(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});  // row_shr:2
  v += shifted(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});  // row_shr:4
  v += shifted(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});  // row_shr:8  -> lane 15 of every row: the row's sum
  v += shifted(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1 and 3
  v += shifted(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});  // row_bcast:31 into rows 2 and 3 -> lane 63: the total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
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
static __constant__ double kExpCoef[16] = {1.6059043836821613e-10, 2.08767569878681e-09,  2.505210838544172e-08, 2.755731922398589e-07,
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

// exp(x) for x <= 0 with a table: n = rint(x 64 / ln2), r = x - n ln2 / 64 (two-term constant, |r| <= ln2 / 128 = 0.0054), j = n mod 64,
// m = floor(n / 64):  exp(x) = 2^m * T[j] * e^r,  T[j] = 2^(j/64) correctly rounded,  e^r - 1 = r + r^2 (1/2 + r/6 + r^2/24 + r^3/120)
// (truncation r^6 / 720 <= 3.5e-17), combined as fma(T, e^r - 1, T): one rounding of T (half an ulp) and the final one.  13 fp64-rate
// instructions instead of the 19 of exp_nonpos (the kernel build is bound by its fp64 VALU work next to its stores);
// the 64-entry table sits in LDS (512 bytes = every bank pair exactly once: a wave's 64 different indices read without conflicts).
// Used by the kernel-matrix build only; the gradient passes keep exp_nonpos.  Measured against numpy's exp on 1e7 arguments in
// tests/test_gpu_blocks.py (<= 1 ulp).
static __constant__ double kExp2Tab[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0
};
constexpr double EXP_TAB_SCALE = 0x1.71547652b82fep+6;   // 64 / ln 2
constexpr double EXP_TAB_C_HI = -0x1.62e42fef00000p-7;   // -(ln 2 / 64), 33 significant bits: n * C_HI is exact for |n| < 2^20
constexpr double EXP_TAB_C_LO = -0x1.473de6af278edp-40;  // -(ln 2 / 64 - hi)

__device__ __forceinline__ void exp_tab_fill(double* __restrict__ tab) {
  if (threadIdx.x < 64) tab[threadIdx.x] = kExp2Tab[threadIdx.x];
}

__device__ __forceinline__ double exp_nonpos_tab(double x, const double* __restrict__ tab) {
  x = x < -750.0 ? -750.0 : x;  // exp underflows to 0 below -745.2: n stays inside the exact range of the reduction for ANY x <= 0, -inf
                                // included; a compare-select, not v_max_f64, so that a NaN argument stays a NaN
  const double n = __builtin_rint(x * EXP_TAB_SCALE);
  double r = __builtin_fma(n, EXP_TAB_C_HI, x);
  r = __builtin_fma(n, EXP_TAB_C_LO, r);
  double q = __builtin_fma(1.0 / 120.0, r, 1.0 / 24.0);
  q = __builtin_fma(q, r, 1.0 / 6.0);
  q = __builtin_fma(q, r, 0.5);
  const double p = __builtin_fma(r * r, q, r);  // e^r - 1
  const int ni = (int)n;
  const double t = tab[ni & 63];
  return __builtin_amdgcn_ldexp(__builtin_fma(t, p, t), ni >> 6);  // (x >= -745.2 gives ni >= -68800: underflows gracefully to 0 below)
}

}  // namespace gprx
