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

}  // namespace gprx
