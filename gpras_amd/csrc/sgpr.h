// Small kernels of the sparse (SGPR, Titsias bound) path.  Everything O(M^2 N) runs on the MFMA
// GEMM / triangular-solve building blocks; what lives here is the O(M^2), O(N) glue.
#pragma once
#include "gprx_common.h"

namespace gprx {

// B[i][i] += value for i < n
// (every kernel here: blockIdx.y = cell, per-cell pointers advance by cs doubles; cs = 0 and gridDim.y = 1 for one model)
__global__ void add_diag_kernel(double* B, int64_t ld, int n, double value, int64_t cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  B += (int64_t)blockIdx.y * cs;
  if (i < n) B[(int64_t)i * ld + i] += value;
}

// out[0] = sum_{i<n} (B[i][i] - minus)
__global__ __launch_bounds__(256) void diag_sum_kernel(const double* B, int64_t ld, int n, double minus, double* out, int64_t cs) {
  __shared__ double s[4];
  B += (int64_t)blockIdx.y * cs;
  out += (int64_t)blockIdx.y * cs;
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += B[(int64_t)i * ld + i] - minus;
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = s[0] + s[1] + s[2] + s[3];
}

// partial[b] = sum of squares of the rows handled by workgroup b (row-strided); a second launch with one
// workgroup adds the partials in a fixed order (deterministic)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const double* A, int64_t ld, int rows, int cols, double* partial, int64_t cs) {
  __shared__ double s[4];
  A += (int64_t)blockIdx.y * cs;
  partial += (int64_t)blockIdx.y * cs;
  double a = 0.0;
  for (int r = blockIdx.x; r < rows; r += gridDim.x)
    for (int c = threadIdx.x; c < cols; c += 256) {
      const double v = A[(int64_t)r * ld + c];
      a = __builtin_fma(v, v, a);
    }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(64) void sum_partials_kernel(const double* partial, int n, double* out, int64_t cs) {
  partial += (int64_t)blockIdx.y * cs;
  out += (int64_t)blockIdx.y * cs;
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) a += partial[i];
  a = wave_sum(a);
  if (threadIdx.x == 0) out[0] = a;
}

// out[0] = sum_{i<n} (y[i] - q[i])^2
__global__ __launch_bounds__(256) void resid_sumsq_kernel(const double* y, const double* q, int n, double* out, int64_t ys, int64_t cs) {
  __shared__ double s[4];
  y += (int64_t)blockIdx.y * ys;
  q += (int64_t)blockIdx.y * cs;
  out += (int64_t)blockIdx.y * cs;
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double r = y[i] - q[i];
    a = __builtin_fma(r, r, a);
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = s[0] + s[1] + s[2] + s[3];
}

// W  = Qinv - Sinv - m m^T                         (weights of dELBO/dKuf, before the 1/s)
// GQ = (2 Qinv - Sinv - T - m m^T) / 2            (dELBO/dKuu;  T = Linv^T B Linv)
__global__ void sgpr_combine_kernel(const double* Qinv, const double* Sinv, const double* T, const double* m, int mp, double* W,
                                    double* GQ, int64_t cs) {
  {
    const int64_t off = (int64_t)blockIdx.y * cs;
    Qinv += off;
    Sinv += off;
    T += off;
    m += off;
    W += off;
    GQ += off;
  }
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= mp * mp) return;
  const int i = e / mp, j = e % mp;
  const double mm = m[i] * m[j];
  const double q = Qinv[e], s = Sinv[e];
  W[e] = q - s - mm;
  GQ[e] = 0.5 * (2.0 * q - s - T[e] - mm);
}

__global__ void copy_matrix_kernel(const double* src, int64_t lds, double* dst, int64_t ldd, int rows, int cols) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= (int64_t)rows * cols) return;
  const int r = (int)(e / cols), c = (int)(e % cols);
  dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds + c];
}

__global__ void set_identity_kernel(double* dst, int64_t ld, int n) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * n) return;
  const int r = (int)(e / n), c = (int)(e % n);
  dst[(int64_t)r * ld + c] = r == c ? 1.0 : 0.0;
}

// y of each cell's unit into its cell block: dst[c] + i = Y[unit_c][i], unit_c from the cell-parameter table
__global__ void gather_y_kernel(const double* __restrict__ Y, int np, const double* __restrict__ cell_par, int par_stride, double* __restrict__ dst,
                                int64_t cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  const int unit = (int)cell_par[(int64_t)blockIdx.y * par_stride + 2];
  dst[(int64_t)blockIdx.y * cs + i] = Y[(int64_t)unit * np + i];
}

}  // namespace gprx
