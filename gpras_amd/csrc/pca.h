// EOF (PCA) projection either side of the GP path -- row N1 of SURVEY.md section 8(f):
// PreProcessor.transform / reverse_transform / _linear_transform_for_var (gpras/preprocess.py:1009-1039, 1052-1094).
//
//   transform:  z[t, k] = ( sum_c ((g(x[t, c]) - mu_c) w_c) E[k, c]  -  xm_k ) / xs_k        over the wet cells c
//   reverse:    full[t, c] = (sum_k (mean[t, k] xs_k + xm_k) E[k, c]) / w_c + mu_c             (wet), fill_c (dry)
//               vfull[t, c] = sum_k var[t, k] (xs_k E[k, c] / w_c)^2                            (wet), 0 (dry)
//
// g = identity (wse, velocity) or max(x - elevation, 0) (depth).  All per-cell parameters are expanded to the full
// cell axis when the projector is created (dry cells: weight 0, E = 0), so no gather / scatter pass exists:
//   * transform = Z = Xc E^T with the fp64 MFMA GEMM, K cut into slices (split-K, slabs summed in a fixed order), the
//     centring / weighting applied to the operand on its way from memory to LDS (gemm_f64_kernel AXF; same operation
//     order per element as the reference: subtract, then weight) -- x is read exactly once -- then the
//     standardisation of the (t, k) result.  HBM-bound: 8 bytes per element of x for 2 k flops.
//   * reverse = one pass that writes every output element once (HBM-write bound): a thread owns one cell and keeps its
//     k EOF entries in registers, the rows' mode values come from LDS; mean and variance in the same pass.
#pragma once
#include "gprx_common.h"

namespace gprx {

// z[t, k] <- (z[t, k] - xm_k) / xs_k
__global__ __launch_bounds__(256) void pca_standardize_kernel(double* __restrict__ z, int64_t rows, int k, const double* __restrict__ xm,
                                                              const double* __restrict__ xs) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * k) return;
  const int kk = (int)(e % k);
  z[e] = (z[e] - xm[kk]) / xs[kk];
}

// Reverse projection.  Eb: (k, cells) EOFs expanded to all cells (0 on dry cells), leading dimension lde; w: weights (1 on dry cells; divided by,
// as the reference does); base: mu_c on wet cells, the fill value on dry cells.  One thread per cell, PCA_RB rows per workgroup pass.
constexpr int PCA_RB = 32;
template <int KMAX>
__global__ __launch_bounds__(256) void pca_reverse_kernel(const double* __restrict__ mean, const double* __restrict__ var, int64_t rows, int k,
                                                          int64_t cells, const double* __restrict__ Eb, int64_t lde, const double* __restrict__ w,
                                                          const double* __restrict__ base, const double* __restrict__ xm,
                                                          const double* __restrict__ xs, double* __restrict__ full,
                                                          double* __restrict__ vfull) {
  __shared__ double sM[PCA_RB][KMAX];
  __shared__ double sV[PCA_RB][KMAX];
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = c < cells;
  double e[KMAX], a2[KMAX];
  const double wc = live ? w[c] : 1.0, b = live ? base[c] : 0.0;
#pragma unroll
  for (int kk = 0; kk < KMAX; ++kk) {
    e[kk] = (live && kk < k) ? Eb[(int64_t)kk * lde + c] : 0.0;
    const double a = (kk < k) ? xs[kk] * e[kk] / wc : 0.0;  // _linear_transform_for_var: (diag(xs) E / w)^2
    a2[kk] = a * a;
  }
  for (int64_t t0 = (int64_t)blockIdx.y * PCA_RB; t0 < rows; t0 += (int64_t)gridDim.y * PCA_RB) {
    __syncthreads();
    for (int q = threadIdx.x; q < PCA_RB * KMAX; q += 256) {
      const int r = q / KMAX, kk = q % KMAX;
      const bool ok = t0 + r < rows && kk < k;
      sM[r][kk] = ok ? mean[(t0 + r) * k + kk] * xs[kk] + xm[kk] : 0.0;
      sV[r][kk] = (ok && var) ? var[(t0 + r) * k + kk] : 0.0;
    }
    __syncthreads();
    if (!live) continue;
    const int nr = (int)((rows - t0 < PCA_RB) ? rows - t0 : PCA_RB);
    for (int r = 0; r < nr; ++r) {
      double s = 0.0, v = 0.0;
#pragma unroll
      for (int kk = 0; kk < KMAX; ++kk) {
        s = __builtin_fma(sM[r][kk], e[kk], s);
        v = __builtin_fma(sV[r][kk], a2[kk], v);
      }
      full[(t0 + r) * cells + c] = s / wc + b;
      if (vfull) vfull[(t0 + r) * cells + c] = v;
    }
  }
}

// What production/analysis/pipeline.py:262-277 does to the reconstructed fields before the metrics, in place on the device:
//   mode 1 ("depth" models):  y += elevations;  then wse_2_depth: y = max(y - elevations, 0)   (literally: (y + e) - e)
//   mode 0 ("wse" models and the truth field):  y = max(y - elevations, 0)                     (PreProcessor.wse_2_depth, :1040-1044)
// and  conf = sqrt(var)  (pipeline.py:286).
__global__ __launch_bounds__(256) void field_to_depth_kernel(double* __restrict__ f, int64_t rows, int64_t cells, const double* __restrict__ elev,
                                                            int add_first) {
  const int64_t total = rows * cells;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const double el = elev[e % cells];
    double v = f[e];
    {
#pragma clang fp contract(off)
      if (add_first) v = v + el;
      v = v - el;
    }
    f[e] = v < 0.0 ? 0.0 : v;  // d[d < 0] = 0: a NaN stays a NaN
  }
}
__global__ __launch_bounds__(256) void field_sqrt_kernel(double* __restrict__ f, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) f[e] = sqrt(f[e]);
}
// dst (cols, rows) <- src (rows, cols)^T  (small: the (modes, points) block of a batched predict -> (points, modes))
__global__ __launch_bounds__(256) void transpose_small_kernel(const double* __restrict__ src, int64_t rows, int64_t cols, double* __restrict__ dst) {
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t c = e / rows, r = e - c * rows;  // consecutive threads: consecutive r -> coalesced writes of dst row c
    dst[c * rows + r] = src[r * cols + c];
  }
}

}  // namespace gprx
