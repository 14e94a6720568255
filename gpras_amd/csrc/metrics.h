// Fused error metrics over two fields x (truth) and y (prediction) and a confidence field, each (T, cells) row-major --
// row N3 of SURVEY.md section 8(f): all reductions of gpras/metrics.py:85-318 from two streaming passes over the data.
//
// Two kernels, each streaming the fields once with the mapping that suits its reduction:
//   metrics_cells_kernel: a thread owns one cell (column) and walks down the timesteps (loads coalesced across cells):
//     per cell sum e, sum e^2, sum conf (e = x - y), the first maximum of x and of y over time with its timestep (numpy
//     argmax semantics: first occurrence; NaN counts as a maximum as in numpy), and the fidelity matches
//     matching[t] = |y_t - x_t| <= v_tol, OR over lags 1..t_tol of |y_t - x_{t+i}| <= v_tol and |x_t - y_{t+i}| <= v_tol
//     while t + i < T (metrics.py:194-204), with a register window of the next rows;
//   metrics_rows_kernel: a workgroup owns one timestep (row) and sums over the cells: sum e, sum e^2, sum conf, sum |e|.
// (A single kernel doing both needed four cross-lane reductions and two barriers per row and ran at 1.2 TB/s.)
// HBM-bound: 2 x 24 bytes per element (2 x 16 without confidence).
#pragma once
#include "gprx_common.h"

namespace gprx {

constexpr int MET_TMAX = 8;  // largest supported t_tol

struct MetricsArgs {
  const double* x;
  const double* y;
  const double* conf;  // may be null
  int64_t rows, cells;
  double v_tol;
  int t_tol;
  // per cell outputs (cells each)
  double* c_sum_e;
  double* c_sum_e2;
  double* c_sum_conf;
  double* c_xpeak;
  double* c_ypeak;
  int* c_xarg;
  int* c_yarg;
  unsigned long long* match_partial;  // one count per workgroup
};

__global__ __launch_bounds__(256) void metrics_cells_kernel(MetricsArgs p) {
  __shared__ unsigned long long smatch[4];
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = c < p.cells;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long nmatch = 0;
  if (live) {
    double se = 0.0, se2 = 0.0, sc = 0.0, xpk = 0.0, ypk = 0.0;
    int xarg = 0, yarg = 0;
    double wx[MET_TMAX + 1], wy[MET_TMAX + 1];  // rows t .. t + t_tol
#pragma unroll
    for (int i = 0; i <= MET_TMAX; ++i) {
      const bool ok = i <= p.t_tol && i < p.rows;
      wx[i] = ok ? p.x[(int64_t)i * p.cells + c] : 0.0;
      wy[i] = ok ? p.y[(int64_t)i * p.cells + c] : 0.0;
    }
    double cnext = p.conf ? p.conf[c] : 0.0;
    for (int64_t t = 0; t < p.rows; ++t) {
      // next row of the window and of conf: issued before this row's arithmetic
      const int64_t tn = t + 1 + p.t_tol;
      double nx = 0.0, ny = 0.0, cn = 0.0;
      if (tn < p.rows) {
        nx = p.x[tn * p.cells + c];
        ny = p.y[tn * p.cells + c];
      }
      if (p.conf && t + 1 < p.rows) cn = p.conf[(t + 1) * p.cells + c];
      const double xv = wx[0], yv = wy[0];
      const double e = xv - yv;
      se += e;
      se2 = __builtin_fma(e, e, se2);
      sc += cnext;
      // first maximum; a NaN is taken as the maximum and sticks, as numpy's argmax does
      if (t == 0 || (!(xpk != xpk) && (xv > xpk || xv != xv))) {
        xpk = xv;
        xarg = (int)t;
      }
      if (t == 0 || (!(ypk != ypk) && (yv > ypk || yv != yv))) {
        ypk = yv;
        yarg = (int)t;
      }
      bool m = fabs(yv - xv) <= p.v_tol;
#pragma unroll
      for (int i = 1; i <= MET_TMAX; ++i)
        if (i <= p.t_tol && t + i < p.rows) m = m || (fabs(yv - wx[i]) <= p.v_tol) || (fabs(xv - wy[i]) <= p.v_tol);
      nmatch += m ? 1ull : 0ull;
#pragma unroll
      for (int i = 0; i < MET_TMAX; ++i) {
        wx[i] = wx[i + 1];
        wy[i] = wy[i + 1];
      }
#pragma unroll
      for (int i = 0; i <= MET_TMAX; ++i)
        if (i == p.t_tol) {
          wx[i] = nx;
          wy[i] = ny;
        }
      cnext = cn;
    }
    p.c_sum_e[c] = se;
    p.c_sum_e2[c] = se2;
    p.c_sum_conf[c] = sc;
    p.c_xpeak[c] = xpk;
    p.c_ypeak[c] = ypk;
    p.c_xarg[c] = xarg;
    p.c_yarg[c] = yarg;
  }
  unsigned long long mm = nmatch;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mm += __shfl_xor(mm, off, 64);
  if (lane == 0) smatch[wave] = mm;
  __syncthreads();
  if (threadIdx.x == 0) p.match_partial[blockIdx.x] = smatch[0] + smatch[1] + smatch[2] + smatch[3];
}

// one workgroup per timestep: row_out[t] = {sum e, sum e^2, sum conf, sum |e|} over the cells (fixed reduction tree)
__global__ __launch_bounds__(256) void metrics_rows_kernel(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ conf,
                                                           int64_t cells, double* __restrict__ row_out) {
  __shared__ double sred[4][4];
  const int64_t t = blockIdx.x;
  const double* xr = x + t * cells;
  const double* yr = y + t * cells;
  const double* cr = conf ? conf + t * cells : nullptr;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int64_t c = threadIdx.x; c < cells; c += 256) {
    const double e = xr[c] - yr[c];
    s0 += e;
    s1 = __builtin_fma(e, e, s1);
    s3 += fabs(e);
    if (cr) s2 += cr[c];
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  s3 = wave_sum(s3);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sred[wave][0] = s0;
    sred[wave][1] = s1;
    sred[wave][2] = s2;
    sred[wave][3] = s3;
  }
  __syncthreads();
  if (threadIdx.x < 4) row_out[t * 4 + threadIdx.x] = sred[0][threadIdx.x] + sred[1][threadIdx.x] + sred[2][threadIdx.x] + sred[3][threadIdx.x];
}

// out[c] = field[idx[c], c]: the cached-argmax gathers x[x_mts, np.arange(cells)] of gpras/metrics.py:119-121 (and
// every *_mts function after it) for CALLER-SUPPLIED timesteps; idx already wrapped to [0, rows) on the host.
__global__ __launch_bounds__(256) void gather_rows_kernel(const double* __restrict__ field, int64_t cells, const int64_t* __restrict__ idx,
                                                          double* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c < cells) out[c] = field[idx[c] * cells + c];
}

}  // namespace gprx
