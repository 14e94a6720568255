// Small kernels of the sparse (SGPR, Titsias bound) path.  Everything O(M^2 N) runs on the MFMA
// GEMM / triangular-solve building blocks; what lives here is the O(M^2), O(N) glue.
#pragma once
#include "gprx_common.h"
#include "sgpr_small_ops.h"

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
  double w, gq;
  sgpr_combine(Qinv[e], Sinv[e], T[e], m[i], m[j], w, gq);
  W[e] = w;
  GQ[e] = gq;
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

// One workgroup per cell (mp <= 128), the values of add_diag_kernel + diag_sum_kernel + two copies in one launch:
// B += I, trace_out = sum_i (B[i][i] - 1) in diag_sum_kernel's order, Bfull = B, and the NB right-hand-side rows under B
// cleared (a failed factorisation leaves NaN there, which the next evaluation must not inherit).
__global__ __launch_bounds__(256) void sgpr_b_finish_kernel(double* __restrict__ B, int mp, double* __restrict__ trace_out,
                                                            double* __restrict__ Bfull, int64_t cs) {
  __shared__ double s[4];
  const int64_t off = (int64_t)blockIdx.x * cs;
  B += off;
  trace_out += off;
  Bfull += off;
  double a = 0.0;
  for (int i = threadIdx.x; i < mp; i += 256) {
    const double v = B[(int64_t)i * mp + i] + 1.0;
    B[(int64_t)i * mp + i] = v;
    a += v - 1.0;
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = a;
  __syncthreads();  // (also: the diagonal is final before the copy below reads it)
  if (threadIdx.x == 0) trace_out[0] = s[0] + s[1] + s[2] + s[3];
  for (int e = threadIdx.x; e < mp * mp; e += 256) Bfull[e] = B[e];
  double* crow = B + (int64_t)mp * mp;
  for (int e = threadIdx.x; e < NB * mp; e += 256) crow[e] = 0.0;
}

// One workgroup per cell, cols <= 64 and rows <= 64: the values of sumsq_partial_kernel (one row per workgroup) followed by
// sum_partials_kernel, in one launch: out[0] = sum over rows of (sum of squares of the row), same trees.
__global__ __launch_bounds__(256) void sumsq_small_kernel(const double* __restrict__ A, int64_t ld, int rows, int cols, double* __restrict__ out,
                                                          int64_t cs) {
  __shared__ double partial[64];
  A += (int64_t)blockIdx.x * cs;
  out += (int64_t)blockIdx.x * cs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < rows; r += 4) {
    double a = 0.0;
    if (lane < cols) {
      const double v = A[(int64_t)r * ld + lane];
      a = __builtin_fma(v, v, a);
    }
    a = wave_sum(a);
    if (lane == 0) partial[r] = a + 0.0 + 0.0 + 0.0;  // (the partial kernel added its three idle waves' zeros)
  }
  __syncthreads();
  if (wave == 0) {
    double a = 0.0;
    if (lane < rows) a += partial[lane];
    a = wave_sum(a);
    if (lane == 0) out[0] = a;
  }
}

// One launch for: logdet_quad_kernel, R = LB^-1 L^-1, Sinv = R^T R, T2 = B L^-1, T1 = L^-T T2, Qinv = L^-T L^-1 (five
// launch_gemm calls), m = L^-T LB^-T c (a copy and two trsv steps), sgpr_combine_kernel and the sum of squares of LB^-1 --
// eleven dependent launches of ~10 us each whose arithmetic is microseconds.  Every value is computed by the same operations
// in the same order as those kernels (the single-model path still runs them one by one; tests hold the two equal).
// LB: second factor (its diagonal), crow: c, invDL / invDB: L^-1 / LB^-1, Bfull: B.  red: [0] sum log diag LB, [1] |c|^2, [3] |LB^-1|_F^2
__global__ __launch_bounds__(256) void sgpr_small_kernel(const double* __restrict__ LB, const double* __restrict__ crow,
                                                         const double* __restrict__ invDL, const double* __restrict__ invDB,
                                                         const double* __restrict__ Bfull, double* __restrict__ red, double* __restrict__ mvec,
                                                         double* __restrict__ W, double* __restrict__ GQ, int64_t cs) {
  __shared__ double sL[NB * SM_LD];  // L^-1
  __shared__ double sB[NB * SM_LD];  // LB^-1, then B
  __shared__ double sC[NB * SM_LD];  // R, then T2
  __shared__ double sb[NB];
  __shared__ double part[4][NB];
  __shared__ double s0[4], s1[4], rowsq[NB];
  {
    const int64_t off = (int64_t)blockIdx.x * cs;
    LB += off; crow += off; invDL += off; invDB += off; Bfull += off; red += off; mvec += off; W += off; GQ += off;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, r = lane & 15;
  load64(invDL, sL, tid);
  load64(invDB, sB, tid);
  // logdet_quad_kernel(LB, 64, crow, 64)
  {
    double a = 0.0, q = 0.0;
    if (tid < NB) {
      a += log(LB[(int64_t)tid * NB + tid]);
      const double v = crow[tid];
      q = __builtin_fma(v, v, q);
      sb[tid] = v;
    }
    a = wave_sum(a);
    q = wave_sum(q);
    if (lane == 0) {
      s0[wave] = a;
      s1[wave] = q;
    }
  }
  __syncthreads();
  if (tid == 0) {
    red[0] = s0[0] + s0[1] + s0[2] + s0[3];
    red[1] = s1[0] + s1[1] + s1[2] + s1[3];
  }
  // sumsq_small_kernel(LB^-1): rows over the waves, then the 64 row sums
  for (int row = wave; row < NB; row += 4) {
    double a = 0.0;
    const double v = sB[row * SM_LD + lane];
    a = __builtin_fma(v, v, a);
    a = wave_sum(a);
    if (lane == 0) rowsq[row] = a + 0.0 + 0.0 + 0.0;
  }
  // m = L^-T (LB^-T c)
  trsv_t64(sB, sb, part, tid);  // (its barriers also publish rowsq)
  if (wave == 0) {
    double a = 0.0;
    a += rowsq[lane];
    a = wave_sum(a);
    if (lane == 0) red[3] = a;
  }
  trsv_t64(sL, sb, part, tid);
  if (tid < NB) mvec[tid] = sb[tid];
  d4 accR[2][2], accS[2][2], accT[2][2], accQ[2][2];
  mm64<false>(sB, sL, accR, wm, wn, g, r);  // R = LB^-1 L^-1
  mm64_store(accR, sC, wm, wn, g, r);
  __syncthreads();                           // R complete; LB^-1 no longer needed
  load64(Bfull, sB, tid);
  mm64<true>(sC, sC, accS, wm, wn, g, r);   // Sinv = R^T R
  __syncthreads();                           // B in place, all reads of R done
  mm64<false>(sB, sL, accR, wm, wn, g, r);  // T2 = B L^-1
  mm64_store(accR, sC, wm, wn, g, r);
  __syncthreads();
  mm64<true>(sL, sC, accT, wm, wn, g, r);   // T1 = L^-T T2
  mm64<true>(sL, sL, accQ, wm, wn, g, r);   // Qinv = L^-T L^-1
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = wm * 32 + a * 16 + g + 4 * q, col = wn * 32 + b * 16 + r;
        double w, gq;
        sgpr_combine(accQ[a][b][q], accS[a][b][q], accT[a][b][q], sb[row], sb[col], w, gq);
        W[row * NB + col] = w;
        GQ[row * NB + col] = gq;
      }
}

// Stage-in of one batched evaluation, straight from the pinned host block (device-visible): the cell's row of the parameter
// table, its Z, its y (unit from the host row: other workgroups of this launch may not have written the device table yet),
// and the cell's result block cleared.  Replaces two host-to-device copies, a memset and the gather kernel: copy nodes cost
// more than kernels in a replayed graph.  grid = (ceil(max(np, m d, par_doubles) / 256), cells)
__global__ void sgpr_stage_in_kernel(const double* __restrict__ Y, int np, const double* __restrict__ par_host, int par_doubles,
                                     double* __restrict__ cell_par, const double* __restrict__ z_host, int zn, double* __restrict__ z_dst,
                                     double* __restrict__ y_dst, int64_t cs, double* __restrict__ cell_res, int res_doubles) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t cell = blockIdx.y;
  if (i < res_doubles) cell_res[cell * res_doubles + i] = 0.0;
  if (i < par_doubles) cell_par[cell * par_doubles + i] = par_host[cell * par_doubles + i];
  if (i < zn) z_dst[cell * cs + i] = z_host[cell * zn + i];
  if (i < np) {
    const int unit = (int)par_host[cell * par_doubles + 2];
    y_dst[cell * cs + i] = Y[(int64_t)unit * np + i];
  }
}

// Stage-out: the cell's result block, its 8 reductions and (with the gradient) its 2 * width trace sums -- formed here from the
// contraction kernels' partials in trace_final's order: one wave per output, lanes over the partials -- and dZ, into the pinned
// host block (visible to the host once the stream has drained).  grid = (cells), 256 threads
__global__ __launch_bounds__(256) void sgpr_stage_out_kernel(const double* __restrict__ cell_res, int res_doubles, double* __restrict__ res_host,
                                                             const double* __restrict__ red, double* __restrict__ red_host,
                                                             const double* __restrict__ part_a, int nwg_a, const double* __restrict__ part_b,
                                                             int nwg_b, int width, double* __restrict__ sums_host,
                                                             const double* __restrict__ dz, int zn, double* __restrict__ dz_host, int64_t cs) {
  const int64_t cell = blockIdx.x;
  const int t = threadIdx.x;
  if (t < res_doubles) res_host[cell * res_doubles + t] = cell_res[cell * res_doubles + t];
  if (t < 8) red_host[cell * 8 + t] = red[cell * cs + t];
  if (part_a) {
    const int lane = t & 63;
    for (int o = t >> 6; o < 2 * width; o += 4) {
      const bool second = o >= width;
      const int e = second ? o - width : o;
      const double* partial = (second ? part_b : part_a) + cell * cs;
      const int nwg = second ? nwg_b : nwg_a;
      double s = 0.0;
      for (int w = lane; w < nwg; w += 64) s += partial[(int64_t)w * width + e];
      s = wave_sum(s);
      if (lane == 0) sums_host[cell * 2 * width + o] = s;
    }
  }
  if (dz)
    for (int e = t; e < zn; e += 256) dz_host[cell * zn + e] = dz[cell * cs + e];
}

}  // namespace gprx
