// Triangular solves, inversion and reductions built on the 64 x 64 diagonal-block inverses
// that potrf_panel_kernel leaves behind.
#pragma once
#include "gemm_f64.h"
#include "gprx_common.h"

namespace gprx {

// ---- vector right-hand side ------------------------------------------------------------------
// One launch per diagonal block.  Step i of the forward solve L x = b:
//   every workgroup j > i:  b_j -= L[j, i] x_i   (64 rows x 64 cols, 16 lanes per row)
//   workgroup j == i + 1 then forms x_{i+1} = invD_{i+1} b_{i+1}.
// Step -1 (first launch) only forms x_0.  x overwrites b.
__global__ __launch_bounds__(256) void trsv_fwd_step(const double* __restrict__ L, int64_t lda, const double* __restrict__ inv_diag,
                                                     double* __restrict__ b, int i, int nblocks, int64_t cs) {
  __shared__ double sx[NB];
  __shared__ double sb[NB];
  L += (int64_t)blockIdx.y * cs;  // batched: blockIdx.y = cell, cs = cell stride (0 for a single system)
  inv_diag += (int64_t)blockIdx.y * cs;
  b += (int64_t)blockIdx.y * cs;
  const int t = threadIdx.x;
  const int j = i + 1 + blockIdx.x;  // block row handled here
  if (j >= nblocks) return;
  if (i >= 0) {
    if (t < NB) sx[t] = b[i * NB + t];
    __syncthreads();
    const int part = t & 15, rr = t >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = rr + 16 * pass;
      const double* lp = L + (int64_t)(j * NB + r) * lda + i * NB + part * 4;
      const d2 v0 = *reinterpret_cast<const d2*>(lp);
      const d2 v1 = *reinterpret_cast<const d2*>(lp + 2);
      double s = v0.x * sx[part * 4] + v0.y * sx[part * 4 + 1] + v1.x * sx[part * 4 + 2] + v1.y * sx[part * 4 + 3];
      s += __shfl_xor(s, 8, 64);
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 1, 64);
      if (part == 0) {
        const double nb_ = b[j * NB + r] - s;
        b[j * NB + r] = nb_;
        sb[r] = nb_;
      }
    }
  } else {
    if (t < NB) sb[t] = b[j * NB + t];
  }
  if (j != i + 1) return;
  __syncthreads();
  if (t < NB) {
    const double* ip = inv_diag + (int64_t)j * NB * NB + t * NB;
    double s = 0.0;
    for (int m = 0; m <= t; ++m) s = __builtin_fma(ip[m], sb[m], s);
    b[j * NB + t] = s;
  }
}

// Step i of the backward solve L^T x = b (i runs from nblocks-1 down):
//   every workgroup j < i:  b_j -= L[i, j]^T x_i   (64 x 64 block; 4 row groups of 16, all 16 loads of a
//                                                   thread in flight at once, then an LDS reduction)
//   workgroup j == i - 1 then forms x_{i-1} = invD_{i-1}^T b_{i-1}.
// Step i == nblocks only forms x_{nblocks-1}.
// x: where the solution blocks live -- b itself (in place, the default: xsep == nullptr) or a separate vector (trsv_bwd_pair_step below)
__global__ __launch_bounds__(256) void trsv_bwd_step(const double* __restrict__ L, int64_t lda, const double* __restrict__ inv_diag,
                                                     double* b, int i, int nblocks, int64_t cs, double* xsep = nullptr) {
  __shared__ double sx[NB];
  __shared__ double sb[NB];
  __shared__ double part[4][NB];
  L += (int64_t)blockIdx.y * cs;
  inv_diag += (int64_t)blockIdx.y * cs;
  b += (int64_t)blockIdx.y * cs;
  double* x = xsep ? xsep : b;
  const int tid = threadIdx.x;
  const int t = tid & 63, grp = tid >> 6;
  const int j = i - 1 - (int)blockIdx.x;
  if (j < 0) return;
  if (i < nblocks) {
    if (tid < NB) sx[tid] = x[i * NB + tid];
    const double* lp = L + (int64_t)(i * NB + grp * 16) * lda + j * NB + t;
    double lv[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) lv[m] = lp[(int64_t)m * lda];
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) s = __builtin_fma(lv[m], sx[grp * 16 + m], s);
    part[grp][t] = s;
    __syncthreads();
    if (tid < NB) {
      const double v = b[j * NB + tid] - (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]);
      b[j * NB + tid] = v;
      sb[tid] = v;
    }
  } else {
    if (tid < NB) sb[tid] = b[j * NB + tid];
  }
  if (j != i - 1) return;
  __syncthreads();
  // x_j[t] = sum_{m >= t} invD_j[m][t] b_j[m]; 4 groups split the rows m
  const double* ip = inv_diag + (int64_t)j * NB * NB;
  double s = 0.0;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int mm = grp * 16 + m;
    s = __builtin_fma(ip[mm * NB + t], sb[mm], s);  // invD is lower triangular with explicit zeros above
  }
  part[grp][t] = s;
  __syncthreads();
  if (tid < NB) x[j * NB + tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
}

// TWO block steps of the backward solve in one launch, for ONE system (a lone fit's alpha = L^-T beta is a chain of N / 64 dependent
// launches of ~4 us each: 0.26 ms of a 2.5 ms fit at N = 4096).  On entry x_i is final and w_j (j < i) holds beta_j minus what the blocks
// above i contribute; workgroup j (j = i - 2 ... 0) forms x_{i-1} = invD_{i-1}^T (w_{i-1} - L(i, i-1)^T x_i) ITSELF -- two extra 64 x 64
// tiles from L2 per workgroup instead of a launch boundary --, then w_j -= L(i, j)^T x_i, w_j -= L(i-1, j)^T x_{i-1}, and workgroup i - 2
// goes on to x_{i-2}.  Every sum is the one trsv_bwd_step forms, in its order: the solution is the same bit for bit (tested).  x and w
// are separate vectors (everybody reads w_{i-1} while x_{i-1} is being written).
__global__ __launch_bounds__(256) void trsv_bwd_pair_step(const double* __restrict__ L, int64_t lda, const double* __restrict__ inv_diag,
                                                          double* __restrict__ w, double* __restrict__ x, int i) {
  __shared__ double sx[NB];    // x_i
  __shared__ double sx1[NB];   // x_{i-1}
  __shared__ double sb[NB];
  __shared__ double part[4][NB];
  const int tid = threadIdx.x;
  const int t = tid & 63, grp = tid >> 6;
  const int j = i - 2 - (int)blockIdx.x;
  if (j < 0) return;
  // every load up front: the own tiles of block rows i and i - 1, and what x_{i-1} needs
  const double* lp_i = L + (int64_t)(i * NB + grp * 16) * lda + j * NB + t;
  const double* lp_1 = L + (int64_t)((i - 1) * NB + grp * 16) * lda + j * NB + t;
  const double* lp_d = L + (int64_t)(i * NB + grp * 16) * lda + (i - 1) * NB + t;
  const double* ip = inv_diag + (int64_t)(i - 1) * NB * NB;
  const double* ip2 = inv_diag + (int64_t)j * NB * NB;  // (used by workgroup i - 2 only; everybody loads: uniform code, L2-resident lines)
  double li[16], l1[16], ld_[16], iv[16], iv2[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    li[m] = lp_i[(int64_t)m * lda];
    l1[m] = lp_1[(int64_t)m * lda];
    ld_[m] = lp_d[(int64_t)m * lda];
    iv[m] = ip[(grp * 16 + m) * NB + t];
    iv2[m] = ip2[(grp * 16 + m) * NB + t];
  }
  const double w_i1 = w[(i - 1) * NB + t], w_j = w[j * NB + t];  // (every memory round trip of the launch is in flight now)
  if (tid < NB) sx[tid] = x[i * NB + tid];
  __syncthreads();
  // w_{i-1} - L(i, i-1)^T x_i  (what workgroup i - 1 of trsv_bwd_step(i) computes)
  double s = 0.0;
#pragma unroll
  for (int m = 0; m < 16; ++m) s = __builtin_fma(ld_[m], sx[grp * 16 + m], s);
  part[grp][t] = s;
  __syncthreads();
  if (tid < NB) sb[tid] = w_i1 - (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]);
  __syncthreads();
  // x_{i-1} = invD_{i-1}^T (that)
  s = 0.0;
#pragma unroll
  for (int m = 0; m < 16; ++m) s = __builtin_fma(iv[m], sb[grp * 16 + m], s);
  part[grp][t] = s;
  __syncthreads();
  if (tid < NB) {
    const double v = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
    sx1[tid] = v;
    if (blockIdx.x == 0) x[(i - 1) * NB + tid] = v;
  }
  __syncthreads();
  // own block: the two updates, one after the other (two roundings, as the two launches made them)
  s = 0.0;
#pragma unroll
  for (int m = 0; m < 16; ++m) s = __builtin_fma(li[m], sx[grp * 16 + m], s);
  part[grp][t] = s;
  __syncthreads();
  double v1 = 0.0;
  if (tid < NB) v1 = w_j - (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]);
  __syncthreads();
  s = 0.0;
#pragma unroll
  for (int m = 0; m < 16; ++m) s = __builtin_fma(l1[m], sx1[grp * 16 + m], s);
  part[grp][t] = s;
  __syncthreads();
  if (tid < NB) {
    const double v2 = v1 - (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]);
    w[j * NB + tid] = v2;
    sb[tid] = v2;
  }
  if (j != i - 2) return;
  __syncthreads();
  // x_{i-2} = invD_{i-2}^T w_{i-2}
  s = 0.0;
#pragma unroll
  for (int m = 0; m < 16; ++m) s = __builtin_fma(iv2[m], sb[grp * 16 + m], s);
  part[grp][t] = s;
  __syncthreads();
  if (tid < NB) x[j * NB + tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
}

// ---- alpha = L^-T beta from the explicit inverse X = L^-1 (the gradient has just built it): alpha_j = sum_{i >= j} X[i][j] beta_i.
// One pass over the lower tiles of X in two launches instead of the 64 dependent launches of the backward substitution (0.27 ms of
// a lone N = 4096 evaluation).  Deterministic: workgroup (column block, row chunk) sums its rows in a fixed order (thread group
// `grp` takes rows grp, grp + 4, ...; the four groups are added in order), alpha_final adds the chunks in order.
// grid = (np / 64, chunks, cells); part: chunks x np doubles per cell.
constexpr int ALPHA_CHUNK = 512;  // rows per workgroup
__global__ __launch_bounds__(256) void alpha_partial_kernel(const double* __restrict__ X, int64_t ldx, const double* __restrict__ beta,
                                                            double* __restrict__ part, int np, int64_t cs_x, int64_t cs_b, int64_t cs_p) {
  __shared__ double red[4][NB];
  X += (int64_t)blockIdx.z * cs_x;
  beta += (int64_t)blockIdx.z * cs_b;
  part += (int64_t)blockIdx.z * cs_p;
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int j0 = blockIdx.x * NB;
  int r0 = blockIdx.y * ALPHA_CHUNK;
  const int r1 = min(r0 + ALPHA_CHUNK, np);
  if (r0 < j0) r0 = j0;  // (tiles above the diagonal hold nothing; the diagonal block has explicit zeros above its diagonal)
  double s = 0.0;
  for (int i = r0 + grp; i < r1; i += 4) s = __builtin_fma(X[(int64_t)i * ldx + j0 + c], beta[i], s);
  red[grp][c] = s;
  __syncthreads();
  if (threadIdx.x < NB) part[(int64_t)blockIdx.y * np + j0 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
__global__ __launch_bounds__(256) void alpha_final_kernel(const double* __restrict__ part, int chunks, double* __restrict__ alpha, int np,
                                                          int64_t cs_p, int64_t cs_a) {
  part += (int64_t)blockIdx.y * cs_p;
  alpha += (int64_t)blockIdx.y * cs_a;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= np) return;
  double s = 0.0;
  for (int k = 0; k < chunks; ++k) s += part[(int64_t)k * np + j];
  alpha[j] = s;
}
// part: cells * ceil(np / ALPHA_CHUNK) * np doubles
inline hipError_t alpha_from_inverse(hipStream_t st, const double* X, int64_t ldx, const double* beta, double* part, double* alpha, int np,
                                     int cells = 1, int64_t cs_x = 0, int64_t cs_b = 0, int64_t cs_a = 0) {
  const int chunks = (np + ALPHA_CHUNK - 1) / ALPHA_CHUNK;
  const int64_t cs_p = (int64_t)chunks * np;
  hipLaunchKernelGGL(alpha_partial_kernel, dim3(np / NB, chunks, cells), dim3(256), 0, st, X, ldx, beta, part, np, cs_x, cs_b, cs_p);
  hipLaunchKernelGGL(alpha_final_kernel, dim3((np + 255) / 256, cells), dim3(256), 0, st, (const double*)part, chunks, alpha, np, cs_p, cs_a);
  return hipGetLastError();
}

// work (transpose, ONE system only): a vector that holds the right-hand side on entry and is used up; b then only receives the solution,
// and the block steps run two per launch (trsv_bwd_pair_step).  GPRX_TRSV_PAIR=0 falls back to one step per launch on the same two vectors.
inline hipError_t trsv_lower(hipStream_t st, const double* L, int64_t lda, const double* inv_diag, double* b, int np, bool transpose,
                             int batch = 1, int64_t cs = 0, double* work = nullptr) {
  const int nblocks = np / NB;
  if (transpose && work && batch == 1) {
    static const bool pair = !(getenv("GPRX_TRSV_PAIR") && atoi(getenv("GPRX_TRSV_PAIR")) == 0);
    hipLaunchKernelGGL(trsv_bwd_step, dim3(1, 1), dim3(256), 0, st, L, lda, inv_diag, work, nblocks, nblocks, (int64_t)0, b);
    int i = nblocks - 1;
    for (; pair && i >= 2; i -= 2) hipLaunchKernelGGL(trsv_bwd_pair_step, dim3(i - 1), dim3(256), 0, st, L, lda, inv_diag, work, b, i);
    for (; i >= 1; --i) hipLaunchKernelGGL(trsv_bwd_step, dim3(i, 1), dim3(256), 0, st, L, lda, inv_diag, work, i, nblocks, (int64_t)0, b);
    return hipGetLastError();
  }
  if (!transpose) {
    for (int i = -1; i < nblocks - 1; ++i)
      hipLaunchKernelGGL(trsv_fwd_step, dim3(i < 0 ? 1 : nblocks - 1 - i, batch), dim3(256), 0, st, L, lda, inv_diag, b, i, nblocks, cs);
  } else {
    for (int i = nblocks; i >= 1; --i)
      hipLaunchKernelGGL(trsv_bwd_step, dim3(i == nblocks ? 1 : i, batch), dim3(256), 0, st, L, lda, inv_diag, b, i, nblocks, cs);
  }
  return hipGetLastError();
}

// ---- matrix right-hand side: L X = B (in place), recursive halving onto MFMA GEMMs -------------
// cells > 1: the same solve for `cells` systems whose L, inv_diag and B are all cs doubles apart
// src (n == NB only): the right-hand side is read from there instead of B (same leading dimension), B only receives the
// solution -- saves the copy a caller would make to keep the right-hand side.
inline hipError_t trsm_lower_left(hipStream_t st, const double* L, int64_t lda, const double* inv_diag, double* B, int64_t ldb,
                                  int n, int ncols, int cells = 1, int64_t cs = 0, const double* src = nullptr) {
  if (n == NB)
    return launch_gemm(st, 0, 0, NB, ncols, NB, 1.0, inv_diag, NB, src ? src : B, ldb, 0.0, B, ldb, GEMM_A_LOWER, 64, 1, 0, 0, 0, cells, cs, cs, cs);
  const int n1 = (n / NB / 2) * NB, n2 = n - n1;
  hipError_t e = trsm_lower_left(st, L, lda, inv_diag, B, ldb, n1, ncols, cells, cs);
  if (e != hipSuccess) return e;
  e = launch_gemm(st, 0, 0, n2, ncols, n1, -1.0, L + (int64_t)n1 * lda, lda, B, ldb, 1.0, B + (int64_t)n1 * ldb, ldb, 0, 0, 1, 0, 0, 0, cells, cs,
                  cs, cs);
  if (e != hipSuccess) return e;
  return trsm_lower_left(st, L + (int64_t)n1 * lda + n1, lda, inv_diag + (int64_t)(n1 / NB) * NB * NB, B + (int64_t)n1 * ldb, ldb, n2,
                         ncols, cells, cs);
}

// L^T X = B (in place)
inline hipError_t trsm_lower_left_t(hipStream_t st, const double* L, int64_t lda, const double* inv_diag, double* B, int64_t ldb,
                                    int n, int ncols) {
  if (n == NB) return launch_gemm(st, 1, 0, NB, ncols, NB, 1.0, inv_diag, NB, B, ldb, 0.0, B, ldb, GEMM_A_UPPER, 64);
  const int n1 = (n / NB / 2) * NB, n2 = n - n1;
  hipError_t e = trsm_lower_left_t(st, L + (int64_t)n1 * lda + n1, lda, inv_diag + (int64_t)(n1 / NB) * NB * NB,
                                   B + (int64_t)n1 * ldb, ldb, n2, ncols);
  if (e != hipSuccess) return e;
  // B1 -= L21^T X2 : op(A) = L21^T with L21 stored (n2 x n1)
  e = launch_gemm(st, 1, 0, n1, ncols, n2, -1.0, L + (int64_t)n1 * lda, lda, B + (int64_t)n1 * ldb, ldb, 1.0, B, ldb, 0);
  if (e != hipSuccess) return e;
  return trsm_lower_left_t(st, L, lda, inv_diag, B, ldb, n1, ncols);
}

// ---- inverse of the Cholesky factor: X = L^-1 (lower), recursive, two GEMMs per node ------------
__global__ __launch_bounds__(256) void scatter_inv_diag(const double* __restrict__ inv_diag, double* __restrict__ X, int64_t ldx,
                                                        int64_t cs_in, int64_t cs_out) {
  const int blk = blockIdx.x;
  inv_diag += (int64_t)blockIdx.y * cs_in;  // batched: blockIdx.y = cell
  X += (int64_t)blockIdx.y * cs_out;
  const double* src = inv_diag + (int64_t)blk * NB * NB;
  double* dst = X + (int64_t)blk * NB * ldx + blk * NB;
  for (int e = threadIdx.x; e < NB * NB; e += 256) {
    const int r = e / NB, c = e % NB;
    dst[(int64_t)r * ldx + c] = src[e];
  }
}

// Bottom-up doubling: at level s (64, 128, ...) adjacent diagonal blocks X11, X22 of size s are already
// inverted; X21 = -X22 (L21 X11) for all pairs at once (batched GEMMs: two launches per level instead of
// two per tree node).  A ragged last pair (n not a multiple of 2 s) gets its own launches.
// cells > 1: the same inversion for `cells` factors; L and inv_diag are csL doubles apart, X and T csX doubles apart.
// tile: GEMM tile (0 = launch_gemm's choice).
inline hipError_t trtri_lower(hipStream_t st, const double* L, int64_t lda, const double* inv_diag, double* X, int64_t ldx, double* T,
                              int64_t ldt, int np, int cells = 1, int64_t csL = 0, int64_t csX = 0, int tile = 0) {
  hipLaunchKernelGGL(scatter_inv_diag, dim3(np / NB, cells), dim3(256), 0, st, inv_diag, X, ldx, csL, csX);
  for (int s = NB; s < np; s *= 2) {
    const int full = np / (2 * s);            // pairs with two complete blocks
    const int rem = np - full * 2 * s;        // leftover columns/rows after the complete pairs
    auto level = [&](int off, int n1, int n2, int batch) -> hipError_t {
      // blocks start at `off`; pair i covers [off + 2 s i, off + 2 s i + n1 + n2)
      const int64_t sa = (int64_t)2 * s * lda + 2 * s, sx = (int64_t)2 * s * ldx + 2 * s, stt = (int64_t)2 * s * ldt + 2 * s;
      const double* L21 = L + (int64_t)(off + n1) * lda + off;
      double* X11 = X + (int64_t)off * ldx + off;
      double* X22 = X + (int64_t)(off + n1) * ldx + (off + n1);
      double* X21 = X + (int64_t)(off + n1) * ldx + off;
      double* T21 = T + (int64_t)(off + n1) * ldt + off;
      hipError_t e = launch_gemm(st, 0, 0, n2, n1, n1, 1.0, L21, lda, X11, ldx, 0.0, T21, ldt, GEMM_B_LOWER, tile, batch, sa, sx, stt, cells, csL,
                                 csX, csX);
      if (e != hipSuccess) return e;
      return launch_gemm(st, 0, 0, n2, n1, n2, -1.0, X22, ldx, T21, ldt, 0.0, X21, ldx, GEMM_A_LOWER, tile, batch, sx, stt, sx, cells, csX, csX,
                         csX);
    };
    if (full > 0) {
      hipError_t e = level(0, s, s, full);
      if (e != hipSuccess) return e;
    }
    if (rem > s) {  // one more pair: a complete first block and a shorter second one
      hipError_t e = level(full * 2 * s, s, rem - s, 1);
      if (e != hipSuccess) return e;
    }
  }
  return hipGetLastError();
}

// ---- in-place transpose of a square matrix (n x n, n a multiple of 64) ----------------------------------------
// L^-1 (lower) -> L^-T (upper), so that K^-1 = L^-T L^-1 = Xt Xt^T runs as an NT product (both operands k-contiguous: the
// LDS-DMA kernel, 58-60 TFLOP/s) instead of the TN form (both operands m-contiguous, 41 TFLOP/s).  One workgroup per tile
// pair (ti >= tj): both 64 x 64 tiles go through one padded LDS image, 16-byte global accesses; 16 n^2 bytes of traffic.
__global__ __launch_bounds__(256) void transpose_inplace_kernel(double* __restrict__ X, int64_t ld, int tiles, int64_t cs) {
  __shared__ double sT[64][65];
  X += (int64_t)blockIdx.y * cs;  // batched: blockIdx.y = cell
  int bid = blockIdx.x;
  int ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= bid) ++ti;
  while (ti * (ti + 1) / 2 > bid) --ti;
  const int tj = bid - ti * (ti + 1) / 2;
  (void)tiles;
  double* A = X + (int64_t)ti * 64 * ld + tj * 64;  // tile (ti, tj)
  double* B = X + (int64_t)tj * 64 * ld + ti * 64;  // tile (tj, ti)
  const int tid = threadIdx.x;
  d2 ra[8], rb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = tid + 256 * i, row = q >> 5, cc = q & 31;
    ra[i] = *reinterpret_cast<const d2*>(A + (int64_t)row * ld + 2 * cc);
    rb[i] = *reinterpret_cast<const d2*>(B + (int64_t)row * ld + 2 * cc);
  }
  auto through_lds = [&](const d2 (&src)[8], double* dst) {  // dst tile <- transpose of the tile held in src
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = tid + 256 * i, row = q >> 5, cc = q & 31;
      sT[row][2 * cc] = src[i].x;
      sT[row][2 * cc + 1] = src[i].y;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = tid + 256 * i, row = q >> 5, cc = q & 31;
      *reinterpret_cast<d2*>(dst + (int64_t)row * ld + 2 * cc) = d2{sT[2 * cc][row], sT[2 * cc + 1][row]};
    }
  };
  through_lds(ra, B);
  if (ti != tj) through_lds(rb, A);
}

inline hipError_t transpose_inplace(hipStream_t st, double* X, int64_t ld, int n, int cells = 1, int64_t cs = 0) {
  const int tiles = n / 64;
  hipLaunchKernelGGL(transpose_inplace_kernel, dim3(tiles * (tiles + 1) / 2, cells), dim3(256), 0, st, X, ld, tiles, cs);
  return hipGetLastError();
}

// ---- reductions ----------------------------------------------------------------------------------
// out[0] = sum_i log L[i,i] (i < n), out[1] = sum_i v[i]^2 (i < n)
__global__ __launch_bounds__(256) void logdet_quad_kernel(const double* __restrict__ L, int64_t lda, const double* __restrict__ v, int n,
                                                          double* __restrict__ out, int64_t cs = 0, int out_stride = 0) {
  __shared__ double s0[4], s1[4];
  L += (int64_t)blockIdx.x * cs;  // batched: blockIdx.x = cell
  if (v) v += (int64_t)blockIdx.x * cs;
  out += (int64_t)blockIdx.x * out_stride;
  double a = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    a += log(L[(int64_t)i * lda + i]);
    if (v) q = __builtin_fma(v[i], v[i], q);
  }
  a = wave_sum(a);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) {
    s0[threadIdx.x >> 6] = a;
    s1[threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = s0[0] + s0[1] + s0[2] + s0[3];
    out[1] = s1[0] + s1[1] + s1[2] + s1[3];
  }
}

// partial[chunk][t] = sum over the chunk's rows of  w[row] * M[row, t]   (w != null)
//                                             or  M[row, t]^2            (w == null)
// (batched: blockIdx.z = cell; M, w and partial advance by their per-cell strides)
__global__ __launch_bounds__(256) void colreduce_partial(const double* __restrict__ M, int64_t ldm, const double* __restrict__ w, int nrows,
                                                         int ncols, int rows_per_chunk, double* __restrict__ partial, int64_t m_cell = 0,
                                                         int64_t w_cell = 0, int64_t p_cell = 0) {
  M += (int64_t)blockIdx.z * m_cell;
  if (w) w += (int64_t)blockIdx.z * w_cell;
  partial += (int64_t)blockIdx.z * p_cell;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ncols) return;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(nrows, r0 + rows_per_chunk);
  double s0 = 0.0, s1 = 0.0;
  int r = r0;
  if (w) {
    for (; r + 1 < r1; r += 2) {
      s0 = __builtin_fma(w[r], M[(int64_t)r * ldm + t], s0);
      s1 = __builtin_fma(w[r + 1], M[(int64_t)(r + 1) * ldm + t], s1);
    }
    if (r < r1) s0 = __builtin_fma(w[r], M[(int64_t)r * ldm + t], s0);
  } else {
    for (; r + 1 < r1; r += 2) {
      const double a = M[(int64_t)r * ldm + t], b = M[(int64_t)(r + 1) * ldm + t];
      s0 = __builtin_fma(a, a, s0);
      s1 = __builtin_fma(b, b, s1);
    }
    if (r < r1) {
      const double a = M[(int64_t)r * ldm + t];
      s0 = __builtin_fma(a, a, s0);
    }
  }
  partial[(int64_t)blockIdx.y * ncols + t] = s0 + s1;
}

// out[t] = (accumulate ? out[t] : base) + scale * sum_chunks partial[chunk][t]
// (batched: blockIdx.y = cell; base = base_tab[cell * base_stride] + (base_tab2 ? base_tab2[cell * base_stride] : 0) when given)
__global__ __launch_bounds__(256) void colreduce_final(const double* __restrict__ partial, int nchunks, int ncols, double base, double scale,
                                                       int accumulate, double* __restrict__ out, int64_t p_cell = 0, int64_t out_cell = 0,
                                                       const double* __restrict__ base_tab = nullptr, const double* __restrict__ base_tab2 = nullptr,
                                                       int base_stride = 0) {
  partial += (int64_t)blockIdx.y * p_cell;
  out += (int64_t)blockIdx.y * out_cell;
  if (base_tab) {
    base = base_tab[(int64_t)blockIdx.y * base_stride];
    if (base_tab2) base += base_tab2[(int64_t)blockIdx.y * base_stride];
  }
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ncols) return;
  double s = 0.0;
  for (int c = 0; c < nchunks; ++c) s += partial[(int64_t)c * ncols + t];
  out[t] = (accumulate ? out[t] : base) + scale * s;
}

// out[row] = base - sum over the `nparts` slabs of partial[part * ld + row]  (the row sums of squares a GEMM launch with
// GemmArgs::rowsq left behind), fixed order
__global__ __launch_bounds__(256) void rowsq_final_kernel(const double* __restrict__ partial, int nparts, int64_t ld, int nrows, double base,
                                                          double* __restrict__ out) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= nrows) return;
  double s = 0.0;
  for (int p = 0; p < nparts; ++p) s += partial[(int64_t)p * ld + row];
  out[row] = base - s;
}

// out[row] = base + scale * sum_c (w ? w[c] * M[row, c] : M[row, c]^2): one wave per row, 16-byte loads along the row
// (ncols a multiple of 2, rows 16-byte aligned).  Used by the transposed predict (rows = test points).
__global__ __launch_bounds__(256) void rowreduce_kernel(const double* __restrict__ M, int64_t ldm, const double* __restrict__ w, int nrows,
                                                        int ncols, double base, double scale, double* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int lane = threadIdx.x & 63;
  const double* mp = M + (int64_t)row * ldm;
  double s0 = 0.0, s1 = 0.0;
  if (w) {
    for (int c = 2 * lane; c < ncols; c += 128) {
      const d2 v = *reinterpret_cast<const d2*>(mp + c);
      const d2 ww = *reinterpret_cast<const d2*>(w + c);
      s0 = __builtin_fma(v.x, ww.x, s0);
      s1 = __builtin_fma(v.y, ww.y, s1);
    }
  } else {
    for (int c = 2 * lane; c < ncols; c += 128) {
      const d2 v = *reinterpret_cast<const d2*>(mp + c);
      s0 = __builtin_fma(v.x, v.x, s0);
      s1 = __builtin_fma(v.y, v.y, s1);
    }
  }
  const double s = wave_sum(s0 + s1);
  if (lane == 0) out[row] = base + scale * s;
}

}  // namespace gprx
