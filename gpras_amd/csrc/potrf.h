// Blocked right-looking Cholesky (lower, in place, row-major) for gfx950.
//
// Per 64-column panel two launches:
//   1. potrf_panel_kernel  -- factor the 64 x 64 diagonal block AND solve every row below it
//      (L21 = A21 L11^-T, including the right-hand-side rows appended under the matrix) in one
//      launch.  A workgroup keeps 256 panel rows in LDS (stride 65: row-per-lane access is
//      conflict-free): the 64 rows of the diagonal block -- every workgroup re-factors it
//      redundantly, so no inter-workgroup hand-off exists -- plus 192 rows of A21.  Work proceeds
//      in 8-column sub-panels: each thread factors the 8 x 8 diagonal sub-block (register math)
//      and solves its own row, then the columns right of the sub-panel are updated on MFMA
//      (rows x 8 times (64 x 8)^T); two barriers per sub-panel.  One extra workgroup carries the
//      64 identity rows, which come out as L11^-1 (inverse of the diagonal block, used by the
//      triangular solves).
//   2. gemm_f64 (NT, C_LOWER, alpha = -1, beta = 1) -- trailing update A22 -= L21 L21^T on MFMA.
#pragma once
#include <utility>
#include <vector>

#include "gemm_f64.h"
#include "gprx_common.h"

namespace gprx {

constexpr int PANEL_ROWS = 192;  // rows of A21 per workgroup (LDS rows 64..255)
constexpr int PLD = 65;          // LDS row stride (doubles): lane t -> bank 2t, conflict-free row-per-lane b64 access

// A points at the diagonal block (c, c).  rows_below = rows under the block to solve.
// LDS image: sRow[256][65]: rows 0..63 = diagonal block (every workgroup factors it redundantly),
// rows 64..255 = this workgroup's 192 rows of A21 (or, in the last workgroup, the 64 identity rows).
__global__ __launch_bounds__(256) void potrf_panel_kernel(double* __restrict__ A, int64_t lda, int rows_below, int nchunks,
                                                          double* __restrict__ inv_diag, int* __restrict__ info, int col0) {
  __shared__ __attribute__((aligned(16))) double sRow[256 * PLD];
  __shared__ __attribute__((aligned(16))) double sD[8][8];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const bool last = (int)blockIdx.x == nchunks;

  // ---- coalesced load: 256 rows x 32 chunks of 16 B ----
#pragma unroll 4
  for (int i = 0; i < 32; ++i) {
    const int q = tid + 256 * i;
    const int row = q >> 5, cc = q & 31;
    d2 v = d2{0.0, 0.0};
    if (row < NB) {
      v = *reinterpret_cast<const d2*>(A + (int64_t)row * lda + 2 * cc);
      if (2 * cc > row) v.x = 0.0;
      if (2 * cc + 1 > row) v.y = 0.0;
    } else if (!last) {
      const int idx = blockIdx.x * PANEL_ROWS + (row - NB);
      if (idx < rows_below) v = *reinterpret_cast<const d2*>(A + (int64_t)(NB + idx) * lda + 2 * cc);
    } else {
      const int e = row - NB;  // identity rows (only the first 64 are meaningful)
      if (2 * cc == e) v.x = 1.0;
      if (2 * cc + 1 == e) v.y = 1.0;
    }
    sRow[row * PLD + 2 * cc] = v.x;
    sRow[row * PLD + 2 * cc + 1] = v.y;
  }
  __syncthreads();

  double* myrow = sRow + tid * PLD;
  const int zero_above = tid < NB ? tid : (1 << 30);  // diagonal-block rows: entries right of the diagonal are zero
  int bad = 0;

  for (int p = 0; p < 8; ++p) {
    const int C0 = 8 * p;
    if (tid >= C0 && tid < C0 + 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) sD[tid - C0][k] = myrow[C0 + k];
    }
    __syncthreads();
    // every thread factors the 8 x 8 diagonal sub-block
    double l[8][8], rinv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int k = 0; k <= j; ++k) l[j][k] = sD[j][k];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      double s = l[j][j];
#pragma unroll
      for (int m = 0; m < j; ++m) s = __builtin_fma(-l[j][m], l[j][m], s);
      if (!(s > 0.0)) {
        if (bad == 0) bad = C0 + j + 1;
        s = 1.0;
      }
      const double ri = rsqrt_f64(s);
      rinv[j] = ri;
      l[j][j] = s * ri;
#pragma unroll
      for (int i = j + 1; i < 8; ++i) {
        double t = l[i][j];
#pragma unroll
        for (int m = 0; m < j; ++m) t = __builtin_fma(-l[i][m], l[j][m], t);
        l[i][j] = t * ri;
      }
    }
    // own row: x = a[C0 .. C0+7] L_dd^-T
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double t = myrow[C0 + k];
#pragma unroll
      for (int m = 0; m < k; ++m) t = __builtin_fma(-x[m], l[k][m], t);
      x[k] = (C0 + k > zero_above) ? 0.0 : t * rinv[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) myrow[C0 + k] = x[k];
    __syncthreads();
    // trailing columns [C0 + 8, 64): rows(256 x 8) * Lpanel(64 x 8)^T on MFMA.  Wave w owns rows 64w .. 64w+63.
    if (C0 + 8 < NB) {
      double fa[4][2];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[rt][ks] = -sRow[(wave * 64 + rt * 16 + r) * PLD + C0 + 4 * ks + g];
      for (int kt = (C0 + 8) >> 4; kt < 4; ++kt) {
        const int kk = kt * 16 + r;
        double fb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb[ks] = (kk >= C0 + 8) ? sRow[kk * PLD + C0 + 4 * ks + g] : 0.0;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          double* cp = sRow + (wave * 64 + rt * 16 + g) * PLD + kt * 16 + r;
          d4 c = d4{cp[0], cp[4 * PLD], cp[8 * PLD], cp[12 * PLD]};
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][0], fb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][1], fb[1], c, 0, 0, 0);
          cp[0] = c[0];
          cp[4 * PLD] = c[1];
          cp[8 * PLD] = c[2];
          cp[12 * PLD] = c[3];
        }
      }
    }
  }
  __syncthreads();

  // ---- coalesced store ----
  if (!last) {
#pragma unroll 4
    for (int i = 8; i < 32; ++i) {  // rows 64..255
      const int q = tid + 256 * i;
      const int row = q >> 5, cc = q & 31;
      const int idx = blockIdx.x * PANEL_ROWS + (row - NB);
      if (idx < rows_below)
        *reinterpret_cast<d2*>(A + (int64_t)(NB + idx) * lda + 2 * cc) = d2{sRow[row * PLD + 2 * cc], sRow[row * PLD + 2 * cc + 1]};
    }
  } else {
#pragma unroll 4
    for (int i = 0; i < 8; ++i) {  // the factored diagonal block, zeros right of the diagonal
      const int q = tid + 256 * i;
      const int row = q >> 5, cc = q & 31;
      *reinterpret_cast<d2*>(A + (int64_t)row * lda + 2 * cc) = d2{sRow[row * PLD + 2 * cc], sRow[row * PLD + 2 * cc + 1]};
    }
    // identity row i came out as column i of L11^-1
    for (int e = tid; e < NB * NB; e += 256) {
      const int kk = e >> 6, i = e & 63;
      inv_diag[e] = sRow[(NB + i) * PLD + kk];
    }
    if (tid == 0 && bad != 0) atomicCAS(info, 0, col0 + bad);
  }
}

// Optional per-launch timing of the two kernels of the factorisation (HIP events on the launch stream).
struct PotrfProfile {
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  std::vector<std::pair<size_t, double>> gemm_marks;  // (index of start event, algorithmic flops of the launch)
  std::vector<size_t> panel_marks;
  hipEvent_t next() {
    if (used == pool.size()) {
      hipEvent_t e;
      hipEventCreate(&e);
      pool.push_back(e);
    }
    return pool[used++];
  }
  void reset() {
    used = 0;
    gemm_marks.clear();
    panel_marks.clear();
  }
  ~PotrfProfile() {
    for (auto e : pool) hipEventDestroy(e);
  }
};

// Factor the (np x np) matrix in place; `extra` rows below it are carried as right-hand sides.
// inv_diag: np/64 blocks of 64 x 64.  info (device int) must be zeroed by the caller.
inline hipError_t potrf_lower(hipStream_t st, double* A, int64_t lda, int np, int extra, double* inv_diag, int* info,
                              PotrfProfile* prof = nullptr) {
  for (int c = 0; c < np; c += NB) {
    const int rows_below = np - c - NB + extra;
    const int nchunks = (rows_below + PANEL_ROWS - 1) / PANEL_ROWS;
    double* Acc = A + (int64_t)c * lda + c;
    if (prof) {
      prof->panel_marks.push_back(prof->used);
      hipEventRecord(prof->next(), st);
    }
    hipLaunchKernelGGL(potrf_panel_kernel, dim3(nchunks + 1), dim3(256), 0, st, Acc, lda, rows_below, nchunks,
                       inv_diag + (int64_t)(c / NB) * NB * NB, info, c);
    if (prof) hipEventRecord(prof->next(), st);
    const int ncols = np - c - NB;
    if (ncols > 0) {
      double* L21 = A + (int64_t)(c + NB) * lda + c;
      double* A22 = A + (int64_t)(c + NB) * lda + (c + NB);
      if (prof) {
        // algorithmic flops: 2 K per updated element of the lower trapezoid (diagonal included)
        const double elems = 0.5 * (double)ncols * (ncols + 1) + (double)extra * ncols;
        prof->gemm_marks.push_back({prof->used, 2.0 * NB * elems});
        hipEventRecord(prof->next(), st);
      }
      hipError_t e = launch_gemm(st, 0, 1, rows_below, ncols, NB, -1.0, L21, lda, L21, lda, 1.0, A22, lda, GEMM_C_LOWER);
      if (prof) hipEventRecord(prof->next(), st);
      if (e != hipSuccess) return e;
    }
  }
  return hipGetLastError();
}

}  // namespace gprx
