// Blocked right-looking Cholesky (lower, in place, row-major) for gfx950.
//
// Per 64-column panel two launches:
//   1. potrf_panel_kernel  -- factor the 64 x 64 diagonal block AND solve every row below it
//      (L21 = A21 L11^-T, including the right-hand-side rows appended under the matrix) in one
//      launch.  A workgroup owns 128 panel rows: the 64 rows of the diagonal block -- every workgroup
//      re-factors it redundantly, so no inter-workgroup hand-off exists -- plus 64 rows of A21.
//      The 128 x 64 panel lives in the MFMA accumulators (wave w: rows 32 w .. 32 w + 31).  Work
//      proceeds in 8-column sub-panels: the sub-panel's columns go accumulators -> LDS, each thread
//      factors the 8 x 8 diagonal sub-block (register math) and solves its own row, then the
//      columns right of the sub-panel are updated on MFMA (rows x 8 times (64 x 8)^T) with the
//      operands read once from LDS and no read-modify-write of C; two barriers per sub-panel.  One
//      extra workgroup carries the 64 identity rows, which come out as L11^-1 (inverse of the
//      diagonal block, used by the triangular solves).
//   2. gemm_f64 (NT, C_LOWER, alpha = -1, beta = 1) -- trailing update A22 -= L21 L21^T on MFMA.
#pragma once
#include <cstdlib>
#include <utility>
#include <vector>

#include "chain64.h"
#include "gemm_f64.h"
#include "gprx_common.h"

namespace gprx {

// GPRX_PANEL_ACC (development builds, tools/panel_acc.sh): phase durations of panel workgroup 0 (a rows workgroup) summed over
// every panel launch of a factorisation -- [0] launches, [1] loads, [2] sub-panel 0, [3] sub-panels 1-3, [4] sub-panels 4-7,
// [5] stores (shader clocks)
#ifdef GPRX_PANEL_ACC
__device__ unsigned long long g_panel_acc[8];
#define PACC_DECL unsigned long long pacc_prev_ = 0;
#define PACC(i)                                                                                                       \
  {                                                                                                                   \
    if ((i) == 5) __builtin_amdgcn_s_waitcnt(0);                                                                      \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                       \
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && gridDim.x > 2) {                                    \
      if ((i) > 0) atomicAdd(&g_panel_acc[i], t_ - pacc_prev_);                                                       \
      else atomicAdd(&g_panel_acc[0], 1ull);                                                                          \
    }                                                                                                                 \
    pacc_prev_ = t_;                                                                                                  \
  }
#else
#define PACC_DECL
#define PACC(i)
#endif
#ifdef GPRX_PANEL_STAMPS
__device__ unsigned long long g_panel_stamps[64];
#define PSTAMP(i)                                                                                                     \
  if (threadIdx.x == 0) {                                                                                             \
    if (blockIdx.x == 0) g_panel_stamps[i] = __builtin_amdgcn_s_memtime();                                            \
    if (blockIdx.x == gridDim.x / 2 && (i) < 6) g_panel_stamps[32 + (i)] = __builtin_amdgcn_s_memtime();               \
    if (blockIdx.x == gridDim.x - 2 && (i) < 6) g_panel_stamps[48 + (i)] = __builtin_amdgcn_s_memtime();               \
  }
#else
#define PSTAMP(i)
#endif

// RT = 16-row tiles per wave: 2 -> 128 rows per workgroup (64 of A21), 4 -> 256 rows per workgroup (192 of A21).
// Fewer, fatter workgroups leave CUs to other cells' GEMMs when several cells are in flight.
template <int RT>
struct PanelGeom {
  static constexpr int kWgRows = 64 * RT;      // rows held by one workgroup (rows 0..63 = the diagonal block)
  static constexpr int kOwnRows = 64 * RT - NB;  // rows of A21 per workgroup
};
// (PSUB, the LDS row stride of the 8-column sub-panel buffers, lives in chain64.h)

// State shared by the unrolled sub-panel steps.
struct PanelCtx {
  double* sIn;    // [rows][9]  current sub-panel, as updated so far (written from the accumulators)
  double* sX;     // [rows][9]  current sub-panel, solved (MFMA operands of the trailing update)
  double* out;    // this thread's output row (global memory or the diagonal-block staging area); nullptr: none
  double* inv_diag;
  double* rinv_out;  // last workgroup: the 64 reciprocal pivots, stored behind the staged diagonal block
  int ident;      // >= 0: this thread carries identity row `ident` (last workgroup)
  int tid, wave, g, r;
  int zero_above;  // diagonal-block rows: entries right of the diagonal are zero
  int bad;
};

// One 8-column sub-panel.  acc[rt][kt]: this wave's 16 RT rows x 64 columns in MFMA C/D layout
// (lane (g, r) holds rows 16 RT w + 16 rt + g + 4 q, column 16 kt + r).
template <int P, int RT>
__device__ __forceinline__ void panel_step(d4 (&acc)[RT][4], PanelCtx& c) {
  constexpr int WROWS = 16 * RT;                 // rows per wave
  constexpr int PWG_ROWS = PanelGeom<RT>::kWgRows;
  constexpr int C0 = 8 * P;
  constexpr int KT = C0 / 16;      // tile column holding this sub-panel
  constexpr int HALF = P & 1;      // which 8 columns of that tile
  if constexpr (P == 1) { PSTAMP(10) }
  // A: accumulators -> LDS (only the lanes that hold these 8 columns)
  if ((c.r >> 3) == HALF) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) c.sIn[(WROWS * c.wave + 16 * rt + c.g + 4 * q) * PSUB + (c.r & 7)] = acc[rt][KT][q];
  }
  __syncthreads();
  if constexpr (P == 1) { PSTAMP(11) }
  // B: every thread factors the 8 x 8 diagonal sub-block (rows C0 .. C0+7 of the diagonal block)
  double l[8][8], rinv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k <= j; ++k) l[j][k] = c.sIn[(C0 + j) * PSUB + k];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    double s = l[j][j];
#pragma unroll
    for (int m = 0; m < j; ++m) s = __builtin_fma(-l[j][m], l[j][m], s);
    if (!(s > 0.0)) {
      if (c.bad == 0) c.bad = C0 + j + 1;
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
  if (c.rinv_out && c.tid == 0) {  // reciprocal pivots for potrf_rows_kernel (split panel): exactly the values used here
#pragma unroll
    for (int j = 0; j < 8; ++j) c.rinv_out[C0 + j] = rinv[j];
  }
  if constexpr (P == 1) { PSTAMP(12) }
  if (c.tid < PWG_ROWS) {
    // own row: x = a[C0 .. C0+7] L_dd^-T
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double t = c.sIn[c.tid * PSUB + k];
#pragma unroll
      for (int m = 0; m < k; ++m) t = __builtin_fma(-x[m], l[k][m], t);
      x[k] = (C0 + k > c.zero_above) ? 0.0 : t * rinv[k];
    }
    if constexpr (P == 1) { PSTAMP(16) }
#pragma unroll
    for (int k = 0; k < 8; ++k) c.sX[c.tid * PSUB + k] = x[k];
    if constexpr (P == 1) { PSTAMP(17) }
    // solved values leave through memory directly (64 contiguous bytes per row and sub-panel): no 66-KiB output
    // image in LDS, so several panel workgroups -- of this or of other cells -- fit on one CU beside GEMM tiles
    if (c.out) {
#pragma unroll
      for (int k = 0; k < 8; k += 2) *reinterpret_cast<d2*>(c.out + C0 + k) = d2{x[k], x[k + 1]};
    }
    if (c.ident >= 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) c.inv_diag[(C0 + k) * NB + c.ident] = x[k];  // identity row i -> column i of L11^-1
    }
  }
  if constexpr (P == 1) { PSTAMP(13) }
  __syncthreads();
  if constexpr (P == 1) { PSTAMP(14) }
  // C: trailing columns [C0 + 8, 64) of this wave's rows: acc -= X_rows (32 x 8) * X_diag(16 kt .. +15, 8)^T
  if constexpr (C0 + 8 < NB) {
    constexpr int KT0 = (C0 + 8) / 16;
    double fa[RT][2], fb[4][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[rt][ks] = -c.sX[(WROWS * c.wave + 16 * rt + c.r) * PSUB + 4 * ks + c.g];
#pragma unroll
    for (int kt = KT0; kt < 4; ++kt) {
      const int kk = kt * 16 + c.r;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb[kt][ks] = (kk >= C0 + 8) ? c.sX[kk * PSUB + 4 * ks + c.g] : 0.0;
    }
    // the tile column that the next sub-panel reads goes first; the rest may still be in the MFMA pipe
    // while the next sub-panel's factorisation runs on the VALU
#pragma unroll
    for (int kt = KT0; kt < 4; ++kt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][0], fb[kt][0], acc[rt][kt], 0, 0, 0);
        acc[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][1], fb[kt][1], acc[rt][kt], 0, 0, 0);
      }
  }
  // the solved values return to the accumulators of the lanes that hold these columns: the panel leaves through
  // one coalesced store pass at the end (16 lanes = one 128-byte line) instead of 16-byte stores scattered over 64
  // rows per instruction -- 4096 write requests per workgroup that the L2 had to merge.  Issued after the MFMAs so
  // that the LDS reads run under them (the MFMA on this tile column added exact zeros to the solved columns).
  if ((c.r >> 3) == HALF) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[rt][KT][q] = c.sX[(WROWS * c.wave + 16 * rt + c.g + 4 * q) * PSUB + (c.r & 7)];
  }
  if constexpr (P == 1) { PSTAMP(15) }
}

// A points at the diagonal block (c, c).  rows_below = rows under the block to solve.
// Workgroup rows 0..63 = the diagonal block (every workgroup factors it redundantly), rows 64..127 =
// this workgroup's 64 rows of A21 (or, in the last workgroup, the 64 identity rows -> L11^-1).
//
// In-place hazard: every workgroup reads the diagonal block, so none of them may overwrite it during
// this launch (a late-dispatched workgroup would read L11 instead of A11).  The last workgroup writes
// L11 to `stage_out` instead, and copies the PREVIOUS panel's staged block (`prev_stage`, prev_pw x
// prev_pw) to its place `prev_dst` in the matrix -- all readers of that block finished with the
// previous launch.  potrf_lower flushes the final block with copy_block_kernel.
__device__ __forceinline__ void flush_staged_block(const double* __restrict__ prev_stage, double* __restrict__ prev_dst, int64_t lda,
                                                   int prev_pw, int tid) {
  if (!prev_stage) return;
  const int chunks = prev_pw / 2;  // 16-byte chunks per row
  for (int e = tid; e < prev_pw * chunks; e += 256) {
    const int row = e / chunks, cc = e % chunks;
    *reinterpret_cast<d2*>(prev_dst + (int64_t)row * lda + 2 * cc) = *reinterpret_cast<const d2*>(prev_stage + row * prev_pw + 2 * cc);
  }
}

__global__ __launch_bounds__(256) void copy_block_kernel(const double* __restrict__ stage, double* __restrict__ dst, int64_t lda, int pw,
                                                         int64_t cs) {
  flush_staged_block(stage + (int64_t)blockIdx.x * cs, dst + (int64_t)blockIdx.x * cs, lda, pw, threadIdx.x);
}

template <int RT, int OCC, bool FUSE_K64 = false>
__global__ __launch_bounds__(256, OCC) void potrf_panel_kernel(double* __restrict__ A, int64_t lda, int rows_below, int nchunks,
                                                          double* __restrict__ inv_diag, int* __restrict__ info, int col0,
                                                          double* __restrict__ stage_out, const double* __restrict__ prev_stage,
                                                          double* __restrict__ prev_dst, int prev_pw, int64_t cs, int info_stride,
                                                          double* __restrict__ yv = nullptr) {
  constexpr int PWG_ROWS = PanelGeom<RT>::kWgRows;
  constexpr int PANEL_ROWS = PanelGeom<RT>::kOwnRows;
  constexpr int WROWS = 16 * RT;
#ifndef GPRX_PANEL_NO_SETPRIO
  // the panel is the dependent chain: its waves outrank the bulk update's waves they share SIMDs with (instruction issue is
  // arbitrated by priority, then age -- MI355X_MICROARCH.md, two waves per SIMD).  Measured: N = 16384 29.79 -> 29.46 ms,
  // N = 8192 6.46 -> 6.30 ms, N = 4096 2.17 -> 2.15 ms (tools/setprio_probe.sh): small, consistent, free.
  __builtin_amdgcn_s_setprio(3);
#endif
  {
    // batched: blockIdx.y = cell; every per-cell pointer lives in one cell block, `cs` doubles apart
    const int64_t off = (int64_t)blockIdx.y * cs;
    A += off;
    inv_diag += off;
    stage_out += off;
    if (prev_stage) prev_stage += off;
    if (prev_dst) prev_dst += off;
    info += (int64_t)blockIdx.y * info_stride;
  }
  __shared__ __attribute__((aligned(16))) double sIn[PWG_ROWS * PSUB];
  __shared__ __attribute__((aligned(16))) double sX[PWG_ROWS * PSUB];
  PanelCtx c;
  c.sIn = sIn;
  c.sX = sX;
  c.inv_diag = inv_diag;
  c.tid = threadIdx.x;
  const int lane = c.tid & 63;
  c.wave = c.tid >> 6;
  c.g = lane >> 4;
  c.r = lane & 15;
  c.zero_above = c.tid < NB ? c.tid : (1 << 30);
  c.bad = 0;
  const bool last = (int)blockIdx.x == nchunks;
  c.rinv_out = last ? stage_out + NB * NB : nullptr;
  if (last) flush_staged_block(prev_stage, prev_dst, lda, prev_pw, c.tid);
  // where this thread's solved row goes (thread t < PWG_ROWS owns workgroup row t)
  c.out = nullptr;
  c.ident = -1;
  if (c.tid < NB) {
    if (last) c.out = stage_out + c.tid * NB;  // the factored diagonal block is staged (in-place hazard, see above)
  } else if (c.tid < PWG_ROWS) {
    if (last && c.tid < 2 * NB) c.ident = c.tid - NB;  // (rows of A21 leave through the final store pass)
  }
  PSTAMP(0)
  PACC_DECL
  PACC(0)

  // ---- load straight into the accumulator layout: 16 RT loads per lane, all in flight ----
  // Every load is unconditional (rows outside the matrix read row 0 of the diagonal block instead) and the
  // triangle / padding / identity masks are applied afterwards with selects: with the conditions around the
  // loads the compiler emitted a branch and a full wait per load, i.e. 32 serialised memory round trips.
  d4 acc[RT][4];
  {
    const double* rowp[RT][4];
    bool valid[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int wrow = WROWS * c.wave + 16 * rt + c.g + 4 * q;  // workgroup row
        const int idx = (int)blockIdx.x * PANEL_ROWS + (wrow - NB);
        const bool diag = wrow < NB;
        const bool ok = diag || (!last && idx < rows_below);
        const int64_t mrow = diag ? wrow : (ok ? NB + idx : 0);
        rowp[rt][q] = A + mrow * lda + c.r;
        valid[rt][q] = ok;
      }
    d4 upd[FUSE_K64 ? RT : 1][4];
    if constexpr (FUSE_K64) {
      // The K = 64 update that the schedule would launch between the previous panel and this one (these 64 columns, every
      // row from the diagonal block down, by the 64 columns left of them), applied here to the rows this workgroup holds: one
      // dependent launch less per odd panel (~9 us of a lone matrix's chain).  Operation for operation what gemm_f64 does
      // (accumulators from zero, stages of 16 along k, instruction j of a stage takes k = k0 + 4 g + j; C + (-1) * sum with one
      // rounding), so the factor is the same bit for bit as with the separate launch (batched cells keep that launch).
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) upd[rt][kt] = d4{0.0, 0.0, 0.0, 0.0};
      const double* arow[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int wrow = WROWS * c.wave + 16 * rt + c.r;  // A operand: lane (g, r) supplies row r of the tile, k = 4 g + j
        const int idx = (int)blockIdx.x * PANEL_ROWS + (wrow - NB);
        const bool diag = wrow < NB;
        const bool ok = diag || (!last && idx < rows_below);
        arow[rt] = A + (diag ? wrow : (ok ? NB + idx : 0)) * lda - NB + 4 * c.g;
      }
      const double* brow = A + (int64_t)c.r * lda - NB + 4 * c.g;  // B operand: rows of the diagonal block (row 16 kt + r)
#pragma unroll
      for (int k0 = 0; k0 < NB; k0 += 16) {
        double fa[RT][4], fb[4][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const d2 lo = *reinterpret_cast<const d2*>(arow[rt] + k0), hi = *reinterpret_cast<const d2*>(arow[rt] + k0 + 2);
          fa[rt][0] = lo.x; fa[rt][1] = lo.y; fa[rt][2] = hi.x; fa[rt][3] = hi.y;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const double* bp = brow + (int64_t)(16 * kt) * lda + k0;
          const d2 lo = *reinterpret_cast<const d2*>(bp), hi = *reinterpret_cast<const d2*>(bp + 2);
          fb[kt][0] = lo.x; fb[kt][1] = lo.y; fb[kt][2] = hi.x; fb[kt][3] = hi.y;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) upd[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][j], fb[kt][j], upd[rt][kt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) acc[rt][kt][q] = rowp[rt][q][kt * 16];
    if constexpr (FUSE_K64) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double v = -1.0 * upd[rt][kt][q];
            acc[rt][kt][q] = __builtin_fma(1.0, acc[rt][kt][q], v);
          }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int wrow = WROWS * c.wave + 16 * rt + c.g + 4 * q;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const int col = kt * 16 + c.r;
          double v = valid[rt][q] ? acc[rt][kt][q] : 0.0;
          if (wrow < NB) {
            v = (col > wrow) ? 0.0 : v;
          } else if (last) {
            v = (col == wrow - NB) ? 1.0 : 0.0;
          }
          acc[rt][kt][q] = v;
        }
      }
  }
  PSTAMP(1)
  PACC(1)

  panel_step<0, RT>(acc, c);
  PSTAMP(2)
  PACC(2)
  panel_step<1, RT>(acc, c);
  panel_step<2, RT>(acc, c);
  panel_step<3, RT>(acc, c);
  PSTAMP(3)
  PACC(3)
  panel_step<4, RT>(acc, c);
  panel_step<5, RT>(acc, c);
  panel_step<6, RT>(acc, c);
  panel_step<7, RT>(acc, c);
  PSTAMP(4)
  PACC(4)
  if (!last) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int wrow = WROWS * c.wave + 16 * rt + c.g + 4 * q;
        const int idx = (int)blockIdx.x * PANEL_ROWS + (wrow - NB);
        if (wrow >= NB && idx < rows_below) {
          double* dst = A + (int64_t)(NB + idx) * lda + c.r;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) dst[kt * 16] = acc[rt][kt][q];
        }
      }
  }
  if (last && c.tid == 0 && c.bad != 0) atomicCAS(info, 0, col0 + c.bad);
  if (yv != nullptr && last) {
    // Right-hand side as a vector (potrf_rows_kernel<..., YVEC>): beta_j = L11^-1 y_j by the diagonal workgroup itself, from the inverse
    // its identity rows have just stored (every wave's stores acknowledged, then the barrier; nobody read those lines before).  Row a of
    // the inverse times y_j: four partial sums of 16 terms, added in a fixed order.
    yv += (int64_t)blockIdx.y * cs;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double* sv = sIn;         // (the sub-panel buffers are free now)
    double* sred = sIn + NB;
    static_assert(PWG_ROWS * PSUB >= NB + 256, "the sub-panel buffer holds y_j and the partial sums");
    if (c.tid < NB) sv[c.tid] = yv[c.tid];
    __syncthreads();
    const int a = c.tid & 63, qq = c.tid >> 6;
    const double* row = inv_diag + a * NB + 16 * qq;
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum = __builtin_fma(row[i], sv[16 * qq + i], sum);
    sred[c.tid] = sum;
    __syncthreads();
    if (c.tid < NB) yv[c.tid] = ((sred[c.tid] + sred[64 + c.tid]) + sred[128 + c.tid]) + sred[192 + c.tid];
  }
  PSTAMP(5)
  PACC(5)
}

// ---- split panel (many cells per launch): rows only ---------------------------------------------------
// potrf_panel_kernel launched with ONE workgroup per cell factors the diagonal block (staged L11, L11^-1 and the
// 64 reciprocal pivots); this kernel then solves the rows below it, 128 rows per workgroup and no redundant
// factorisation: L11 comes from the staging area into LDS.  The arithmetic per row -- substitution order, the
// MFMA updates of the columns right of each 8-column sub-panel and their order -- is that of potrf_panel_kernel
// with the operands it would have recomputed, so the results are bit-identical; the registers that held the 8 x 8
// factor are free (3 workgroups per CU instead of 2) and a launch has half as many workgroups.
constexpr int ROWS_WG = 128;
constexpr int ROWS_LSTR = NB + 1;  // LDS row stride of the L11 image

// RT = 16-row tiles per wave: 2 -> 128 rows per workgroup (half of the threads substitute), 4 -> 256 rows per workgroup
// (every thread substitutes one row; one L11 image in LDS serves twice the rows).
// WAVE_LOCAL: a wave substitutes its OWN 16 RT rows (lanes < 16 RT), so nothing crosses waves and the two workgroup barriers
// of a sub-panel become wave-scope fences: the four waves of a workgroup drift apart and overlap their phases.
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int P, int RT, bool SCALAR_L = false, bool WAVE_LOCAL = false>
__device__ __forceinline__ void rows_step(d4 (&acc)[RT][4], double* __restrict__ sIn, double* __restrict__ sX, const double* __restrict__ sL,
                                          const double* __restrict__ sRinv, int tid, int wave, int g, int r,
                                          const double* __restrict__ gstage = nullptr) {
  constexpr int C0 = 8 * P;
  constexpr int KT = C0 / 16;
  constexpr int HALF = P & 1;
  constexpr int WROWS = 16 * RT;
  if ((r >> 3) == HALF) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) sIn[(WROWS * wave + 16 * rt + g + 4 * q) * PSUB + (r & 7)] = acc[rt][KT][q];
  }
  if constexpr (WAVE_LOCAL) wave_sync_lds(); else __syncthreads();
  const int srow = WAVE_LOCAL ? WROWS * wave + (tid & 63) : tid;
  if (WAVE_LOCAL ? (tid & 63) < WROWS : tid < 64 * RT) {
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double t = sIn[srow * PSUB + k];
#pragma unroll
      for (int m = 0; m < k; ++m) {
        // SCALAR_L: the 8 x 8 diagonal sub-block and the pivots through scalar loads from the staged block (uniform addresses:
        // s_load into SGPRs that v_fma_f64 takes directly) instead of 44 LDS broadcast reads per row and sub-panel
        const double lkm = SCALAR_L ? gstage[(C0 + k) * NB + C0 + m] : sL[(C0 + k) * ROWS_LSTR + C0 + m];
        t = __builtin_fma(-x[m], lkm, t);
      }
      x[k] = t * (SCALAR_L ? gstage[NB * NB + C0 + k] : sRinv[C0 + k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sX[srow * PSUB + k] = x[k];
  }
  if constexpr (WAVE_LOCAL) wave_sync_lds(); else __syncthreads();
  if constexpr (C0 + 8 < NB) {
    constexpr int KT0 = (C0 + 8) / 16;
    double fa[RT][2], fb[4][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[rt][ks] = -sX[(WROWS * wave + 16 * rt + r) * PSUB + 4 * ks + g];
#pragma unroll
    for (int kt = KT0; kt < 4; ++kt) {
      const int kk = kt * 16 + r;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        fb[kt][ks] = (kk >= C0 + 8) ? (SCALAR_L && sL == nullptr ? gstage[kk * NB + C0 + 4 * ks + g] : sL[kk * ROWS_LSTR + C0 + 4 * ks + g]) : 0.0;
    }
#pragma unroll
    for (int kt = KT0; kt < 4; ++kt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][0], fb[kt][0], acc[rt][kt], 0, 0, 0);
        acc[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][1], fb[kt][1], acc[rt][kt], 0, 0, 0);
      }
  }
  if ((r >> 3) == HALF) {  // solved values back into the accumulators (see potrf_panel_kernel): one coalesced store pass at the end
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[rt][KT][q] = sX[(WROWS * wave + 16 * rt + g + 4 * q) * PSUB + (r & 7)];
  }
}

// A21: first row below the diagonal block (rows_below rows, lda); stage: this panel's staged L11 (64 x 64) followed by
// the 64 reciprocal pivots.  grid = (ceil(rows_below / (64 RT)), cells).
// FUSE_K64: as in potrf_panel_kernel -- the K = 64 update of these 64 columns by the 64 columns left of them (which the schedule
// would launch between the previous panel and this one) is applied to this workgroup's rows on their way in: with many cells per
// launch that update is HBM-bound (16 bytes of C traffic per 128 flops), and this kernel reads and writes the very same columns
// anyway.  Operation for operation the general NT kernel's arithmetic (accumulators from zero, stages of 16 along k, instruction j
// takes k = k0 + 4 g + j, C + (-1) * sum with one rounding): bit-identical to the separate launch.
// YVEC (round 4): the right-hand side of the cell travels as a VECTOR instead of a 64-row tile below the matrix (one useful row of 64:
// T^2 / 2 tile products of a factorisation with T block columns, 4.5 % of the flops at N = 4096).  yv points at this panel's 64 entries
// of it -- beta_j = L11^-1 y_j, written by potrf_beta_block_kernel between the diagonal workgroup and this launch -- followed by the
// entries of the rows below, which this kernel updates: y_i -= sum_c L(i, c) beta_j[c], the 64 products of a row summed over the tile
// columns in a lane (k t ascending) and then over the 16 lanes that hold the row (xor 1, 2, 4, 8): a fixed order.
template <int RT, int OCC, bool SCALAR_L = false, bool NO_LDS_L = false, bool WAVE_LOCAL = false, bool FUSE_K64 = false, bool YVEC = false>
__global__ __launch_bounds__(256, OCC) void potrf_rows_kernel(double* __restrict__ A21, int64_t lda, int rows_below,
                                                              const double* __restrict__ stage, int64_t cs, double* __restrict__ yv = nullptr) {
  constexpr int WG_ROWS = 64 * RT, WROWS = 16 * RT;
  __shared__ __attribute__((aligned(16))) double sIn[WG_ROWS * PSUB];
  __shared__ __attribute__((aligned(16))) double sX[WG_ROWS * PSUB];
  __shared__ __attribute__((aligned(16))) double sLbuf[NO_LDS_L ? 1 : NB * ROWS_LSTR];
  __shared__ double sRinv[NB];
  double* sL = NO_LDS_L ? nullptr : sLbuf;
  A21 += (int64_t)blockIdx.y * cs;
  stage += (int64_t)blockIdx.y * cs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
  const int row0 = blockIdx.x * WG_ROWS;
  // own rows -> accumulator layout, every load unconditional (rows past the end re-read row 0 and are masked)
  d4 acc[RT][4];
  {
    const double* rowp[RT][4];
    bool valid[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = row0 + WROWS * wave + 16 * rt + g + 4 * q;
        valid[rt][q] = idx < rows_below;
        rowp[rt][q] = A21 + (int64_t)(valid[rt][q] ? idx : 0) * lda + r;
      }
    d4 upd[FUSE_K64 ? RT : 1][4];
    if constexpr (FUSE_K64) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) upd[rt][kt] = d4{0.0, 0.0, 0.0, 0.0};
      const double* arow[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int idx = row0 + WROWS * wave + 16 * rt + r;  // A operand: lane (g, r) supplies row r of the tile, k = 4 g + j
        arow[rt] = A21 + (int64_t)(idx < rows_below ? idx : 0) * lda - NB + 4 * g;
      }
      const double* brow = A21 - (int64_t)NB * lda + (int64_t)r * lda - NB + 4 * g;  // B operand: the diagonal block's rows, previous 64 columns
#pragma unroll
      for (int k0 = 0; k0 < NB; k0 += 16) {
        double fa[RT][4], fb[4][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const d2 lo = *reinterpret_cast<const d2*>(arow[rt] + k0), hi = *reinterpret_cast<const d2*>(arow[rt] + k0 + 2);
          fa[rt][0] = lo.x; fa[rt][1] = lo.y; fa[rt][2] = hi.x; fa[rt][3] = hi.y;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const double* bp = brow + (int64_t)(16 * kt) * lda + k0;
          const d2 lo = *reinterpret_cast<const d2*>(bp), hi = *reinterpret_cast<const d2*>(bp + 2);
          fb[kt][0] = lo.x; fb[kt][1] = lo.y; fb[kt][2] = hi.x; fb[kt][3] = hi.y;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) upd[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][j], fb[kt][j], upd[rt][kt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) acc[rt][kt][q] = rowp[rt][q][kt * 16];
    if constexpr (FUSE_K64) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double v = -1.0 * upd[rt][kt][q];
            acc[rt][kt][q] = __builtin_fma(1.0, acc[rt][kt][q], v);
          }
    }
    // L11 image and the reciprocal pivots (staged by the diagonal workgroup of this panel)
    if constexpr (!NO_LDS_L) {
      for (int e = tid; e < NB * NB / 2; e += 256) {
        const int row = e / (NB / 2), cc = e % (NB / 2);
        const d2 v = *reinterpret_cast<const d2*>(stage + row * NB + 2 * cc);
        sL[row * ROWS_LSTR + 2 * cc] = v.x;
        sL[row * ROWS_LSTR + 2 * cc + 1] = v.y;
      }
      if (tid < NB) sRinv[tid] = stage[NB * NB + tid];
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) acc[rt][kt][q] = valid[rt][q] ? acc[rt][kt][q] : 0.0;
  }
  // (the first barrier inside rows_step<0> also publishes sL / sRinv)
  rows_step<0, RT, SCALAR_L, WAVE_LOCAL>(acc, sIn, sX, sL, sRinv, tid, wave, g, r, stage);
  rows_step<1, RT, SCALAR_L, WAVE_LOCAL>(acc, sIn, sX, sL, sRinv, tid, wave, g, r, stage);
  rows_step<2, RT, SCALAR_L, WAVE_LOCAL>(acc, sIn, sX, sL, sRinv, tid, wave, g, r, stage);
  rows_step<3, RT, SCALAR_L, WAVE_LOCAL>(acc, sIn, sX, sL, sRinv, tid, wave, g, r, stage);
  rows_step<4, RT, SCALAR_L, WAVE_LOCAL>(acc, sIn, sX, sL, sRinv, tid, wave, g, r, stage);
  rows_step<5, RT, SCALAR_L, WAVE_LOCAL>(acc, sIn, sX, sL, sRinv, tid, wave, g, r, stage);
  // (YVEC: beta_j and this lane group's entries of y are requested before the last two sub-panel steps -- their latency hides behind
  // those -- and only stored at the end; not earlier, the registers are needed)
  double ybj[YVEC ? 4 : 1], yold[YVEC ? RT : 1][4];
  if constexpr (YVEC) {
    yv += (int64_t)blockIdx.y * cs;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) ybj[kt] = yv[16 * kt + r];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = row0 + WROWS * wave + 16 * rt + g + 4 * q;
        yold[rt][q] = yv[NB + (idx < rows_below ? idx : 0)];
      }
  }
  rows_step<6, RT, SCALAR_L, WAVE_LOCAL>(acc, sIn, sX, sL, sRinv, tid, wave, g, r, stage);
  rows_step<7, RT, SCALAR_L, WAVE_LOCAL>(acc, sIn, sX, sL, sRinv, tid, wave, g, r, stage);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = row0 + WROWS * wave + 16 * rt + g + 4 * q;
      if (idx < rows_below) {
        double* dst = A21 + (int64_t)idx * lda + r;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) dst[kt * 16] = acc[rt][kt][q];
      }
    }
  if constexpr (YVEC) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double sum = acc[rt][0][q] * ybj[0];
#pragma unroll
        for (int kt = 1; kt < 4; ++kt) sum = __builtin_fma(acc[rt][kt][q], ybj[kt], sum);
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) sum += __shfl_xor(sum, m, 64);
        const int idx = row0 + WROWS * wave + 16 * rt + g + 4 * q;
        if (r == 0 && idx < rows_below) yv[NB + idx] = yold[rt][q] - sum;
      }
  }
}

// beta_j = L11^-1 y_j for the panel at hand (YVEC above): one workgroup per cell, between the diagonal workgroup (which leaves L11^-1 in
// inv_diag) and the rows kernel.  Row a of the inverse times y_j, four partial sums of 16 terms added in a fixed order.
__global__ __launch_bounds__(256) void potrf_beta_block_kernel(const double* __restrict__ inv, double* __restrict__ yv, int64_t cs) {
  __shared__ double sv[NB];
  __shared__ double sred[256];
  inv += (int64_t)blockIdx.y * cs;
  yv += (int64_t)blockIdx.y * cs;
  const int tid = threadIdx.x, a = tid & 63, q = tid >> 6;
  if (tid < NB) sv[tid] = yv[tid];
  __syncthreads();
  const double* row = inv + a * NB + 16 * q;
  double sum = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) sum = __builtin_fma(row[i], sv[16 * q + i], sum);
  sred[tid] = sum;
  __syncthreads();
  if (tid < NB) yv[tid] = ((sred[tid] + sred[64 + tid]) + sred[128 + tid]) + sred[192 + tid];
}

// ---- split panel, rows by ONE tile product against the diagonal block's inverse (round 4) ----------------------------------
// X = A21 L11^-T with the explicit L11^-1 that the diagonal workgroup leaves in inv_diag anyway (row-major, exact zeros above
// the diagonal): per 16-row tile 40 MFMAs (the k range of output tile column kt ends at 16 kt + 15) and NO fp64 vector chain --
// potrf_rows_kernel's 8 x (LDS round trip, 44-FMA substitution, MFMA update) is what runs 1.7-2.3x slower beside a bulk update,
// because fp64 vector and matrix instructions share one pipe on gfx950; this kernel is loads -> MFMA -> stores.
// Operands come straight from global memory in MFMA operand layout (lane (g, r): row r of the tile, 4 consecutive k = 32 bytes,
// the four lane groups of a row = one 128-byte line); the 32 KB inverse is L1/L2 resident.  Results equal the substitution's to
// rounding (||L11^-1|| eps instead of the substitution's backward-stable rows), not bit for bit: include/gprx.h says where.
// FUSE_K64: as potrf_rows_kernel -- the K = 64 update by the 64 columns left of these is applied on the way in (same operation
// order as the separate launch, so fused == unfused bit for bit); the updated rows pass through a wave-private LDS image
// (accumulator layout -> operand layout), 16 rows at a time.
constexpr int RINV_LD = NB;  // doubles per row of the wave-private transposition image: 4 x 16 x 64 doubles = 32 KB per workgroup (the
                             // slot a capped bulk update leaves free on a CU); 32-byte groups XOR-ed with the row against bank conflicts
template <int RT, int OCC, bool FUSE_K64 = false>
__global__ __launch_bounds__(256, OCC) void potrf_rows_inv_kernel(double* __restrict__ A21, int64_t lda, int rows_below,
                                                                  const double* __restrict__ inv, int64_t cs) {
  constexpr int WG_ROWS = 64 * RT, WROWS = 16 * RT;
  __shared__ __attribute__((aligned(16))) double sT[FUSE_K64 ? 4 * 16 * RINV_LD : 2];
  A21 += (int64_t)blockIdx.y * cs;
  inv += (int64_t)blockIdx.y * cs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
  const int row0 = blockIdx.x * WG_ROWS + WROWS * wave;
  double fa[RT][4][4];  // [tile][stage of 16 along k][instruction]: A operand of instruction j of stage s = A[row r][16 s + 4 g + j]
  if constexpr (!FUSE_K64) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int idx = row0 + 16 * rt + r;
      const double* p = A21 + (int64_t)(idx < rows_below ? idx : 0) * lda + 4 * g;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const d2 lo = *reinterpret_cast<const d2*>(p + 16 * s), hi = *reinterpret_cast<const d2*>(p + 16 * s + 2);
        fa[rt][s][0] = lo.x; fa[rt][s][1] = lo.y; fa[rt][s][2] = hi.x; fa[rt][s][3] = hi.y;
      }
    }
  } else {
    // the K = 64 update exactly as potrf_rows_kernel<..., FUSE_K64> applies it (accumulators from zero, stages of 16, C - sum)
    d4 upd[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) upd[rt][kt] = d4{0.0, 0.0, 0.0, 0.0};
    const double* arow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int idx = row0 + 16 * rt + r;
      arow[rt] = A21 + (int64_t)(idx < rows_below ? idx : 0) * lda - NB + 4 * g;
    }
    const double* brow = A21 - (int64_t)NB * lda + (int64_t)r * lda - NB + 4 * g;
#pragma unroll
    for (int k0 = 0; k0 < NB; k0 += 16) {
      double ua[RT][4], ub[4][4];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const d2 lo = *reinterpret_cast<const d2*>(arow[rt] + k0), hi = *reinterpret_cast<const d2*>(arow[rt] + k0 + 2);
        ua[rt][0] = lo.x; ua[rt][1] = lo.y; ua[rt][2] = hi.x; ua[rt][3] = hi.y;
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const double* bp = brow + (int64_t)(16 * kt) * lda + k0;
        const d2 lo = *reinterpret_cast<const d2*>(bp), hi = *reinterpret_cast<const d2*>(bp + 2);
        ub[kt][0] = lo.x; ub[kt][1] = lo.y; ub[kt][2] = hi.x; ub[kt][3] = hi.y;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) upd[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[rt][j], ub[kt][j], upd[rt][kt], 0, 0, 0);
    }
    double* img = sT + wave * 16 * RINV_LD;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (rt) wave_sync_lds();  // (the image is reused: the previous tile's reads are done)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = row0 + 16 * rt + g + 4 * q;
        const double* src = A21 + (int64_t)(idx < rows_below ? idx : 0) * lda + r;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const double v = -1.0 * upd[rt][kt][q];
          img[(g + 4 * q) * RINV_LD + 4 * (((4 * kt + (r >> 2)) ^ (g + 4 * q)) & 15) + (r & 3)] = __builtin_fma(1.0, src[kt * 16], v);
        }
      }
      wave_sync_lds();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double* gp = img + r * RINV_LD + 4 * (((4 * s + g) ^ r) & 15);
        const d2 lo = *reinterpret_cast<const d2*>(gp), hi = *reinterpret_cast<const d2*>(gp + 2);
        fa[rt][s][0] = lo.x; fa[rt][s][1] = lo.y; fa[rt][s][2] = hi.x; fa[rt][s][3] = hi.y;
      }
    }
  }
  d4 acc[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) acc[rt][kt] = d4{0.0, 0.0, 0.0, 0.0};
  const double* bp = inv + r * NB + 4 * g;  // B operand: (L11^-T)[k][c] = L11^-1[c][k], c = 16 kt + r, k = 16 s + 4 g + j
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    double fb[4][4];
#pragma unroll
    for (int kt = s; kt < 4; ++kt) {
      const d2 lo = *reinterpret_cast<const d2*>(bp + 16 * kt * NB + 16 * s), hi = *reinterpret_cast<const d2*>(bp + 16 * kt * NB + 16 * s + 2);
      fb[kt][0] = lo.x; fb[kt][1] = lo.y; fb[kt][2] = hi.x; fb[kt][3] = hi.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int kt = s; kt < 4; ++kt) acc[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][s][j], fb[kt][j], acc[rt][kt], 0, 0, 0);
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = row0 + 16 * rt + g + 4 * q;
      if (idx < rows_below) {
        double* dst = A21 + (int64_t)idx * lda + r;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) dst[kt * 16] = acc[rt][kt][q];
      }
    }
}

// ---- 128-column panel: the same algorithm with a 128 x 128 diagonal block ------------------------
// Workgroup rows 0..127 = the diagonal block, rows 128..255 = 128 rows of A21 (last workgroup: 128
// identity rows).  Wave w owns workgroup rows 64 w .. 64 w + 63: acc[4][8] = 64 rows x 128 columns.
// LDS holds only the two 8-column sub-panel buffers (18 KiB); solved values go straight to memory
// (one thread = one row, 64 contiguous bytes per sub-panel).
constexpr int PW = 128;
// staging area: STAGE_LD doubles per matrix column; the panel at column c stages its diagonal block (pw x pw, row-major)
// at c * STAGE_LD and its pw reciprocal pivots right behind it (64 * 64 + 64 and 128 * 128 + 128 both fit)
constexpr int STAGE_LD = PW + 1;

struct Panel128Ctx {
  double* sIn;   // [256][9]
  double* sX;    // [256][9]
  double* out;   // this thread's output row in global memory (nullptr: nothing to write)
  double* inv_diag;
  double* rinv_out;  // last workgroup: the 128 reciprocal pivots behind the staged block (read by potrf_rows128_kernel)
  int tid, wave, g, r;
  int zero_above;
  int ident;     // >= 0: this thread carries identity row `ident` (last workgroup)
  int bad;
};

// Operands of the trailing updates that were NOT needed by the next sub-panel: they are issued one
// step later, interleaved with that step's scalar factorisation (MFMA pipe and VALU overlap; issued
// back-to-back they would hold the wave for 64 cycles each, in order, before the factorisation starts).
struct Deferred128 {
  double fa[4][2];
  double fb[8][2];
};

template <int P>
__device__ __forceinline__ void panel128_step(d4 (&acc)[4][8], Panel128Ctx& c, Deferred128& df) {
  constexpr int C0 = 8 * P;
  constexpr int KT = C0 / 16;
  constexpr int HALF = P & 1;
  if ((c.r >> 3) == HALF) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) c.sIn[(64 * c.wave + 16 * rt + c.g + 4 * q) * PSUB + (c.r & 7)] = acc[rt][KT][q];
  }
  __syncthreads();
  double l[8][8], rinv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k <= j; ++k) l[j][k] = c.sIn[(C0 + j) * PSUB + k];
  const double own0 = c.sIn[c.tid * PSUB + 0], own1 = c.sIn[c.tid * PSUB + 1], own2 = c.sIn[c.tid * PSUB + 2],
               own3 = c.sIn[c.tid * PSUB + 3], own4 = c.sIn[c.tid * PSUB + 4], own5 = c.sIn[c.tid * PSUB + 5],
               own6 = c.sIn[c.tid * PSUB + 6], own7 = c.sIn[c.tid * PSUB + 7];
  const double own[8] = {own0, own1, own2, own3, own4, own5, own6, own7};
  // deferred updates of the previous step: tile columns right of the one this step has just consumed
  if constexpr (P >= 1) {
    constexpr int KD0 = (8 * (P - 1) + 8) / 16 + 1;  // first deferred tile column of step P - 1
#pragma unroll
    for (int kt = KD0; kt < 8; ++kt)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        acc[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(df.fa[rt][0], df.fb[kt][0], acc[rt][kt], 0, 0, 0);
        acc[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(df.fa[rt][1], df.fb[kt][1], acc[rt][kt], 0, 0, 0);
      }
  }
  int bad = c.bad;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    double s = l[j][j];
#pragma unroll
    for (int m = 0; m < j; ++m) s = __builtin_fma(-l[j][m], l[j][m], s);
    const bool ok = s > 0.0;
    bad = (!ok && bad == 0) ? C0 + j + 1 : bad;
    s = ok ? s : 1.0;
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
  c.bad = bad;
  if (c.rinv_out && c.tid == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c.rinv_out[C0 + j] = rinv[j];
  }
  double x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double t = own[k];
#pragma unroll
    for (int m = 0; m < k; ++m) t = __builtin_fma(-x[m], l[k][m], t);
    x[k] = (C0 + k > c.zero_above) ? 0.0 : t * rinv[k];
  }
  if constexpr (P >= 1) {
    // one MFMA, then a run of VALU instructions, repeated: keeps the matrix pipe fed without stalling the chain
    constexpr int KD0 = (8 * (P - 1) + 8) / 16 + 1;
    constexpr int NM = (8 - KD0) * 8;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NM >= 40 ? 6 : (NM >= 24 ? 10 : 16), 0);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) c.sX[c.tid * PSUB + k] = x[k];
  if (c.out) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) *reinterpret_cast<d2*>(c.out + C0 + k) = d2{x[k], x[k + 1]};
  }
  if (c.ident >= 0) {
    // identity row i: x[k] = (L11^-1)[C0 + k][i]; keep the two 64 x 64 diagonal blocks of the inverse
    constexpr int BLK = C0 / NB;
    if ((c.ident / NB) == BLK) {
      const int i = c.ident - BLK * NB;
#pragma unroll
      for (int k = 0; k < 8; ++k) c.inv_diag[BLK * NB * NB + (C0 - BLK * NB + k) * NB + i] = x[k];
    }
  }
  __syncthreads();
  if constexpr (C0 + 8 < PW) {
    constexpr int KT0 = (C0 + 8) / 16;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) df.fa[rt][ks] = -c.sX[(64 * c.wave + 16 * rt + c.r) * PSUB + 4 * ks + c.g];
#pragma unroll
    for (int kt = KT0; kt < 8; ++kt) {
      const int kk = kt * 16 + c.r;
      df.fb[kt][0] = (kk >= C0 + 8) ? c.sX[kk * PSUB + c.g] : 0.0;
      df.fb[kt][1] = (kk >= C0 + 8) ? c.sX[kk * PSUB + 4 + c.g] : 0.0;
    }
    // only the tile column that the next sub-panel reads is updated now
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      acc[rt][KT0] = __builtin_amdgcn_mfma_f64_16x16x4f64(df.fa[rt][0], df.fb[KT0][0], acc[rt][KT0], 0, 0, 0);
      acc[rt][KT0] = __builtin_amdgcn_mfma_f64_16x16x4f64(df.fa[rt][1], df.fb[KT0][1], acc[rt][KT0], 0, 0, 0);
    }
  }
}

constexpr int PANEL128_ROWS = 128;  // rows of A21 per workgroup

__global__ __launch_bounds__(256) void potrf_panel128_kernel(double* __restrict__ A, int64_t lda, int rows_below, int nchunks,
                                                             double* __restrict__ inv_diag, int* __restrict__ info, int col0,
                                                             double* __restrict__ stage_out, const double* __restrict__ prev_stage,
                                                             double* __restrict__ prev_dst, int prev_pw, int64_t cs, int info_stride) {
  {
    const int64_t off = (int64_t)blockIdx.y * cs;
    A += off;
    inv_diag += off;
    stage_out += off;
    if (prev_stage) prev_stage += off;
    if (prev_dst) prev_dst += off;
    info += (int64_t)blockIdx.y * info_stride;
  }
  __shared__ __attribute__((aligned(16))) double sIn[256 * PSUB];
  __shared__ __attribute__((aligned(16))) double sX[256 * PSUB];
  Panel128Ctx c;
  c.sIn = sIn;
  c.sX = sX;
  c.tid = threadIdx.x;
  const int lane = c.tid & 63;
  c.wave = c.tid >> 6;
  c.g = lane >> 4;
  c.r = lane & 15;
  c.zero_above = c.tid < PW ? c.tid : (1 << 30);
  c.bad = 0;
  c.inv_diag = inv_diag;
  const bool last = (int)blockIdx.x == nchunks;
  c.rinv_out = last ? stage_out + PW * PW : nullptr;
  // output row of this thread
  c.out = nullptr;
  c.ident = -1;
  if (last) flush_staged_block(prev_stage, prev_dst, lda, prev_pw, c.tid);
  if (c.tid < PW) {
    if (last) c.out = stage_out + c.tid * PW;  // staged: see the in-place hazard note above
  } else if (!last) {
    const int idx = blockIdx.x * PANEL128_ROWS + (c.tid - PW);
    if (idx < rows_below) c.out = A + (int64_t)(PW + idx) * lda;
  } else {
    c.ident = c.tid - PW;
  }

  d4 acc[4][8];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int wrow = 64 * c.wave + 16 * rt + c.g + 4 * q;  // workgroup row 0..255
      const double* src = nullptr;
      if (wrow < PW) {
        src = A + (int64_t)wrow * lda;
      } else if (!last) {
        const int idx = blockIdx.x * PANEL128_ROWS + (wrow - PW);
        if (idx < rows_below) src = A + (int64_t)(PW + idx) * lda;
      }
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        const int col = kt * 16 + c.r;
        double v = 0.0;
        if (src) v = src[col];
        if (wrow < PW) {
          if (col > wrow) v = 0.0;
        } else if (last) {
          v = (col == wrow - PW) ? 1.0 : 0.0;
        }
        acc[rt][kt][q] = v;
      }
    }

  Deferred128 df;
  panel128_step<0>(acc, c, df);
  panel128_step<1>(acc, c, df);
  panel128_step<2>(acc, c, df);
  panel128_step<3>(acc, c, df);
  panel128_step<4>(acc, c, df);
  panel128_step<5>(acc, c, df);
  panel128_step<6>(acc, c, df);
  panel128_step<7>(acc, c, df);
  panel128_step<8>(acc, c, df);
  panel128_step<9>(acc, c, df);
  panel128_step<10>(acc, c, df);
  panel128_step<11>(acc, c, df);
  panel128_step<12>(acc, c, df);
  panel128_step<13>(acc, c, df);
  panel128_step<14>(acc, c, df);
  panel128_step<15>(acc, c, df);
  if (last && c.tid == 0 && c.bad != 0) atomicCAS(info, 0, col0 + c.bad);
}

// ---- split panel, 128 columns: rows only ------------------------------------------------------------------------
// potrf_panel128_kernel launched with ONE workgroup per cell factors the 128 x 128 diagonal block (staged L11, the two
// 64 x 64 blocks of L11^-1, 128 reciprocal pivots); this kernel solves the rows below it, 128 rows x 128 columns per
// workgroup.  It replaces [rows of panel c] + [K = 64 update of columns c + 64 .. c + 127] + [rows of panel c + 64] of
// the 64-column scheme with ONE pass over the 128 columns (4 instead of 7 column-block transfers per pair, 1-KiB row
// segments): the update of the second half happens through the 8-column sub-panel updates, which reach an element in the
// same ascending groups of four k as syrk_k64_kernel does, so the factor is bit-identical to the 64-column paths.
// L11 is not kept in LDS (128 KiB): the 8-column strip of L11 a step needs (its 8 x 8 diagonal sub-block for the
// substitution, the rows below it as MFMA operands) is fetched one step ahead into a double-buffered 128 x 8 image.
constexpr int R128_ROWS = 128;

template <int P>
__device__ __forceinline__ void rows128_step(d4 (&acc)[2][8], double* __restrict__ sIn, double* __restrict__ sX, double* __restrict__ sS,
                                             const double* __restrict__ sRinv, const double* __restrict__ stage, int tid, int wave, int g, int r) {
  constexpr int C0 = 8 * P;
  constexpr int KT = C0 / 16;
  constexpr int HALF = P & 1;
  double* cur = sS + (P & 1) * (PW * PSUB);         // strip of this step: cur[row * PSUB + m] = L11[row][C0 + m]
  double* nxt = sS + ((P + 1) & 1) * (PW * PSUB);
  // prefetch the next strip (rows 0..127, columns C0 + 8 .. C0 + 15): 4 doubles per thread
  d2 pf0 = d2{0.0, 0.0}, pf1 = d2{0.0, 0.0};
  if constexpr (P < 15) {
    const int row = tid >> 1, half = tid & 1;
    const double* src = stage + row * PW + C0 + 8 + 4 * half;
    pf0 = *reinterpret_cast<const d2*>(src);
    pf1 = *reinterpret_cast<const d2*>(src + 2);
  }
  if ((r >> 3) == HALF) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) sIn[(32 * wave + 16 * rt + g + 4 * q) * PSUB + (r & 7)] = acc[rt][KT][q];
  }
  __syncthreads();
  if (tid < R128_ROWS) {
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double t = sIn[tid * PSUB + k];
#pragma unroll
      for (int m = 0; m < k; ++m) t = __builtin_fma(-x[m], cur[(C0 + k) * PSUB + m], t);
      x[k] = t * sRinv[C0 + k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sX[tid * PSUB + k] = x[k];
  }
  __syncthreads();
  if constexpr (C0 + 8 < PW) {
    constexpr int KT0 = (C0 + 8) / 16;
    double fa[2][2], fb[8][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[rt][ks] = -sX[(32 * wave + 16 * rt + r) * PSUB + 4 * ks + g];
#pragma unroll
    for (int kt = KT0; kt < 8; ++kt) {
      const int kk = kt * 16 + r;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb[kt][ks] = (kk >= C0 + 8) ? cur[kk * PSUB + 4 * ks + g] : 0.0;
    }
#pragma unroll
    for (int kt = KT0; kt < 8; ++kt)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        acc[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][0], fb[kt][0], acc[rt][kt], 0, 0, 0);
        acc[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][1], fb[kt][1], acc[rt][kt], 0, 0, 0);
      }
  }
  if ((r >> 3) == HALF) {  // solved values back into the accumulators: one coalesced store pass at the end
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[rt][KT][q] = sX[(32 * wave + 16 * rt + g + 4 * q) * PSUB + (r & 7)];
  }
  if constexpr (P < 15) {  // publish the next strip (the other buffer: nobody reads it in this step)
    const int row = tid >> 1, half = tid & 1;
    double* dst = nxt + row * PSUB + 4 * half;
    dst[0] = pf0.x;
    dst[1] = pf0.y;
    dst[2] = pf1.x;
    dst[3] = pf1.y;
  }
}

// A21: first row below the 128 x 128 diagonal block (rows_below rows, lda); stage: staged L11 (128 x 128, row-major)
// followed by the 128 reciprocal pivots.  grid = (ceil(rows_below / 128), cells).
__global__ __launch_bounds__(256, 2) void potrf_rows128_kernel(double* __restrict__ A21, int64_t lda, int rows_below,
                                                               const double* __restrict__ stage, int64_t cs) {
  __shared__ __attribute__((aligned(16))) double sIn[R128_ROWS * PSUB];
  __shared__ __attribute__((aligned(16))) double sX[R128_ROWS * PSUB];
  __shared__ __attribute__((aligned(16))) double sS[2 * PW * PSUB];
  __shared__ double sRinv[PW];
  A21 += (int64_t)blockIdx.y * cs;
  stage += (int64_t)blockIdx.y * cs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
  const int row0 = blockIdx.x * R128_ROWS;
  d4 acc[2][8];
  {
    const double* rowp[2][4];
    bool valid[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = row0 + 32 * wave + 16 * rt + g + 4 * q;
        valid[rt][q] = idx < rows_below;
        rowp[rt][q] = A21 + (int64_t)(valid[rt][q] ? idx : 0) * lda + r;
      }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) acc[rt][kt][q] = rowp[rt][q][kt * 16];
    // first strip (columns 0..7 of L11) and the reciprocal pivots
    {
      const int row = tid >> 1, half = tid & 1;
      const double* src = stage + row * PW + 4 * half;
      const d2 v0 = *reinterpret_cast<const d2*>(src), v1 = *reinterpret_cast<const d2*>(src + 2);
      double* dst = sS + row * PSUB + 4 * half;
      dst[0] = v0.x;
      dst[1] = v0.y;
      dst[2] = v1.x;
      dst[3] = v1.y;
    }
    if (tid < PW) sRinv[tid] = stage[PW * PW + tid];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) acc[rt][kt][q] = valid[rt][q] ? acc[rt][kt][q] : 0.0;
  }
  rows128_step<0>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<1>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<2>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<3>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<4>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<5>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<6>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<7>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<8>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<9>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<10>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<11>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<12>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<13>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<14>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
  rows128_step<15>(acc, sIn, sX, sS, sRinv, stage, tid, wave, g, r);
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = row0 + 32 * wave + 16 * rt + g + 4 * q;
      if (idx < rows_below) {
        double* dst = A21 + (int64_t)idx * lda + r;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) dst[kt * 16] = acc[rt][kt][q];
      }
    }
}

// Optional per-launch timing of the two kernels of the factorisation (HIP events on the launch stream).
struct PotrfProfile {
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  std::vector<std::pair<size_t, double>> gemm_marks;   // launches of the main GEMM kernel (bulk HEAD / TAIL updates and in-block updates with K > 128):
                                                       // (index of start event, algorithmic flops)
  std::vector<std::pair<size_t, double>> strip_marks;  // short-K in-block updates (K = 64: syrk_k64_kernel, K = 128: GEMM with C prefetch)
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
    strip_marks.clear();
    panel_marks.clear();
  }
  ~PotrfProfile() {
    for (auto e : pool) hipEventDestroy(e);
  }
};

// run-time tuning knobs (gprx_set_tuning): 0 = default heuristics
struct PotrfTuning {
  int panel_width = 0;   // 64 or 128
  int outer_block = 0;   // multiple of 128
  int update_tile = 0;   // tile of the TAIL GEMM: 64 or 128
  int no_lookahead = 0;  // 1: everything on the main stream (debugging)
  int panel_rows = 0;    // rows per panel workgroup: 128 (default) or 256
  int panel_occ = 0;     // 3: panel kernel compiled for 3 workgroups per CU (168 registers, small spills) instead of 2
  int inblock = 0;       // 1: right-looking K = 64 strips inside an outer block instead of the recursive halving
  int split_panel = 0;   // 1: always the split panel (diagonal workgroup + rows kernel), -1: never, 0: from 24 cells per launch on
  int poison_workspace = 0;  // testing: the gradient's L^-1 workspace starts as NaN patterns (nothing may depend on its old contents)
  int split_updates = 0; // lone matrix: 1 = look-ahead split of the K >= 256 updates over a side stream (see potrf_lower; measured
                         // slower: 2.17 -> 2.66 ms at N = 4096), 0 = every update whole on the main stream (default)
  int cell_kernel = 0;   // batched cells: 1 = always one workgroup per cell (potrf_cell.h), -1 never, 0 = for np <= 1024 and >= 256 cells
  int rhs_vector = 0;    // batched cells, split panel: -1 = the right-hand side always rides as a 64-row tile; 0 / 1 = as a vector where the schedule knows it
  int rows_inv = 0;      // split panel: 1 = rows below the diagonal block by one MFMA tile product against L11^-1 (potrf_rows_inv_kernel), 0 / -1 = substitution
  int rows_inv_rt = 0;   // 16-row tiles per wave of that kernel: 1 (default) or 2
  int rows_inv_lone = 0; // 1 = a lone matrix takes the split panel + rows_inv too (experiments at N >= 8192)
  int dag = 0;           // lone matrices: 1 = the tile-DAG factorisation (potrf_dag.h); 0 / -1 = the launch-per-panel schedule (default)
};
inline PotrfTuning& potrf_tuning() {
  static PotrfTuning t = [] {
    PotrfTuning v;
    if (const char* e = getenv("GPRX_OUTER_BLOCK")) v.outer_block = atoi(e);  // experiments without recompiling callers
    if (const char* e = getenv("GPRX_UPDATE_TILE")) v.update_tile = atoi(e);
    if (const char* e = getenv("GPRX_PANEL_ROWS")) v.panel_rows = atoi(e);
    if (const char* e = getenv("GPRX_PANEL_WIDTH")) v.panel_width = atoi(e);
    if (const char* e = getenv("GPRX_PANEL_OCC")) v.panel_occ = atoi(e);
    if (const char* e = getenv("GPRX_INBLOCK")) v.inblock = atoi(e);
    if (const char* e = getenv("GPRX_SPLIT_PANEL")) v.split_panel = atoi(e);
    if (const char* e = getenv("GPRX_DAG")) v.dag = atoi(e);
    if (const char* e = getenv("GPRX_RHS_VECTOR")) v.rhs_vector = atoi(e);
    if (const char* e = getenv("GPRX_ROWS_INV")) v.rows_inv = atoi(e);
    if (const char* e = getenv("GPRX_ROWS_INV_RT")) v.rows_inv_rt = atoi(e);
    if (const char* e = getenv("GPRX_ROWS_INV_LONE")) v.rows_inv_lone = atoi(e);
    if (const char* e = getenv("GPRX_SPLIT_UPDATES")) v.split_updates = atoi(e);
    if (const char* e = getenv("GPRX_CELL_KERNEL")) v.cell_kernel = atoi(e);
    return v;
  }();
  return t;
}

// Streams and events of the look-ahead schedule (owned by the caller, reused across factorisations).
struct PotrfStreams {
  hipStream_t aux = nullptr;
  hipStream_t side = nullptr;  // look-ahead inside a block: the parts of an update that the next panel does not need yet
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  hipEvent_t block_done = nullptr, tail_done = nullptr;
  hipEvent_t next_event() {
    if (used == pool.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[used++];
  }
  hipError_t init() {
    hipError_t e = hipStreamCreateWithFlags(&aux, hipStreamNonBlocking);  // normal priority (see gprx_create)
    if (e != hipSuccess) return e;
    if ((e = hipStreamCreateWithFlags(&side, hipStreamNonBlocking)) != hipSuccess) return e;
    if ((e = hipEventCreateWithFlags(&block_done, hipEventDisableTiming)) != hipSuccess) return e;
    return hipEventCreateWithFlags(&tail_done, hipEventDisableTiming);
  }
  void destroy() {
    if (aux) hipStreamDestroy(aux);
    if (side) hipStreamDestroy(side);
    for (auto e : pool) hipEventDestroy(e);
    pool.clear();
    used = 0;
    if (block_done) hipEventDestroy(block_done);
    if (tail_done) hipEventDestroy(tail_done);
    aux = side = nullptr;
    block_done = tail_done = nullptr;
  }
};

// Can potrf_lower carry ONE right-hand side as a vector (yvec; then extra = 0) under this tuning and batch size?  Only the default split
// panel of batched cells knows the form (the diagonal workgroup + potrf_beta_block_kernel + potrf_rows_kernel<..., YVEC>); a lone or
// small-batch factorisation (fused panel) keeps the 64-row tile, whose arithmetic is the one single calls are bit-identical in.
// "rhs_vector": 1 / 0 = where possible (default), -1 = never.
inline bool potrf_rhs_vector_ok(const PotrfTuning& tune, int batch) {
  static const bool rows_lds = getenv("GPRX_ROWS_LDS") != nullptr;
  const bool split_panel = tune.split_panel ? tune.split_panel > 0 : batch >= 24;
  return tune.rhs_vector >= 0 && split_panel && batch > 1 && !rows_lds && tune.rows_inv <= 0 && tune.panel_width != PW && tune.panel_rows != 256 &&
         tune.panel_occ != 3 && tune.inblock != 1;
}

// Factor the (np x np) matrix in place; `extra` rows below it are carried as right-hand sides.
// inv_diag: np/64 blocks of 64 x 64.  info (device int) must be zeroed by the caller.
//
// Two-level right-looking schedule with look-ahead.  Panels are 64 columns wide by default (one launch
// factors the diagonal block and solves all rows below it; the 128-column kernel is selectable with
// gprx_set_tuning and measured slower: its MFMA updates serialise behind the scalar factor chain).
// Outer blocks are `ob` columns (1024 up to n = 4096, else 512, measured): the bulk trailing updates
// run with K = ob, i.e. n / ob passes over the trailing matrix instead of n / 64; inside a block the
// K = 64 strip kernel updates the block's remaining columns after every panel.  For each outer block J =
// columns [C, C + w):
//   main stream: its panels, each followed by the strip update of the block's remaining columns;
//                then HEAD(J): the update of the NEXT block's columns by block J (K = w);
//   aux stream : TAIL(J): the update of every column right of the next block (K = w, the bulk of the
//                flops), overlapping the next block's panel chain on the main stream.
// Order: TAIL(J) waits for HEAD(J) to be enqueued behind block J (event) and follows TAIL(J-1)
// (stream order); HEAD(J) waits for TAIL(J-1), the last writer of the next block's columns.
// diag_stage: scratch of np * STAGE_LD doubles (staged diagonal blocks and reciprocal pivots, see potrf_panel_kernel)
// col_base: added to the failing-pivot index reported through `info` (the matrix is a diagonal block of a larger one).
// batch > 1: `batch` matrices at A + c * cs (inv_diag and diag_stage likewise: all live in cell blocks `cs` doubles apart),
// info words info_stride ints apart; every launch carries the cell index in blockIdx.y.
inline hipError_t potrf_lower(hipStream_t st, double* A, int64_t lda, int np, int extra, double* inv_diag, int* info,
                              double* diag_stage, PotrfProfile* prof = nullptr, PotrfStreams* ps = nullptr, int batch = 1, int64_t cs = 0,
                              int info_stride = 0, const PotrfTuning* tune_in = nullptr, int col_base = 0, hipEvent_t first_block_evt = nullptr,
                              double* yvec = nullptr) {
  const double* prev_stage = nullptr;
  double* prev_dst = nullptr;
  int prev_pw = 0;
  auto mark_gemm = [&](hipStream_t s, int ncols_lower, int rows_rect, int ncols, int k, bool strip = false) {
    if (!prof) return;
    // algorithmic flops: 2 K per updated element (lower triangle incl. diagonal of the square part + rectangle)
    const double elems = 0.5 * (double)ncols_lower * (ncols_lower + 1) + (double)rows_rect * ncols;
    (strip ? prof->strip_marks : prof->gemm_marks).push_back({prof->used, 2.0 * k * elems * batch});
    hipEventRecord(prof->next(), s);
  };
  auto mark_end = [&](hipStream_t s) {
    if (prof) hipEventRecord(prof->next(), s);
  };
  const int total_rows = np + extra;
  const PotrfTuning& tune = tune_in ? *tune_in : potrf_tuning();  // a handle's own knobs, or the process defaults
  if (tune.no_lookahead) ps = nullptr;
  const int ob = tune.outer_block ? tune.outer_block : (np > 4096 ? 512 : 1024);  // measured: N=2048/4096 -> 1024, N=8192/16384 -> 512
  // panel width: 64.  128 (gprx_set_tuning) selects the 128-column kernels -- fused, or in split mode one diagonal
  // workgroup + potrf_rows128_kernel per 128 columns; both bit-identical to the 64-column paths and both measured
  // slower (64 cells of N = 4096: 203 us per 128 columns against 173 us: the 128 x 128 diagonal workgroup takes 100 us
  // and the rows kernel, 16 dependent sub-panel steps at 2 workgroups per CU, is latency- rather than HBM-bound)
  const int pwidth = tune.panel_width ? tune.panel_width : NB;
  // bulk-update tile: batched cells fill the chip with 64 x 64 tiles already (4 workgroups per CU hide the C
  // read-modify-write; measured 1529 vs 1513 fits/s at 16 cells of N = 4096); a single matrix lets launch_gemm choose
  // (a single matrix too since the 64 x 64 kernel takes its operands by LDS-DMA: N = 16384 30.7 ms against 32.4 ms with the
  // 128 x 128 tile, whose ragged row counts keep it on the register-staged kernel)
  const int bulk_tile = tune.update_tile ? tune.update_tile : 64;
  bool tail_pending = false;
  // 0 (default): ordinary launches.  Measured dead end: a persistent TAIL grid of 2 / 3 / 4 workgroups per CU (one slot of
  // every CU left to the chain) gave N = 16384 31.0 / 30.7 / 31.5 ms against 30.05 ms, N = 8192 6.44 / 6.51 / 6.55 against 6.35
  static const int tail_slots = getenv("GPRX_TAIL_SLOTS") ? atoi(getenv("GPRX_TAIL_SLOTS")) : 0;
  hipError_t err = hipSuccess;
  // split panel (diagonal workgroup, then a rows-only kernel): pays one more dependent launch per panel and wins once
  // a fused launch would fill the chip with redundant factorisations; bit-identical either way
  const bool rows_inv = tune.rows_inv > 0;  // split panel: rows by one tile product against L11^-1 (potrf_rows_inv_kernel) instead of the substitution
  const int rows_inv_rt = tune.rows_inv_rt == 2 ? 2 : 1;  // 16-row tiles per wave of that kernel (64 or 128 rows per workgroup)
  const bool split_panel = tune.split_panel ? tune.split_panel > 0 : (batch >= 24 || (rows_inv && tune.rows_inv_lone > 0));  // measured at N = 4096: -2 % at 16 cells per launch, +5 % at 32
  // one panel: factor the diagonal block at column c and solve every row below it
  // fuse: the K = 64 update of these 64 columns by the 64 columns left of them happens inside the panel kernel (lone
  // matrices: one dependent launch less); only where the separate launch would be the general NT kernel (same arithmetic)
  static const bool fuse_ok = [] {
    const char* a = getenv("GPRX_K64_GEMM");
    const char* b = getenv("GPRX_GEMM_DMA");
    const char* c = getenv("GPRX_FUSE_K64");
    return !(a && atoi(a) == 0) && !(b && atoi(b) == 0) && !(c && atoi(c) == 0);
  }();
  // ---- look-ahead inside and between blocks (ONE matrix on two extra streams) ------------------------------------------------
  // An update of n > 64 columns with K >= 256 is needed by the NEXT panel only in its first 64 columns; the rest is needed one
  // panel, three panels, seven panels ... later.  So the update is split by columns: [0, 64) stays on the main stream, the rest
  // goes to the side stream in dyadic pieces [64,128), [128,256), [256,512), ... each followed by an event; the main stream waits
  // for a piece only before its first launch that touches those columns.  The chain (panel, fused K = 64 update, panel, ...) no
  // longer carries the K = 256 / 512 in-block updates (17 / 43 us each at N = 4096) nor the HEAD updates (113 us) whole.  Same
  // tiles, same K range per tile, same arithmetic: the factor is bit-identical to the unsplit schedule (tested).
  // OPT-IN ("split_updates" = 1): measured on MI355X it LOSES -- N = 4096 2.17 -> 2.66 ms, N = 8192 6.2 -> 7.0, N = 16384 29.5 ->
  // 30.5: about 40 cross-stream event edges per factorisation, and every piece that runs beside a panel slows that panel the way
  // the TAIL update does at N = 16384 (81 us instead of 20).  Capping the update kernel at 3 or 2 workgroups per CU
  // (GPRX_GEMM_PAD_LDS, so that a 200-224-register panel workgroup always finds room) changes neither number: the slow-down of a
  // panel beside a bulk update is not a residency effect.
  struct PendingCols {
    int begin, end;
    hipEvent_t ev;
  };
  std::vector<PendingCols> pending;
  const bool split_ok = ps && ps->side && batch == 1 && !prof && tune.split_updates > 0 && tune.inblock != 1;
  if (split_ok) ps->used = 0;
  auto main_wait = [&](int a, int b) {  // the main stream is about to touch columns [a, b)
    for (size_t q = 0; q < pending.size();) {
      if (pending[q].begin < b && pending[q].end > a) {
        hipStreamWaitEvent(st, pending[q].ev, 0);
        pending[q] = pending.back();
        pending.pop_back();
      } else {
        ++q;
      }
    }
  };
  auto panel = [&](int c, int pw, bool fuse = false) {
    main_wait(c, c + pw);
    const int rows_below = total_rows - c - pw;
    double* Acc = A + (int64_t)c * lda + c;
    if (prof) {
      prof->panel_marks.push_back(prof->used);
      hipEventRecord(prof->next(), st);
    }
    double* stage_out = diag_stage + (int64_t)c * STAGE_LD;
    double* invd = inv_diag + (int64_t)(c / NB) * NB * NB;
    if (pw == PW && split_panel) {
      hipLaunchKernelGGL(potrf_panel128_kernel, dim3(1, batch), dim3(256), 0, st, Acc, lda, 0, 0, invd, info, col_base + c, stage_out, prev_stage, prev_dst,
                         prev_pw, cs, info_stride);
      if (rows_below > 0)
        hipLaunchKernelGGL(potrf_rows128_kernel, dim3((rows_below + R128_ROWS - 1) / R128_ROWS, batch), dim3(256), 0, st,
                           Acc + (int64_t)PW * lda, lda, rows_below, (const double*)stage_out, cs);
    } else if (pw == PW) {
      const int nchunks = (rows_below + PANEL128_ROWS - 1) / PANEL128_ROWS;
      hipLaunchKernelGGL(potrf_panel128_kernel, dim3(nchunks + 1, batch), dim3(256), 0, st, Acc, lda, rows_below, nchunks, invd, info, col_base + c,
                         stage_out, prev_stage, prev_dst, prev_pw, cs, info_stride);
    } else if (tune.panel_rows == 256) {
      const int own = PanelGeom<4>::kOwnRows;
      const int nchunks = (rows_below + own - 1) / own;
      hipLaunchKernelGGL((potrf_panel_kernel<4, 2>), dim3(nchunks + 1, batch), dim3(256), 0, st, Acc, lda, rows_below, nchunks, invd, info, col_base + c,
                         stage_out, prev_stage, prev_dst, prev_pw, cs, info_stride);
    } else if (split_panel) {
      // diagonal block: one workgroup per cell (the `last` role of the panel kernel: L11 staged, L11^-1, pivots) ...
      // (right-hand side as a vector: beta_j = L11^-1 y_j at the end of the diagonal workgroup -- or, GPRX_BETA_BLOCK_KERNEL=1, by a launch
      // of its own: same sums --, then the rows kernel takes L21 beta_j off the entries below)
      static const bool beta_launch = getenv("GPRX_BETA_BLOCK_KERNEL") && atoi(getenv("GPRX_BETA_BLOCK_KERNEL")) != 0;
      const bool yv_on = yvec != nullptr;
      double* yv = yv_on ? yvec + c : nullptr;
      double* yv_diag = (yv_on && !beta_launch) ? yv : nullptr;
      if (fuse)
        hipLaunchKernelGGL((potrf_panel_kernel<2, 2, true>), dim3(1, batch), dim3(256), 0, st, Acc, lda, 0, 0, invd, info, col_base + c, stage_out,
                           prev_stage, prev_dst, prev_pw, cs, info_stride, yv_diag);
      else
        hipLaunchKernelGGL((potrf_panel_kernel<2, 2>), dim3(1, batch), dim3(256), 0, st, Acc, lda, 0, 0, invd, info, col_base + c, stage_out, prev_stage,
                           prev_dst, prev_pw, cs, info_stride, yv_diag);
      if (yv_on && beta_launch) hipLaunchKernelGGL(potrf_beta_block_kernel, dim3(1, batch), dim3(256), 0, st, (const double*)invd, yv, cs);
      // ... then the rows below it, 128 per workgroup
      if (rows_below > 0) {
        // default: no L11 image in LDS (the 8 x 8 diagonal sub-blocks and pivots through scalar loads, the MFMA operands straight
        // from the staged block, L1 / L2 resident): 18 KB of LDS and 116 VGPRs, four workgroups per CU instead of three.  The
        // kernel is bound by its own dependent chain (8 sub-panels x (LDS round trip, 44-FMA substitution, MFMA update)), not by
        // bandwidth (3.3 TB/s) -- rows in flight per CU are what counts: -0.5 ms per 128-cell step at N = 4096.  Measured without
        // effect: 256-row workgroups, scalar loads alone.  With nothing shared in LDS a wave substitutes its own 32 rows and the
        // workgroup barriers become wave-scope fences (another -0.3 ms).  GPRX_ROWS_LDS=1 restores the LDS image and the barriers
        // (same values either way).
        static const bool rows_lds = getenv("GPRX_ROWS_LDS") != nullptr;
        const dim3 grid_inv((rows_below + 64 * rows_inv_rt - 1) / (64 * rows_inv_rt), batch);
        if (rows_inv && rows_inv_rt == 1 && fuse)
          hipLaunchKernelGGL((potrf_rows_inv_kernel<1, 4, true>), grid_inv, dim3(256), 0, st, Acc + (int64_t)NB * lda, lda, rows_below, (const double*)invd, cs);
        else if (rows_inv && rows_inv_rt == 1)
          hipLaunchKernelGGL((potrf_rows_inv_kernel<1, 4>), grid_inv, dim3(256), 0, st, Acc + (int64_t)NB * lda, lda, rows_below, (const double*)invd, cs);
        else if (rows_inv && fuse)
          hipLaunchKernelGGL((potrf_rows_inv_kernel<2, 2, true>), grid_inv, dim3(256), 0, st, Acc + (int64_t)NB * lda, lda, rows_below, (const double*)invd, cs);
        else if (rows_inv)
          hipLaunchKernelGGL((potrf_rows_inv_kernel<2, 2>), grid_inv, dim3(256), 0, st, Acc + (int64_t)NB * lda, lda, rows_below, (const double*)invd, cs);
        else if (fuse && yv_on)
          hipLaunchKernelGGL((potrf_rows_kernel<2, 2, true, true, true, true, true>), dim3((rows_below + ROWS_WG - 1) / ROWS_WG, batch), dim3(256), 0, st,
                             Acc + (int64_t)NB * lda, lda, rows_below, (const double*)stage_out, cs, yv);
        else if (yv_on)
          hipLaunchKernelGGL((potrf_rows_kernel<2, 4, true, true, true, false, true>), dim3((rows_below + ROWS_WG - 1) / ROWS_WG, batch), dim3(256), 0, st,
                             Acc + (int64_t)NB * lda, lda, rows_below, (const double*)stage_out, cs, yv);
        else if (fuse)
          hipLaunchKernelGGL((potrf_rows_kernel<2, 2, true, true, true, true>), dim3((rows_below + ROWS_WG - 1) / ROWS_WG, batch), dim3(256), 0, st,
                             Acc + (int64_t)NB * lda, lda, rows_below, (const double*)stage_out, cs);
        else if (rows_lds)
          hipLaunchKernelGGL((potrf_rows_kernel<2, 3, false, false>), dim3((rows_below + ROWS_WG - 1) / ROWS_WG, batch), dim3(256), 0, st,
                             Acc + (int64_t)NB * lda, lda, rows_below, (const double*)stage_out, cs);
        else
          hipLaunchKernelGGL((potrf_rows_kernel<2, 4, true, true, true>), dim3((rows_below + ROWS_WG - 1) / ROWS_WG, batch), dim3(256), 0, st,
                             Acc + (int64_t)NB * lda, lda, rows_below, (const double*)stage_out, cs);
      }
    } else {
      const int own = PanelGeom<2>::kOwnRows;
      const int nchunks = (rows_below + own - 1) / own;
      if (tune.panel_occ == 3)
        hipLaunchKernelGGL((potrf_panel_kernel<2, 3>), dim3(nchunks + 1, batch), dim3(256), 0, st, Acc, lda, rows_below, nchunks, invd, info,
                           col_base + c, stage_out, prev_stage, prev_dst, prev_pw, cs, info_stride);
      else if (fuse)
        hipLaunchKernelGGL((potrf_panel_kernel<2, 2, true>), dim3(nchunks + 1, batch), dim3(256), 0, st, Acc, lda, rows_below, nchunks, invd, info,
                           col_base + c, stage_out, prev_stage, prev_dst, prev_pw, cs, info_stride);
      else
        hipLaunchKernelGGL((potrf_panel_kernel<2, 2>), dim3(nchunks + 1, batch), dim3(256), 0, st, Acc, lda, rows_below, nchunks, invd, info,
                           col_base + c, stage_out, prev_stage, prev_dst, prev_pw, cs, info_stride);
    }
    prev_stage = stage_out;
    prev_dst = Acc;
    prev_pw = pw;
    if (prof) hipEventRecord(prof->next(), st);
  };
  // in-block update: columns [c1, c1 + n) and every row from c1 down, by the k columns [c0, c0 + k) factored before
  auto update_on = [&](hipStream_t s_, int c0, int k, int c1, int n, int tile) {
    const int rows = total_rows - c1;
    const double* L21 = A + (int64_t)c1 * lda + c0;
    double* A22 = A + (int64_t)c1 * lda + c1;
    mark_gemm(s_, n, rows - n, n, k, k <= 128);  // K <= 128 runs the short-K kernels (syrk_k64 / C-prefetch GEMM), longer K the main GEMM kernel
    hipError_t e = (k == NB) ? launch_update_k64(s_, rows, n, L21, lda, A22, lda, batch, cs)
                             : launch_gemm(s_, 0, 1, rows, n, k, -1.0, L21, lda, L21, lda, 1.0, A22, lda, GEMM_C_LOWER, tile, batch, cs, cs, cs);
    mark_end(s_);
    if (e != hipSuccess && err == hipSuccess) err = e;
  };
  // the update as a whole on the main stream, or split (see above): `side_wait` = an event the side stream must see first (the
  // last writer of these columns on another stream), may be null
  auto split_update = [&](int c0, int k, int c1, int n, int tile, hipEvent_t side_wait) {
    if (!split_ok || n <= NB || k < 256) {
      main_wait(c1, c1 + n);
      if (side_wait) hipStreamWaitEvent(st, side_wait, 0);
      update_on(st, c0, k, c1, n, tile);
      return;
    }
    hipEvent_t ready = ps->next_event();  // the k columns [c0, c0 + k) are final on the main stream
    hipEventRecord(ready, st);
    hipStreamWaitEvent(ps->side, ready, 0);
    if (side_wait) hipStreamWaitEvent(ps->side, side_wait, 0);
    for (int start = NB; start < n;) {  // pieces [64,128), [128,256), [256,512), ...: a piece is as wide as everything left of it
      const int wdt = n - start < start ? n - start : start;
      update_on(ps->side, c0, k, c1 + start, wdt, tile);
      hipEvent_t done = ps->next_event();
      hipEventRecord(done, ps->side);
      pending.push_back({c1 + start, c1 + start + wdt, done});  // (older pieces over the same columns stay listed: all are waited for)
      start += wdt;
    }
    main_wait(c1, c1 + NB);
    if (side_wait) hipStreamWaitEvent(st, side_wait, 0);
    update_on(st, c0, k, c1, NB, tile);
  };
  auto inblock_update = [&](int c0, int k, int c1, int n) { split_update(c0, k, c1, n, 64, nullptr); };
  // Inside an outer block the panels are combined recursively: factor the left half, update the right half with
  // it (K = half the width), factor the right half.  The block's columns are rewritten log2(w / 64) times
  // instead of w / 64 times (a K = 64 update moves 16 bytes of C per 128 flops -- the strips were HBM-bound once
  // many cells are batched), and most in-block flops run at K >= 128.  tune.inblock == 1: the old right-looking
  // strips (every panel followed by a K = 64 update of all remaining columns of the block).
  auto factor_range = [&](auto&& self, int c0, int w) -> void {
    const int base = (pwidth == PW && w == PW) ? PW : NB;
    if (w <= base) {
      panel(c0, w);
      return;
    }
    const int h = ((w / NB + 1) / 2) * NB;
    self(self, c0, h);
    // (split panels fuse too since round 3, GPRX_FUSE_K64_SPLIT=0 restores the separate launch: the update is HBM-bound there)
    static const bool fuse_split = !(getenv("GPRX_FUSE_K64_SPLIT") && atoi(getenv("GPRX_FUSE_K64_SPLIT")) == 0);
    if (h == NB && w - h == NB && pwidth != PW && (!split_panel || fuse_split) && fuse_ok && tune.panel_rows != 256 && tune.panel_occ != 3) {
      panel(c0 + h, NB, true);  // the K = 64 update of the right panel rides in its own kernel
      return;
    }
    inblock_update(c0, h, c0 + h, w - h);
    self(self, c0 + h, w - h);
  };
  for (int C = 0; C < np; C += ob) {
    const int w = (np - C < ob) ? np - C : ob;
    if (tune.inblock == 1) {
      for (int c = C; c < C + w; c += NB) {
        panel(c, NB);
        const int strip = C + w - c - NB;
        if (strip > 0) inblock_update(c, NB, c + NB, strip);
      }
    } else {
      factor_range(factor_range, C, w);
    }
    if (err != hipSuccess) return err;
    // (phase-staggered cell groups: another group's stream starts its factorisation when this one leaves its first in-block phase)
    if (C == 0 && first_block_evt) hipEventRecord(first_block_evt, st);
    const int R = C + w;  // first column right of this block
    if (R >= np) break;
    const int wn = (np - R < ob) ? np - R : ob;     // width of the next block
    const double* Lpan = A + (int64_t)R * lda + C;  // L[R:, C:C+w]
    // HEAD(J): columns [R, R + wn), rows [R, total_rows); the last writer of these columns is TAIL(J-1) on the aux stream
    split_update(C, w, R, wn, batch > 1 ? bulk_tile : 64, (ps && tail_pending) ? ps->tail_done : nullptr);
    if (err != hipSuccess) return err;
    if (ps) hipEventRecord(ps->block_done, st);
    // TAIL(J): columns [R + wn, np), rows [R + wn, total_rows)
    const int R2 = R + wn;
    if (R2 < np) {
      hipStream_t ts = ps ? ps->aux : st;
      if (ps) hipStreamWaitEvent(ts, ps->block_done, 0);
      const int rows = total_rows - R2, cols = np - R2;
      const double* Lrow = A + (int64_t)R2 * lda + C;  // L[R2:, C:C+w]
      mark_gemm(ts, cols, rows - cols, cols, w);
      // (option GPRX_TAIL_SLOTS: beside the panel chain of the next block the bulk update can run from a persistent grid that
      // leaves workgroup slots of every CU to the chain's small kernels)
      const int persist = (ps && batch == 1) ? tail_slots : 0;
      hipError_t e = launch_gemm(ts, 0, 1, rows, cols, w, -1.0, Lrow, lda, Lrow, lda, 1.0, A + (int64_t)R2 * lda + R2, lda, GEMM_C_LOWER,
                                   bulk_tile, batch, cs, cs, cs, 1, 0, 0, 0, nullptr, 0, persist);
      mark_end(ts);
      if (e != hipSuccess) return e;
      if (ps) {
        hipEventRecord(ps->tail_done, ts);
        tail_pending = true;
      }
    }
  }
  main_wait(0, 1 << 30);  // (nothing is left in flight on the side stream)
  // the last panel's diagonal block is still staged
  if (prev_stage) hipLaunchKernelGGL(copy_block_kernel, dim3(batch), dim3(256), 0, st, prev_stage, prev_dst, lda, prev_pw, cs);
  // everything later on `st` must see the aux stream's last update
  if (ps && tail_pending) hipStreamWaitEvent(st, ps->tail_done, 0);
  return hipGetLastError();
}

}  // namespace gprx
