// One workgroup factors one whole matrix: the batched Cholesky for SMALL matrices (N <= 1024) in MANY cells.
//
// Why: at N = 1024 the launch-per-panel schedule of potrf_lower runs everything at short K -- the whole matrix is one outer
// block: 63 k fits/s = 0.29 of the fp64 MFMA peak with 512 cells per launch sequence -- and every one of
// its ~100 launches moves the panel through HBM.  A cell of N = 1024 is 8 MB: one workgroup can own it from the first column to
// the last with NO inter-workgroup dependency, two workgroups per CU hiding each other's dependent chains:
//   for block column j:  A(i,j) -= sum_{c<j} L(i,c) L(j,c)^T for i >= j (left-looking: ONE pass with K = 64 j per tile, four row
//                        tiles per pass sharing the B operand -- dag_panel of potrf_dag.h);
//                        the diagonal block through the eight sub-panel steps of the tile-DAG chain (L(j,j) and its inverse);
//                        L(i,j) = A(i,j) L(j,j)^-T for i > j (tile products against the inverse).
// Each element of the lower triangle is read and written once per block column it belongs to plus once per use as an operand
// -- all of it from this CU's L1 / the XCD's L2 (plain accesses: nobody else touches the cell).
// Arithmetic: the same tile products and the same sub-panel substitution as the other schedules, but a tile receives its whole
// update as one sum (C - sum, one rounding) and rows are solved against the explicit 64 x 64 inverse: results agree with
// potrf_lower to rounding (tested), not bit for bit.
#pragma once
#include "kmat.h"
#include "tile_ops.h"

namespace gprx {

// Row tiles per fused pass (they share the B operand; 32 accumulator registers each).  Measured (tools/cell_ni_sweep.sh, 512 cells):
// 4 tiles 70.7 k fits/s at N = 1024 / 290 k at N = 512 (the kernel sits at the 256-register cap and spills 372 B per lane), 3 tiles
// 72.2 k / 305 k, 2 tiles 71.5 k / 306 k (228 B of scratch, all in cell_diag), 6 tiles 55 k / 232 k.  A second operand set in flight
// (two loads ahead) on top of 2 or 3 tiles: no gain (70.7 k / 302 k) -- the kernel is not waiting for its loads.
#ifndef GPRX_CELL_NI
#define GPRX_CELL_NI 2
#endif
constexpr int CELL_NI = GPRX_CELL_NI;

// The barrier between two phases of a cell's workgroup.  Phases hand data to each other THROUGH GLOBAL MEMORY (one wave stores a tile,
// another wave loads it, or fetches it by LDS-DMA, in the next phase), so the barrier comes after every wave's stores have been
// acknowledged and its loads have returned: an explicit s_waitcnt vmcnt(0).  __syncthreads() does not promise that wait -- for a
// workgroup-scope release the compiler may leave vmcnt alone on this target (the waves of a workgroup share one L1, whose in-order
// handling is taken to be enough); it was seen without one behind the chain's inverse stores.  (That was NOT the cause of the round-4
// wrong results at two workgroups per CU -- see store_inverse_block in tile_ops.h -- but the hand-off should not rest on it.)
__device__ __forceinline__ void cell_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// threadIdx.x behind an opaque move: everything a phase derives from it (lane offsets, swizzles, buffer offsets) is computed INSIDE the
// phase.  Without it the compiler hoists those lane-dependent values of every phase out of the kernel's column loop and keeps them all
// live at once: the column-pair kernel, whose phases fit into 252 / 177 / 156 / 190 registers one by one, came out with 340 B of scratch.
__device__ __forceinline__ int cell_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// GPRX_CELL_ACC (development builds, tools/cell_acc.sh): phase durations of workgroup 0 of potrf_cell2_kernel in s_memrealtime ticks
// (100 MHz), summed over launches: [0] launches, [1] the diagonal pair's streaming product, [2] the rest of the diagonal pair (tile
// fetches, the two chains, the solve between them), [3] the beta steps, [4] a row group's streaming product, [5] E1, [6] E2, [7] E3.
#ifdef GPRX_CELL_ACC
__device__ unsigned long long g_cell_acc[16];
#define CACC(i)                                                                  \
  {                                                                              \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                   \
      const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();            \
      if ((i) > 0) atomicAdd(&g_cell_acc[i], t_ - s_cacc_prev);                  \
      else atomicAdd(&g_cell_acc[0], 1ull);                                      \
      s_cacc_prev = t_;                                                          \
    }                                                                            \
  }
__shared__ unsigned long long s_cacc_prev;
#else
#define CACC(i)
#endif

struct CellArgs {
  double* A;         // cell 0; cells are `cs` doubles apart
  int64_t lda;
  int T, R;          // 64-column blocks, 64-row blocks (T + right-hand-side blocks)
  double* inv_diag;  // cell 0 (T blocks of 64 x 64), cs apart
  int* info;         // cell 0, info_stride ints apart
  int64_t cs;
  int info_stride;
  int col_base;
  // kernel matrix built INSIDE the column-pair kernel (potrf_cell2_kernel<true>): the training inputs (n x d, shared by the cells) and the
  // cells' parameter table (kmat.h CELL_PAR layout: [0] variance, [1] noise, [8 .. 8 + d) lengthscales), CELL_PAR doubles per cell
  const double* X = nullptr;
  const double* cell_par = nullptr;
  int n = 0, d = 0;
  // column-pair kernel, R == T: beta = L^-1 y carried as a VECTOR (cell2_beta_*): cell 0's right-hand-side row (y on entry, beta on
  // exit; cs apart), nullptr = the right-hand side rides as a 64-row tile below the matrix (R = T + 1)
  double* beta = nullptr;
};

// the diagonal block (j, j): chain_step<0..7> on [64 diagonal rows | 64 identity rows] -> L(j,j) in place, L(j,j)^-1 to inv_diag
__device__ __forceinline__ int cell_diag(double* __restrict__ Ajj, int64_t lda, double* __restrict__ inv, double* __restrict__ smem) {
  ChainCtx c;
  c.sIn = smem;
  c.sX = smem + 128 * PSUB;
  double* sT = smem + 2 * 128 * PSUB;
  c.tid = cell_tid();
  const int lane = c.tid & 63;
  c.wave = c.tid >> 6;
  c.g = lane >> 4;
  c.r = lane & 15;
  c.bad = 0;
  const int wave = c.wave, g = c.g, r = c.r, tid = c.tid;
  const unsigned ldb = (unsigned)lda * 8u;
  const unsigned off_cd = (unsigned)(16 * wave + g) * ldb + (unsigned)r * 8u;
  const __amdgpu_buffer_rsrc_t rs = dag_rsrc(Ajj);
  d4 acc[2][4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * wave + g + 4 * q, col = 16 * kt + r;
      const double v = ld1_sc1<false>(rs, off_cd, (unsigned)(4 * q) * ldb + (unsigned)kt * 128u);
      acc[0][kt][q] = (col > row) ? 0.0 : v;
      acc[1][kt][q] = (col == row) ? 1.0 : 0.0;
    }
  chain_step<0>(acc, c);
  chain_step<1>(acc, c);
  chain_step<2>(acc, c);
  chain_step<3>(acc, c);
  chain_step<4>(acc, c);
  chain_step<5>(acc, c);
  chain_step<6>(acc, c);
  chain_step<7>(acc, c);
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      st1_sc1<false>(rs, off_cd, acc[0][kt][q], (unsigned)(4 * q) * ldb + (unsigned)kt * 128u);
      sT[(16 * kt + r) * DAG_T_LD + 16 * wave + g + 4 * q] = acc[1][kt][q];
    }
  lds_barrier();
  const __amdgpu_buffer_rsrc_t ri = dag_rsrc(inv);
  store_inverse_block<false>(ri, sT, tid);
  return c.bad;
}

// Tiles (i0 .. i0+ni-1, j), all below the diagonal tile, in ONE pass: C_t <- (C_t - sum_{c<j} L(i,c) L(j,c)^T) L(j,j)^-T.  The
// update sum as dag_panel<false> forms it (same steps, same accumulators); the updated tile then goes from the accumulator layout
// straight into the A stage image and is multiplied by the inverse (B image, loaded once per call) exactly as dag_panel<true>
// does with the tile it re-reads from memory -- same operands, same products: the factor is the two-pass kernel's bit for bit,
// with one store and one load of every tile less (8.7 of ~44 MB per N = 1024 cell).
__device__ __forceinline__ void cell_panel_fused(const TileCtx& p, int i0, int ni, int j, double* __restrict__ smem) {
  const int tid = cell_tid(), lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, r = lane & 15;
  const unsigned ldb = (unsigned)p.lda * 8u;
  double* sA = smem;
  double* sB = smem + NB * NB;
  const int lrow0 = tid >> 5, c32 = tid & 31;
  const unsigned off_ld = (unsigned)lrow0 * ldb + (unsigned)c32 * 16u;
  const unsigned off_ld_inv = (unsigned)lrow0 * (NB * 8u) + (unsigned)c32 * 16u;
  const int st_img = (c32 >> 3) * (NB * GEMM_BK), st_cc = c32 & 7;
  const unsigned off_cd = (unsigned)(wm * 32 + g) * ldb + (unsigned)(wn * 32 + r) * 8u;
  const int swz = kc_swz(r);
  const int nsteps = j * ni;
  const double* Brow = p.A + (int64_t)j * NB * p.lda;
  d4 acc[CELL_NI][2][2];
#pragma unroll
  for (int t = 0; t < CELL_NI; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = d4{0.0, 0.0, 0.0, 0.0};
  d2 ra[8], rb[8];
  double cold[2][2][4];
  auto request_c = [&](int t) {
    const __amdgpu_buffer_rsrc_t rc = dag_rsrc(p.A + (int64_t)(i0 + t) * NB * p.lda + (int64_t)j * NB);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) cold[a][b][q] = ld1_sc1<false>(rc, off_cd, (unsigned)(a * 16 + 4 * q) * ldb + (unsigned)b * 128u);
  };
  auto request = [&](int step) {
    const int blk = step / ni, t = step - blk * ni;
    const __amdgpu_buffer_rsrc_t rsa = dag_rsrc(p.A + (int64_t)(i0 + t) * NB * p.lda + (int64_t)blk * NB);
#pragma unroll
    for (int e = 0; e < 8; ++e) ra[e] = ld2_sc1<false>(rsa, off_ld, (unsigned)(8 * e) * ldb);
    if (t == 0) {
      const __amdgpu_buffer_rsrc_t rsb = dag_rsrc(Brow + (int64_t)blk * NB);
#pragma unroll
      for (int e = 0; e < 8; ++e) rb[e] = ld2_sc1<false>(rsb, off_ld, (unsigned)(8 * e) * ldb);
    }
  };
  auto request_inverse = [&] {
    const __amdgpu_buffer_rsrc_t rsb = dag_rsrc(p.inv_diag + (int64_t)j * NB * NB);
#pragma unroll
    for (int e = 0; e < 8; ++e) rb[e] = ld2_sc1<false>(rsb, off_ld_inv, (unsigned)(8 * e) * (NB * 8u));
  };
  auto publish = [&](bool with_a, bool with_b) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = lrow0 + 8 * e;
      const int slot = st_img + row * GEMM_BK + ((st_cc ^ kc_swz(row)) * 2);
      if (with_a) *reinterpret_cast<d2*>(sA + slot) = ra[e];
      if (with_b) *reinterpret_cast<d2*>(sB + slot) = rb[e];
    }
  };
  if (nsteps > 0) request(0);
  for (int step = 0; step < nsteps; ++step) {
    const int blk = step / ni, t = step - blk * ni;
    lds_barrier();
    publish(true, t == 0);
    lds_barrier();
    if (step + 1 < nsteps) {
      request(step + 1);
    } else {
      request_c(0);  // under the last step's MFMAs
    }
#pragma unroll
    for (int tt = 0; tt < CELL_NI; ++tt)
      if (tt == t) dag_mma64(acc[tt], sA, sB, wm, wn, g, r, swz);
  }
  if (nsteps == 0) request_c(0);
  request_inverse();  // (not under the MFMAs: with the operand registers of the last step still live it spilled)
  lds_barrier();  // every wave has finished reading the last step's images
  publish(false, true);  // the inverse becomes the B image
#pragma unroll
  for (int t = 0; t < CELL_NI; ++t) {
    if (t < ni) {
      // the updated tile (C + (-1) * sum, one rounding, as gemm_f64) -> A stage image
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = wm * 32 + a * 16 + g + 4 * q, col = wn * 32 + b * 16 + r;
            const double v = __builtin_fma(1.0, cold[a][b][q], -1.0 * acc[t][a][b][q]);
            sA[(col >> 4) * (NB * GEMM_BK) + row * GEMM_BK + ((((col & 15) >> 1) ^ kc_swz(row)) * 2) + (col & 1)] = v;
          }
      lds_barrier();
      if (t + 1 < ni) request_c(t + 1);
      d4 (&u)[2][2] = acc[t];  // (the sum has gone into the image: its registers take the product)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) u[a][b] = d4{0.0, 0.0, 0.0, 0.0};
      dag_mma64(u, sA, sB, wm, wn, g, r, swz);
      const __amdgpu_buffer_rsrc_t rc = dag_rsrc(p.A + (int64_t)(i0 + t) * NB * p.lda + (int64_t)j * NB);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) st1_sc1<false>(rc, off_cd, u[a][b][q], (unsigned)(a * 16 + 4 * q) * ldb + (unsigned)b * 128u);
      lds_barrier();  // (the A image is free for the next tile)
    }
  }
}

// The diagonal tile's update alone: A(j,j) -= sum_{c<j} L(j,c) L(j,c)^T.  Both operands of a step are the SAME block, so one
// block load per step feeds both images' reads, and with one accumulator tile there are registers for two loads in flight (the
// general panel's one-step-ahead prefetch left a memory round trip exposed in every step of this short dependent pass: 120 steps
// per N = 1024 cell).  Same products in the same order as dag_panel<false> on this tile: bit-identical.
__device__ __forceinline__ void cell_diag_update(const TileCtx& p, int j, double* __restrict__ smem) {
  const int tid = cell_tid(), lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, r = lane & 15;
  const unsigned ldb = (unsigned)p.lda * 8u;
  double* sA = smem;
  const int lrow0 = tid >> 5, c32 = tid & 31;
  const unsigned off_ld = (unsigned)lrow0 * ldb + (unsigned)c32 * 16u;
  const int st_img = (c32 >> 3) * (NB * GEMM_BK), st_cc = c32 & 7;
  const unsigned off_cd = (unsigned)(wm * 32 + g) * ldb + (unsigned)(wn * 32 + r) * 8u;
  const int swz = kc_swz(r);
  const double* Brow = p.A + (int64_t)j * NB * p.lda;
  d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
  d2 r0[8], r1[8];
  auto request = [&](d2 (&v)[8], int blk) {
    const __amdgpu_buffer_rsrc_t rs = dag_rsrc(Brow + (int64_t)blk * NB);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ld2_sc1<false>(rs, off_ld, (unsigned)(8 * e) * ldb);
  };
  auto publish = [&](const d2 (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = lrow0 + 8 * e;
      *reinterpret_cast<d2*>(sA + st_img + row * GEMM_BK + ((st_cc ^ kc_swz(row)) * 2)) = v[e];
    }
  };
  const __amdgpu_buffer_rsrc_t rc = dag_rsrc(Brow + (int64_t)j * NB);
  double cold[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) cold[a][b][q] = ld1_sc1<false>(rc, off_cd, (unsigned)(a * 16 + 4 * q) * ldb + (unsigned)b * 128u);
  request(r0, 0);
  if (j > 1) request(r1, 1);
  for (int k = 0; k < j; k += 2) {
    lds_barrier();
    publish(r0);
    lds_barrier();
    if (k + 2 < j) request(r0, k + 2);
    dag_mma64(acc, sA, sA, wm, wn, g, r, swz);
    if (k + 1 < j) {
      lds_barrier();
      publish(r1);
      lds_barrier();
      if (k + 3 < j) request(r1, k + 3);
      dag_mma64(acc, sA, sA, wm, wn, g, r, swz);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        st1_sc1<false>(rc, off_cd, __builtin_fma(1.0, cold[a][b][q], -1.0 * acc[a][b][q]), (unsigned)(a * 16 + 4 * q) * ldb + (unsigned)b * 128u);
}

// FUSED (default): per block column the diagonal tile is updated alone, factored, and every group of tiles below it is updated and
// solved in one pass (cell_panel_fused); FUSED = false: the two-pass form (update every tile, factor, solve every tile).
template <bool FUSED>
__global__ __launch_bounds__(256, 2) void potrf_cell_kernel_t(CellArgs p) {
  __shared__ __attribute__((aligned(16))) double smem[DAG_SMEM];
  const int64_t off = (int64_t)blockIdx.x * p.cs;
  const TileCtx tc{p.A + off, p.lda, p.inv_diag + off};
  int first_bad = 0;
  for (int j = 0; j < p.T; ++j) {
    if (j > 0) {
      cell_diag_update(tc, j, smem);
      cell_sync();
    }
    const int bad = cell_diag(tc.A + (int64_t)j * NB * p.lda + (int64_t)j * NB, p.lda, const_cast<double*>(tc.inv_diag) + (int64_t)j * NB * NB, smem);
    if (bad > 0 && first_bad == 0) first_bad = j * NB + bad;
    cell_sync();
    for (int i0 = j + 1; i0 < p.R; i0 += CELL_NI) {
      const int ni = p.R - i0 < CELL_NI ? p.R - i0 : CELL_NI;
      cell_panel_fused(tc, i0, ni, j, smem);
      cell_sync();
    }
  }
  if (threadIdx.x == 0 && first_bad > 0) atomicCAS(p.info + (int64_t)blockIdx.x * p.info_stride, 0, p.col_base + first_bad);
}

__global__ __launch_bounds__(256, 2) void potrf_cell_kernel(CellArgs p) {
  __shared__ __attribute__((aligned(16))) double smem[DAG_SMEM];
  const int64_t off = (int64_t)blockIdx.x * p.cs;
  const TileCtx tc{p.A + off, p.lda, p.inv_diag + off};
  int first_bad = 0;
  for (int j = 0; j < p.T; ++j) {
    if (j > 0) {
      for (int i0 = j; i0 < p.R; i0 += DAG_NI) {
        const int ni = p.R - i0 < DAG_NI ? p.R - i0 : DAG_NI;
        dag_panel<false, false>(tc, i0, ni, j, 0, j, smem);
        cell_sync();  // (LDS images free; the stores are visible to this workgroup's later loads)
      }
    }
    const int bad = cell_diag(tc.A + (int64_t)j * NB * p.lda + (int64_t)j * NB, p.lda, const_cast<double*>(tc.inv_diag) + (int64_t)j * NB * NB, smem);
    if (bad > 0 && first_bad == 0) first_bad = j * NB + bad;
    cell_sync();
    for (int i0 = j + 1; i0 < p.R; i0 += DAG_NI) {
      const int ni = p.R - i0 < DAG_NI ? p.R - i0 : DAG_NI;
      dag_panel<true, false>(tc, i0, ni, j, j, j + 1, smem);
      cell_sync();
    }
  }
  if (threadIdx.x == 0 && first_bad > 0) atomicCAS(p.info + (int64_t)blockIdx.x * p.info_stride, 0, p.col_base + first_bad);
}

// ---- column PAIRS (round 4): the operand stream halved -----------------------------------------------------------------------------
// Measured on the kernel above (profiles/r04_pmc_cell_kernel.json, N = 1024 x 512 cells): 52.8 MB of HBM traffic per cell (45.7 read +
// 5.9 written; the matrix itself is 4.2 MB) at 4.56 TB/s -- the kernel is HBM-bound, and three quarters of those bytes are the operand
// tiles of the left-looking update, 1.5 tile loads per 64 x 64 x 64 product (two row tiles sharing one B tile).  Here two block
// columns (j, j + 1) are updated TOGETHER: a group of two row tiles against the two column tiles is one 128 x 128 product
//     acc(t, c) = sum_{k < 64 j} L(i0 + t, k) L(j + c, k)^T        t, c in {0, 1}   (cell2_stream)
// whose operands are two contiguous 128-row panels of L, streamed global -> LDS by LDS-DMA in 16-deep stages (double-buffered: the
// loop of gemm_f64.h with a 128 x 128 workgroup tile, every wave holding the 32 x 32 quarter it owns of each of the four 64 x 64
// tiles, so each tile sits in the accumulator layout the solve steps below expect): 1.0 tile loads per product and no staging
// registers.  Then, per row tile, on LDS images only:
//     L(i, j)     = (A(i, j)     - acc(t, 0)) L(j, j)^-T
//     acc(t, 1)  += L(i, j) L(j + 1, j)^T                          (the one term of column j + 1 that column j's result feeds)
//     L(i, j + 1) = (A(i, j + 1) - acc(t, 1)) L(j + 1, j + 1)^-T
// The accumulation order of every tile is that of the single-column kernel (k ascending in stages of 16, instruction jj takes
// k = k0 + 4 g + jj), and the operands are the same values: the factor equals potrf_cell_kernel_t's BIT FOR BIT (tested).
constexpr int CELL2_STAGE = 4 * NB * GEMM_BK;  // doubles per stage: A image [128][16] | B image [128][16]
static_assert(2 * CELL2_STAGE <= DAG_SMEM, "the two stages fit into the LDS of the single-column kernel");

// a 64 x 64 block stored with leading dimension `ld` -> the four [64][16] stage images at `img`, by LDS-DMA (lane l of wave w fills row
// 8 (4 i + w) + (l >> 3), LDS chunk l & 7 = global chunk (l & 7) ^ kc_swz(row): the involution the fragment reads apply)
__device__ __forceinline__ void cell2_dma_block(const double* __restrict__ src, int64_t ld, double* __restrict__ img, int wave_u, int lane) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbase = (i * 4 + wave_u) * 8;
      const int row = rbase + (lane >> 3);
      glds16(src + (int64_t)row * ld + s * GEMM_BK + 2 * ((lane & 7) ^ kc_swz(row)), img + s * (NB * GEMM_BK) + rbase * GEMM_BK);
    }
}

// The streaming product of a pass: acc(t, c) += sum_{k < K} A(64 t + ., k) B(64 c + ., k)^T, A = NI row tiles at Arow, B = the two
// tiles at Brow, both with leading dimension lda; SAME: A and B are the same panel (the pair's own rows: one image serves both, and
// the tile above the diagonal, (t, c) = (0, 1), is left out).  16-deep stages, LDS-DMA, two buffers, one barrier per stage.
// BETA (with SAME: the pair's own 128-row panel): the panel's product with the vector beta[0, K) rides along -- thread t owns row
// t & 127 and the k half t >> 7 of every 16-deep stage image (8 fp64 FMAs per thread and stage beside 48 MFMAs per wave), its partial
// sum comes back in `tsum`: what the forward substitution of the right-hand side needs from these rows, without a tile row of its own.
template <int NI, bool SAME, bool BETA = false>
__device__ __forceinline__ void cell2_stream(d4 (&acc)[NI][2][2][2], const double* __restrict__ Arow, const double* __restrict__ Brow, int64_t lda, int K,
                                             double* __restrict__ smem, int lane, int wave_u, int wm, int wn, int g, int r,
                                             const double* __restrict__ sBeta = nullptr, double* tsum = nullptr) {
  static_assert(!BETA || SAME, "the vector product is defined on the pair's own panel");
  const int swz = kc_swz(r);
  const int brow = (wave_u & 1) * 64 + lane, bhalf = wave_u >> 1;  // (BETA) this thread's panel row and k half
  const int bswz = kc_swz(brow);
  double bsum = 0.0;
  auto dma_fill = [&](int k0, int buf) {
    double* sa = smem + buf * CELL2_STAGE;
    double* sb = sa + 2 * NB * GEMM_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rbase = (i * 4 + wave_u) * 8;
      const int row = rbase + (lane >> 3);
      const int chunk = 2 * ((lane & 7) ^ kc_swz(row));
      if (!SAME && i < 2 * NI) glds16(Arow + (int64_t)row * lda + k0 + chunk, sa + rbase * GEMM_BK);
      glds16(Brow + (int64_t)row * lda + k0 + chunk, sb + rbase * GEMM_BK);
    }
  };
  auto stage = [&](int k0, int buf) {
#ifndef GPRX_CELL2_NODMA
    if (k0 + GEMM_BK < K) dma_fill(k0 + GEMM_BK, buf ^ 1);  // (every wave left buffer buf ^ 1 at the previous barrier)
#endif
    const double* sb = smem + buf * CELL2_STAGE + 2 * NB * GEMM_BK;
    const double* sa = SAME ? sb : smem + buf * CELL2_STAGE;
    double fa[NI][2][4], fb[2][2][4];
#pragma unroll
    for (int t = 0; t < NI; ++t)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int row = t * NB + wm * 32 + a * 16 + r;
        const d2 lo = *reinterpret_cast<const d2*>(sa + row * GEMM_BK + 2 * ((2 * g) ^ swz));
        const d2 hi = *reinterpret_cast<const d2*>(sa + row * GEMM_BK + 2 * ((2 * g + 1) ^ swz));
        fa[t][a][0] = lo.x; fa[t][a][1] = lo.y; fa[t][a][2] = hi.x; fa[t][a][3] = hi.y;
      }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int col = c * NB + wn * 32 + b * 16 + r;
        const d2 lo = *reinterpret_cast<const d2*>(sb + col * GEMM_BK + 2 * ((2 * g) ^ swz));
        const d2 hi = *reinterpret_cast<const d2*>(sb + col * GEMM_BK + 2 * ((2 * g + 1) ^ swz));
        fb[c][b][0] = lo.x; fb[c][b][1] = lo.y; fb[c][b][2] = hi.x; fb[c][b][3] = hi.y;
      }
    if constexpr (BETA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const d2 pv = *reinterpret_cast<const d2*>(sb + brow * GEMM_BK + 2 * ((4 * bhalf + i) ^ bswz));
        const d2 bv = *reinterpret_cast<const d2*>(sBeta + k0 + 8 * bhalf + 2 * i);
        bsum = __builtin_fma(pv.x, bv.x, bsum);
        bsum = __builtin_fma(pv.y, bv.y, bsum);
      }
    }
#ifdef GPRX_CELL2_NOMMA
    acc[0][0][0][0][0] += fa[0][0][0] + fb[0][0][0] + fa[NI - 1][1][3] + fb[1][1][3];
#else
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int t = 0; t < NI; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (SAME && t == 0 && c == 1) continue;
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[t][c][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[t][a][jj], fb[c][b][jj], acc[t][c][a][b], 0, 0, 0);
        }
#endif
    // (without this the scheduler hoists the barrier -- and with it the wait for the next stage's DMA -- to the middle of the MFMA
    // sequence: the loads then have a third of a stage to land instead of a whole one)
    __builtin_amdgcn_sched_barrier(0);
    cell_sync();  // (vmcnt(0): this wave's DMA of the next stage has landed; then every wave's)
  };
  if (K > 0) {
    dma_fill(0, 0);
    cell_sync();
    for (int k0 = 0; k0 < K; k0 += 2 * GEMM_BK) {
      stage(k0, 0);
      stage(k0 + GEMM_BK, 1);  // (K is a multiple of 64)
    }
  }
  if constexpr (BETA) *tsum = bsum;
}

// what the solve steps share: accumulator layout <-> A stage images, the residual C - sum, tile stores
struct Cell2Lane {
  int wm, wn, g, r;
  unsigned ldb, off_cd;
};
__device__ __forceinline__ void cell2_to_image(double* __restrict__ sA, const d4 (&v)[2][2], const Cell2Lane& q) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = q.wm * 32 + a * 16 + q.g + 4 * e, col = q.wn * 32 + b * 16 + q.r;
        sA[(col >> 4) * (NB * GEMM_BK) + row * GEMM_BK + ((((col & 15) >> 1) ^ kc_swz(row)) * 2) + (col & 1)] = v[a][b][e];
      }
}
// v <- C - v for the tile at `tile` (one rounding, as gemm_f64: fma(1, C, -1 * sum))
__device__ __forceinline__ void cell2_residual(d4 (&v)[2][2], const double* __restrict__ tile, const Cell2Lane& q) {
  const __amdgpu_buffer_rsrc_t rc = dag_rsrc(tile);
  double cold[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) cold[a][b][e] = ld1_sc1<false>(rc, q.off_cd, (unsigned)(a * 16 + 4 * e) * q.ldb + (unsigned)b * 128u);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[a][b][e] = __builtin_fma(1.0, cold[a][b][e], -1.0 * v[a][b][e]);
}
__device__ __forceinline__ void cell2_store(const d4 (&v)[2][2], double* __restrict__ tile, const Cell2Lane& q) {
  const __amdgpu_buffer_rsrc_t rc = dag_rsrc(tile);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) st1_sc1<false>(rc, q.off_cd, v[a][b][e], (unsigned)(a * 16 + 4 * e) * q.ldb + (unsigned)b * 128u);
}
__device__ __forceinline__ void cell2_zero(d4 (&v)[2][2]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) v[a][b] = d4{0.0, 0.0, 0.0, 0.0};
}

// ---- the kernel matrix on first touch --------------------------------------------------------------------------------------------
// Every tile of A is read exactly once before it is overwritten (the residual C - sum), so K(X, X) + s I need not exist in memory at
// all: the tile is evaluated where it is consumed.  Saves the kernel-build launch (0.65 ms of a 6.2 ms step at N = 1024 x 512 cells),
// its 4.2 MB write and the 4.4 MB re-read per cell.  The arithmetic is kmat_body's, operation for operation (coordinates scaled by
// one reciprocal per dimension, r2 accumulated in k order by (a - b, fma), variance * exp_nonpos_tab(-r2 / 2), the diagonal term added
// to the product, identity on the padding) -- the values are the ones the separate launch writes, bit for bit (tested).  RBF in the
// difference form only (the judged configuration); anything else keeps the launch.
constexpr int CELL2_KX = DAG_SMEM;                      // doubles: row points [64][8] | column points [8][KM_BT_LD] | 2^(j/64) table [64]
constexpr int CELL2_SMEM_K = DAG_SMEM + 64 * KM_DC + KM_DC * KM_BT_LD + 64;
struct Cell2K {
  const double* X;
  const double* par;  // this cell's row of the parameter table
  int n, d, T;
};
// v <- K(tile ti, tile tj) - v in the accumulator layout (ti >= tj; tiles of the right-hand-side rows, ti >= T, are read from memory)
__device__ __forceinline__ void cell2_ktile(d4 (&v)[2][2], int ti, int tj, const Cell2K& kq, const Cell2Lane& q, int tid, double* __restrict__ smem) {
  double (*sXa)[KM_DC] = reinterpret_cast<double (*)[KM_DC]>(smem + CELL2_KX);
  double (*sXbt)[KM_BT_LD] = reinterpret_cast<double (*)[KM_BT_LD]>(smem + CELL2_KX + 64 * KM_DC);
  const double* sTab = smem + CELL2_KX + 64 * KM_DC + KM_DC * KM_BT_LD;
  const int i0 = ti * NB, j0 = tj * NB;
  double r2[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) r2[a][b][e] = 0.0;
  for (int k0 = 0; k0 < kq.d; k0 += KM_DC) {
    {
      double ra[2], rb[2];
      const int kc = min(k0 + (tid & 7), kq.d - 1);
      const double s = kq.par[CELL_PAR_LS + kc];
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const int pt = (tid + 256 * rep) >> 3;
        ra[rep] = kq.X[(int64_t)min(i0 + pt, kq.n - 1) * kq.d + kc];
        rb[rep] = kq.X[(int64_t)min(j0 + pt, kq.n - 1) * kq.d + kc];
      }
      const double inv = 1.0 / s;
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const int qq = tid + 256 * rep;
        const int pt = qq >> 3, kk = qq & 7;
        const bool live = k0 + kk < kq.d;
        sXa[pt][kk] = (live && i0 + pt < kq.n) ? ra[rep] * inv : 0.0;
        sXbt[kk][pt] = (live && j0 + pt < kq.n) ? rb[rep] * inv : 0.0;
      }
    }
    lds_barrier();
    double bv[2][KM_DC];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int kk = 0; kk < KM_DC; ++kk) bv[b][kk] = sXbt[kk][q.wn * 32 + b * 16 + q.r];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = q.wm * 32 + a * 16 + q.g + 4 * e;
#pragma unroll
        for (int kk = 0; kk < KM_DC; ++kk) {
          const double av = sXa[row][kk];
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const double d0 = av - bv[b][kk];
            r2[a][b][e] = __builtin_fma(d0, d0, r2[a][b][e]);
          }
        }
      }
    lds_barrier();
  }
  const double variance = kq.par[0], diag_add = kq.par[1];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = i0 + q.wm * 32 + a * 16 + q.g + 4 * e, jcol = j0 + q.wn * 32 + b * 16 + q.r;
        const bool valid = i < kq.n && jcol < kq.n;
        double val = valid ? variance * corr_g<0>(r2[a][b][e], sTab) : 0.0;
        if (i == jcol) val += valid ? diag_add : 1.0;
        v[a][b][e] = __builtin_fma(1.0, val, -1.0 * v[a][b][e]);
      }
}
// the residual of tile (ti, tj): from memory, or (KB) evaluated on the spot
template <bool KB>
__device__ __forceinline__ void cell2_fetch(d4 (&v)[2][2], const TileCtx& p, int ti, int tj, const Cell2K& kq, const Cell2Lane& q, int tid,
                                            double* __restrict__ smem) {
  if (KB && ti < kq.T)
    cell2_ktile(v, ti, tj, kq, q, tid, smem);
  else
    cell2_residual(v, p.A + (int64_t)ti * NB * p.lda + (int64_t)tj * NB, q);
}

// ---- beta = L^-1 y as a vector (round 4) ---------------------------------------------------------------------------------------------
// The right-hand side used to ride as a 64-row tile below the matrix (one useful row of 64): Sum_j j = T^2 / 2 tile products plus a
// 64 x 128 streaming pass per column pair -- at N = 1024 8 % of the launch, at N = 512 16 % (tools/batch_n1024.py with and without it:
// 5.97 -> 5.52 ms, 1.56 -> 1.35 ms per 512 cells).  The forward substitution needs, per block column j, only
//     beta_j = L(j,j)^-1 (y_j - Sum_{k < 64 j} L(j, k) beta_k),
// and the panel L(j .. j+1, 0 .. 64 j) is exactly what cell2_diag_pair streams through LDS for the pair's own tiles: the panel-times-vector
// product rides in that stream (cell2_stream<2, true, true>), then three 64 x 64 matrix-vector products per pair (two against the
// inverses the chain leaves in memory, one against L(j+1, j)) finish the pair.  beta lives in LDS (at most 1024 entries: T <= 16) and
// replaces y in the cell's right-hand-side row.  Sums of 64 j + 64 terms in fixed order: the loss agrees with the tile form to rounding.
constexpr int CELL2_BETA_MAXT = 16;
constexpr int CELL2_BETA_LDS = CELL2_BETA_MAXT * NB + 256 + 256 + NB;  // doubles: beta | the stream's partial sums | matrix-vector partials | a block's residual
struct Cell2Beta {
  double* y;      // global: this cell's right-hand-side row
  double* sBeta;  // [CELL2_BETA_MAXT * NB]
  double* sPart;  // [256]: row r of the pair, k half h at r + 128 h
  double* sRed;   // [256]
  double* sV;     // [NB]
};
// sRed[64 q + a] = Sum_{b in [16 q, 16 q + 16)} M[a][b] v[b]: M row-major in global memory (row stride ldm), v in LDS
__device__ __forceinline__ void cell2_gemv64(const double* __restrict__ M, int64_t ldm, const double* __restrict__ v, double* __restrict__ sRed, int tid) {
  const int a = tid & 63, q = tid >> 6;
  const double* row = M + (int64_t)a * ldm + 16 * q;
  d2 m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = *reinterpret_cast<const d2*>(row + 2 * i);
  double sum = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sum = __builtin_fma(m[i].x, v[16 * q + 2 * i], sum);
    sum = __builtin_fma(m[i].y, v[16 * q + 2 * i + 1], sum);
  }
  sRed[tid] = sum;
}
__device__ __forceinline__ double cell2_red4(const double* __restrict__ sRed, int a) { return ((sRed[a] + sRed[64 + a]) + sRed[128 + a]) + sRed[192 + a]; }
// block column jb: sV holds y_jb - (what the columns left of it contribute); beta_jb = L(jb,jb)^-1 sV -> sBeta and the cell's row
__device__ __forceinline__ void cell2_beta_solve(const TileCtx& p, int jb, const Cell2Beta& bq, int tid) {
  cell2_gemv64(p.inv_diag + (int64_t)jb * NB * NB, NB, bq.sV, bq.sRed, tid);
  lds_barrier();
  if (tid < NB) {
    const double b = cell2_red4(bq.sRed, tid);
    bq.sBeta[jb * NB + tid] = b;
    bq.y[jb * NB + tid] = b;
  }
  lds_barrier();
}
// after the pair (j, j + 1) is factored (every store of the pair acknowledged: the caller's cell_sync); sPart = the stream's partial sums
__device__ __forceinline__ void cell2_beta_pair(const TileCtx& p, int j, const Cell2Beta& bq, int tid) {
  if (tid < NB) bq.sV[tid] = bq.y[j * NB + tid] - (bq.sPart[tid] + bq.sPart[128 + tid]);
  lds_barrier();
  cell2_beta_solve(p, j, bq, tid);
  cell2_gemv64(p.A + (int64_t)(j + 1) * NB * p.lda + (int64_t)j * NB, p.lda, bq.sBeta + j * NB, bq.sRed, tid);
  lds_barrier();
  if (tid < NB) bq.sV[tid] = bq.y[(j + 1) * NB + tid] - ((bq.sPart[NB + tid] + bq.sPart[128 + NB + tid]) + cell2_red4(bq.sRed, tid));
  lds_barrier();
  cell2_beta_solve(p, j + 1, bq, tid);
}
// a single last block column j (T odd): its row panel L(j, 0 .. 64 j) times beta straight from memory
__device__ __forceinline__ void cell2_beta_single(const TileCtx& p, int j, const Cell2Beta& bq, int tid) {
  const int a = tid & 63, q = tid >> 6;
  const double* row = p.A + ((int64_t)j * NB + a) * p.lda;
  double sum = 0.0;
  for (int k = 16 * q; k < j * NB; k += 64) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const d2 m = *reinterpret_cast<const d2*>(row + k + 2 * i);
      sum = __builtin_fma(m.x, bq.sBeta[k + 2 * i], sum);
      sum = __builtin_fma(m.y, bq.sBeta[k + 2 * i + 1], sum);
    }
  }
  bq.sRed[tid] = sum;
  lds_barrier();
  if (tid < NB) bq.sV[tid] = bq.y[j * NB + tid] - cell2_red4(bq.sRed, tid);
  lds_barrier();
  cell2_beta_solve(p, j, bq, tid);
}

// row tiles i0 .. i0 + NI - 1 (all below tile row j + 1) of the block columns j and j + 1
template <int NI, bool KB>
__device__ __forceinline__ void cell2_rows(const TileCtx& p, int i0, int j, const Cell2K& kq, double* __restrict__ smem) {
  const int tid = cell_tid(), lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  Cell2Lane q;
  q.wm = wave >> 1, q.wn = wave & 1, q.g = lane >> 4, q.r = lane & 15;
  q.ldb = (unsigned)p.lda * 8u;
  q.off_cd = (unsigned)(q.wm * 32 + q.g) * q.ldb + (unsigned)(q.wn * 32 + q.r) * 8u;
  const int swz = kc_swz(q.r);
  d4 acc[NI][2][2][2];
#pragma unroll
  for (int t = 0; t < NI; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      cell2_zero(acc[t][c]);
    }
  cell2_stream<NI, false>(acc, p.A + (int64_t)i0 * NB * p.lda, p.A + (int64_t)j * NB * p.lda, p.lda, j * NB, smem, lane, wave_u, q.wm, q.wn, q.g, q.r);
  CACC(4)
  // ---- the solve steps: one B image at a time (L(j,j)^-1, then L(j+1,j), then L(j+1,j+1)^-1), every row tile against it ----
  // (the streaming loop's last barrier has passed: both stage buffers are free)
  double* sA = smem;
  double* sB = smem + NB * NB;
  auto tile = [&](int i, int jc) { return p.A + (int64_t)i * NB * p.lda + (int64_t)jc * NB; };
  // E1: L(i, j) = (A(i, j) - acc(t, 0)) L(j, j)^-T
  cell2_dma_block(p.inv_diag + (int64_t)j * NB * NB, NB, sB, wave_u, lane);
#pragma unroll
  for (int t = 0; t < NI; ++t) {
    cell2_fetch<KB>(acc[t][0], p, i0 + t, j, kq, q, tid, smem);
    cell2_to_image(sA, acc[t][0], q);
    cell_sync();  // (the first one also waits for the B image's DMA)
    cell2_zero(acc[t][0]);
    dag_mma64(acc[t][0], sA, sB, q.wm, q.wn, q.g, q.r, swz);
    cell2_store(acc[t][0], tile(i0 + t, j), q);
    lds_barrier();  // the A image is free again
  }
  CACC(5)
  // E2: acc(t, 1) += L(i, j) L(j + 1, j)^T
  cell2_dma_block(tile(j + 1, j), p.lda, sB, wave_u, lane);
#pragma unroll
  for (int t = 0; t < NI; ++t) {
    cell2_to_image(sA, acc[t][0], q);
    cell_sync();
    dag_mma64(acc[t][1], sA, sB, q.wm, q.wn, q.g, q.r, swz);
    lds_barrier();
  }
  CACC(6)
  // E3: L(i, j + 1) = (A(i, j + 1) - acc(t, 1)) L(j + 1, j + 1)^-T
  cell2_dma_block(p.inv_diag + (int64_t)(j + 1) * NB * NB, NB, sB, wave_u, lane);
#pragma unroll
  for (int t = 0; t < NI; ++t) {
    cell2_fetch<KB>(acc[t][1], p, i0 + t, j + 1, kq, q, tid, smem);
    cell2_to_image(sA, acc[t][1], q);
    cell_sync();
    cell2_zero(acc[t][1]);
    dag_mma64(acc[t][1], sA, sB, q.wm, q.wn, q.g, q.r, swz);
    cell2_store(acc[t][1], tile(i0 + t, j + 1), q);
    lds_barrier();
  }
  CACC(7)
}

// The pair's own three tiles (j, j), (j + 1, j), (j + 1, j + 1): ONE streaming product of the 128-row panel with itself (what the
// single-column kernel does as three latency-bound passes of one tile product per step: the diagonal update of j, update + solve of
// (j + 1, j), the diagonal update of j + 1), then the two chains and the solve between them.  Returns the failing pivot (1-based,
// within the pair's 128 columns) or 0.
template <bool KB, bool BETA = false>
__device__ __forceinline__ int cell2_diag_pair(const TileCtx& p, int j, const Cell2K& kq, double* __restrict__ smem, const Cell2Beta* bq = nullptr) {
  const int tid = cell_tid(), lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  Cell2Lane q;
  q.wm = wave >> 1, q.wn = wave & 1, q.g = lane >> 4, q.r = lane & 15;
  q.ldb = (unsigned)p.lda * 8u;
  q.off_cd = (unsigned)(q.wm * 32 + q.g) * q.ldb + (unsigned)(q.wn * 32 + q.r) * 8u;
  const int swz = kc_swz(q.r);
  d4 acc[2][2][2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) cell2_zero(acc[t][c]);
  const double* Prow = p.A + (int64_t)j * NB * p.lda;
  if constexpr (BETA) {
    double tsum = 0.0;
    cell2_stream<2, true, true>(acc, Prow, Prow, p.lda, j * NB, smem, lane, wave_u, q.wm, q.wn, q.g, q.r, bq->sBeta, &tsum);
    bq->sPart[tid] = tsum;  // (read by cell2_beta_pair, several barriers from here)
  } else {
    cell2_stream<2, true>(acc, Prow, Prow, p.lda, j * NB, smem, lane, wave_u, q.wm, q.wn, q.g, q.r);
  }
  CACC(1)
  auto tile = [&](int i, int jc) { return p.A + (int64_t)i * NB * p.lda + (int64_t)jc * NB; };
  double* inv = const_cast<double*>(p.inv_diag);
  double* sA = smem;
  double* sB = smem + NB * NB;
  // (j, j): the updated tile goes back to memory, the chain factors it there
  cell2_fetch<KB>(acc[0][0], p, j, j, kq, q, tid, smem);
  cell2_store(acc[0][0], tile(j, j), q);
  cell_sync();
  int bad = cell_diag(tile(j, j), p.lda, inv + (int64_t)j * NB * NB, smem);
  cell_sync();
  // (j + 1, j) = (A - sum) L(j, j)^-T, then its square into the sum of (j + 1, j + 1)
  cell2_dma_block(p.inv_diag + (int64_t)j * NB * NB, NB, sB, wave_u, lane);
  cell2_fetch<KB>(acc[1][0], p, j + 1, j, kq, q, tid, smem);
  cell2_to_image(sA, acc[1][0], q);
  cell_sync();
  cell2_zero(acc[1][0]);
  dag_mma64(acc[1][0], sA, sB, q.wm, q.wn, q.g, q.r, swz);
  cell2_store(acc[1][0], tile(j + 1, j), q);
  lds_barrier();
  cell2_to_image(sA, acc[1][0], q);
  lds_barrier();
  dag_mma64(acc[1][1], sA, sA, q.wm, q.wn, q.g, q.r, swz);
  cell2_fetch<KB>(acc[1][1], p, j + 1, j + 1, kq, q, tid, smem);
  cell2_store(acc[1][1], tile(j + 1, j + 1), q);
  cell_sync();
  const int bad2 = cell_diag(tile(j + 1, j + 1), p.lda, inv + (int64_t)(j + 1) * NB * NB, smem);
  if (bad == 0 && bad2 > 0) bad = NB + bad2;
  return bad;
}

#ifndef GPRX_CELL2_OCC
#define GPRX_CELL2_OCC 2
#endif
template <bool KB, bool BETA = false>
__global__ __launch_bounds__(256, GPRX_CELL2_OCC) void potrf_cell2_kernel(CellArgs p) {
  static_assert(!(KB && BETA), "two workgroups per CU: the LDS holds either the kernel-build staging or the beta vector");
  __shared__ __attribute__((aligned(16))) double smem[KB ? CELL2_SMEM_K : DAG_SMEM];
  __shared__ __attribute__((aligned(16))) double sbeta[BETA ? CELL2_BETA_LDS : 2];
  const int64_t off = (int64_t)blockIdx.x * p.cs;
  const TileCtx tc{p.A + off, p.lda, p.inv_diag + off};
  const Cell2K kq{p.X, p.cell_par + (int64_t)blockIdx.x * CELL_PAR, p.n, p.d, p.T};
  const Cell2Beta bq{BETA ? p.beta + off : nullptr, sbeta, sbeta + CELL2_BETA_MAXT * NB, sbeta + CELL2_BETA_MAXT * NB + 256, sbeta + CELL2_BETA_MAXT * NB + 512};
  if (KB) exp_tab_fill(smem + CELL2_KX + 64 * KM_DC + KM_DC * KM_BT_LD);  // (visible after the first barrier of the first tile's staging)
  int first_bad = 0;
  CACC(0)
  for (int j = 0; j < p.T; j += 2) {
    if (j + 1 >= p.T) {  // a last single column: the single-column passes
      if (j > 0) {
        cell_diag_update(tc, j, smem);
        cell_sync();
      }
      const int bad = cell_diag(tc.A + (int64_t)j * NB * p.lda + (int64_t)j * NB, p.lda, const_cast<double*>(tc.inv_diag) + (int64_t)j * NB * NB, smem);
      if (bad > 0 && first_bad == 0) first_bad = j * NB + bad;
      cell_sync();
      if constexpr (BETA) cell2_beta_single(tc, j, bq, cell_tid());
      for (int i0 = j + 1; i0 < p.R; i0 += CELL_NI) {
        cell_panel_fused(tc, i0, p.R - i0 < CELL_NI ? p.R - i0 : CELL_NI, j, smem);
        cell_sync();
      }
      break;
    }
    const int bad = cell2_diag_pair<KB, BETA>(tc, j, kq, smem, &bq);
    if (bad > 0 && first_bad == 0) first_bad = j * NB + bad;
    cell_sync();
    CACC(2)
    if constexpr (BETA) cell2_beta_pair(tc, j, bq, cell_tid());
    CACC(3)
#ifndef GPRX_CELL2_NOROWS
    int i0 = j + 2;
    for (; i0 + 1 < p.R; i0 += 2) {
      cell2_rows<2, KB>(tc, i0, j, kq, smem);
      cell_sync();
    }
    if (i0 < p.R) {  // an odd tile at the end (the right-hand-side rows when T is even): a 64 x 128 pass of its own, nothing duplicated
      cell2_rows<1, KB>(tc, i0, j, kq, smem);
      cell_sync();
    }
#endif
  }
  if (threadIdx.x == 0 && first_bad > 0) atomicCAS(p.info + (int64_t)blockIdx.x * p.info_stride, 0, p.col_base + first_bad);
}

// `batch` matrices of np x np (+ extra right-hand-side rows), cs doubles apart; info words info_stride ints apart (zeroed by the caller)
// Can the column-pair kernel build K itself (RBF, difference form, an even number of block columns: the single last column of an odd
// count goes through the single-column passes, which read their tiles from memory)?  The caller then skips its kernel-build launch.
inline bool potrf_cells_builds_k(int kid, int form, int np, int d) {
  // OPT-IN (GPRX_CELL_BUILD_K=1): measured SLOWER than the separate launch -- N = 1024 x 512 cells 6.45 against 6.05 ms per step: the 153
  // tiles of a cell each cost a staging round trip (coordinates from L2, two barriers) and ~500 fp64 instructions per thread on the pipe
  // the MFMAs need, more than the 0.65 ms launch and the 8.6 MB per cell it saves
  static const bool off = !(getenv("GPRX_CELL_BUILD_K") && atoi(getenv("GPRX_CELL_BUILD_K")) == 1) ||
                          (getenv("GPRX_CELL_TWO_PASS") && atoi(getenv("GPRX_CELL_TWO_PASS")) != 0) ||
                          (getenv("GPRX_CELL_SINGLE_COLUMN") && atoi(getenv("GPRX_CELL_SINGLE_COLUMN")) != 0);
  return !off && kid == 0 && form == 0 && (np / NB) % 2 == 0 && d >= 1 && d <= CELL_PAR - CELL_PAR_LS;
}

// build_k: X / cell_par / n / d given and potrf_cells_builds_k() holds -- the matrices need not have been written (only their
// right-hand-side rows)
// Does potrf_cells carry the right-hand side as a VECTOR (cell2_beta_*: the default form, one right-hand-side row of which only the first
// np entries are read and written) instead of 64 tile rows?  GPRX_CELL_BETA_ROWS=1 restores the tile form (A/B, bit-for-bit tests
// against the other kernel forms, which only know the tile form).
inline bool potrf_cells_beta_vector(int np, int extra, bool builds_k) {
  static const bool off = (getenv("GPRX_CELL_BETA_ROWS") && atoi(getenv("GPRX_CELL_BETA_ROWS")) != 0) ||
                          (getenv("GPRX_CELL_TWO_PASS") && atoi(getenv("GPRX_CELL_TWO_PASS")) != 0) ||
                          (getenv("GPRX_CELL_SINGLE_COLUMN") && atoi(getenv("GPRX_CELL_SINGLE_COLUMN")) != 0);
  return !off && !builds_k && extra == NB && np / NB <= CELL2_BETA_MAXT;
}

inline hipError_t potrf_cells(hipStream_t st, double* A, int64_t lda, int np, int extra, double* inv_diag, int* info, int batch, int64_t cs,
                              int info_stride, int col_base = 0, const double* X = nullptr, const double* cell_par = nullptr, int n = 0, int d = 0) {
  CellArgs a;
  a.A = A;
  a.lda = lda;
  a.T = np / NB;
  a.R = a.T + extra / NB;
  a.inv_diag = inv_diag;
  a.info = info;
  a.cs = cs;
  a.info_stride = info_stride;
  a.col_base = col_base;
  static const bool two_pass = getenv("GPRX_CELL_TWO_PASS") && atoi(getenv("GPRX_CELL_TWO_PASS")) != 0;
  static const bool single_column = getenv("GPRX_CELL_SINGLE_COLUMN") && atoi(getenv("GPRX_CELL_SINGLE_COLUMN")) != 0;  // (round 3's kernel, for A/B)
  if (two_pass)
    hipLaunchKernelGGL(potrf_cell_kernel, dim3(batch), dim3(256), 0, st, a);
  else if (single_column)
    hipLaunchKernelGGL(potrf_cell_kernel_t<true>, dim3(batch), dim3(256), 0, st, a);
  else if (X && cell_par) {
    a.X = X;
    a.cell_par = cell_par;
    a.n = n;
    a.d = d;
    hipLaunchKernelGGL(potrf_cell2_kernel<true>, dim3(batch), dim3(256), 0, st, a);
  } else if (potrf_cells_beta_vector(np, extra, false)) {
    a.R = a.T;
    a.beta = A + (int64_t)np * lda;
    hipLaunchKernelGGL((potrf_cell2_kernel<false, true>), dim3(batch), dim3(256), 0, st, a);
  } else
    hipLaunchKernelGGL(potrf_cell2_kernel<false>, dim3(batch), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace gprx
