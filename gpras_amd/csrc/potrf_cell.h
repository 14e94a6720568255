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
#include "potrf_dag.h"

namespace gprx {

// Row tiles per fused pass (they share the B operand; 32 accumulator registers each).  Measured (tools/cell_ni_sweep.sh, 512 cells):
// 4 tiles 70.7 k fits/s at N = 1024 / 290 k at N = 512 (the kernel sits at the 256-register cap and spills 372 B per lane), 3 tiles
// 72.2 k / 305 k, 2 tiles 71.5 k / 306 k (228 B of scratch, all in cell_diag), 6 tiles 55 k / 232 k.  A second operand set in flight
// (two loads ahead) on top of 2 or 3 tiles: no gain (70.7 k / 302 k) -- the kernel is not waiting for its loads.
#ifndef GPRX_CELL_NI
#define GPRX_CELL_NI 2
#endif
constexpr int CELL_NI = GPRX_CELL_NI;

struct CellArgs {
  double* A;         // cell 0; cells are `cs` doubles apart
  int64_t lda;
  int T, R;          // 64-column blocks, 64-row blocks (T + right-hand-side blocks)
  double* inv_diag;  // cell 0 (T blocks of 64 x 64), cs apart
  int* info;         // cell 0, info_stride ints apart
  int64_t cs;
  int info_stride;
  int col_base;
};

// the diagonal block (j, j): chain_step<0..7> on [64 diagonal rows | 64 identity rows] -> L(j,j) in place, L(j,j)^-1 to inv_diag
__device__ __forceinline__ int cell_diag(double* __restrict__ Ajj, int64_t lda, double* __restrict__ inv, double* __restrict__ smem) {
  ChainCtx c;
  c.sIn = smem;
  c.sX = smem + 128 * PSUB;
  double* sT = smem + 2 * 128 * PSUB;
  c.tid = threadIdx.x;
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
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int q = tid + 256 * e;
    st2_sc1<false>(ri, (unsigned)tid * 16u, *reinterpret_cast<const d2*>(sT + (q >> 5) * DAG_T_LD + 2 * (q & 31)), (unsigned)e * 4096u);
  }
  return c.bad;
}

// Tiles (i0 .. i0+ni-1, j), all below the diagonal tile, in ONE pass: C_t <- (C_t - sum_{c<j} L(i,c) L(j,c)^T) L(j,j)^-T.  The
// update sum as dag_panel<false> forms it (same steps, same accumulators); the updated tile then goes from the accumulator layout
// straight into the A stage image and is multiplied by the inverse (B image, loaded once per call) exactly as dag_panel<true>
// does with the tile it re-reads from memory -- same operands, same products: the factor is the two-pass kernel's bit for bit,
// with one store and one load of every tile less (8.7 of ~44 MB per N = 1024 cell).
__device__ __forceinline__ void cell_panel_fused(const TileCtx& p, int i0, int ni, int j, double* __restrict__ smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
      __syncthreads();
    }
    const int bad = cell_diag(tc.A + (int64_t)j * NB * p.lda + (int64_t)j * NB, p.lda, const_cast<double*>(tc.inv_diag) + (int64_t)j * NB * NB, smem);
    if (bad > 0 && first_bad == 0) first_bad = j * NB + bad;
    __syncthreads();
    for (int i0 = j + 1; i0 < p.R; i0 += CELL_NI) {
      const int ni = p.R - i0 < CELL_NI ? p.R - i0 : CELL_NI;
      cell_panel_fused(tc, i0, ni, j, smem);
      __syncthreads();
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
        __syncthreads();  // (LDS images free; the stores are visible to this workgroup's later loads)
      }
    }
    const int bad = cell_diag(tc.A + (int64_t)j * NB * p.lda + (int64_t)j * NB, p.lda, const_cast<double*>(tc.inv_diag) + (int64_t)j * NB * NB, smem);
    if (bad > 0 && first_bad == 0) first_bad = j * NB + bad;
    __syncthreads();
    for (int i0 = j + 1; i0 < p.R; i0 += DAG_NI) {
      const int ni = p.R - i0 < DAG_NI ? p.R - i0 : DAG_NI;
      dag_panel<true, false>(tc, i0, ni, j, j, j + 1, smem);
      __syncthreads();
    }
  }
  if (threadIdx.x == 0 && first_bad > 0) atomicCAS(p.info + (int64_t)blockIdx.x * p.info_stride, 0, p.col_base + first_bad);
}

// `batch` matrices of np x np (+ extra right-hand-side rows), cs doubles apart; info words info_stride ints apart (zeroed by the caller)
inline hipError_t potrf_cells(hipStream_t st, double* A, int64_t lda, int np, int extra, double* inv_diag, int* info, int batch, int64_t cs,
                              int info_stride, int col_base = 0) {
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
  if (two_pass)
    hipLaunchKernelGGL(potrf_cell_kernel, dim3(batch), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(potrf_cell_kernel_t<true>, dim3(batch), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace gprx
