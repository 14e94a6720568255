// The 64 x 64 dependent chain of the Cholesky factorisations: one workgroup turns [64 diagonal rows | 64 identity rows] into L(k,k) and
// L(k,k)^-T, eight sub-panel steps of 8 columns (chain_step).  Kernel-free header (device inline functions only) so that translation units
// other than gprx.hip can use the chain: the one-workgroup-per-cell Cholesky (potrf_cell.h), the tile-DAG scheduler (potrf_dag.h) and the
// fused sparse evaluation (sf_*.hip).  Split out of tile_ops.h / potrf.h in round 5.
#pragma once
#include "gprx_common.h"

namespace gprx {

constexpr int PSUB = 9;  // LDS row stride of the 8-column sub-panel buffers (row-per-lane b64 access conflict-free)

// workgroup barrier for LDS traffic only: __syncthreads() also waits for every outstanding global store and load of the wave
// (s_waitcnt vmcnt(0)), i.e. it would drain the write-through stores at every barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// ---- chain workgroup ---------------------------------------------------------------------------------------------------
// Rows of the chain's panel: workgroup rows 0..63 = the diagonal block, 64..127 = identity rows (they come out as L^-T).
// Wave w holds band w of each: acc[0] = diagonal rows 16 w .. 16 w + 15, acc[1] = identity rows 16 w .. 16 w + 15 -- so the
// band a wave needs for the next step (its rows of the updated diagonal block) is the band it computes.
struct ChainCtx {
  double* sIn;
  double* sX;
  int tid, wave, g, r;
  int bad;
};

#ifdef GPRX_CHAIN_STAMPS
static __device__ unsigned long long g_chain_stamps[16];  // (one copy per translation unit: gprx_chain_stamps reads gprx.hip's)
#define CSTAMP(i) if constexpr (P == 3) { if (c.tid == 0) g_chain_stamps[i] = __builtin_amdgcn_s_memtime(); }
#else
#define CSTAMP(i)
#endif

template <int P>
__device__ __forceinline__ void chain_step(d4 (&acc)[2][4], ChainCtx& c) {
  constexpr int C0 = 8 * P;
  CSTAMP(0)
  constexpr int KT = C0 / 16;
  constexpr int HALF = P & 1;
  // accumulators -> LDS (the lanes that hold these 8 columns)
  if ((c.r >> 3) == HALF) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) c.sIn[(64 * rt + 16 * c.wave + c.g + 4 * q) * PSUB + (c.r & 7)] = acc[rt][KT][q];
  }
  CSTAMP(1)
  lds_barrier();
  CSTAMP(2)
  // every thread factors the 8 x 8 diagonal sub-block (rows C0 .. C0 + 7 of the diagonal block): potrf.h panel_step
  double l[8][8], rinv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k <= j; ++k) l[j][k] = c.sIn[(C0 + j) * PSUB + k];
  CSTAMP(3)
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
  CSTAMP(4)
  if (c.tid < 128) {
    const int zero_above = c.tid < NB ? c.tid : (1 << 30);
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double t = c.sIn[c.tid * PSUB + k];
#pragma unroll
      for (int m = 0; m < k; ++m) t = __builtin_fma(-x[m], l[k][m], t);
      x[k] = (C0 + k > zero_above) ? 0.0 : t * rinv[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) c.sX[c.tid * PSUB + k] = x[k];
  }
  CSTAMP(5)
  lds_barrier();
  CSTAMP(6)
  if constexpr (C0 + 8 < NB) {
    constexpr int KT0 = (C0 + 8) / 16;
    double fa[2][2], fb[4][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[rt][ks] = -c.sX[(64 * rt + 16 * c.wave + c.r) * PSUB + 4 * ks + c.g];
#pragma unroll
    for (int kt = KT0; kt < 4; ++kt) {
      const int kk = kt * 16 + c.r;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb[kt][ks] = (kk >= C0 + 8) ? c.sX[kk * PSUB + 4 * ks + c.g] : 0.0;
    }
#pragma unroll
    for (int kt = KT0; kt < 4; ++kt)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        acc[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][0], fb[kt][0], acc[rt][kt], 0, 0, 0);
        acc[rt][kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[rt][1], fb[kt][1], acc[rt][kt], 0, 0, 0);
      }
  }
  CSTAMP(7)
  if ((c.r >> 3) == HALF) {  // solved values back into the accumulators
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[rt][KT][q] = c.sX[(64 * rt + 16 * c.wave + c.g + 4 * q) * PSUB + (c.r & 7)];
  }
  CSTAMP(8)
}

}  // namespace gprx
