// Tile-DAG Cholesky for ONE matrix (lower, in place, row-major): one persistent launch instead of ~250 dependent ones.
//
// Why: a lone N = 4096 factorisation under potrf_lower is a chain of 64 panel launches (19.9 us each) with every in-block
// update serialised between them -- 2.2 ms for 23 GFLOP (0.13 of the fp64 MFMA peak).  Here the
// dependent chain runs inside ONE workgroup that never leaves its CU and never waits for bulk work; everything else is tile
// tasks claimed from a queue by the other workgroups, ordered by per-tile version counters in global memory (no grid
// barriers, no launches).
//
//   chain workgroup (the first to arrive), step k = 0 .. T-1
//       8 sub-panel steps on [64 diagonal rows | 64 identity rows] in MFMA accumulators (the arithmetic of
//       potrf_panel_kernel's diagonal workgroup) -> L(k,k) and L(k,k)^-1, published (flag diag[k]);
//       L(k+1,k) = A(k+1,k) L(k,k)^-T (one triangular tile product), published;
//       A(k+1,k+1) - L(k+1,k) L(k+1,k)^T stays in registers: it IS the next step's diagonal block.
//   worker workgroups: tasks in a static priority order (column by column, the tiles next to the front first)
//       TRSM(i,k)  : L(i,k) = A(i,k) L(k,k)^-T              needs diag[k] and ver(i,k) == k
//       UPD(i,j,k0,k1): A(i,j) -= L(i,k0:k1) L(j,k0:k1)^T   needs ver(i,c) == c+1, ver(j,c) == c+1 (c in k0..k1-1), ver(i,j) == k0
//   ver(i,j) = number of 64-column blocks already applied to tile (i,j); a tile is final (holds L) at ver == j + 1.  Updates
//   of one tile are applied in ascending k, so the result does not depend on timing or placement (deterministic).
//
// Every worker dependency of the chain has one full chain step of slack: step k+1 needs A(k+2,k+1) through column k-1 ... and
// what column k adds to the diagonal block and to A(k+1,k) the chain computes itself.
//
// Hand-offs follow MI355X_MICROARCH.md (inter-workgroup visibility, valid forms, first table row): every byte another
// workgroup reads is stored write-through (sc1), every storing wave drains (s_waitcnt vmcnt(0)), a workgroup barrier, then
// ONE lane publishes with an agent-scope relaxed store; consumers poll with sc1 loads from one wave, a workgroup barrier,
// then sc1 loads to registers only (no LDS-DMA, no plain loads of handed-off tiles) -- so no acquire fence per task.
// Every spin is bounded (s_memrealtime) and watches a global abort word: a scheduling fault ends the launch with an error
// code instead of hanging the device.  Queue order: every dependency of task t is a chain step or a task before t, and a
// workgroup only ever waits on those -- so the launch drains whatever the residency (workgroups that never start claim nothing).
#pragma once
#include <algorithm>
#include <vector>

#include "gemm_f64.h"
#include "gprx_common.h"
#include "potrf.h"
#include "tile_ops.h"

namespace gprx {

struct DagTask {
  uint16_t i, ni, j, k0, k1, pad;  // tiles (i .. i+ni-1, j); TRSM when j == k0 (then k1 == k0 + 1), else update by column blocks [k0, k1);
                                   // pad == 1: the fused task of dag_critical (row block i = k0 + 2 of column k0)
};

constexpr int DAG_HEAD = 0, DAG_ROLE = 1, DAG_ABORT = 2, DAG_CHAINKEY = 3, DAG_HDR = 16;  // ints
constexpr int DAG_ERR_TIMEOUT = 1;

struct DagArgs {
  double* A;
  int64_t lda;
  int T;  // 64-column blocks
  int R;  // 64-row blocks (T + extra / 64)
  double* inv_diag;
  int* info;
  int* st;  // [DAG_HDR header | diag[T] | ver[R][T]]
  const DagTask* tasks;
  int ntasks;
  int col_base;
  unsigned long long timeout_ticks;  // s_memrealtime ticks (100 MHz)
  unsigned long long* stamps;        // optional (GPRX_DAG_STAMPS): chain [k][4] phase ends, then per workgroup {wait, work, tasks, first}
};
#define DAG_STAMP(slot)                                                                        \
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)4 * k + (slot)] = __builtin_amdgcn_s_memrealtime();

__host__ __device__ inline int dag_diag_off(int T) { return DAG_HDR; }
__host__ __device__ inline int dag_ver_off(int T) { return DAG_HDR + ((T + 15) / 16) * 16; }
__host__ __device__ inline size_t dag_state_ints(int T, int R) { return (size_t)dag_ver_off(T) + (((size_t)R * T + 15) / 16) * 16; }


// bounded wait of ONE wave: `ready()` is evaluated by every lane (lanes without a condition pass true)
template <class F>
__device__ __forceinline__ bool dag_wait(const DagArgs& p, F ready) {
  unsigned long long t0 = 0;
  for (unsigned spins = 0;; ++spins) {
    if (__all(ready())) return true;
    if (ld_agent(p.st + DAG_ABORT) != 0) return false;
    if ((spins & 255u) == 255u) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0)
        t0 = now;
      else if (now - t0 > p.timeout_ticks) {
        st_agent(p.st + DAG_ABORT, DAG_ERR_TIMEOUT);
        return false;
      }
    }
    // back off: the polls of ~500 waiting workgroups bypass L1 and go to the memory side -- at one poll per ~100 ns each they
    // slowed every hand-off on the chip (5 us per hop instead of 2); a satisfied wait returns before its first sleep
    if (spins < 4)
      __builtin_amdgcn_s_sleep(2);
    else if (spins < 16)
      __builtin_amdgcn_s_sleep(8);
    else if (spins < 64)
      __builtin_amdgcn_s_sleep(24);
    else
      __builtin_amdgcn_s_sleep(48);
  }
}


__device__ __forceinline__ int* dag_ver(const DagArgs& p, int i, int j) { return p.st + dag_ver_off(p.T) + (size_t)i * p.T + j; }

// wave 0 polls one word until it equals `want`; the outcome reaches every thread through LDS (uniform exit on abort)
__device__ __forceinline__ bool chain_wait(const DagArgs& p, const int* word, int want, int* s_ok) {
  if (threadIdx.x < 64) {
    const bool ok = dag_wait(p, [&]() { return ld_agent(word) == want; });
    if (threadIdx.x == 0) *s_ok = ok ? 1 : 0;
  }
  lds_barrier();
  const bool ok = *s_ok != 0;
  lds_barrier();  // (the word may be rewritten by the next wait)
  return ok;
}

__device__ void dag_chain(const DagArgs& p, double* __restrict__ smem, int* s_ok) {
  ChainCtx c;
  c.sIn = smem;
  c.sX = smem + 128 * PSUB;
  double* sT = smem + 2 * 128 * PSUB;  // 64 x DAG_T_LD: L^-1 (row c, column m), then L(k+1,k)
  c.tid = threadIdx.x;
  const int lane = c.tid & 63;
  c.wave = c.tid >> 6;
  c.g = lane >> 4;
  c.r = lane & 15;
  c.bad = 0;
  const int wave = c.wave, g = c.g, r = c.r, tid = c.tid;
  const unsigned ldb = (unsigned)p.lda * 8u;  // row pitch in bytes
  // per-lane byte offsets inside a 64 x 64 tile: C/D layout (row 16 w + g [+ 4 q], column r [+ 16 kt]) and A-operand layout
  // (row 16 w + r, k = 4 g [+ 16 s])
  const unsigned off_cd = (unsigned)(16 * wave + g) * ldb + (unsigned)r * 8u;
  const unsigned off_a = (unsigned)(16 * wave + r) * ldb + (unsigned)g * 32u;
  int first_bad = 0;  // 1-based index of the first non-positive pivot
  d4 acc[2][4];
  {  // step 0: the diagonal block straight from the matrix
    const __amdgpu_buffer_rsrc_t rs = dag_rsrc(p.A);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[0][kt][q] = ld1_sc1(rs, off_cd, (unsigned)(4 * q) * ldb + (unsigned)kt * 128u);
  }
  int pending = -1;  // step whose flags are not out yet: its stores are drained and published INSIDE the next step (no stall)
  auto publish = [&](int kk) {  // every wave: its stores of step kk have landed; then one lane sets the flags
    drain_stores();
    lds_barrier();
    if (tid == 0) {
      st_agent(p.st + dag_diag_off(p.T) + kk, 1);
      st_agent(dag_ver(p, kk, kk), kk + 1);
      if (kk + 1 < p.R) st_agent(dag_ver(p, kk + 1, kk), kk + 1);
    }
  };
  for (int k = 0; k < p.T; ++k) {
    // masks: triangle of the diagonal block, identity rows
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * wave + g + 4 * q, col = 16 * kt + r;
        acc[0][kt][q] = (col > row) ? 0.0 : acc[0][kt][q];
        acc[1][kt][q] = (col == row) ? 1.0 : 0.0;
      }
    const bool rows_below = k + 1 < p.R;
    const bool more_cols = k + 1 < p.T;
    chain_step<0>(acc, c);
    if (pending >= 0) publish(pending);  // (about 2 us after those stores were issued: the drain costs nothing here)
    pending = -1;
    chain_step<1>(acc, c);
    chain_step<2>(acc, c);
    chain_step<3>(acc, c);
    chain_step<4>(acc, c);
    // Early look at the two tiles this step needs from the workers -- A(k+1,k) and A(k+1,k+1), both through column k-1: one lane
    // asks now, the answer is taken after the next sub-panel (the round trip hides under it) and reaches every thread through LDS
    int v1 = 0, v2 = 0;
    if (tid == 0) {
      v1 = rows_below ? ld_agent(dag_ver(p, k + 1, k)) : k;
      v2 = more_cols ? ld_agent(dag_ver(p, k + 1, k + 1)) : k;
    }
    chain_step<5>(acc, c);
    if (tid == 0) *s_ok = (v1 == k && v2 == k) ? 1 : 0;
    chain_step<6>(acc, c);  // (its barriers publish *s_ok)
    const bool early = *s_ok != 0;
    double* Lrow = p.A + (int64_t)(k + 1) * NB * p.lda + (int64_t)k * NB;
    d2 fa[4][2];
    if (early && rows_below) {  // operands of the chain's own TRSM: requested under the last sub-panel
      const __amdgpu_buffer_rsrc_t ra = dag_rsrc(Lrow);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        fa[s][0] = ld2_sc1(ra, off_a, (unsigned)s * 128u);
        fa[s][1] = ld2_sc1(ra, off_a, (unsigned)s * 128u + 16u);
      }
    }
    chain_step<7>(acc, c);
    DAG_STAMP(0)
    if (c.bad > 0 && first_bad == 0) first_bad = k * NB + c.bad;
    c.bad = 0;
    double* Lkk = p.A + (int64_t)k * NB * p.lda + (int64_t)k * NB;
    // L(k,k) -> matrix (write-through, not waited for), L^-T band -> sT transposed (sT[c][i] = L^-1[c][i])
    {
      const __amdgpu_buffer_rsrc_t rs = dag_rsrc(Lkk);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          st1_sc1(rs, off_cd, acc[0][kt][q], (unsigned)(4 * q) * ldb + (unsigned)kt * 128u);
          sT[(16 * kt + r) * DAG_T_LD + 16 * wave + g + 4 * q] = acc[1][kt][q];
        }
    }
    bool ok = true;
    if (!early) {
      if (rows_below) ok = chain_wait(p, dag_ver(p, k + 1, k), k, s_ok);
      if (ok && more_cols) ok = chain_wait(p, dag_ver(p, k + 1, k + 1), k, s_ok);
      if (!ok) break;
      if (rows_below) {
        const __amdgpu_buffer_rsrc_t ra = dag_rsrc(Lrow);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          fa[s][0] = ld2_sc1(ra, off_a, (unsigned)s * 128u);
          fa[s][1] = ld2_sc1(ra, off_a, (unsigned)s * 128u + 16u);
        }
      }
    }
    double cold[4][4];
    if (more_cols) {  // the next diagonal block's contents: needed only after the two tile products below
      const __amdgpu_buffer_rsrc_t rc = dag_rsrc(p.A + (int64_t)(k + 1) * NB * p.lda + (int64_t)(k + 1) * NB);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) cold[kt][q] = (kt <= wave) ? ld1_sc1(rc, off_cd, (unsigned)(4 * q) * ldb + (unsigned)kt * 128u) : 0.0;
    }
    lds_barrier();  // sT holds L^-1
    {  // L(k,k)^-1 -> its place behind the matrix (the triangular solves and the workers' TRSMs read it)
      const __amdgpu_buffer_rsrc_t ri = dag_rsrc(p.inv_diag + (int64_t)k * NB * NB);
      store_inverse_block<true>(ri, sT, tid);
    }
    DAG_STAMP(1)
    if (rows_below) {
      // ---- L(k+1,k) = A(k+1,k) L(k,k)^-T : wave w takes rows 16 w .. 16 w + 15, all 64 columns (triangular in k)
      d4 xacc[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) xacc[b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int b = s; b < 4; ++b) {  // L^-1[c][m] = 0 for m > c: column tile b needs stages s <= b only
          const double* bp = sT + (16 * b + r) * DAG_T_LD + 16 * s + 4 * g;
          const d2 lo = *reinterpret_cast<const d2*>(bp), hi = *reinterpret_cast<const d2*>(bp + 2);
          xacc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s][0].x, lo.x, xacc[b], 0, 0, 0);
          xacc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s][0].y, lo.y, xacc[b], 0, 0, 0);
          xacc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s][1].x, hi.x, xacc[b], 0, 0, 0);
          xacc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s][1].y, hi.y, xacc[b], 0, 0, 0);
        }
      lds_barrier();  // every wave has read its L^-1 operands (LDS and the copy to memory): sT becomes the image of L(k+1,k)
      const __amdgpu_buffer_rsrc_t rx = dag_rsrc(Lrow);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          st1_sc1(rx, off_cd, xacc[b][q], (unsigned)(4 * q) * ldb + (unsigned)b * 128u);
          sT[(16 * wave + g + 4 * q) * DAG_T_LD + 16 * b + r] = xacc[b][q];
        }
      lds_barrier();
    }
    pending = k;
    DAG_STAMP(2)
    if (!more_cols) break;
    // ---- next diagonal block: A(k+1,k+1) - L(k+1,k) L(k+1,k)^T, band w in wave w
    {
      d4 u[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) u[kt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double* ap = sT + (16 * wave + r) * DAG_T_LD + 16 * s + 4 * g;
        const d2 alo = *reinterpret_cast<const d2*>(ap), ahi = *reinterpret_cast<const d2*>(ap + 2);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          if (kt <= wave) {
            const double* bp = sT + (16 * kt + r) * DAG_T_LD + 16 * s + 4 * g;
            const d2 lo = *reinterpret_cast<const d2*>(bp), hi = *reinterpret_cast<const d2*>(bp + 2);
            u[kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(alo.x, lo.x, u[kt], 0, 0, 0);
            u[kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(alo.y, lo.y, u[kt], 0, 0, 0);
            u[kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ahi.x, hi.x, u[kt], 0, 0, 0);
            u[kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ahi.y, hi.y, u[kt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[0][kt][q] = __builtin_fma(1.0, cold[kt][q], -1.0 * u[kt][q]);
    }
    lds_barrier();  // sT is rewritten by the next step
    DAG_STAMP(3)
  }
  if (pending >= 0) publish(pending);
  if (tid == 0 && first_bad > 0) atomicCAS(p.info, 0, p.col_base + first_bad);
}

// The task chain step k+1 waits for, fused (one claim, one poll, ONE round of loads, one drain): row block i = k+2 of column k
//   X = A(i,k) L(k,k)^-T  -> L(i,k);   A(i,k+1) -= X L(k+1,k)^T;   A(i,i) -= X X^T   (the last two where those tiles exist).
// X never leaves the workgroup between the three products (it is stored for everyone else, not re-read).
__device__ __forceinline__ void dag_critical(const DagArgs& p, int k, double* __restrict__ smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, r = lane & 15;
  const unsigned ldb = (unsigned)p.lda * 8u;
  const int i = k + 2;
  const bool has1 = k + 1 < p.T, has2 = i < p.T;
  double* sA = smem;
  double* sB = smem + NB * NB;
  const int lrow0 = tid >> 5, c32 = tid & 31;
  const unsigned off_ld = (unsigned)lrow0 * ldb + (unsigned)c32 * 16u;
  const unsigned off_ld_inv = (unsigned)lrow0 * (NB * 8u) + (unsigned)c32 * 16u;
  const int st_img = (c32 >> 3) * (NB * GEMM_BK), st_cc = c32 & 7;
  const unsigned off_cd = (unsigned)(wm * 32 + g) * ldb + (unsigned)(wn * 32 + r) * 8u;
  const int swz = kc_swz(r);
  double* Aik = p.A + (int64_t)i * NB * p.lda + (int64_t)k * NB;
  const __amdgpu_buffer_rsrc_t rsa = dag_rsrc(Aik);
  const __amdgpu_buffer_rsrc_t rsi = dag_rsrc(p.inv_diag + (int64_t)k * NB * NB);
  const __amdgpu_buffer_rsrc_t rsl = dag_rsrc(p.A + (int64_t)(k + 1) * NB * p.lda + (int64_t)k * NB);
  const __amdgpu_buffer_rsrc_t rc1 = dag_rsrc(p.A + (int64_t)i * NB * p.lda + (int64_t)(k + 1) * NB);
  const __amdgpu_buffer_rsrc_t rc2 = dag_rsrc(p.A + (int64_t)i * NB * p.lda + (int64_t)i * NB);
  d2 ra[8], rb[8], rl[8];
  double c1[2][2][4], c2[2][2][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ra[e] = ld2_sc1(rsa, off_ld, (unsigned)(8 * e) * ldb);
    rb[e] = ld2_sc1(rsi, off_ld_inv, (unsigned)(8 * e) * (NB * 8u));
  }
  if (has1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) rl[e] = ld2_sc1(rsl, off_ld, (unsigned)(8 * e) * ldb);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) c1[a][b][q] = ld1_sc1(rc1, off_cd, (unsigned)(a * 16 + 4 * q) * ldb + (unsigned)b * 128u);
  }
  auto to_lds = [&](double* img, const d2 (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = lrow0 + 8 * e;
      *reinterpret_cast<d2*>(img + st_img + row * GEMM_BK + ((st_cc ^ kc_swz(row)) * 2)) = v[e];
    }
  };
  to_lds(sA, ra);
  to_lds(sB, rb);
  lds_barrier();
  d4 x[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) x[a][b] = d4{0.0, 0.0, 0.0, 0.0};
  dag_mma64(x, sA, sB, wm, wn, g, r, swz);
  if (has2) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) c2[a][b][q] = ld1_sc1(rc2, off_cd, (unsigned)(a * 16 + 4 * q) * ldb + (unsigned)b * 128u);
  }
  lds_barrier();  // both images have been read: X becomes the A image, L(k+1,k) the B image
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = wm * 32 + a * 16 + g + 4 * q, col = wn * 32 + b * 16 + r;
        st1_sc1(rsa, off_cd, x[a][b][q], (unsigned)(a * 16 + 4 * q) * ldb + (unsigned)b * 128u);
        sA[(col >> 4) * (NB * GEMM_BK) + row * GEMM_BK + ((((col & 15) >> 1) ^ kc_swz(row)) * 2) + (col & 1)] = x[a][b][q];
      }
  if (has1) to_lds(sB, rl);
  lds_barrier();
  if (has1) {
    d4 u[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) u[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    dag_mma64(u, sA, sB, wm, wn, g, r, swz);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          st1_sc1(rc1, off_cd, __builtin_fma(1.0, c1[a][b][q], -1.0 * u[a][b][q]), (unsigned)(a * 16 + 4 * q) * ldb + (unsigned)b * 128u);
  }
  if (has2) {
    d4 u[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) u[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    dag_mma64(u, sA, sA, wm, wn, g, r, swz);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          st1_sc1(rc2, off_cd, __builtin_fma(1.0, c2[a][b][q], -1.0 * u[a][b][q]), (unsigned)(a * 16 + 4 * q) * ldb + (unsigned)b * 128u);
  }
}


__device__ __forceinline__ unsigned dag_cu_key() {
  // (se, sh, cu) of HW_ID and the XCC id: two workgroups with equal keys share a CU
  const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
  const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
  return ((hw & 0xff00u) | (xcc << 16)) + 1u;
}

__global__ __launch_bounds__(256, 2) void potrf_dag_kernel(DagArgs p) {
  __shared__ __attribute__((aligned(16))) double smem[DAG_SMEM];
  __shared__ int s_word[2];
  const int tid = threadIdx.x;
  const unsigned key = dag_cu_key();
  if (tid == 0) {
    const int role = atomicAdd(p.st + DAG_ROLE, 1);
    if (role == 0) st_agent(p.st + DAG_CHAINKEY, (int)key);
    s_word[0] = role;
  }
  __syncthreads();
  if (s_word[0] == 0) {
    dag_chain(p, smem, &s_word[1]);
    return;
  }
  unsigned long long t_wait = 0, t_work = 0, n_done = 0, t_first = 0;
  for (;;) {
    __syncthreads();  // (s_word and smem of the previous task are free)
    if (tid == 0) {
      int t = -1;
      // the chain owns its CU: a worker that landed beside it retires (the chain's key is published before its first flag)
      if (ld_agent(p.st + DAG_CHAINKEY) != (int)key && ld_agent(p.st + DAG_ABORT) == 0) t = atomicAdd(p.st + DAG_HEAD, 1);
      s_word[0] = t;
    }
    __syncthreads();
    const int t = s_word[0];
    if (t < 0 || t >= p.ntasks) {
      if (p.stamps && tid == 0) {
        unsigned long long* w = p.stamps + (size_t)4 * p.T + (size_t)4 * blockIdx.x;
        w[0] = t_wait; w[1] = t_work; w[2] = n_done; w[3] = t_first;
      }
      return;
    }
    const unsigned long long s0 = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    if (n_done == 0) t_first = s0;
    const DagTask task = p.tasks[t];
    const int i = task.i, ni = task.ni, j = task.j, k0 = task.k0, k1 = task.k1;
    const bool trsm = (j == k0), critical = task.pad == 1;
    if (tid < 64) {
      bool ok;
      if (critical) {
        // flags of chain step k0 (L(k0,k0)^-1 and L(k0+1,k0) go out together) and the three tiles through column k0-1
        const int* d = p.st + dag_diag_off(p.T) + k0;
        const int* v = p.st;
        bool active = false;
        if (tid == 1) { v = dag_ver(p, i, k0); active = true; }
        if (tid == 2 && k0 + 1 < p.T) { v = dag_ver(p, i, k0 + 1); active = true; }
        if (tid == 3 && i < p.T) { v = dag_ver(p, i, i); active = true; }
        ok = dag_wait(p, [&]() { return tid == 0 ? ld_agent(d) != 0 : (active ? ld_agent(v) == k0 : true); });
      } else if (trsm) {
        // lane 0: the diagonal block's inverse is published; lanes 16..16+ni-1: tile (i + l, j) has every earlier update
        const int* d = p.st + dag_diag_off(p.T) + k0;
        const int l = tid - 16;
        const int* v = dag_ver(p, i + (l >= 0 && l < ni ? l : 0), j);
        ok = dag_wait(p, [&]() { return tid == 0 ? ld_agent(d) != 0 : ((l >= 0 && l < ni) ? ld_agent(v) == k0 : true); });
      } else {
        // lanes 0..7: L(j, k0 + l) final; lanes 16..16+ni-1: tile (i + l, j) has every earlier update; lanes 32..63: L(i + t, k0 + l)
        // final for (t, l) = (lane - 32) / 8, (lane - 32) % 8; ranges longer than 8 blocks poll in rounds of 8
        bool all = true;
        for (int base = 0; base < k1 - k0; base += 8) {
          const int* v = p.st;
          int want = 0;
          bool active = false;
          if (tid < 8) {
            const int c = k0 + base + tid;
            active = c < k1;
            if (active) v = dag_ver(p, j, c);
            want = c + 1;
          } else if (tid >= 16 && tid < 32) {
            active = base == 0 && tid - 16 < ni;
            if (active) v = dag_ver(p, i + tid - 16, j);
            want = k0;
          } else if (tid >= 32) {
            const int tt = (tid - 32) >> 3, c = k0 + base + ((tid - 32) & 7);
            active = tt < ni && c < k1;
            if (active) v = dag_ver(p, i + tt, c);
            want = c + 1;
          }
          all = all && dag_wait(p, [&]() { return active ? ld_agent(v) == want : true; });
        }
        ok = all;
      }
      if (tid == 0) s_word[1] = ok ? 1 : 0;
    }
    __syncthreads();
    if (s_word[1] == 0) return;
    const unsigned long long s1 = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    const TileCtx tc{p.A, p.lda, p.inv_diag};
    if (critical)
      dag_critical(p, k0, smem);
    else if (trsm)
      dag_panel<true>(tc, i, ni, j, k0, k1, smem);
    else
      dag_panel<false>(tc, i, ni, j, k0, k1, smem);
    drain_stores();
    __syncthreads();
    if (critical) {
      if (tid == 0) st_agent(dag_ver(p, i, k0), k0 + 1);
      if (tid == 1 && k0 + 1 < p.T) st_agent(dag_ver(p, i, k0 + 1), k0 + 1);
      if (tid == 2 && i < p.T) st_agent(dag_ver(p, i, i), k0 + 1);
    } else if (tid < ni) {
      st_agent(dag_ver(p, i + tid, j), trsm ? k0 + 1 : k1);
    }
    if (p.stamps) {
      const unsigned long long s2 = __builtin_amdgcn_s_memrealtime();
      t_wait += s1 - s0;
      t_work += s2 - s1;
      ++n_done;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
struct DagPlan {
  int T = -1, R = -1;
  DagTask* tasks = nullptr;
  int ntasks = 0;
  int* state = nullptr;
  size_t state_bytes = 0;
  int grid = 0;
  unsigned long long* stamps = nullptr;  // GPRX_DAG_STAMPS=1: 4 T chain stamps + 4 per workgroup (read with gprx_dag_stamps)
  size_t stamp_words = 0;
  void destroy() {
    if (tasks) hipFree(tasks);
    if (state) hipFree(state);
    if (stamps) hipFree(stamps);
    tasks = nullptr;
    state = nullptr;
    stamps = nullptr;
    T = R = -1;
  }
};

// Static priority order.  Column by column (k); inside a column first the TRSMs (rows k+2 .. R-1: row k+1 is the chain's),
// then the updates column-major from the front: tile column j = k+1 first -- the chain needs A(k+2,k+1) next.  The chain
// itself computes the update of (k+1,k+1).  Tasks are panels of up to DAG_NI vertically adjacent tiles.
inline std::vector<DagTask> dag_build_tasks(int T, int R) {
  std::vector<DagTask> v;
  int pn = DAG_NI;
  if (const char* e = getenv("GPRX_DAG_NI")) pn = atoi(e) >= 1 && atoi(e) <= DAG_NI ? atoi(e) : DAG_NI;
  auto panels = [&](int ifirst, int ilast, int j, int k0, int k1) {  // rows ifirst .. ilast-1 in panels of pn tiles
    for (int i = ifirst; i < ilast; i += pn) {
      const int ni = ilast - i < pn ? ilast - i : pn;
      v.push_back(DagTask{(uint16_t)i, (uint16_t)ni, (uint16_t)j, (uint16_t)k0, (uint16_t)k1, 0});
    }
  };
  for (int k = 0; k < T; ++k) {
    // what chain step k+1 waits for, as single tiles at the head of the column's tasks (shortest latency): L(k+2,k), then
    // A(k+2,k+1) and A(k+2,k+2) through column k
    if (k + 2 < R) v.push_back(DagTask{(uint16_t)(k + 2), 1, (uint16_t)k, (uint16_t)k, (uint16_t)(k + 1), 1});  // fused (dag_critical)
    panels(k + 3, R, k, k, k + 1);  // the other TRSMs of column k
    for (int j = k + 1; j < T; ++j) panels(j <= k + 2 ? k + 3 : j, R, j, k, k + 1);  // ((k+1,k+1) is the chain's)
  }
  return v;
}

// "Lazy" schedule (GPRX_DAG_SCHED=1): what the stamps of the eager schedule asked for.  The dependent loop of a row block i --
// TRSM(i,c) -> update of (i,c+1) by column c -> TRSM(i,c+1) -- stays two SINGLE-tile tasks per column (7 us each, measured);
// every other update of a tile is aggregated over K:
//   tile (i,j) with `pend` worker-owned columns (j; j-1 for i = j+1 and j-2 for i = j: the rest belongs to the fused critical task
//   and the chain):  lazy chunks [8m, 8m+8) while 8m+8 <= B,  one catch-up [B, pend-2),  two prompt singles [pend-2, pend-1),
//   [pend-1, pend);   B = 8 floor((pend-2)/8) - 8 (>= 0), so a catch-up spans 8..15 column blocks and has two chain steps of slack.
// Emission (the queue is claimed in order, so this IS the priority): at step k the fused critical task, the TRSMs of column k, the
// prompt singles ending at k+1, the catch-ups ending at k+1, then the slice of lazy chunks due now (chunk m is spread over the
// steps 8m+7 .. 8m+14 by tile column: a burst of ~1500 K = 512 tasks in front of the next column's TRSMs would stall every row).
// Per tile the tasks appear in plan order (a claimed task only ever waits for earlier ones).
inline std::vector<DagTask> dag_build_tasks_lazy(int T, int R) {
  constexpr int G = 8;
  std::vector<std::vector<DagTask>> at(T + 1);  // tasks by emission step, already in intra-step order classes
  std::vector<std::vector<DagTask>> prompt(T + 1), catchup(T + 1), lazy(T + 1);
  auto pend_of = [](int i, int j) { return i == j ? j - 2 : (i == j + 1 ? j - 1 : j); };
  for (int j = 1; j < T; ++j) {
    // rows of this tile column grouped by plan: i == j, i == j + 1 (their own plans), i >= j + 2 (one plan, panels of rows)
    auto emit_plan = [&](int i0, int ni, int pend) {
      if (pend <= 0) return;
      const int pfirst = pend - 2 > 0 ? pend - 2 : 0;  // prompt singles cover [pfirst, pend)
      int B = ((pend - 2) / G) * G - G;
      if (B < 0 || pend - 2 <= 0) B = 0;
      for (int m = 0; G * m + G <= B; ++m) {
        int step = G * m + G - 1 + ((j - (G * m + G)) * G) / (T - (G * m + G) > 0 ? T - (G * m + G) : 1);
        if (step > pend - 4) step = pend - 4;
        if (step < G * m + G - 1) step = G * m + G - 1;
        for (int r0 = 0; r0 < ni; r0 += DAG_NI)
          lazy[step].push_back(DagTask{(uint16_t)(i0 + r0), (uint16_t)(ni - r0 < DAG_NI ? ni - r0 : DAG_NI), (uint16_t)j, (uint16_t)(G * m), (uint16_t)(G * m + G), 0});
      }
      if (pfirst > B)
        for (int r0 = 0; r0 < ni; r0 += DAG_NI)
          catchup[pfirst - 1].push_back(DagTask{(uint16_t)(i0 + r0), (uint16_t)(ni - r0 < DAG_NI ? ni - r0 : DAG_NI), (uint16_t)j, (uint16_t)B, (uint16_t)pfirst, 0});
      for (int c = pfirst; c < pend; ++c)
        for (int r = 0; r < ni; ++r) prompt[c].push_back(DagTask{(uint16_t)(i0 + r), 1, (uint16_t)j, (uint16_t)c, (uint16_t)(c + 1), 0});
    };
    emit_plan(j, 1, pend_of(j, j));
    if (j + 1 < R) emit_plan(j + 1, 1, pend_of(j + 1, j));
    if (j + 2 < R) emit_plan(j + 2, R - (j + 2), j);
  }
  std::vector<DagTask> v;
  for (int k = 0; k < T; ++k) {
    if (k + 2 < R) v.push_back(DagTask{(uint16_t)(k + 2), 1, (uint16_t)k, (uint16_t)k, (uint16_t)(k + 1), 1});  // fused (dag_critical)
    for (int i = k + 3; i < R; ++i) v.push_back(DagTask{(uint16_t)i, 1, (uint16_t)k, (uint16_t)k, (uint16_t)(k + 1), 0});  // TRSMs, singles
    // prompt singles of this step: the tile column next to the front first (it feeds the next TRSMs)
    std::stable_sort(prompt[k].begin(), prompt[k].end(), [](const DagTask& a, const DagTask& b) { return a.j < b.j; });
    v.insert(v.end(), prompt[k].begin(), prompt[k].end());
    v.insert(v.end(), catchup[k].begin(), catchup[k].end());
    v.insert(v.end(), lazy[k].begin(), lazy[k].end());
  }
  return v;
}

inline hipError_t dag_ensure_plan(DagPlan& plan, int T, int R, hipStream_t st) {
  if (plan.T == T && plan.R == R) return hipSuccess;
  plan.destroy();
  const bool lazy_sched = getenv("GPRX_DAG_SCHED") && atoi(getenv("GPRX_DAG_SCHED")) == 1;
  const std::vector<DagTask> tasks = lazy_sched ? dag_build_tasks_lazy(T, R) : dag_build_tasks(T, R);
  hipError_t e = hipMalloc((void**)&plan.tasks, sizeof(DagTask) * (tasks.size() + 1));
  if (e != hipSuccess) return e;
  plan.state_bytes = sizeof(int) * dag_state_ints(T, R);
  if ((e = hipMalloc((void**)&plan.state, plan.state_bytes)) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(plan.tasks, tasks.data(), sizeof(DagTask) * tasks.size(), hipMemcpyHostToDevice, st)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;  // (the host vector dies with this scope)
  plan.ntasks = (int)tasks.size();
  int dev = 0, cus = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  plan.grid = cus > 0 ? cus : 256;  // one workgroup per CU: two per CU tripled the time of a task (LDS, MFMA pipe and polls shared)
  if (const char* g = getenv("GPRX_DAG_GRID")) plan.grid = atoi(g) > 1 ? atoi(g) : plan.grid;
  if (getenv("GPRX_DAG_STAMPS")) {
    plan.stamp_words = (size_t)4 * T + (size_t)4 * plan.grid;
    if ((e = hipMalloc((void**)&plan.stamps, sizeof(unsigned long long) * plan.stamp_words)) != hipSuccess) return e;
  }
  plan.T = T;
  plan.R = R;
  return hipSuccess;
}

// Factor the np x np matrix (np a multiple of 64) with `extra` (0 or 64) right-hand-side rows below it.  info must be zeroed
// by the caller; plan.state[DAG_ABORT] != 0 after the launch means the scheduler gave up (read it back with the results).
inline hipError_t potrf_dag(hipStream_t st, double* A, int64_t lda, int np, int extra, double* inv_diag, int* info, DagPlan& plan,
                            int col_base = 0) {
  const int T = np / NB, R = T + extra / NB;
  hipError_t e = dag_ensure_plan(plan, T, R, st);
  if (e != hipSuccess) return e;
  if ((e = hipMemsetAsync(plan.state, 0, plan.state_bytes, st)) != hipSuccess) return e;
  DagArgs a;
  a.A = A;
  a.lda = lda;
  a.T = T;
  a.R = R;
  a.inv_diag = inv_diag;
  a.info = info;
  a.st = plan.state;
  a.tasks = plan.tasks;
  a.ntasks = plan.ntasks;
  a.col_base = col_base;
  a.timeout_ticks = 300000000ull;  // 3 s of s_memrealtime (100 MHz)
  a.stamps = plan.stamps;
  if (plan.stamps && (e = hipMemsetAsync(plan.stamps, 0, sizeof(unsigned long long) * plan.stamp_words, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(potrf_dag_kernel, dim3(plan.grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace gprx
