// Tile-level building blocks shared by the one-workgroup-per-cell Cholesky (potrf_cell.h) and the opt-in tile-DAG scheduler
// (potrf_dag.h): the write-through / L1-bypassing buffer accessors, the chain's sub-panel step on [64 diagonal rows | 64 identity rows]
// (chain_step: L(k,k) and L(k,k)^-1 in one workgroup), the 64 x 64 x 64 tile product on swizzled LDS stage images (dag_mma64) and the
// panel pass of several row tiles against one B block (dag_panel).  Round 4: split out of potrf_dag.h so that the product path (the
// cell kernel) does not include the scheduler (task queue, version counters, persistent kernel), which is an opt-in of its own.
#pragma once
#include "gemm_f64.h"
#include "chain64.h"
#include "gprx_common.h"
#include "potrf.h"

namespace gprx {

constexpr int DAG_NI = 4;              // tiles per panel task (one claim, one dependency poll, operands prefetched tile by tile)
constexpr int DAG_T_LD = NB + 2;       // LDS row stride of the 64 x 64 operand image of the chain (16-byte aligned rows)
constexpr int DAG_SMEM = 2 * NB * NB;  // doubles: workers: A block | B block (4 stage images each); chain: sIn | sX | sT (52 KB of it)
static_assert(2 * PanelGeom<2>::kWgRows * PSUB + NB * DAG_T_LD <= DAG_SMEM, "the chain's buffers fit into the workers' LDS");

// every shared word goes through GLOBAL (never flat) agent-scope accesses; every handed-off double through buffer accesses with
// the sc1 bit (aux 16): stores write through, loads bypass the CU's L1.  Addresses = descriptor base + per-lane byte offset
// (one VGPR) + wave-uniform byte offset (SGPR): no 64-bit per-lane pointers, so the address arithmetic costs no registers.
typedef __attribute__((address_space(1))) int gint;
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load((gint*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store((gint*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dag_rsrc(const double* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, 0xffffffff, 0x00020000);
}
// SC1 = false: plain (cached) accesses for data no other workgroup touches (potrf_cell.h)
template <bool SC1 = true>
__device__ __forceinline__ d2 ld2_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff = 0) {
  const u4v v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, SC1 ? 16 : 0);
  d2 out;
  __builtin_memcpy(&out, &v, 16);
  return out;
}
template <bool SC1 = true>
__device__ __forceinline__ void st2_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, d2 v, unsigned soff = 0) {
  u4v raw;
  __builtin_memcpy(&raw, &v, 16);
  __builtin_amdgcn_raw_buffer_store_b128(raw, r, voff, soff, SC1 ? 16 : 0);
}
template <bool SC1 = true>
__device__ __forceinline__ double ld1_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff = 0) {
  const u2v v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, SC1 ? 16 : 0);
  double out;
  __builtin_memcpy(&out, &v, 8);
  return out;
}
template <bool SC1 = true>
__device__ __forceinline__ void st1_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, double v, unsigned soff = 0) {
  u2v raw;
  __builtin_memcpy(&raw, &v, 8);
  __builtin_amdgcn_raw_buffer_store_b64(raw, r, voff, soff, SC1 ? 16 : 0);
}
// The 64 x 64 inverse block from its LDS image sT (row c, column m; rows DAG_T_LD apart) to memory, 16 bytes per lane and instruction.
// ALL eight values are read first, the eight stores are issued back to back, and the data registers are kept alive until the stores have
// completed (s_waitcnt vmcnt(0), then an empty asm that still names them).  Round 4: written as a loop { read 16 bytes from LDS; store
// them } the compiler reused the first data register for the next LDS address immediately after each buffer_store_dwordx4 -- it
// inserts no wait state there when the store takes its offset from an SGPR -- and on gfx950 the store unit had not always read its
// data by then: with two workgroups per CU the LOW DWORD of the first double of a store came out as that address in 7-50 % of the
// cells of the workgroups that became resident second (relative error ~5e-7 in a few entries of L(j,j)^-1, differently on every
// run; found with tools/cell_check.hip, which compares the factors of two kernels element by element at full load).
template <bool SC1>
__device__ __forceinline__ void store_inverse_block(__amdgpu_buffer_rsrc_t ri, const double* __restrict__ sT, int tid) {
  d2 iv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int q = tid + 256 * e;  // 2048 chunks of 16 bytes
    iv[e] = *reinterpret_cast<const d2*>(sT + (q >> 5) * (NB + 2) + 2 * (q & 31));
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) st2_sc1<SC1>(ri, (unsigned)tid * 16u, iv[e], (unsigned)e * 4096u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(iv[e].x), "v"(iv[e].y));
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ---- workers -------------------------------------------------------------------------------------------------------------
// acc (this wave's 32 x 32 quarter of a 64 x 64 tile) += A B^T over one 64-deep block; ia / ib: the operands' four stage
// images [64 rows][16 k] in LDS, chunks XOR-swizzled (gemm_f64.h kc_swz)
__device__ __forceinline__ void dag_mma64(d4 (&acc)[2][2], const double* __restrict__ ia, const double* __restrict__ ib, int wm, int wn, int g,
                                          int r, int swz) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const double* pa = ia + s * (NB * GEMM_BK);
    const double* pb = ib + s * (NB * GEMM_BK);
    double fa[2][4], fb[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int row = wm * 32 + a * 16 + r;
      const d2 lo = *reinterpret_cast<const d2*>(pa + row * GEMM_BK + 2 * ((2 * g) ^ swz));
      const d2 hi = *reinterpret_cast<const d2*>(pa + row * GEMM_BK + 2 * ((2 * g + 1) ^ swz));
      fa[a][0] = lo.x; fa[a][1] = lo.y; fa[a][2] = hi.x; fa[a][3] = hi.y;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = wn * 32 + b * 16 + r;
      const d2 lo = *reinterpret_cast<const d2*>(pb + col * GEMM_BK + 2 * ((2 * g) ^ swz));
      const d2 hi = *reinterpret_cast<const d2*>(pb + col * GEMM_BK + 2 * ((2 * g + 1) ^ swz));
      fb[b][0] = lo.x; fb[b][1] = lo.y; fb[b][2] = hi.x; fb[b][3] = hi.y;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a][jj], fb[b][jj], acc[a][b], 0, 0, 0);
  }
}

// One panel task: tiles (i .. i+ni-1, j), C_t = C_t - A_t B^T over the column blocks [k0, k1) (update) or C_t = C_t B^T with
// B = L(k0,k0)^-1 (TRSM, in place).  Steps (k block, tile): the step's 64 x 64 A block (and, on a new k block, the B block) come
// through registers (requested one step ahead) into swizzled LDS stage images; 64 MFMAs per wave and step.  Every global access
// is write-through / L1-bypassing (sc1).
struct TileCtx {
  double* A;
  int64_t lda;
  const double* inv_diag;
};
template <bool TRSM, bool SC1 = true>
__device__ __forceinline__ void dag_panel(const TileCtx& p, int i0, int ni, int j, int k0, int k1, double* __restrict__ smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, r = lane & 15;
  const unsigned ldb = (unsigned)p.lda * 8u;
  double* sA = smem;            // 4 stage images [64 rows][16 k], chunks XOR-swizzled (gemm_f64.h kc_swz)
  double* sB = smem + NB * NB;
  // block load map: chunk q = tid + 256 e (e = 0..7): row = q >> 5, 16-byte chunk c32 = q & 31 of the row's 512 bytes
  const int lrow0 = tid >> 5, c32 = tid & 31;  // row = lrow0 + 8 e
  const unsigned off_ld = (unsigned)lrow0 * ldb + (unsigned)c32 * 16u;
  const unsigned off_ld_inv = (unsigned)lrow0 * (NB * 8u) + (unsigned)c32 * 16u;
  const int st_img = (c32 >> 3) * (NB * GEMM_BK), st_cc = c32 & 7;
  const unsigned off_cd = (unsigned)(wm * 32 + g) * ldb + (unsigned)(wn * 32 + r) * 8u;
  const int swz = kc_swz(r);
  const int nblk = TRSM ? 1 : k1 - k0;
  const int nsteps = nblk * ni;
  const double* Brow = TRSM ? p.inv_diag + (int64_t)k0 * NB * NB : p.A + (int64_t)j * NB * p.lda;
  d4 acc[DAG_NI][2][2];
#pragma unroll
  for (int t = 0; t < DAG_NI; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = d4{0.0, 0.0, 0.0, 0.0};
  d2 ra[8], rb[8];
  double cold[2][2][4];  // C of the tile whose epilogue comes next: tile 0 is requested with the first operands
  auto request_c = [&](int t) {
    if constexpr (!TRSM) {
      const __amdgpu_buffer_rsrc_t rc = dag_rsrc(p.A + (int64_t)(i0 + t) * NB * p.lda + (int64_t)j * NB);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) cold[a][b][q] = ld1_sc1<SC1>(rc, off_cd, (unsigned)(a * 16 + 4 * q) * ldb + (unsigned)b * 128u);
    }
  };
  auto request = [&](int step) {  // A block of (tile step % ni, k block step / ni); B block when the k block changes
    const int blk = step / ni, t = step - blk * ni;
    const int kb = TRSM ? k0 : k0 + blk;
    const __amdgpu_buffer_rsrc_t rsa = dag_rsrc(p.A + (int64_t)(i0 + t) * NB * p.lda + (int64_t)kb * NB);
#pragma unroll
    for (int e = 0; e < 8; ++e) ra[e] = ld2_sc1<SC1>(rsa, off_ld, (unsigned)(8 * e) * ldb);
    if (t == 0) {
      if constexpr (TRSM) {
        const __amdgpu_buffer_rsrc_t rsb = dag_rsrc(Brow);
#pragma unroll
        for (int e = 0; e < 8; ++e) rb[e] = ld2_sc1<SC1>(rsb, off_ld_inv, (unsigned)(8 * e) * (NB * 8u));
      } else {
        const __amdgpu_buffer_rsrc_t rsb = dag_rsrc(Brow + (int64_t)kb * NB);
#pragma unroll
        for (int e = 0; e < 8; ++e) rb[e] = ld2_sc1<SC1>(rsb, off_ld, (unsigned)(8 * e) * ldb);
      }
    }
  };
  auto publish = [&](bool with_b) {  // registers -> LDS stage images
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = lrow0 + 8 * e;
      const int slot = st_img + row * GEMM_BK + ((st_cc ^ kc_swz(row)) * 2);
      *reinterpret_cast<d2*>(sA + slot) = ra[e];
      if (with_b) *reinterpret_cast<d2*>(sB + slot) = rb[e];
    }
  };
  request(0);
  for (int step = 0; step < nsteps; ++step) {
    const int blk = step / ni, t = step - blk * ni;
    lds_barrier();  // every wave has finished reading the previous step's images
    publish(t == 0);
    lds_barrier();
    if (step + 1 < nsteps) request(step + 1);
    else request_c(0);  // under the last step's MFMAs
#pragma unroll
    for (int tt = 0; tt < DAG_NI; ++tt)
      if (tt == t) dag_mma64(acc[tt], sA, sB, wm, wn, g, r, swz);  // (static accumulator index: unrolled, one branch is taken)
  }
  // epilogue, tile by tile: C - acc (one rounding: C + (-1) * sum, as gemm_f64) or the product itself; the next tile's C is
  // requested before this tile's stores
  lds_barrier();  // (TRSM in place: every A block of this panel has been read)
#pragma unroll
  for (int t = 0; t < DAG_NI; ++t) {
    if (t < ni) {
      const __amdgpu_buffer_rsrc_t rc = dag_rsrc(p.A + (int64_t)(i0 + t) * NB * p.lda + (int64_t)j * NB);
      double v[2][2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if constexpr (TRSM)
              v[a][b][q] = acc[t][a][b][q];
            else
              v[a][b][q] = __builtin_fma(1.0, cold[a][b][q], -1.0 * acc[t][a][b][q]);
          }
      if (t + 1 < ni) request_c(t + 1);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) st1_sc1<SC1>(rc, off_cd, v[a][b][q], (unsigned)(a * 16 + 4 * q) * ldb + (unsigned)b * 128u);
    }
  }
}

}  // namespace gprx
