// fp64 MFMA GEMM for gfx950 (v_mfma_f64_16x16x4_f64), row-major operands.
//
//   C[M x N] = alpha * op(A) * op(B) + beta * C
//
// Used for: the Cholesky trailing update (syrk: A = B = panel, C_LOWER), triangular solves
// with matrix right-hand sides (updates + products with inverted diagonal blocks), A A^T of
// the sparse model, K^-1 = L^-T L^-1, and the W P product of the SGPR gradient.
//
// Work decomposition (wave64): a workgroup of 4 waves (2 x 2) owns a BM x BN tile of C; each
// wave owns (BM/2) x (BN/2) = TM x TN MFMA tiles of 16 x 16.  The K loop advances 16 at a
// time through a double-buffered LDS stage (register prefetch of the next stage, one barrier
// per stage).  Operands are staged in the orientation they have in memory, so global loads
// are 16-byte and coalesced along the contiguous dimension for every transpose case:
//   "KC" image  [rows][16 + 2]   when k is the contiguous dimension (A as stored M x K, B as N x K)
//   "MC" image  [16][rows + 4]   when m/n is contiguous            (A stored K x M, B stored K x N)
// MFMA operand lane map (f64 16x16x4): lane l supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; within a 16-deep stage lane group g = l >> 4 takes k = 4g + j for
// the j-th MFMA (the sum over k is order-free as long as A and B agree), so a KC image is read
// with two ds_read_b128 per 16-row fragment.  C/D: col = l & 15, row = (l >> 4) + 4 * reg.
#pragma once
#include <cstdlib>

#include "gprx_common.h"

namespace gprx {

struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  int64_t lda, ldb, ldc;
  int M, N, K;
  double alpha, beta;
  int flags;
  int tiles_n;
  int tiles_m;
  int nwg;  // launched workgroups per batch entry (only tiles that touch the lower triangle when C_LOWER)
  int64_t strideA, strideB, strideC;  // element strides between batch entries (blockIdx.y)
  int ksplit;                          // > 0: blockIdx.z takes K slice [z * ksplit, (z+1) * ksplit) and writes slab z of C
  int64_t slab;                        // elements per slab
  int inner = 0;                       // > 0: blockIdx.y = cell * inner + entry; cells are cellA / cellB / cellC elements apart
  int64_t cellA = 0, cellB = 0, cellC = 0;
  // AXF kernels only (TA == 0): A is read as (g(A[m][k]) - a_sub[k]) * a_mul[k], g = max(. - a_elev[k], 0) if a_elev
  const double* a_sub = nullptr;
  const double* a_mul = nullptr;
  const double* a_elev = nullptr;
  // per-cell alpha (two-level batches): alpha = alpha_tab[cell * alpha_stride] when alpha_tab is set
  const double* alpha_tab = nullptr;
  int alpha_stride = 0;
  int persist_slots = 0;  // > 0 (NT, LDS-DMA eligible, no split-K): persistent grid of persist_slots x #CUs workgroups
  // rowsq != nullptr: C is NOT stored; instead rowsq[(2 tj + wn) * rowsq_ld + row] = sum over the 16 TN columns of this wave
  // of (alpha * acc)^2 -- the row sums of squares of the product in 2 * tiles_n partial slabs (summed by rowsq_final_kernel
  // in a fixed order).  The predictive variance needs only these sums of V^T = Ks^T L^-T, never V itself.
  double* rowsq = nullptr;
  int64_t rowsq_ld = 0;
  // 1: batch entry (cell) -> XCD affinity: workgroups are dealt round-robin over the 8 XCDs in linear order, so linear id L goes to
  // XCD L % 8; XCD x then walks the cells x, x + 8, ... one after the other, all tiles of a cell on it (its operand panels live in
  // ONE L2 instead of eight).  Placement is a speed matter only: any mapping computes the same tiles.
  int cell_xcd = 0;
};

#ifndef GPRX_GEMM_PFC_DEFAULT
#define GPRX_GEMM_PFC_DEFAULT 1
#endif
#ifndef GPRX_GEMM_DMA_DEFAULT
#define GPRX_GEMM_DMA_DEFAULT 1
#endif
constexpr int GEMM_BK = 16;
constexpr int GEMM_LDK = GEMM_BK;  // KC image row stride (doubles): 128 B, unpadded; the eight 16-B chunks of a row are XOR-swizzled
// Swizzle of the KC image: chunk c of row `row` is stored at chunk c ^ kc_swz(row).  ds_read_b128 serves a wave in four
// fixed 16-lane groups ({0-3,12-15,20-27}, ... : MI355X_MICROARCH.md, LDS) and conflicts are counted per 16-B slot
// modulo 256 B; with the fragment map row = lane & 15, chunk = 2 (lane >> 4) + {0, 1} this table (found by search) makes
// every group hit 16 distinct slots.  The padded image [rows][18] it replaces cost 2 cycles per group
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.33) and 12 % more LDS.
__device__ __forceinline__ int kc_swz(int row) { return (int)((0xa7a09f5366f7ull >> (3 * (row & 15))) & 7ull); }

template <int BMN>
struct McStride {
  static constexpr int value = BMN + 4;  // (4 * stride) mod 32 == 16: the two k-rows of a half-wave hit disjoint banks
};

// ---- global -> registers -----------------------------------------------------------------
// KC source: rows = tile rows (m or n), 16 contiguous k per row = 8 chunks of 16 B.
template <int ROWS>
__device__ __forceinline__ void load_kc(d2 (&r)[ROWS / 32], const double* __restrict__ base, int64_t ld, int row0, int nrows_valid,
                                        int k0, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int q = tid + 256 * i;
    const int row = q >> 3, cc = q & 7;
    if (row0 + row < nrows_valid)
      r[i] = *reinterpret_cast<const d2*>(base + (int64_t)(row0 + row) * ld + k0 + cc * 2);
    else
      r[i] = d2{0.0, 0.0};
  }
}
template <int ROWS>
__device__ __forceinline__ void store_kc(double* s, const d2 (&r)[ROWS / 32], int tid) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int q = tid + 256 * i;
    const int row = q >> 3, cc = q & 7;
    *reinterpret_cast<d2*>(s + row * GEMM_LDK + ((cc ^ kc_swz(row)) * 2)) = r[i];
  }
}
// MC source: 16 k-rows, each with COLS contiguous m/n values = COLS/2 chunks of 16 B.
template <int COLS>
__device__ __forceinline__ void load_mc(d2 (&r)[COLS / 32], const double* __restrict__ base, int64_t ld, int col0, int ncols_valid,
                                        int k0, int tid) {
#pragma unroll
  for (int i = 0; i < COLS / 32; ++i) {
    const int q = tid + 256 * i;
    const int krow = q / (COLS / 2), cc = q % (COLS / 2);
    const int col = col0 + cc * 2;
    if (col + 1 < ncols_valid) {
      r[i] = *reinterpret_cast<const d2*>(base + (int64_t)(k0 + krow) * ld + col);
    } else if (col < ncols_valid) {
      r[i] = d2{base[(int64_t)(k0 + krow) * ld + col], 0.0};
    } else {
      r[i] = d2{0.0, 0.0};
    }
  }
}
template <int COLS>
__device__ __forceinline__ void store_mc(double* s, const d2 (&r)[COLS / 32], int tid) {
#pragma unroll
  for (int i = 0; i < COLS / 32; ++i) {
    const int q = tid + 256 * i;
    const int krow = q / (COLS / 2), cc = q % (COLS / 2);
    *reinterpret_cast<d2*>(s + krow * McStride<COLS>::value + cc * 2) = r[i];
  }
}

// ---- the kernel ----------------------------------------------------------------------------
// TA == 0: A stored M x K (KC image);  TA == 1: A stored K x M (MC image), op(A) = A^T.
// TB == 1: B stored N x K (KC image), op(B) = B^T;  TB == 0: B stored K x N (MC image).
// PF: fetch C before the main loop (beta != 0, short K: hides the C latency; costs 32 VGPRs on the 64 x 64 tile,
// i.e. one workgroup of occupancy, so long-K launches use PF = 0)
// AXF: the A operand is transformed element-wise on its way from memory to LDS (fused centring / weighting of the EOF
// projection, pca.h): no separate pass over A.
// DMA: both operands go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write pass; the XOR
// swizzle of the KC image moves to the per-lane SOURCE address, the LDS image is written lane-linear: 1 KiB = 8 rows of
// 128 B per wave-instruction).  Only for the NT form on full tiles (M % BM == 0, N % BN == 0): no bounds predicates exist.
__device__ __forceinline__ void glds16(const double* src, double* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int TA, int TB, int BM, int BN>
struct GemmSmem {
  static constexpr int A_ELEMS = TA ? GEMM_BK * McStride<BM>::value : BM * GEMM_LDK;
  static constexpr int B_ELEMS = TB ? BN * GEMM_LDK : GEMM_BK * McStride<BN>::value;
  static constexpr int doubles = 2 * (A_ELEMS + B_ELEMS);
};

// One output tile: workgroup (bx, by, bz) of the launch grid described in launch_gemm_t.  `smem` is the kernel's single
// shared array (GemmSmem::doubles).
template <int TA, int TB, int BM, int BN, int PF, int AXF, int DMA>
__device__ __forceinline__ void gemm_tile(GemmArgs p, const int bx_in, const int by_in, const int bz, double* __restrict__ smem) {
  int bx = bx_in, by = by_in;
  if (p.cell_xcd) {
    const int L = bx_in + p.nwg * by_in, slot = L >> 3;
    const int cq = slot / p.nwg;
    by = (L & 7) + 8 * cq;
    bx = slot - cq * p.nwg;
  }
  static_assert(!DMA || (AXF == 0 && (TA == 0 || BM == 64) && (TB == 1 || BN == 64)),
                "the LDS-DMA staging exists for k-contiguous operands and for m/n-contiguous operands of 64-wide tiles");
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int A_ELEMS = GemmSmem<TA, TB, BM, BN>::A_ELEMS;
  constexpr int B_ELEMS = GemmSmem<TA, TB, BM, BN>::B_ELEMS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, r = lane & 15;

  // Tile index.  XCD-contiguous remap first (workgroups are dealt round-robin over the 8 XCDs, so give
  // each XCD a contiguous run of tiles: neighbours share operand panels in that XCD's L2), then, for
  // C_LOWER, decode the linear index over the lower trapezoid only -- every launched workgroup has
  // work, and the tiles are spread evenly over the XCDs.
  {
    int entry = by;
    if (p.inner > 0) {
      const int cell = entry / p.inner;
      entry -= cell * p.inner;
      p.A += (int64_t)cell * p.cellA;
      p.B += (int64_t)cell * p.cellB;
      p.C += (int64_t)cell * p.cellC;
      if (p.alpha_tab) p.alpha = p.alpha_tab[(int64_t)cell * p.alpha_stride];
    }
    p.A += (int64_t)entry * p.strideA;
    p.B += (int64_t)entry * p.strideB;
    p.C += (int64_t)entry * p.strideC;
  }
  int bid = bx;
  if (!p.cell_xcd && !(p.flags & (GEMM_A_LOWER | GEMM_A_UPPER | GEMM_B_LOWER | GEMM_B_UPPER))) {
    // (with triangular operands the K range, i.e. the cost, varies along the tile order: keep the
    // hardware's round-robin there, which spreads long and short tiles over all XCDs)
    const int q = p.nwg >> 3, rr = p.nwg & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }
  int ti, tj;
  if (p.flags & GEMM_C_LOWER) {
    // Lower trapezoid in BANDS of 4 tile rows, column-major inside a band: the (up to) 4 tiles of a band column are
    // consecutive, so they are dispatched together to the same XCD (after the remap above), run in step through K and
    // share their B panel in that XCD's L2, while the band's 4 A panels (2 MB at K = 1024, half of the L2) serve every
    // column of the band.  FETCH_SIZE of the batched bulk update: row-major 2.02 GB per launch, bands of 2 / 3 / 4 / 6 /
    // 8 / 16 rows 1.77 / 1.64 / 1.56 / 1.58 / 1.72 / 2.20 GB (algorithmic: about 1.2 GB).
#ifndef GPRX_BAND
#define GPRX_BAND 4
#endif
    constexpr int R = GPRX_BAND;
    const int tn = p.tiles_n;
    const int tri = tn * (tn + 1) / 2;
    if (bid < tri) {
      // tiles before band b: S(b) = R^2 b (b - 1) / 2 + b R (R + 1) / 2
      int b = (int)((sqrtf((float)(R + 1) * (R + 1) / 4.0f - (float)(R + 1) * R / 2.0f + (float)R * R / 4.0f + 2.0f * (float)bid) -
                     (float)(R + 1) / 2.0f + (float)R / 2.0f) / (float)R);
      if (b < 0) b = 0;
      auto S = [](int bb) { return R * R * bb * (bb - 1) / 2 + bb * R * (R + 1) / 2; };
      while (S(b + 1) <= bid) ++b;
      while (S(b) > bid) --b;
      int u = bid - S(b);
      const int r0 = R * b;
      const int h = (tn - r0 < R) ? tn - r0 : R;
      if (u < h * r0) {
        tj = u / h;
        ti = r0 + u % h;
      } else {
        u -= h * r0;
        int c = 0;
        while (u >= h - c) {
          u -= h - c;
          ++c;
        }
        tj = r0 + c;
        ti = r0 + c + u;
      }
    } else {
      const int rest = bid - tri;
      const int b2 = rest / (R * tn);
      const int u = rest - b2 * R * tn;
      const int left = p.tiles_m - tn - R * b2;
      const int h = left < R ? left : R;
      tj = u / h;
      ti = tn + R * b2 + u % h;
    }
  } else {
    const int tri = p.flags & (GEMM_A_LOWER | GEMM_A_UPPER | GEMM_B_LOWER | GEMM_B_UPPER);
    if (tri == GEMM_B_UPPER || tri == GEMM_B_LOWER) {
      // one triangular operand: the K range of a tile grows (UPPER) or shrinks (LOWER) with tj.  Longest tiles
      // first (column-major over the tiles, longest column first): the launch ends with the short tiles instead
      // of a tail of full-K tiles, and the 8 consecutive workgroups that the dispatcher deals to the 8 XCDs all
      // have the same cost (row-major order with tiles_n a multiple of 8 gave XCD 7 54 % more work than XCD 0).
      const int cj = bid / p.tiles_m;
      ti = bid % p.tiles_m;
      tj = (tri == GEMM_B_UPPER) ? p.tiles_n - 1 - cj : cj;
    } else if (tri == GEMM_A_LOWER || tri == GEMM_A_UPPER) {
      const int ri = bid / p.tiles_n;
      tj = bid % p.tiles_n;
      ti = (tri == GEMM_A_LOWER) ? p.tiles_m - 1 - ri : ri;
    } else {
      ti = bid / p.tiles_n;
      tj = bid % p.tiles_n;
      if (tri & (GEMM_B_LOWER | GEMM_B_UPPER)) tj = (tj + ti) % p.tiles_n;  // keep the XCDs balanced (see above)
    }
  }
  const int m0 = ti * BM, n0 = tj * BN;

  int kbeg = 0, kend = p.K;
  if (p.flags & GEMM_A_LOWER) kend = min(kend, m0 + BM);
  if (p.flags & GEMM_A_UPPER) kbeg = max(kbeg, m0);
  if (p.flags & GEMM_B_LOWER) kbeg = max(kbeg, n0);
  if (p.flags & GEMM_B_UPPER) kend = min(kend, n0 + BN);
  if (p.ksplit > 0) {
    kbeg = max(kbeg, bz * p.ksplit);
    kend = min(kend, (bz + 1) * p.ksplit);
    p.C += (int64_t)bz * p.slab;
  }
  kbeg &= ~(GEMM_BK - 1);
  kend = (kend + GEMM_BK - 1) & ~(GEMM_BK - 1);
  if (kend > p.K) kend = p.K;

  d4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

  // beta != 0 on a small tile: fetch C before the main loop so that its latency hides under the MFMAs
  constexpr bool kPrefetchC = PF != 0;
  double cpre[kPrefetchC ? TM : 1][kPrefetchC ? TN : 1][4];
  if constexpr (kPrefetchC) {
    if (p.beta != 0.0) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int col = n0 + wn * (BN / 2) + b * 16 + r;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = m0 + wm * (BM / 2) + a * 16 + g + 4 * q;
            cpre[a][b][q] = (row < p.M && col < p.N) ? p.C[(int64_t)row * p.ldc + col] : 0.0;
          }
        }
    }
  }

  d2 ra[BM / 32], rb[BN / 32];
  auto gload = [&](int k0) {
    if constexpr (TA == 0) {
      load_kc<BM>(ra, p.A, p.lda, m0, p.M, k0, tid);
      if constexpr (AXF != 0) {
        // every chunk of this thread has the same two k (q & 7 is the same for q = tid + 256 i)
        const int kk = k0 + (tid & 7) * 2;
        const d2 sub = *reinterpret_cast<const d2*>(p.a_sub + kk), mul = *reinterpret_cast<const d2*>(p.a_mul + kk);
        d2 el = d2{0.0, 0.0};
        if (p.a_elev) el = *reinterpret_cast<const d2*>(p.a_elev + kk);
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
          d2 v = ra[i];
          if (p.a_elev) {
            v.x = fmax(v.x - el.x, 0.0);
            v.y = fmax(v.y - el.y, 0.0);
          }
          ra[i] = d2{(v.x - sub.x) * mul.x, (v.y - sub.y) * mul.y};
        }
      }
    } else {
      load_mc<BM>(ra, p.A, p.lda, m0, p.M, k0, tid);
    }
    if constexpr (TB == 1)
      load_kc<BN>(rb, p.B, p.ldb, n0, p.N, k0, tid);
    else
      load_mc<BN>(rb, p.B, p.ldb, n0, p.N, k0, tid);
  };
  auto sstore = [&](int buf) {
    double* sa = smem + buf * (A_ELEMS + B_ELEMS);
    double* sb = sa + A_ELEMS;
    if constexpr (TA == 0)
      store_kc<BM>(sa, ra, tid);
    else
      store_mc<BM>(sa, ra, tid);
    if constexpr (TB == 1)
      store_kc<BN>(sb, rb, tid);
    else
      store_mc<BN>(sb, rb, tid);
  };

  // one 16-deep stage: prefetch the next stage's operands into registers, MFMAs on LDS buffer `buf`, publish the
  // prefetch in the other buffer
  const int swz = kc_swz(r);  // rows and columns of this lane's fragments are all congruent to r modulo 16
  auto stage = [&](int k0, int buf) {
    const bool more = (k0 + GEMM_BK) < kend;
    if (more) gload(k0 + GEMM_BK);
    const double* sa = smem + buf * (A_ELEMS + B_ELEMS);
    const double* sb = sa + A_ELEMS;
    double fa[TM][4], fb[TN][4];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int row = wm * (BM / 2) + a * 16 + r;
      if constexpr (TA == 0) {
        const d2 lo = *reinterpret_cast<const d2*>(sa + row * GEMM_LDK + 2 * ((2 * g) ^ swz));
        const d2 hi = *reinterpret_cast<const d2*>(sa + row * GEMM_LDK + 2 * ((2 * g + 1) ^ swz));
        fa[a][0] = lo.x; fa[a][1] = lo.y; fa[a][2] = hi.x; fa[a][3] = hi.y;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) fa[a][j] = sa[(4 * g + j) * McStride<BM>::value + row];
      }
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = wn * (BN / 2) + b * 16 + r;
      if constexpr (TB == 1) {
        const d2 lo = *reinterpret_cast<const d2*>(sb + col * GEMM_LDK + 2 * ((2 * g) ^ swz));
        const d2 hi = *reinterpret_cast<const d2*>(sb + col * GEMM_LDK + 2 * ((2 * g + 1) ^ swz));
        fb[b][0] = lo.x; fb[b][1] = lo.y; fb[b][2] = hi.x; fb[b][3] = hi.y;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[b][j] = sb[(4 * g + j) * McStride<BN>::value + col];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
    if (more) sstore(buf ^ 1);
    __syncthreads();
  };
  if constexpr (DMA != 0) {
    // every wave fills rows 8 (4 i + wave) .. + 7 of each operand image per instruction i; lane l: row + (l >> 3), LDS chunk
    // l & 7, which holds the global chunk (l & 7) ^ kc_swz(row) -- the same involution the fragment reads apply
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma_fill = [&](int k0, int buf) {
      double* sa = smem + buf * (A_ELEMS + B_ELEMS);
      double* sb = sa + A_ELEMS;
      if constexpr (TA == 0) {
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
          const int rbase = (i * 4 + wave_u) * 8;
          const int row = rbase + (lane >> 3);
          glds16(p.A + (int64_t)(m0 + row) * p.lda + k0 + 2 * ((lane & 7) ^ kc_swz(row)), sa + rbase * GEMM_LDK);
        }
      } else {
        // TN: A is stored K x M -- the image and its source-side swizzle are those of the NN form's B operand (below)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int rbase = (i * 4 + wave_u) * 2;
          const int row = rbase + (lane >> 5);
          glds16(p.A + (int64_t)(k0 + row) * p.lda + m0 + 2 * ((lane & 31) ^ (8 * ((row >> 2) & 1))), sa + rbase * BM);
        }
      }
      if constexpr (TB == 1) {
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
          const int rbase = (i * 4 + wave_u) * 8;
          const int row = rbase + (lane >> 3);
          glds16(p.B + (int64_t)(n0 + row) * p.ldb + k0 + 2 * ((lane & 7) ^ kc_swz(row)), sb + rbase * GEMM_LDK);
        }
      } else {
        // NN: B is stored K x N, so a stage is 16 k-rows of 64 contiguous columns = 512 B each: one instruction lands two rows.
        // The image is [16][64] UNPADDED (the DMA writes lane-linear); conflict-free fragment reads come from the SOURCE side:
        // row k keeps its 16-byte chunks at position chunk ^ 8 ((k >> 2) & 1), so the lane groups g and g + 1 of a ds_read_b64
        // half-wave (k = 4 g + j: rows 4 apart) use complementary halves of the 64 banks.
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int rbase = (i * 4 + wave_u) * 2;
          const int row = rbase + (lane >> 5);
          glds16(p.B + (int64_t)(k0 + row) * p.ldb + n0 + 2 * ((lane & 31) ^ (8 * ((row >> 2) & 1))), sb + rbase * BN);
        }
      }
    };
    const int swz_d = kc_swz(r);
    auto dma_stage = [&](int k0, int buf) {
      if (k0 + GEMM_BK < kend) dma_fill(k0 + GEMM_BK, buf ^ 1);  // (all waves left buffer buf ^ 1 at the previous barrier)
      const double* sa = smem + buf * (A_ELEMS + B_ELEMS);
      const double* sb = sa + A_ELEMS;
      double fa[TM][4], fb[TN][4];
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const int row = wm * (BM / 2) + a * 16 + r;
        if constexpr (TA == 0) {
          const d2 lo = *reinterpret_cast<const d2*>(sa + row * GEMM_LDK + 2 * ((2 * g) ^ swz_d));
          const d2 hi = *reinterpret_cast<const d2*>(sa + row * GEMM_LDK + 2 * ((2 * g + 1) ^ swz_d));
          fa[a][0] = lo.x; fa[a][1] = lo.y; fa[a][2] = hi.x; fa[a][3] = hi.y;
        } else {
          const int pos = 2 * ((row >> 1) ^ (8 * (g & 1))) + (row & 1);
#pragma unroll
          for (int j = 0; j < 4; ++j) fa[a][j] = sa[(4 * g + j) * BM + pos];
        }
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = wn * (BN / 2) + b * 16 + r;
        if constexpr (TB == 1) {
          const d2 lo = *reinterpret_cast<const d2*>(sb + col * GEMM_LDK + 2 * ((2 * g) ^ swz_d));
          const d2 hi = *reinterpret_cast<const d2*>(sb + col * GEMM_LDK + 2 * ((2 * g + 1) ^ swz_d));
          fb[b][0] = lo.x; fb[b][1] = lo.y; fb[b][2] = hi.x; fb[b][3] = hi.y;
        } else {
          const int pos = 2 * ((col >> 1) ^ (8 * (g & 1))) + (col & 1);  // (row 4 g + j: (row >> 2) & 1 == g & 1)
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[b][j] = sb[(4 * g + j) * BN + pos];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
      // (the scheduler places this barrier -- and the wait for the next stage's DMA -- in the middle of the stage's MFMAs; pinning it behind
      // them with sched_barrier, which gained 2 % in the one-workgroup-per-cell kernel, measured nothing here: 2219 -> 2225 fits/s, noise)
      __syncthreads();  // (waits for this wave's DMA of the next stage -- vmcnt(0) -- then for every wave)
    };
    if (kbeg < kend) {
      dma_fill(kbeg, 0);
      __syncthreads();
      for (int k0 = kbeg; k0 < kend; k0 += 2 * GEMM_BK) {
        dma_stage(k0, 0);
        if (k0 + GEMM_BK < kend) dma_stage(k0 + GEMM_BK, 1);
      }
    }
  } else if (kbeg < kend) {
    gload(kbeg);
    sstore(0);
    __syncthreads();
    // two stages per loop iteration with constant buffer indices: half the taken branches (measured: a taken branch
    // every 4 MFMAs costs a pure MFMA loop 30 % of its rate, tools/mfma_peak*.hip) and immediate LDS offsets
    for (int k0 = kbeg; k0 < kend; k0 += 2 * GEMM_BK) {  // (four stages per iteration measured no better)
      stage(k0, 0);
      if (k0 + GEMM_BK < kend) stage(k0 + GEMM_BK, 1);
    }
  }

  // epilogue: lane holds rows g + 4q (q = 0..3) of column r of each 16 x 16 tile.  The C reads of a
  // tile row are all issued before the first store (a load -> store -> load chain is latency-bound).
  const double alpha = p.alpha, beta = p.beta;
  if (p.rowsq) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double ssq = 0.0;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int col = n0 + wn * (BN / 2) + b * 16 + r;
          const double v = col < p.N ? alpha * acc[a][b][q] : 0.0;
          ssq = __builtin_fma(v, v, ssq);
        }
        // the 16 lanes that share g hold the 16 columns r = 0..15 of this row
        ssq += __shfl_xor(ssq, 1, 64);
        ssq += __shfl_xor(ssq, 2, 64);
        ssq += __shfl_xor(ssq, 4, 64);
        ssq += __shfl_xor(ssq, 8, 64);
        const int row = m0 + wm * (BM / 2) + a * 16 + g + 4 * q;
        if (r == 0 && row < p.M) p.rowsq[(int64_t)(2 * tj + wn) * p.rowsq_ld + row] = ssq;
      }
    return;
  }
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    double cold[TN][4];
    if constexpr (kPrefetchC) {
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) cold[b][q] = cpre[a][b][q];
    } else if (beta != 0.0) {
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn * (BN / 2) + b * 16 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = m0 + wm * (BM / 2) + a * 16 + g + 4 * q;
          cold[b][q] = (row < p.M && col < p.N) ? p.C[(int64_t)row * p.ldc + col] : 0.0;
        }
      }
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + wn * (BN / 2) + b * 16 + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = m0 + wm * (BM / 2) + a * 16 + g + 4 * q;
        if (row < p.M && col < p.N) {
          double v = alpha * acc[a][b][q];
          if (beta != 0.0) v = __builtin_fma(beta, cold[b][q], v);
          p.C[(int64_t)row * p.ldc + col] = v;
        }
      }
    }
  }
}

template <int TA, int TB, int BM, int BN, int PF = 0, int AXF = 0, int DMA = 0>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128) ? 2 : (((PF && !DMA) || AXF) ? 3 : 4)) void gemm_f64_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) double smem[GemmSmem<TA, TB, BM, BN>::doubles];
  gemm_tile<TA, TB, BM, BN, PF, AXF, DMA>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, smem);
}

// The same tiles from a PERSISTENT grid: workgroup b takes tiles b, b + gridDim.x, ... of the (nwg x batch) tile list.  A
// launch of `slots` x 256 workgroups keeps at most `slots` workgroups of this kernel on a CU, so the wave slots, registers
// and LDS of the remaining slot stay free for the small dependent kernels of the panel chain that runs beside the bulk
// update on another stream (measured at N = 16384: behind an ordinary launch, whose workgroups fill every CU, those kernels
// waited 20-170 us for a slot each).  gridDim.x must be a multiple of 8 (the tile -> XCD affinity of the remap survives).
template <int BM, int BN, int PF>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128) ? 2 : 4) void gemm_f64_nt_dma_persistent_kernel(GemmArgs p, int batch) {
  __shared__ __attribute__((aligned(16))) double smem[GemmSmem<0, 1, BM, BN>::doubles];
  const int total = p.nwg * batch;
  for (int t = (int)blockIdx.x; t < total; t += (int)gridDim.x) {
    const int by = t / p.nwg;
    gemm_tile<0, 1, BM, BN, PF, 0, 1>(p, t - by * p.nwg, by, 0, smem);
    // (the last stage of a tile ends with a barrier behind its LDS reads; the epilogue does not touch LDS)
  }
}

template <int TA, int TB, int BM, int BN>
inline hipError_t launch_gemm_t(hipStream_t st, GemmArgs p, int batch, int nsplit = 1) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  if (p.tiles_m == 0 || p.tiles_n == 0) return hipSuccess;
  if (p.flags & GEMM_C_LOWER) {
    // square tiles: tile (ti, tj) touches the lower triangle iff tj <= ti
    if (p.tiles_n > p.tiles_m) p.tiles_n = p.tiles_m;
    p.nwg = p.tiles_n * (p.tiles_n + 1) / 2 + (p.tiles_m - p.tiles_n) * p.tiles_n;
  } else {
    p.nwg = p.tiles_m * p.tiles_n;
  }
  {
    // default on (GPRX_CELL_XCD=0 restores the per-cell interleave): measured on the batched step at N = 4096, 128 cells -- FETCH /
    // WRITE traffic of the main update kernel 7.57 -> 6.08 GB per launch, 2092 -> 2113 fits/s; same tiles, same arithmetic
    static const int cell_xcd = getenv("GPRX_CELL_XCD") ? atoi(getenv("GPRX_CELL_XCD")) : 1;
    p.cell_xcd = (cell_xcd && p.inner == 0 && nsplit == 1 && batch >= 8 && batch % 8 == 0 && p.persist_slots == 0 &&
                  !(p.flags & (GEMM_A_LOWER | GEMM_A_UPPER | GEMM_B_LOWER | GEMM_B_UPPER))) ? 1 : 0;
  }
  if constexpr (TA == 0 && TB == 1) {
    // NT form on full tiles (every update of the Cholesky): both operands by LDS-DMA.  With the operands off the register
    // file the 64 x 64 kernel has room to fetch its C tile BEFORE the main loop (32 registers: still 4 workgroups per CU),
    // so the read-modify-write of C costs no exposed latency (GPRX_GEMM_PFC: 1 = for short K only, 2 = at any K).
    static const int dma = getenv("GPRX_GEMM_DMA") ? atoi(getenv("GPRX_GEMM_DMA")) : GPRX_GEMM_DMA_DEFAULT;
    static const int pfc = getenv("GPRX_GEMM_PFC") ? atoi(getenv("GPRX_GEMM_PFC")) : GPRX_GEMM_PFC_DEFAULT;
    if (dma && p.M % BM == 0 && p.N % BN == 0 && p.K % GEMM_BK == 0 && p.lda % 2 == 0 && p.ldb % 2 == 0) {
      const bool prefetch_c = BM * BN <= 64 * 64 && p.beta != 0.0 && nsplit == 1 && (pfc >= 2 || (pfc == 1 && p.K <= 128));
      if (p.persist_slots > 0 && nsplit == 1 && !prefetch_c) {
        static const int n_cu = [] {
          int dev = 0, cus = 256;
          hipGetDevice(&dev);
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
          return cus > 0 ? cus : 256;
        }();
        int grid = p.persist_slots * n_cu;
        grid -= grid % 8;
        if (grid > 0 && (int64_t)p.nwg * batch > grid) {
          hipLaunchKernelGGL((gemm_f64_nt_dma_persistent_kernel<BM, BN, 0>), dim3(grid), dim3(256), 0, st, p, batch);
          return hipGetLastError();
        }
      }
      // Experiment knob (GPRX_GEMM_PAD_LDS = bytes of unused dynamic LDS for ONE matrix's updates, default 0): caps this kernel at
      // 3 (9216) or 2 (22000) workgroups per CU, so that a panel workgroup of the chain (200-224 VGPRs per lane; five resident
      // workgroups of this kernel leave 72 per SIMD) always finds room.  Measured without effect on the panels' 81 us beside the
      // bulk update at N = 16384 (29.5 / 29.6 / 31.3 ms for 5 / 3 / 2 workgroups per CU): not a residency effect.
      static const int pad_env = getenv("GPRX_GEMM_PAD_LDS") ? atoi(getenv("GPRX_GEMM_PAD_LDS")) : -1;
      const unsigned pad = (batch == 1 && BM * BN <= 64 * 64) ? (unsigned)(pad_env >= 0 ? pad_env : 0) : 0u;
      if (prefetch_c)
        hipLaunchKernelGGL((gemm_f64_kernel<0, 1, (BM > 64 ? 64 : BM), (BN > 64 ? 64 : BN), 1, 0, 1>), dim3(p.nwg, batch, nsplit), dim3(256), pad, st, p);
      else
        hipLaunchKernelGGL((gemm_f64_kernel<0, 1, BM, BN, 0, 0, 1>), dim3(p.nwg, batch, nsplit), dim3(256), pad, st, p);
      return hipGetLastError();
    }
  }
  if constexpr (TA == 0 && TB == 0 && BM == 64 && BN == 64) {
    // NN on full 64 x 64 tiles (the L^-1 build, triangular solves with matrix right-hand sides): both operands by LDS-DMA as well
    static const int dma_nn = getenv("GPRX_GEMM_DMA_NN") ? atoi(getenv("GPRX_GEMM_DMA_NN")) : 1;
    if (dma_nn && p.M % BM == 0 && p.N % BN == 0 && p.K % GEMM_BK == 0 && p.lda % 2 == 0 && p.ldb % 2 == 0 && nsplit == 1 &&
        (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.B) & 15) == 0 && p.strideA % 2 == 0 && p.strideB % 2 == 0 &&
        p.cellA % 2 == 0 && p.cellB % 2 == 0) {
      hipLaunchKernelGGL((gemm_f64_kernel<0, 0, 64, 64, 0, 0, 1>), dim3(p.nwg, batch, nsplit), dim3(256), 0, st, p);
      return hipGetLastError();
    }
  }
  if constexpr (TA == 1 && TB == 0 && BM == 64 && BN == 64) {
    // TN on full 64 x 64 tiles (K^-1 = X^T X of the gradient, X = L^-1): both operands are m / n-contiguous, both by LDS-DMA
    static const int dma_tn = getenv("GPRX_GEMM_DMA_TN") ? atoi(getenv("GPRX_GEMM_DMA_TN")) : 1;
    if (dma_tn && p.M % BM == 0 && p.N % BN == 0 && p.K % GEMM_BK == 0 && p.lda % 2 == 0 && p.ldb % 2 == 0 && nsplit == 1 &&
        (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.B) & 15) == 0 && p.strideA % 2 == 0 && p.strideB % 2 == 0 &&
        p.cellA % 2 == 0 && p.cellB % 2 == 0) {
      hipLaunchKernelGGL((gemm_f64_kernel<1, 0, 64, 64, 0, 0, 1>), dim3(p.nwg, batch, nsplit), dim3(256), 0, st, p);
      return hipGetLastError();
    }
  }
  if (BM * BN <= 64 * 64 && p.beta != 0.0 && p.K <= 128 && nsplit == 1) {
    hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, (BM > 64 ? 64 : BM), (BN > 64 ? 64 : BN), 1>), dim3(p.nwg, batch, nsplit), dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, BM, BN, 0>), dim3(p.nwg, batch, nsplit), dim3(256), 0, st, p);
  }
  return hipGetLastError();
}

// ---- single-stage K = 64 update: C(M x N) -= A(M x 64) * B(N x 64)^T on the lower trapezoid -----------
// The in-block "strip" updates of the Cholesky sit on its critical path and are pure latency: with K = 64
// the whole operand panel of a 64 x 64 tile is 2 x 32 KiB, so every global load (operands and the C tile)
// is issued before anything is waited for, followed by one barrier, 64 MFMAs per wave and the store.
constexpr int S64_LD = 64;  // LDS row stride (doubles), unpadded: 16-B chunk c of row `row` sits at chunk c ^ (row & 15) -- conflict-free
                            // for the four 16-lane groups of ds_read_b128 (the padded stride 66 cost two cycles per group)

__global__ __launch_bounds__(256) void syrk_k64_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C,
                                                       int64_t lda, int64_t ldc, int M, int N, int tiles_n, int64_t cs) {
  __shared__ __attribute__((aligned(16))) double sA[64 * S64_LD];
  __shared__ __attribute__((aligned(16))) double sB[64 * S64_LD];
  A += (int64_t)blockIdx.y * cs;  // batched: blockIdx.y = cell, cs = cell stride (0 for a single matrix)
  B += (int64_t)blockIdx.y * cs;
  C += (int64_t)blockIdx.y * cs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, r = lane & 15;
  // lower-trapezoid tile decode (tiles are square: tile (ti, tj) has work iff tj <= ti)
  int bid = blockIdx.x, ti, tj;
  const int tri = tiles_n * (tiles_n + 1) / 2;
  if (bid < tri) {
    ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= bid) ++ti;
    while (ti * (ti + 1) / 2 > bid) --ti;
    tj = bid - ti * (ti + 1) / 2;
  } else {
    const int rest = bid - tri;
    ti = tiles_n + rest / tiles_n;
    tj = rest % tiles_n;
  }
  const int m0 = ti * 64, n0 = tj * 64;
  // operands: 64 rows x 32 chunks of 16 B each = 2048 chunks -> 8 per thread per operand
  d2 ra[8], rb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = tid + 256 * i;
    const int row = q >> 5, cc = q & 31;
    ra[i] = (m0 + row < M) ? *reinterpret_cast<const d2*>(A + (int64_t)(m0 + row) * lda + 2 * cc) : d2{0.0, 0.0};
    rb[i] = (n0 + row < N) ? *reinterpret_cast<const d2*>(B + (int64_t)(n0 + row) * lda + 2 * cc) : d2{0.0, 0.0};
  }
  d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = n0 + wn * 32 + b * 16 + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = m0 + wm * 32 + a * 16 + g + 4 * q;
        acc[a][b][q] = (row < M && col < N) ? C[(int64_t)row * ldc + col] : 0.0;
      }
    }
  // LDS image: inside every block of 16 k the elements are stored 4 x 4 transposed (k_local -> (k_local % 4) * 4 +
  // k_local / 4), so that the four consecutive doubles lane group g reads for a step are k = 16 ks + {g, 4 + g, 8 + g,
  // 12 + g}: MFMA j of a step then sums the four CONSECUTIVE k = 16 ks + 4 j + {0..3}, and the 16 MFMAs of a tile walk k
  // in ascending groups of four -- the same order in which the 8-column sub-panel updates of the panel kernels reach an
  // element, which is what keeps 64- and 128-column panels bit-identical.
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = tid + 256 * i;
    const int row = q >> 5, cc = q & 31;
    const int c8 = cc & 7;
    const int pos0 = 16 * (cc >> 3) + 8 * (c8 & 1) + (c8 >> 1), pos1 = pos0 + 4;
    const int sw = row & 15;
    const int o0 = 2 * ((pos0 >> 1) ^ sw) + (pos0 & 1), o1 = 2 * ((pos1 >> 1) ^ sw) + (pos1 & 1);
    sA[row * S64_LD + o0] = ra[i].x;
    sA[row * S64_LD + o1] = ra[i].y;
    sB[row * S64_LD + o0] = rb[i].x;
    sB[row * S64_LD + o1] = rb[i].y;
  }
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {  // 16 k per step; MFMA j of the step takes k = 16 ks + 4 j + (lane group)
    double fa[2][4], fb[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const double* pa = sA + (wm * 32 + a * 16 + r) * S64_LD;
      const d2 lo = *reinterpret_cast<const d2*>(pa + 2 * ((8 * ks + 2 * g) ^ r)), hi = *reinterpret_cast<const d2*>(pa + 2 * ((8 * ks + 2 * g + 1) ^ r));
      fa[a][0] = -lo.x; fa[a][1] = -lo.y; fa[a][2] = -hi.x; fa[a][3] = -hi.y;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const double* pb = sB + (wn * 32 + b * 16 + r) * S64_LD;
      const d2 lo = *reinterpret_cast<const d2*>(pb + 2 * ((8 * ks + 2 * g) ^ r)), hi = *reinterpret_cast<const d2*>(pb + 2 * ((8 * ks + 2 * g + 1) ^ r));
      fb[b][0] = lo.x; fb[b][1] = lo.y; fb[b][2] = hi.x; fb[b][3] = hi.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = n0 + wn * 32 + b * 16 + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = m0 + wm * 32 + a * 16 + g + 4 * q;
        if (row < M && col < N) C[(int64_t)row * ldc + col] = acc[a][b][q];
      }
    }
}

// C(M x N) -= A(M x 64) A(first N rows)^T, lower trapezoid only (M >= N)
inline hipError_t launch_syrk_k64(hipStream_t st, int M, int N, const double* A, int64_t lda, double* C, int64_t ldc, int batch = 1,
                                  int64_t cs = 0) {
  const int tm = (M + 63) / 64;
  int tn = (N + 63) / 64;
  if (tm == 0 || tn == 0) return hipSuccess;
  if (tn > tm) tn = tm;
  const int nwg = tn * (tn + 1) / 2 + (tm - tn) * tn;
  hipLaunchKernelGGL(syrk_k64_kernel, dim3(nwg, batch), dim3(256), 0, st, A, A, C, lda, ldc, M, N, tn, cs);
  return hipGetLastError();
}
inline hipError_t launch_gemm(hipStream_t st, int ta, int tb, int M, int N, int K, double alpha, const double* A, int64_t lda, const double* B,
                              int64_t ldb, double beta, double* C, int64_t ldc, int flags, int tile, int batch, int64_t strideA, int64_t strideB,
                              int64_t strideC, int cells, int64_t cellA, int64_t cellB, int64_t cellC, const double* alpha_tab, int alpha_stride,
                              int persist_slots, double* rowsq, int64_t rowsq_ld);
// The K = 64 in-block update: the general NT kernel (LDS-DMA operands, C prefetched: 32 KiB of LDS), or with GPRX_K64_GEMM=0
// the single-stage kernel above.  syrk_k64_kernel holds both whole operand panels in 64 KiB of LDS; beside the bulk update
// of a large matrix (whose workgroups own all LDS of every CU) each of its launches waited for TWO of them to retire on
// one CU: 166 us per launch at N = 16384 (rocprofv3), 21 of the 32 ms of that factorisation.  Measured with the general
// kernel: N = 16384 31.8 -> 30.2 ms, N = 8192 6.73 -> 6.50 ms, 128 cells of N = 4096 +0.5 %, N <= 4096 single unchanged.
// (The 128-column panel option keeps syrk_k64's k order inside its sub-panel updates: it is no longer bit-identical to the
// 64-column default, only equal to rounding.)
inline hipError_t launch_update_k64(hipStream_t st, int M, int N, const double* A, int64_t lda, double* C, int64_t ldc, int batch, int64_t cs) {
  static const int via_gemm = getenv("GPRX_K64_GEMM") ? atoi(getenv("GPRX_K64_GEMM")) : 1;
  if (!via_gemm) return launch_syrk_k64(st, M, N, A, lda, C, ldc, batch, cs);
  return launch_gemm(st, 0, 1, M, N, 64, -1.0, A, lda, A, lda, 1.0, C, ldc, GEMM_C_LOWER, 64, batch, cs, cs, cs, 1, 0, 0, 0, nullptr, 0, 0, nullptr, 0);
}

// tile: 0 = choose, 128 or 64 (square workgroup tiles)
inline hipError_t launch_gemm(hipStream_t st, int ta, int tb, int M, int N, int K, double alpha, const double* A, int64_t lda,
                              const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int flags, int tile = 0, int batch = 1,
                              int64_t strideA = 0, int64_t strideB = 0, int64_t strideC = 0, int cells = 1, int64_t cellA = 0,
                              int64_t cellB = 0, int64_t cellC = 0, const double* alpha_tab = nullptr, int alpha_stride = 0, int persist_slots = 0,
                              double* rowsq = nullptr, int64_t rowsq_ld = 0) {
  GemmArgs p{A, B, C, lda, ldb, ldc, M, N, K, alpha, beta, flags, 0, 0, 0, strideA, strideB, strideC, 0, 0};
  p.persist_slots = persist_slots;
  p.rowsq = rowsq;
  p.rowsq_ld = rowsq_ld;
  if (M <= 0 || N <= 0 || batch <= 0 || cells <= 0) return hipSuccess;
  if (cells > 1 || alpha_tab) {  // two-level batch: `batch` entries per cell
    p.inner = batch;
    p.cellA = cellA;
    p.cellB = cellB;
    p.cellC = cellC;
    p.alpha_tab = alpha_tab;
    p.alpha_stride = alpha_stride;
    batch *= cells;
  }
  static const int force_tile = getenv("GPRX_FORCE_TILE") ? atoi(getenv("GPRX_FORCE_TILE")) : 0;  // experiments
  if (force_tile) tile = force_tile;
  if (tile == 0) {
    // 128 x 128 tiles (2 workgroups per CU) once they fill the chip more than twice over; otherwise
    // 64 x 64 tiles (4 per CU), which keep the tail short on the small updates of a factorisation
    const int64_t t128 = (int64_t)batch * ((M + 127) / 128) * ((N + 127) / 128) / ((flags & GEMM_C_LOWER) ? 2 : 1);
    tile = (t128 >= 1024) ? 128 : 64;
  }
#define GPRX_GEMM_CASE(TA_, TB_)                                                  \
  if (ta == TA_ && tb == TB_) {                                                   \
    if (tile == 128) return launch_gemm_t<TA_, TB_, 128, 128>(st, p, batch);      \
    return launch_gemm_t<TA_, TB_, 64, 64>(st, p, batch);                         \
  }
  GPRX_GEMM_CASE(0, 1)
  GPRX_GEMM_CASE(0, 0)
  GPRX_GEMM_CASE(1, 0)
#undef GPRX_GEMM_CASE
  return hipErrorInvalidValue;
}

// ---- split-K for skinny products (few output tiles, long K) -------------------------------------------
// out[i][j] = beta * out[i][j] + sum_z slab_z[i][j]   (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const double* __restrict__ ws, int nsplit, int M, int N, double beta,
                                                            double* __restrict__ C, int64_t ldc, int64_t ws_cell = 0, int64_t c_cell = 0) {
  ws += (int64_t)blockIdx.y * ws_cell;  // blockIdx.y = cell
  C += (int64_t)blockIdx.y * c_cell;
  const int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (e >= (int64_t)M * N) return;
  const int i = (int)(e / N), j = (int)(e % N);
  double s = 0.0;
  for (int z = 0; z < nsplit; ++z) s += ws[(int64_t)z * M * N + e];
  double* cp = C + (int64_t)i * ldc + j;
  *cp = (beta != 0.0) ? beta * (*cp) + s : s;
}

// C = alpha op(A) op(B) + beta C with K cut into slices of `kchunk` (multiple of 16) over blockIdx.z; ws must hold
// ceil(K / kchunk) * M * N doubles.  No triangular flags.
// cells > 1: the same product for `cells` operand sets cellA / cellB / cellC doubles apart, workspaces ws_cell apart
// (each >= ceil(K / kchunk) * M * N), per-cell alpha from alpha_tab[cell * alpha_stride] if given.
inline hipError_t launch_gemm_splitk(hipStream_t st, int ta, int tb, int M, int N, int K, double alpha, const double* A, int64_t lda,
                                     const double* B, int64_t ldb, double beta, double* C, int64_t ldc, double* ws, int kchunk, int cells = 1,
                                     int64_t cellA = 0, int64_t cellB = 0, int64_t cellC = 0, int64_t ws_cell = 0,
                                     const double* alpha_tab = nullptr, int alpha_stride = 0) {
  if (M <= 0 || N <= 0 || cells <= 0) return hipSuccess;
  const int nsplit = (K + kchunk - 1) / kchunk;
  GemmArgs p{A, B, ws, lda, ldb, (int64_t)N, M, N, K, alpha, 0.0, 0, 0, 0, 0, 0, 0, 0, kchunk, (int64_t)M * N};
  if (cells > 1 || alpha_tab) {
    p.inner = 1;
    p.cellA = cellA;
    p.cellB = cellB;
    p.cellC = ws_cell;
    p.alpha_tab = alpha_tab;
    p.alpha_stride = alpha_stride;
  }
  hipError_t e = hipErrorInvalidValue;
  if (ta == 0 && tb == 1) e = launch_gemm_t<0, 1, 64, 64>(st, p, cells, nsplit);
  if (ta == 0 && tb == 0) e = launch_gemm_t<0, 0, 64, 64>(st, p, cells, nsplit);
  if (ta == 1 && tb == 0) e = launch_gemm_t<1, 0, 64, 64>(st, p, cells, nsplit);
  if (e != hipSuccess) return e;
  const int64_t total = (int64_t)M * N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256), cells), dim3(256), 0, st, (const double*)ws, nsplit, M, N, beta, C,
                     ldc, ws_cell, cellC);
  return hipGetLastError();
}

// split-K NT product with the fused A transform (EOF projection): out = ((g(A) - a_sub) * a_mul) B^T
inline hipError_t launch_gemm_splitk_axf(hipStream_t st, int M, int N, int K, const double* A, int64_t lda, const double* B, int64_t ldb,
                                         double* C, int64_t ldc, double* ws, int kchunk, const double* a_sub, const double* a_mul,
                                         const double* a_elev) {
  if (M <= 0 || N <= 0) return hipSuccess;
  const int nsplit = (K + kchunk - 1) / kchunk;
  GemmArgs p{A, B, ws, lda, ldb, (int64_t)N, M, N, K, 1.0, 0.0, 0, 0, 0, 0, 0, 0, 0, kchunk, (int64_t)M * N};
  p.a_sub = a_sub;
  p.a_mul = a_mul;
  p.a_elev = a_elev;
  p.tiles_m = (M + 63) / 64;
  if (N <= 32) {
    // few output columns (EOF modes): 64 x 32 tiles halve the MFMA work spent on padding columns -- with N = 10 the
    // 64-wide tile made the projection MFMA-bound instead of HBM-bound
    p.tiles_n = 1;
    p.nwg = p.tiles_m;
    hipLaunchKernelGGL((gemm_f64_kernel<0, 1, 64, 32, 0, 1>), dim3(p.nwg, 1, nsplit), dim3(256), 0, st, p);
  } else {
    p.tiles_n = (N + 63) / 64;
    p.nwg = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((gemm_f64_kernel<0, 1, 64, 64, 0, 1>), dim3(p.nwg, 1, nsplit), dim3(256), 0, st, p);
  }
  const int64_t total = (int64_t)M * N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const double*)ws, nsplit, M, N, 0.0, C, ldc);
  return hipGetLastError();
}

}  // namespace gprx
