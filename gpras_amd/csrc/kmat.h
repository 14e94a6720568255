// Stationary kernel-matrix build (RBF, Matern-1/2, -3/2, -5/2, gpflow "Exponential").
//
//   out[i, j] = variance * g(r2(a_i, b_j)) + (i == j ? diag_add : 0),
//   r2 = sum_k ((a_ik - b_jk) / l_k)^2      (difference form, the default: no cancellation, r2(a, a) == 0)
//   r2 = |a_i/l|^2 + |b_j/l|^2 - 2 (a_i/l).(b_j/l)   (form == 1: gpflow's literal arithmetic, utilities/ops.py
//        square_distance: squares rounded then summed in k order, the dot product as an fma chain in k order)
//
// Replaces gpflow's kernel evaluation inside SGPR (Kuf, Kuu, Kus; k(X, X) + s I for the exact
// specialisation).  HBM-write bound: each 64 x 64 output tile costs 32 KiB of stores against
// 2 x 64 x d inputs, so the layout goal is full-line stores -- a wave stores two rows of
// 32 x 16 B = 512 contiguous bytes per instruction -- and no reload of the inputs: the
// (scaled) coordinates of the 64 + 64 points of a tile are staged once through LDS.
#pragma once
#include "gprx_common.h"
#include "kfun.h"

namespace gprx {

constexpr int KM_T = 64;   // tile edge
constexpr int KM_DC = 8;   // coordinates staged per LDS pass
// row stride of the transposed staging image sBt[k][point]: 66 doubles = 528 bytes (33 x 16: the 16-byte reads stay aligned).  With 64
// the eight lanes of a staging store that differ in k sat 512 bytes apart -- one bank, an 8-way conflict (SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE = 0.24); with 66 a store instruction's 16-lane groups cover 16 distinct bank pairs.
#ifdef GPRX_KMAT_NOPAD
constexpr int KM_BT_LD = KM_T;
#else
constexpr int KM_BT_LD = KM_T + 2;
#endif

struct KmatArgs {
  const double* a;       // (n1, d)
  const double* b;       // (n2, d)
  const double* ls;      // d lengthscales, device (coordinates are divided by them, as gpflow's scale())
  double* out;
  int64_t ld;
  int n1, n2, d;
  int n1p, n2p;          // padded extents actually written
  double variance, diag_add;
  int mode;              // 0: rectangle (zero padding), 1: symmetric, tiles on/below the diagonal only, 2: symmetric, all tiles
  double pad_diag;       // value written on the diagonal of the padding (1 for symmetric modes, else 0)
  const double* dparams; // optional device-resident {variance, diag_add} overriding the two fields above (graph replay)
  int tiles_n;
  // batched build (blockIdx.y = cell): cell c writes out + c * out_stride with the parameters of row c of the
  // cell-parameter table, CELL_PAR doubles per row: [0] variance, [1] diag_add, [2] unit, [8 .. 8 + d) lengthscales
  const double* cell_par = nullptr;
  int64_t out_stride = 0;
  int64_t a_stride = 0, b_stride = 0;  // per-cell point sets (inducing inputs); 0: shared
  int diag_const = 0;                  // 1: keep diag_add as given (jitter) instead of the table's [1]
  int form = 0;                        // 0: difference form, 1: gpflow's expanded form (see the header comment)
  int tri = 0;                         // mode 1 launched on the T (T + 1) / 2 lower tiles only (launch_kmat sets it): blockIdx.x counts them row by row
};

// Stage KM_DC scaled coordinates of 64 points: sA[point][KM_DC] (row broadcast reads),
// sBt[KM_DC][64] (lane-contiguous reads).
// FORM (compile time: the default difference form keeps the registers and code of the kernel it always was -- as a run-time
// branch the expanded form's extra accumulators cost the default path 50 %: 2.8 -> 4.2 ms for 128 cells of N = 4096)
template <int KID, int FORM = 0>
__device__ __forceinline__ void kmat_body(KmatArgs p, int bx, double (*sA)[KM_DC], double (*sBt)[KM_BT_LD], double* __restrict__ sTab) {
  int ti, tj;
  if (p.tri) {
    // bx = ti (ti + 1) / 2 + tj, tj <= ti: only the tiles the Cholesky reads are launched (the full T x T grid returned at once in 49 %
    // of its workgroups).  ti from the float square root, corrected by one step either way.
    ti = (int)((__builtin_sqrtf(8.0f * (float)bx + 1.0f) - 1.0f) * 0.5f);
    if ((ti + 1) * (ti + 2) / 2 <= bx) ++ti;
    if (ti * (ti + 1) / 2 > bx) --ti;
    tj = bx - ti * (ti + 1) / 2;
  } else {
    ti = bx / p.tiles_n;
    tj = bx % p.tiles_n;
    if (p.mode == 1 && tj > ti) return;
  }
  exp_tab_fill(sTab);  // (visible after the first staging barrier below: every path has d >= 1)
  const int i0 = ti * KM_T, j0 = tj * KM_T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cp = lane & 31, rsub = lane >> 5;
  double diag_keep = p.diag_add;
  if (p.cell_par) {
    const double* par = p.cell_par + (int64_t)blockIdx.y * CELL_PAR;
    p.ls = par + CELL_PAR_LS;
    p.dparams = par;
    p.out += (int64_t)blockIdx.y * p.out_stride;
    p.a += (int64_t)blockIdx.y * p.a_stride;
    p.b += (int64_t)blockIdx.y * p.b_stride;
  }

  double acc[8][2];
#pragma unroll
  for (int it = 0; it < 8; ++it) acc[it][0] = acc[it][1] = 0.0;
  double na[FORM ? 8 : 1], nb[2] = {0.0, 0.0};  // expanded form only: squared norms of this thread's 8 row points / 2 column points
#pragma unroll
  for (int it = 0; it < (FORM ? 8 : 1); ++it) na[it] = 0.0;

  for (int k0 = 0; k0 < p.d; k0 += KM_DC) {
    // 64 points x 8 coords for each side = 512 + 512 values, 256 threads -> 2 + 2 each
    {
      // unconditional loads from clamped indices, selected afterwards (predicated loads each cost a full memory round trip)
      double ra[2], rb[2];
      const int kc = min(k0 + (tid & 7), p.d - 1);
      const double s = p.ls[kc];
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const int pt = (tid + 256 * rep) >> 3;
        ra[rep] = p.a[(int64_t)min(i0 + pt, p.n1 - 1) * p.d + kc];
        rb[rep] = p.b[(int64_t)min(j0 + pt, p.n2 - 1) * p.d + kc];
      }
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const int q = tid + 256 * rep;
        const int pt = q >> 3, kk = q & 7;
        const bool live = k0 + kk < p.d;
        double va, vb;
        if constexpr (FORM == 0) {
          // one division per thread and pass (its two points share the coordinate k) instead of four: the staging divisions were
          // ~9 of the ~61 fp64 instructions per output element.  (FORM 1 keeps x / l: it restates gpflow's arithmetic to the bit.)
          const double inv = 1.0 / s;
          va = ra[rep] * inv;
          vb = rb[rep] * inv;
        } else {
          va = ra[rep] / s;
          vb = rb[rep] / s;
        }
        sA[pt][kk] = (live && i0 + pt < p.n1) ? va : 0.0;
        sBt[kk][pt] = (live && j0 + pt < p.n2) ? vb : 0.0;
      }
    }
    __syncthreads();
    double b0[KM_DC], b1[KM_DC];
#pragma unroll
    for (int kk = 0; kk < KM_DC; ++kk) {
      const d2 v = *reinterpret_cast<const d2*>(&sBt[kk][2 * cp]);
      b0[kk] = v.x;
      b1[kk] = v.y;
    }
    if constexpr (FORM == 0) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = wave * 16 + 2 * it + rsub;
#pragma unroll
        for (int kk = 0; kk < KM_DC; ++kk) {
          const double a = sA[row][kk];
          const double d0 = a - b0[kk], d1 = a - b1[kk];
          acc[it][0] = __builtin_fma(d0, d0, acc[it][0]);
          acc[it][1] = __builtin_fma(d1, d1, acc[it][1]);
        }
      }
    } else {
      sqnorm_accumulate(nb[0], b0);
      sqnorm_accumulate(nb[1], b1);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = wave * 16 + 2 * it + rsub;
        double av[KM_DC];
#pragma unroll
        for (int kk = 0; kk < KM_DC; ++kk) av[kk] = sA[row][kk];
        sqnorm_accumulate(na[it], av);
#pragma unroll
        for (int kk = 0; kk < KM_DC; ++kk) {  // (padding coordinates are zero: they add exact zeros)
          acc[it][0] = __builtin_fma(av[kk], b0[kk], acc[it][0]);
          acc[it][1] = __builtin_fma(av[kk], b1[kk], acc[it][1]);
        }
      }
    }
    __syncthreads();
  }
  if constexpr (FORM != 0) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      acc[it][0] = expanded_r2(na[it], nb[0], acc[it][0]);
      acc[it][1] = expanded_r2(na[it], nb[1], acc[it][1]);
    }
  }

  if (p.dparams) {
    p.variance = p.dparams[0];
    p.diag_add = p.diag_const ? diag_keep : p.dparams[1];
  }
  // interior tiles (every row and column a real point, no diagonal element): no bounds or diagonal selects -- they were ~10 of the
  // ~61 fp64-rate instructions per element, and all but 2 T - 1 of the T (T + 1) / 2 lower tiles of a T x T matrix are interior
  if (i0 + KM_T <= p.n1 && j0 + KM_T <= p.n2 && ti != tj) {  // (i == j only happens inside tiles with ti == tj)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int i = i0 + wave * 16 + 2 * it + rsub;
      d2 v;
#ifdef GPRX_KMAT_NOEXP  // (timing experiments only: where does the time of this kernel go)
      v.x = p.variance * acc[it][0];
      v.y = p.variance * acc[it][1];
#else
      v.x = p.variance * corr_g<KID>(acc[it][0], sTab);
      v.y = p.variance * corr_g<KID>(acc[it][1], sTab);
#endif
#ifdef GPRX_KMAT_NOSTORE
      if (v.x == 123.456) *reinterpret_cast<d2*>(p.out + (int64_t)i * p.ld + j0 + 2 * cp) = v;
#elif defined(GPRX_KMAT_NT)
      __builtin_nontemporal_store(v, reinterpret_cast<d2*>(p.out + (int64_t)i * p.ld + j0 + 2 * cp));
#else
      *reinterpret_cast<d2*>(p.out + (int64_t)i * p.ld + j0 + 2 * cp) = v;
#endif
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int i = i0 + wave * 16 + 2 * it + rsub;
    const int j = j0 + 2 * cp;
    if (i >= p.n1p || j >= p.n2p) continue;
    d2 v;
    const bool vi = i < p.n1;
    v.x = (vi && j < p.n2) ? p.variance * corr_g<KID>(acc[it][0], sTab) : 0.0;
    v.y = (vi && j + 1 < p.n2) ? p.variance * corr_g<KID>(acc[it][1], sTab) : 0.0;
    if (i == j) v.x += (vi && j < p.n2) ? p.diag_add : p.pad_diag;
    if (i == j + 1) v.y += (vi && j + 1 < p.n2) ? p.diag_add : p.pad_diag;
    *reinterpret_cast<d2*>(p.out + (int64_t)i * p.ld + j) = v;
  }
}

template <int KID, int FORM = 0>
__global__ __launch_bounds__(256) void kmat_kernel(KmatArgs p) {
  __shared__ __attribute__((aligned(16))) double sA[KM_T][KM_DC];
  __shared__ __attribute__((aligned(16))) double sBt[KM_DC][KM_BT_LD];
  __shared__ __attribute__((aligned(16))) double sTab[64];
  kmat_body<KID, FORM>(p, (int)blockIdx.x, sA, sBt, sTab);
}
// Two builds in one launch (the sparse model's Kuf and Kuu): workgroups [0, first) take p, the rest q.  A dependent launch
// costs ~10 us on this part whatever it computes.
template <int KID, int FORM = 0>
__global__ __launch_bounds__(256) void kmat_pair_kernel(KmatArgs p, KmatArgs q, int first) {
  __shared__ __attribute__((aligned(16))) double sA[KM_T][KM_DC];
  __shared__ __attribute__((aligned(16))) double sBt[KM_DC][KM_BT_LD];
  __shared__ __attribute__((aligned(16))) double sTab[64];
  if ((int)blockIdx.x < first)
    kmat_body<KID, FORM>(p, (int)blockIdx.x, sA, sBt, sTab);
  else
    kmat_body<KID, FORM>(q, (int)blockIdx.x - first, sA, sBt, sTab);
}

inline hipError_t launch_kmat(hipStream_t st, int kid, KmatArgs p, int batch = 1) {
  const int tiles_m = (p.n1p + KM_T - 1) / KM_T;
  p.tiles_n = (p.n2p + KM_T - 1) / KM_T;
  if (tiles_m == 0 || p.tiles_n == 0) return hipSuccess;
  p.tri = 0;
#ifndef GPRX_KMAT_FULLGRID
  if (p.mode == 1 && tiles_m == p.tiles_n && tiles_m <= 4096) p.tri = 1;  // (the float decode of the tile index is exact far beyond 4096 (4096 + 1) / 2)
#endif
  dim3 grid(p.tri ? tiles_m * (tiles_m + 1) / 2 : tiles_m * p.tiles_n, batch), block(256);
#define GPRX_KMAT_CASE(K_)                                                          \
  case K_:                                                                          \
    if (p.form)                                                                     \
      hipLaunchKernelGGL((kmat_kernel<K_, 1>), grid, block, 0, st, p);              \
    else                                                                            \
      hipLaunchKernelGGL((kmat_kernel<K_, 0>), grid, block, 0, st, p);              \
    break;
  switch (kid) {
    GPRX_KMAT_CASE(0)
    GPRX_KMAT_CASE(1)
    GPRX_KMAT_CASE(2)
    GPRX_KMAT_CASE(3)
    GPRX_KMAT_CASE(4)
    default: return hipErrorInvalidValue;
  }
#undef GPRX_KMAT_CASE
  return hipGetLastError();
}

inline hipError_t launch_kmat_pair(hipStream_t st, int kid, KmatArgs p, KmatArgs q, int batch = 1) {
  p.tiles_n = (p.n2p + KM_T - 1) / KM_T;
  q.tiles_n = (q.n2p + KM_T - 1) / KM_T;
  const int np_ = ((p.n1p + KM_T - 1) / KM_T) * p.tiles_n, nq = ((q.n1p + KM_T - 1) / KM_T) * q.tiles_n;
  if (np_ == 0 || nq == 0 || p.form != q.form) return hipErrorInvalidValue;
  dim3 grid(np_ + nq, batch), block(256);
#define GPRX_KMAT_CASE(K_)                                                                       \
  case K_:                                                                                       \
    if (p.form)                                                                                  \
      hipLaunchKernelGGL((kmat_pair_kernel<K_, 1>), grid, block, 0, st, p, q, np_);              \
    else                                                                                         \
      hipLaunchKernelGGL((kmat_pair_kernel<K_, 0>), grid, block, 0, st, p, q, np_);              \
    break;
  switch (kid) {
    GPRX_KMAT_CASE(0)
    GPRX_KMAT_CASE(1)
    GPRX_KMAT_CASE(2)
    GPRX_KMAT_CASE(3)
    GPRX_KMAT_CASE(4)
    default: return hipErrorInvalidValue;
  }
#undef GPRX_KMAT_CASE
  return hipGetLastError();
}

}  // namespace gprx
