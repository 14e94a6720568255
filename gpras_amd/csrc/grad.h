// Gradient reductions: traces of W against dK/dtheta with the kernel derivative recomputed from
// the inputs on the fly (nothing of size N x N x n_theta is ever stored).
//
// Exact GP:  dLML/dtheta = 1/2 sum_ij W_ij dK_ij/dtheta,  W = alpha alpha^T - K^-1  (lower tiles
// of K^-1 are read once: 8 N^2 / 2 bytes for all 2 + n_len traces together).
// Sparse GP: the same contraction with W = dELBO/dKuf (M x N) or dELBO/dKuu (M x M) and, in
// addition, the row-wise contraction that gives dELBO/dZ.
//
// For a stationary kernel K = v g(r2), r2 = sum_k ds_k^2, ds_k = (a_k - b_k) / l_k:
//   dK/dv = g,   dK/dl_k = -v h ds_k^2 / l_k,   dK/da_k = v h ds_k / l_k,   h = 2 dg/dr2.
#pragma once
#include "gprx_common.h"
#include "kmat.h"

namespace gprx {

struct TraceArgs {
  const double* a;       // (n1, d) row points (exact: X; sparse: Z)
  const double* b;       // (n2, d) column points
  const double* ls;      // d lengthscales
  const double* W;       // weights, row-major, ldw
  int64_t ldw;
  const double* u;       // optional rank-1 term: w_ij = w_scale * W[i][j] + uv_scale * u[i] * v[j]
  const double* v;       //   (exact GP: u = v = alpha, w_scale = -1, uv_scale = 1)
  double w_scale, uv_scale;
  int n1, n2, d;
  double variance;
  int sym;               // 1: a == b, only tiles on/below the diagonal are visited, off-diagonal weights doubled
  double* partial;       // [grid][2 + d]: S_g, S_trace, S_len[k]
  double* wh_out;        // optional (n1, n2) store of w * variance * h (feeds the dZ GEMM), leading dimension ldwh
  int64_t ldwh;
  int tiles_n;
  // batched (blockIdx.y = cell): lengthscales and variance from row `cell` of the cell-parameter table (kmat.h), W / u, v /
  // partial advanced by their per-cell strides
  const double* cell_par = nullptr;
  int64_t w_stride = 0, uv_stride = 0, partial_stride = 0;
  int64_t a_stride = 0, b_stride = 0, v_stride = -1, wh_stride = 0;  // v_stride < 0: v advances like u
  int scale_inv_noise = 0;  // 1: w_scale = uv_scale = 1 / table[1] (the sparse model's 1 / s)
  int form = 0;             // distance form of r2 inside g and h (kmat.h); the factors ds_k of the derivatives stay differences
  int iso = 0;              // 1: ONE lengthscale for every dimension (the reference's default kernels; any d, with or without wh_out): only the sum over k of the
                            // lengthscale traces is wanted -- partial[wg][2] = -sum w v h r2 / l, partial[wg][3 ..] = 0
};

// One workgroup per 64 x 64 tile.  Thread mapping as kmat_kernel: 8 rows x 2 columns per thread.
// Output per workgroup (deterministic two-stage reduction, no atomics):
//   partial[wg][0] = sum w g          partial[wg][1] = sum_{i == j} w      partial[wg][2 + k] = -sum w v h ds_k^2 / l_k
// ISO (difference form only): sum_k ds_k^2 IS the r2 of pass 1, so the per-dimension pass 2 (a third of the kernel's fp64
// instructions) reduces to one FMA per element.
template <int KID, int FORM = 0, int ISO = 0>
__device__ __forceinline__ void trace_body(TraceArgs p, int bx, double (*sA)[KM_DC], double (*sBt)[KM_T], double (*sRed)[KM_DC + 2]) {
  if (p.cell_par) {
    const double* par = p.cell_par + (int64_t)blockIdx.y * CELL_PAR;
    p.ls = par + CELL_PAR_LS;
    p.variance = par[0];
    p.W += (int64_t)blockIdx.y * p.w_stride;
    if (p.u) {
      p.u += (int64_t)blockIdx.y * p.uv_stride;
      p.v += (int64_t)blockIdx.y * (p.v_stride < 0 ? p.uv_stride : p.v_stride);
    }
    p.partial += (int64_t)blockIdx.y * p.partial_stride;
    p.a += (int64_t)blockIdx.y * p.a_stride;
    p.b += (int64_t)blockIdx.y * p.b_stride;
    if (p.wh_out) p.wh_out += (int64_t)blockIdx.y * p.wh_stride;
    if (p.scale_inv_noise) p.w_scale = p.uv_scale = par[3];
  }
  const int ti = bx / p.tiles_n, tj = bx % p.tiles_n;
  double* out = p.partial + (int64_t)bx * (2 + p.d);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (p.sym && tj > ti) {
    for (int e = tid; e < 2 + p.d; e += 256) out[e] = 0.0;
    return;
  }
  const int i0 = ti * KM_T, j0 = tj * KM_T;
  const int cp = lane & 31, rsub = lane >> 5;

  auto stage = [&](int k0) {
    // (unconditional loads from clamped indices, selected afterwards: predicated loads each cost a full memory round trip)
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
      if constexpr (FORM == 0) {  // the scaled coordinates of kmat_kernel's difference form: x * (1 / l)
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
  };

  // The weights of this thread's 8 x 2 elements, requested BEFORE the distance pass so that their latency hides under it.
  // Every load is unconditional (indices clamped into the matrix, masks applied afterwards): with the bounds tests around the
  // loads hipcc emitted a branch and a full wait per load -- 16 serialised memory round trips per workgroup, 12 ms per 128-cell
  // launch for a kernel whose HBM floor is under 2 ms.  Interior tiles read two neighbouring weights with one 16-byte load.
  double wraw[8][2], urow[8], vcol[2] = {0.0, 0.0};
  {
    const bool wide = i0 + KM_T <= p.n1 && j0 + KM_T <= p.n2 && (p.ldw & 1) == 0 && ((reinterpret_cast<uintptr_t>(p.W) & 15) == 0);
    const int jc0 = min(j0 + 2 * cp, p.n2 - 1), jc1 = min(j0 + 2 * cp + 1, p.n2 - 1);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int ic = min(i0 + wave * 16 + 2 * it + rsub, p.n1 - 1);
      if (wide) {
        const d2 t = *reinterpret_cast<const d2*>(p.W + (int64_t)ic * p.ldw + j0 + 2 * cp);
        wraw[it][0] = t.x;
        wraw[it][1] = t.y;
      } else {
        wraw[it][0] = p.W[(int64_t)ic * p.ldw + jc0];
        wraw[it][1] = p.W[(int64_t)ic * p.ldw + jc1];
      }
      urow[it] = p.u ? p.u[ic] : 0.0;
    }
    if (p.u) {
      vcol[0] = p.v[jc0];
      vcol[1] = p.v[jc1];
    }
  }

  // pass 1: r2
  double r2[8][2];
#pragma unroll
  for (int it = 0; it < 8; ++it) r2[it][0] = r2[it][1] = 0.0;
  double na[FORM ? 8 : 1], nb[2] = {0.0, 0.0};  // expanded form only (kmat.h)
#pragma unroll
  for (int it = 0; it < (FORM ? 8 : 1); ++it) na[it] = 0.0;
  for (int k0 = 0; k0 < p.d; k0 += KM_DC) {
    stage(k0);
    __syncthreads();
    if constexpr (FORM == 0) {
#pragma unroll
      for (int kk = 0; kk < KM_DC; ++kk) {
        const d2 bv = *reinterpret_cast<const d2*>(&sBt[kk][2 * cp]);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const double av = sA[wave * 16 + 2 * it + rsub][kk];
          const double d0 = av - bv.x, d1 = av - bv.y;
          r2[it][0] = __builtin_fma(d0, d0, r2[it][0]);
          r2[it][1] = __builtin_fma(d1, d1, r2[it][1]);
        }
      }
    } else {
      double b0[KM_DC], b1[KM_DC];
#pragma unroll
      for (int kk = 0; kk < KM_DC; ++kk) {
        const d2 bv = *reinterpret_cast<const d2*>(&sBt[kk][2 * cp]);
        b0[kk] = bv.x;
        b1[kk] = bv.y;
      }
      sqnorm_accumulate(nb[0], b0);
      sqnorm_accumulate(nb[1], b1);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        double av[KM_DC];
#pragma unroll
        for (int kk = 0; kk < KM_DC; ++kk) av[kk] = sA[wave * 16 + 2 * it + rsub][kk];
        sqnorm_accumulate(na[it], av);
#pragma unroll
        for (int kk = 0; kk < KM_DC; ++kk) {
          r2[it][0] = __builtin_fma(av[kk], b0[kk], r2[it][0]);
          r2[it][1] = __builtin_fma(av[kk], b1[kk], r2[it][1]);
        }
      }
    }
    __syncthreads();
  }
  if constexpr (FORM != 0) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      r2[it][0] = expanded_r2(na[it], nb[0], r2[it][0]);
      r2[it][1] = expanded_r2(na[it], nb[1], r2[it][1]);
    }
  }

  // weights and correlation terms
  double wh[8][2];  // w * v * h
  double sg = 0.0, str = 0.0;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int i = i0 + wave * 16 + 2 * it + rsub;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = j0 + 2 * cp + c;
      double w = p.w_scale * wraw[it][c];
      if (p.u) w = __builtin_fma(p.uv_scale * urow[it], vcol[c], w);
      if (p.sym) w = (j > i) ? 0.0 : ((j < i) ? w * 2.0 : w);
      w = (i < p.n1 && j < p.n2) ? w : 0.0;
      double g, h;
      corr_gh<KID>(r2[it][c], g, h);
      sg = __builtin_fma(w, g, sg);
      if (i == j) str += w;
      wh[it][c] = w * p.variance * h;
    }
    if (p.wh_out && i < p.n1) {
      const int j = j0 + 2 * cp;
      if (j + 1 < p.n2) {
        *reinterpret_cast<d2*>(p.wh_out + (int64_t)i * p.ldwh + j) = d2{wh[it][0], wh[it][1]};
      } else if (j < p.n2) {
        p.wh_out[(int64_t)i * p.ldwh + j] = wh[it][0];
      }
    }
  }

  auto block_sum_store = [&](double v, int slot) {
    v = wave_sum(v);
    if (lane == 0) sRed[wave][slot] = v;
  };
  block_sum_store(sg, 0);
  block_sum_store(str, 1);
  __syncthreads();
  if (tid < 2) out[tid] = sRed[0][tid] + sRed[1][tid] + sRed[2][tid] + sRed[3][tid];
  __syncthreads();

  if constexpr (ISO != 0) {
    static_assert(FORM == 0, "the isotropic shortcut needs r2 as the plain sum of squared differences");
    double acc = 0.0;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      acc = __builtin_fma(wh[it][0], r2[it][0], acc);
      acc = __builtin_fma(wh[it][1], r2[it][1], acc);
    }
    block_sum_store(acc, 2);
    __syncthreads();
    if (tid == 0) out[2] = -(sRed[0][2] + sRed[1][2] + sRed[2][2] + sRed[3][2]) / p.ls[0];
    if (tid >= 1 && tid < p.d) out[2 + tid] = 0.0;
    return;
  }
  // pass 2: per-dimension sums  -sum wh ds_k^2 / l_k   (d <= 8: the staged coordinates of pass 1 are still in LDS)
  for (int k0 = 0; k0 < p.d; k0 += KM_DC) {
    if (p.d > KM_DC) {
      stage(k0);
      __syncthreads();
    }
    double sk[KM_DC];
#pragma unroll
    for (int kk = 0; kk < KM_DC; ++kk) {
      const d2 bv = *reinterpret_cast<const d2*>(&sBt[kk][2 * cp]);
      double acc = 0.0;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const double av = sA[wave * 16 + 2 * it + rsub][kk];
        const double d0 = av - bv.x, d1 = av - bv.y;
        acc = __builtin_fma(wh[it][0] * d0, d0, acc);
        acc = __builtin_fma(wh[it][1] * d1, d1, acc);
      }
      sk[kk] = acc;
    }
#pragma unroll
    for (int kk = 0; kk < KM_DC; ++kk) block_sum_store(sk[kk], 2 + kk);
    __syncthreads();
    if (tid < KM_DC && k0 + tid < p.d)
      out[2 + k0 + tid] = -(sRed[0][2 + tid] + sRed[1][2 + tid] + sRed[2][2 + tid] + sRed[3][2 + tid]) / p.ls[k0 + tid];
    __syncthreads();
  }
}

template <int KID, int FORM = 0, int ISO = 0>
__global__ __launch_bounds__(256) void trace_kernel(TraceArgs p) {  // (216 VGPRs, two workgroups per CU; capped at 168 it spills: 8.9 -> 9.3 ms)
  __shared__ __attribute__((aligned(16))) double sA[KM_T][KM_DC];
  __shared__ __attribute__((aligned(16))) double sBt[KM_DC][KM_T];
  __shared__ double sRed[4][KM_DC + 2];
  trace_body<KID, FORM, ISO>(p, (int)blockIdx.x, sA, sBt, sRed);
}
// Two contractions in one launch (the sparse model's Kuf and Kuu terms): workgroups [0, first) take p, the rest q.
template <int KID, int FORM = 0, int ISO = 0>
__global__ __launch_bounds__(256) void trace_pair_kernel(TraceArgs p, TraceArgs q, int first) {
  __shared__ __attribute__((aligned(16))) double sA[KM_T][KM_DC];
  __shared__ __attribute__((aligned(16))) double sBt[KM_DC][KM_T];
  __shared__ double sRed[4][KM_DC + 2];
  if ((int)blockIdx.x < first)
    trace_body<KID, FORM, ISO>(p, (int)blockIdx.x, sA, sBt, sRed);
  else
    trace_body<KID, FORM, ISO>(q, (int)blockIdx.x - first, sA, sBt, sRed);
}

// dELBO/dZ of the sparse model:
//   dZ[i][k] = ( sum_n WHP[i][n] (z_ik - x_nk) + 2 sum_j WHQ[i][j] (z_ik - z_jk) ) / l_k^2
// with WHP = dELBO/dKuf * v h and WHQ = dELBO/dKuu * v h as stored by trace_kernel.  The differences
// are formed explicitly: for kernels whose h is singular at r -> 0 (Matern-1/2, "Exponential") the
// algebraically equal  z rowsum(WH) - WH X  cancels catastrophically on near-coincident points.
// One workgroup per (DZ_IG inducing points, DZ_DC dimensions): a point's coordinates are loaded once for all of them and
// a row of WHP once per chunk of dimensions (one workgroup per (i, k) re-read x for every i and WHP for every k: 0.5 GB
// of L2 traffic and 96 us for 16 cells of M = 50, N = 4096, d = 10).  Per output the operation order is unchanged:
// thread t accumulates n = t, t + 256, ... with one FMA each, wave sums, the four waves added in order.
// Round 5: four inducing points per workgroup (X is re-read half as often), the Kuu terms in the same accumulators with doubled weights, wave sums
// by DPP moves: 16 cells of M = 300, N = 4096, d = 10 157 -> 111 us, M = 128 56.6 -> 40.2 us.
constexpr int DZ_IG = 4, DZ_DC = 16;
__global__ __launch_bounds__(256) void dz_kernel(const double* __restrict__ Z, const double* __restrict__ X, const double* __restrict__ WHP,
                                                 int64_t ldp, const double* __restrict__ WHQ, int64_t ldq, const double* __restrict__ ls,
                                                 int m, int n, int d, double* __restrict__ dZ, int64_t cs = 0,
                                                 const double* __restrict__ cell_par = nullptr) {
  __shared__ double sred[4][DZ_IG * DZ_DC];
  if (cell_par) {  // batched: blockIdx.y = cell; Z, WHP, WHQ, dZ live in the cell block, lengthscales in the table
    const int64_t off = (int64_t)blockIdx.y * cs;
    Z += off;
    WHP += off;
    WHQ += off;
    dZ += off;
    ls = cell_par + (int64_t)blockIdx.y * CELL_PAR + CELL_PAR_LS;
  }
  const int kchunks = (d + DZ_DC - 1) / DZ_DC;
  const int i0 = ((int)blockIdx.x / kchunks) * DZ_IG, k0 = ((int)blockIdx.x % kchunks) * DZ_DC;
  const int ni = min(DZ_IG, m - i0), nk = min(DZ_DC, d - k0);
  double z[DZ_IG][DZ_DC];  // uniform over the workgroup: scalar registers
#pragma unroll
  for (int ii = 0; ii < DZ_IG; ++ii)
#pragma unroll
    for (int kk = 0; kk < DZ_DC; ++kk) z[ii][kk] = (ii < ni && kk < nk) ? Z[(int64_t)(i0 + ii) * d + k0 + kk] : 0.0;
  double acc[DZ_IG][DZ_DC];  // (the Kuu terms enter the same sums with their weights doubled: G_Q is symmetric, z_i sits at both index positions)
#pragma unroll
  for (int ii = 0; ii < DZ_IG; ++ii)
#pragma unroll
    for (int kk = 0; kk < DZ_DC; ++kk) acc[ii][kk] = 0.0;
  for (int c = threadIdx.x; c < n; c += 256) {
    double x[DZ_DC], w[DZ_IG];
#pragma unroll
    for (int kk = 0; kk < DZ_DC; ++kk) x[kk] = kk < nk ? X[(int64_t)c * d + k0 + kk] : 0.0;
#pragma unroll
    for (int ii = 0; ii < DZ_IG; ++ii) w[ii] = ii < ni ? WHP[(int64_t)(i0 + ii) * ldp + c] : 0.0;
#pragma unroll
    for (int ii = 0; ii < DZ_IG; ++ii)
#pragma unroll
      for (int kk = 0; kk < DZ_DC; ++kk) acc[ii][kk] = __builtin_fma(w[ii], z[ii][kk] - x[kk], acc[ii][kk]);
  }
  for (int c = threadIdx.x; c < m; c += 256) {
    double x[DZ_DC], w[DZ_IG];
#pragma unroll
    for (int kk = 0; kk < DZ_DC; ++kk) x[kk] = kk < nk ? Z[(int64_t)c * d + k0 + kk] : 0.0;
#pragma unroll
    for (int ii = 0; ii < DZ_IG; ++ii) w[ii] = ii < ni ? 2.0 * WHQ[(int64_t)(i0 + ii) * ldq + c] : 0.0;
#pragma unroll
    for (int ii = 0; ii < DZ_IG; ++ii)
#pragma unroll
      for (int kk = 0; kk < DZ_DC; ++kk) acc[ii][kk] = __builtin_fma(w[ii], z[ii][kk] - x[kk], acc[ii][kk]);
  }
#pragma unroll
  for (int ii = 0; ii < DZ_IG; ++ii)
#pragma unroll
    for (int kk = 0; kk < DZ_DC; ++kk) {
      const double a = wave_sum_dpp(acc[ii][kk]);  // (32 sums per workgroup: by ds_bpermute they were most of its time)
      if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6][ii * DZ_DC + kk] = a;
    }
  __syncthreads();
  const int e = threadIdx.x;
  if (e < DZ_IG * DZ_DC) {
    const int ii = e / DZ_DC, kk = e % DZ_DC;
    if (ii < ni && kk < nk) {
      const double l = ls[k0 + kk];
      dZ[(int64_t)(i0 + ii) * d + k0 + kk] = (sred[0][e] + sred[1][e] + sred[2][e] + sred[3][e]) / (l * l);
    }
  }
}
inline int dz_grid(int m, int d) { return ((m + DZ_IG - 1) / DZ_IG) * ((d + DZ_DC - 1) / DZ_DC); }

// out[e] = sum over workgroups of partial[wg][e]
__global__ __launch_bounds__(64) void trace_final(const double* __restrict__ partial, int nwg, int width, double* __restrict__ out,
                                                  int64_t partial_stride = 0, int64_t out_stride = -1) {
  const int e = blockIdx.x;
  partial += (int64_t)blockIdx.y * partial_stride;  // batched: blockIdx.y = cell, results width (or out_stride) apart
  out += (int64_t)blockIdx.y * (out_stride < 0 ? width : out_stride);
  double s = 0.0;
  for (int w = threadIdx.x; w < nwg; w += 64) s += partial[(int64_t)w * width + e];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[e] = s;
}

// trace_final for two partial blocks in one launch: blockIdx.x < width -> (partial_a, nwg_a) into out[e], else (partial_b, nwg_b)
// into out[width + e]
__global__ __launch_bounds__(64) void trace_final2(const double* __restrict__ partial_a, int nwg_a, const double* __restrict__ partial_b, int nwg_b,
                                                   int width, double* __restrict__ out, int64_t partial_stride, int64_t out_stride) {
  const bool second = (int)blockIdx.x >= width;
  const int e = second ? (int)blockIdx.x - width : (int)blockIdx.x;
  const double* partial = (second ? partial_b : partial_a) + (int64_t)blockIdx.y * partial_stride;
  const int nwg = second ? nwg_b : nwg_a;
  out += (int64_t)blockIdx.y * out_stride + (second ? width : 0);
  double s = 0.0;
  for (int w = threadIdx.x; w < nwg; w += 64) s += partial[(int64_t)w * width + e];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[e] = s;
}

inline hipError_t launch_trace(hipStream_t st, int kid, TraceArgs p, int grid, int cells = 1) {
  dim3 g(grid, cells), b(256);
#define GPRX_TRACE_CASE(K_)                                                   \
  case K_:                                                                    \
    if (p.form)                                                               \
      hipLaunchKernelGGL((trace_kernel<K_, 1>), g, b, 0, st, p);              \
    else if (p.iso)                                                           \
      hipLaunchKernelGGL((trace_kernel<K_, 0, 1>), g, b, 0, st, p);           \
    else                                                                      \
      hipLaunchKernelGGL((trace_kernel<K_, 0>), g, b, 0, st, p);              \
    break;
  switch (kid) {
    GPRX_TRACE_CASE(0)
    GPRX_TRACE_CASE(1)
    GPRX_TRACE_CASE(2)
    GPRX_TRACE_CASE(3)
    GPRX_TRACE_CASE(4)
    default: return hipErrorInvalidValue;
  }
#undef GPRX_TRACE_CASE
  return hipGetLastError();
}

inline hipError_t launch_trace_pair(hipStream_t st, int kid, TraceArgs p, int grid_p, TraceArgs q, int grid_q, int cells = 1) {
  if (p.form != q.form) return hipErrorInvalidValue;
  dim3 g(grid_p + grid_q, cells), b(256);
#define GPRX_TRACE_CASE(K_)                                                                \
  case K_:                                                                                 \
    if (p.form)                                                                            \
      hipLaunchKernelGGL((trace_pair_kernel<K_, 1>), g, b, 0, st, p, q, grid_p);           \
    else if (p.iso && q.iso)                                                               \
      hipLaunchKernelGGL((trace_pair_kernel<K_, 0, 1>), g, b, 0, st, p, q, grid_p);        \
    else                                                                                   \
      hipLaunchKernelGGL((trace_pair_kernel<K_, 0>), g, b, 0, st, p, q, grid_p);           \
    break;
  switch (kid) {
    GPRX_TRACE_CASE(0)
    GPRX_TRACE_CASE(1)
    GPRX_TRACE_CASE(2)
    GPRX_TRACE_CASE(3)
    GPRX_TRACE_CASE(4)
    default: return hipErrorInvalidValue;
  }
#undef GPRX_TRACE_CASE
  return hipGetLastError();
}

}  // namespace gprx
