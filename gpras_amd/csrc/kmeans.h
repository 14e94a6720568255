// Lloyd iterations of the k-means inducing-point initialisation -- SURVEY.md section 8(f) row N4, the work behind
// KMeans(n_clusters=M, random_state=0, n_init="auto").fit(x) at gpras/gpr.py:312-315 after its k-means++ seeding
// (scikit-learn's _kmeans_single_lloyd; restated on the CPU by oracle/kmeans.py).
//
//   E-step  kmeans_assign_kernel: a thread owns a point; the centres pass through LDS in chunks; squared distances in the
//           difference form, accumulated in k order WITHOUT fused multiply-add (the oracle's and numpy's arithmetic, so
//           that near-ties resolve identically); the first nearest centre wins (strict <, ascending index: numpy argmin).
//   M-step  kmeans_update_kernel: a workgroup owns a cluster: members summed in a fixed order (thread-strided over the
//           points, then a fixed tree), mean, squared shift against the old centre, member count (0 raises the empty flag:
//           scikit-learn relocates empty clusters, the caller falls back to it).
// N x M x d is tiny (4096 x 50 x 10): the cost is the dependent launches and one small read-back per iteration.
#pragma once
#include "gprx_common.h"

namespace gprx {

constexpr int KME_LDS = 4096;  // doubles of centre coordinates per LDS pass

// stat[0] = 1.0 if any label changed, stat[1] = 1.0 if a cluster is empty, stat[2 + j] = squared shift of centre j
__global__ __launch_bounds__(256) void kmeans_assign_kernel(const double* __restrict__ X, int n, int d, const double* __restrict__ C, int m,
                                                            int* __restrict__ labels, double* __restrict__ stat) {
  __shared__ double sC[KME_LDS];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const double* xi = X + (int64_t)(i < n ? i : 0) * d;
  const int chunk = KME_LDS / d;  // centres per pass (d <= 64 is checked by the host: at least 64)
  double best = 0.0;
  int bestj = -1;
  for (int j0 = 0; j0 < m; j0 += chunk) {
    const int cnt = min(chunk, m - j0);
    for (int e = threadIdx.x; e < cnt * d; e += 256) sC[e] = C[(int64_t)j0 * d + e];
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      double acc = 0.0;
      {
#pragma clang fp contract(off)
        for (int k = 0; k < d; ++k) {
          const double diff = xi[k] - sC[j * d + k];
          const double sq = diff * diff;
          acc = acc + sq;
        }
      }
      if (bestj < 0 || acc < best) {
        best = acc;
        bestj = j0 + j;
      }
    }
    __syncthreads();
  }
  if (i < n) {
    if (labels[i] != bestj) stat[0] = 1.0;  // (every writer stores the same value)
    labels[i] = bestj;
  }
}

__global__ __launch_bounds__(256) void kmeans_update_kernel(const double* __restrict__ X, int n, int d, const int* __restrict__ labels,
                                                            const double* __restrict__ Cold, double* __restrict__ Cnew, double* __restrict__ stat) {
  __shared__ double sred[4][9];
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double shift2 = 0.0;
  double count = 0.0;
  for (int k0 = 0; k0 < d; k0 += 8) {
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double c = 0.0;
    for (int i = tid; i < n; i += 256) {
      if (labels[i] != j) continue;
      c += 1.0;
      const double* xi = X + (int64_t)i * d + k0;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
        if (k0 + kk < d) a[kk] += xi[kk];
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) a[kk] = wave_sum(a[kk]);
    c = wave_sum(c);
    __syncthreads();  // (sred of the previous pass has been consumed)
    if (lane == 0) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) sred[wave][kk] = a[kk];
      sred[wave][8] = c;
    }
    __syncthreads();
    count = sred[0][8] + sred[1][8] + sred[2][8] + sred[3][8];
    if (tid < 8 && k0 + tid < d) {
      const double sum = sred[0][tid] + sred[1][tid] + sred[2][tid] + sred[3][tid];
      const double old = Cold[(int64_t)j * d + k0 + tid];
      const double cen = count > 0.0 ? sum / count : old;
      Cnew[(int64_t)j * d + k0 + tid] = cen;
      const double diff = cen - old;
      sred[0][tid] = diff * diff;  // (only thread tid read sred[*][tid]: safe to reuse its own slot)
    }
    __syncthreads();
    if (tid == 0)
      for (int kk = 0; kk < 8 && k0 + kk < d; ++kk) shift2 += sred[0][kk];
  }
  if (tid == 0) {
    stat[2 + j] = shift2;
    if (count == 0.0) stat[1] = 1.0;
  }
}

}  // namespace gprx
