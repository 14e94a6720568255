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

// ---- k-means++ seeding on the device (sklearn.cluster._kmeans._kmeans_plusplus, behind gpr.py:313) -----------------------------
// The host keeps only what cannot move: the RandomState(0) draws (first index, then n_local_trials uniforms per centre -- they
// do not depend on the data).  Per centre two launches:
//   kpp_select_kernel (one workgroup): potentials of the previous candidates (partial sums added in a fixed order), the best
//       one (first minimum: numpy argmin) becomes a centre and its distance array the current closest_dist_sq; then
//       searchsorted(cumsum(closest_dist_sq), uniforms * potential): 256 chunk sums, a sequential walk over the chunk sums and
//       then inside the chunk (first i with cumsum[i] >= v, clipped to n - 1);
//   kpp_dist_kernel (n / 256 x trials workgroups): squared distances of every point to the candidates in scikit-learn's
//       expanded form ((-2 x.c + |c|^2) + |x|^2, clamped at 0, the norms as the host's row_norms gave them), minimum with the
//       current closest distances, partial potentials per workgroup.
// Rounding differs from scikit-learn's BLAS products in the last bits; an index can only differ when a uniform draw falls
// within that rounding of a cumulative sum (probability ~ n 1e-16 per draw): the test matrix ends at identical centres.
constexpr int KPP_MAX_TRIALS = 16;
struct KppState {
  int best;                   // index (among the previous candidates) of the one that became a centre
  int cand[KPP_MAX_TRIALS];   // candidate point indices of the current centre
  double pot;
};

// mode 0: distances to ONE given point (the first centre), no minimum.  grid (blocks, trials)
__global__ __launch_bounds__(256) void kpp_dist_kernel(const double* __restrict__ X, int n, int d, const double* __restrict__ xsq,
                                                       const KppState* __restrict__ st, const double* __restrict__ prev, int first_id,
                                                       double* __restrict__ out, double* __restrict__ partial, int nblocks) {
  __shared__ double sc[64];
  __shared__ double sred[4];
  const int l = blockIdx.y;
  const int cand = prev ? st->cand[l] : first_id;
  const double* closest = prev ? prev + (int64_t)st->best * n : nullptr;
  if (threadIdx.x < d) sc[threadIdx.x] = X[(int64_t)cand * d + threadIdx.x];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  double v = 0.0;
  if (i < n) {
    const double* xi = X + (int64_t)i * d;
    double dot = 0.0;
    for (int k = 0; k < d; ++k) dot = __builtin_fma(sc[k], xi[k], dot);
    {
#pragma clang fp contract(off)
      double t = -2.0 * dot;
      t = t + xsq[cand];
      t = t + xsq[i];
      v = t > 0.0 ? t : 0.0;
    }
    if (closest) v = closest[i] < v ? closest[i] : v;
    out[(int64_t)l * n + i] = v;
  }
  double s = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)l * nblocks + blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

// prev_trials candidates of the previous centre (buffers prev[l][n], partial sums) -> the chosen one (indices_out[c - 1]); then,
// if uniforms != nullptr, the `trials` candidates of the next centre into st_out
__global__ __launch_bounds__(256) void kpp_select_kernel(int n, const double* __restrict__ prev, const double* __restrict__ partial, int nblocks,
                                                         int prev_trials, const KppState* __restrict__ st_in, int first_id, KppState* __restrict__ st_out,
                                                         const double* __restrict__ uniforms, int trials, long long* __restrict__ index_out) {
  __shared__ double spot[KPP_MAX_TRIALS];
  __shared__ double schunk[256];
  __shared__ int sbest;
  const int tid = threadIdx.x;
  if (tid < prev_trials) {
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)tid * nblocks + b];
    spot[tid] = s;
  }
  __syncthreads();
  if (tid == 0) {
    int best = 0;
    for (int l = 1; l < prev_trials; ++l)
      if (spot[l] < spot[best]) best = l;
    sbest = best;
    st_out->best = best;
    st_out->pot = spot[best];
    *index_out = st_in ? st_in->cand[best] : first_id;
  }
  __syncthreads();
  if (!uniforms) return;
  const double* closest = prev + (int64_t)sbest * n;
  const double pot = spot[sbest];
  const int chunk = (n + 255) / 256;
  {
    double s = 0.0;
    const int lo = tid * chunk, hi = min(n, lo + chunk);
    for (int i = lo; i < hi; ++i) s += closest[i];
    schunk[tid] = s;
  }
  __syncthreads();
  if (tid < trials) {
    const double v = uniforms[tid] * pot;
    double run = 0.0;
    int c = 0;
    for (; c < 256; ++c) {  // first chunk whose end reaches v
      if (run + schunk[c] >= v) break;
      run += schunk[c];
    }
    int idx = n - 1;  // np.clip(candidate_ids, None, n - 1)
    if (c < 256) {
      const int lo = c * chunk, hi = min(n, lo + chunk);
      for (int i = lo; i < hi; ++i) {
        run += closest[i];
        if (run >= v) {
          idx = i;
          break;
        }
      }
      if (idx == n - 1 && hi < n) idx = hi < n ? hi : n - 1;  // (rounding between the chunk sum and the walk: the next element)
    }
    st_out->cand[tid] = idx;
  }
}

}  // namespace gprx
