// Standalone check of the one-workgroup-per-cell kernels at FULL load (development aid): `cells` SPD matrices of order n are factored by
// the single-column kernel and by the column-pair kernel; the two factors (and the inverse diagonal blocks) are compared element by
// element on the host and the tiles that differ are listed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/cc tools/cell_check.hip && /tmp/cc [n=1024] [cells=512] [reps=3]
#include "../gpras_amd/csrc/potrf_cell.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
using namespace gprx;
#define CK(e)                                                   \
  do {                                                          \
    hipError_t e_ = (e);                                        \
    if (e_ != hipSuccess) {                                     \
      fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));   \
      return 1;                                                 \
    }                                                           \
  } while (0)

__global__ void rhs_kernel(double* A, int64_t ld, int np, int64_t cs) {
  double* dst = A + (int64_t)blockIdx.y * cs + (int64_t)np * ld;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 64 * np; e += gridDim.x * blockDim.x) {
    const int r = e / np, c = e % np;
    dst[(int64_t)r * ld + c] = r == 0 ? sin(0.01 * c + blockIdx.y) : 0.0;
  }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1024, cells = argc > 2 ? atoi(argv[2]) : 512, reps = argc > 3 ? atoi(argv[3]) : 3, d = 8;
  const int np = (n + 63) / 64 * 64, T = np / 64;
  const int64_t ld = np, cs = (int64_t)(np + 64) * np + (int64_t)np * 64 + 64;
  const int64_t off_inv = (int64_t)(np + 64) * np;
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nd;
  std::vector<double> x((size_t)n * d), par((size_t)cells * CELL_PAR, 0.0);
  for (auto& v : x) v = nd(rng);
  for (int c = 0; c < cells; ++c) {
    par[(size_t)c * CELL_PAR + 0] = 1.0 + 0.001 * c;
    par[(size_t)c * CELL_PAR + 1] = 1.0;
    for (int k = 0; k < d; ++k) par[(size_t)c * CELL_PAR + CELL_PAR_LS + k] = 0.8 + 0.0005 * c;
  }
  double *dx, *dpar, *a0, *a1, *a2;
  int* info;
  CK(hipMalloc((void**)&dx, x.size() * 8));
  CK(hipMalloc((void**)&dpar, par.size() * 8));
  CK(hipMalloc((void**)&a0, (size_t)cs * cells * 8));
  CK(hipMalloc((void**)&a1, (size_t)cs * cells * 8));
  CK(hipMalloc((void**)&a2, (size_t)cs * cells * 8));
  CK(hipMalloc((void**)&info, sizeof(int) * cells * 2));
  CK(hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dpar, par.data(), par.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(a0, 0, (size_t)cs * cells * 8));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  KmatArgs ka{dx, dx, nullptr, a0, ld, n, n, d, np, np, 0.0, 0.0, 1, 1.0, nullptr, 0};
  ka.cell_par = dpar;
  ka.out_stride = cs;
  CK(launch_kmat(st, 0, ka, cells));
  hipLaunchKernelGGL(rhs_kernel, dim3(64, cells), dim3(256), 0, st, a0, ld, np, cs);
  CK(hipStreamSynchronize(st));
  CellArgs ca;
  ca.lda = ld;
  ca.T = T;
  ca.R = T + 1;
  ca.cs = cs;
  ca.info_stride = 1;
  ca.col_base = 0;
  std::vector<double> h1((size_t)cs), h2((size_t)cs);
  for (int rep = 0; rep < reps; ++rep) {
    CK(hipMemcpyAsync(a1, a0, (size_t)cs * cells * 8, hipMemcpyDeviceToDevice, st));
    CK(hipMemcpyAsync(a2, a0, (size_t)cs * cells * 8, hipMemcpyDeviceToDevice, st));
    CK(hipMemsetAsync(info, 0, sizeof(int) * cells * 2, st));
    ca.A = a1;
    ca.inv_diag = a1 + off_inv;
    ca.info = info;
    hipLaunchKernelGGL(potrf_cell_kernel_t<true>, dim3(cells), dim3(256), 0, st, ca);
    ca.A = a2;
    ca.inv_diag = a2 + off_inv;
    ca.info = info + cells;
    if (getenv("CC_BUILD_K")) {
      ca.X = dx;
      ca.cell_par = dpar;
      ca.n = n;
      ca.d = d;
      hipLaunchKernelGGL(potrf_cell2_kernel<true>, dim3(cells), dim3(256), 0, st, ca);
    } else {
      hipLaunchKernelGGL(potrf_cell2_kernel<false>, dim3(cells), dim3(256), 0, st, ca);
    }
    CK(hipStreamSynchronize(st));
    int bad_cells = 0, shown = 0;
    for (int c = 0; c < cells; ++c) {
      CK(hipMemcpy(h1.data(), a1 + (size_t)c * cs, (size_t)cs * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h2.data(), a2 + (size_t)c * cs, (size_t)cs * 8, hipMemcpyDeviceToHost));
      bool bad = false;
      for (int ti = 0; ti <= T; ++ti)
        for (int tj = 0; tj <= (ti < T ? ti : T - 1); ++tj) {
          int cnt = 0, fr = -1, fc = -1;
          double mx = 0.0;
          for (int r = 0; r < 64; ++r)
            for (int q = 0; q < 64; ++q) {
              if (ti == tj && q > r) continue;
              const size_t e = (size_t)(ti * 64 + r) * ld + tj * 64 + q;
              if (memcmp(&h1[e], &h2[e], 8) != 0) {
                if (!cnt) fr = r, fc = q;
                ++cnt;
                mx = fmax(mx, fabs(h1[e] - h2[e]));
              }
            }
          if (cnt) {
            bad = true;
            if (shown < 12) {
              printf("  rep %d cell %d tile (%d,%d): %d entries differ, max |d| %.2e, first at row %d col %d\n", rep, c, ti, tj, cnt, mx, fr, fc);
              ++shown;
            }
          }
        }
      for (int j = 0; j < T; ++j) {
        int cnt = 0;
        for (int e = 0; e < 4096; ++e) cnt += memcmp(&h1[off_inv + (size_t)j * 4096 + e], &h2[off_inv + (size_t)j * 4096 + e], 8) != 0;
        if (cnt) {
          bad = true;
          if (shown < 12) {
            printf("  rep %d cell %d inverse block %d: %d entries differ\n", rep, c, j, cnt);
            ++shown;
          }
        }
      }
      bad_cells += bad;
    }
    printf("rep %d: %d of %d cells differ between the two kernels\n", rep, bad_cells, cells);
    // ground truth for the first rows of the first differing cell: which kernel is off?
    for (int c = 0; c < cells && rep == 0; ++c) {
      CK(hipMemcpy(h1.data(), a1 + (size_t)c * cs, (size_t)cs * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h2.data(), a2 + (size_t)c * cs, (size_t)cs * 8, hipMemcpyDeviceToHost));
      if (memcmp(h1.data(), h2.data(), (size_t)np * ld * 8) == 0) continue;
      const int m = 192;
      std::vector<double> k0((size_t)cs), L((size_t)m * m, 0.0);
      CK(hipMemcpy(k0.data(), a0 + (size_t)c * cs, (size_t)cs * 8, hipMemcpyDeviceToHost));
      for (int i = 0; i < m; ++i)
        for (int j = 0; j <= i; ++j) {
          long double sum = k0[(size_t)i * ld + j];
          for (int k = 0; k < j; ++k) sum -= (long double)L[(size_t)i * m + k] * L[(size_t)j * m + k];
          L[(size_t)i * m + j] = (i == j) ? (double)sqrtl(sum) : (double)(sum / L[(size_t)j * m + j]);
        }
      double e1 = 0, e2 = 0;
      for (int i = 0; i < m; ++i)
        for (int j = 0; j <= i; ++j) {
          e1 = fmax(e1, fabs(h1[(size_t)i * ld + j] - L[(size_t)i * m + j]));
          e2 = fmax(e2, fabs(h2[(size_t)i * ld + j] - L[(size_t)i * m + j]));
        }
      {  // where in tile (1, 0) do the two kernels differ, and is the stored inverse of block 0 the inverse of the stored L(0,0)?
        int colcnt[64] = {0}, rowcnt[64] = {0};
        for (int r = 0; r < 64; ++r)
          for (int q = 0; q < 64; ++q)
            if (memcmp(&h1[(size_t)(64 + r) * ld + q], &h2[(size_t)(64 + r) * ld + q], 8) != 0) ++colcnt[q], ++rowcnt[r];
        printf("tile (1,0) differing entries per column:");
        for (int q = 0; q < 64; ++q)
          if (colcnt[q]) printf(" c%d:%d", q, colcnt[q]);
        printf("\n");
        {  // the stored inverse of block 0 of the two kernels, entry by entry (row c, column m of L00^-1)
          int shown2 = 0;
          for (int cc = 0; cc < 64; ++cc) {
            int cnt = 0, first = -1, last = -1;
            double mx = 0.0, rel = 0.0;
            for (int m = 0; m <= cc; ++m) {
              const double a = h1[off_inv + (size_t)cc * 64 + m], b = h2[off_inv + (size_t)cc * 64 + m];
              if (memcmp(&a, &b, 8) != 0) {
                if (first < 0) first = m;
                last = m;
                ++cnt;
                mx = fmax(mx, fabs(a - b));
                rel = fmax(rel, fabs(a - b) / fabs(a));
              }
            }
            if (cnt && shown2++ < 8) printf("inverse block 0 row %d: %d entries differ (columns %d..%d), max |d| %.2e, max rel %.2e\n", cc, cnt, first, last, mx, rel);
          }
        }
        for (int which = 0; which < 2; ++which) {
          const std::vector<double>& hh = which ? h2 : h1;
          double worst = 0.0;
          for (int i = 0; i < 64; ++i)
            for (int j = 0; j <= i; ++j) {
              long double sum = 0.0L;
              for (int k = j; k <= i; ++k) sum += (long double)hh[off_inv + (size_t)i * 64 + k] * hh[(size_t)k * ld + j];
              worst = fmax(worst, fabs((double)sum - (i == j ? 1.0 : 0.0)));
            }
          printf("%s: |inv0 * L00 - I| max %.2e\n", which ? "column-pair" : "single-column", worst);
        }
      }
      printf("cell %d, first %d rows against a long-double host Cholesky: single-column max |err| %.2e, column-pair max |err| %.2e\n", c, m, e1, e2);
      break;
    }
  }
  return 0;
}
