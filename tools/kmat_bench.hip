// Standalone timing of the batched kernel-matrix build (kmat.h) for A/B variants (development aid):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DGPRX_KMAT_OLDEXP] [-DGPRX_KMAT_NOPAD] [-DGPRX_KMAT_FULLGRID] [-DGPRX_KMAT_NT] -o /tmp/kb tools/kmat_bench.hip
//   /tmp/kb [cells=128] [n=4096] [d=8]
#include "../gpras_amd/csrc/kmat.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace gprx;
#define CK(e)                                                                   \
  do {                                                                          \
    hipError_t e_ = (e);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));                   \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

int main(int argc, char** argv) {
  const int cells = argc > 1 ? atoi(argv[1]) : 128, n = argc > 2 ? atoi(argv[2]) : 4096, d = argc > 3 ? atoi(argv[3]) : 8;
  const int np = (n + 63) / 64 * 64;
  const int64_t ld = np, cs = (int64_t)(np + 64) * np + 64 * 1024;
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nd;
  std::vector<double> x((size_t)n * d), par((size_t)cells * CELL_PAR, 0.0);
  for (auto& v : x) v = nd(rng);
  for (int c = 0; c < cells; ++c) {
    par[(size_t)c * CELL_PAR + 0] = 1.0 + 0.01 * c;
    par[(size_t)c * CELL_PAR + 1] = 1.0;
    for (int k = 0; k < d; ++k) par[(size_t)c * CELL_PAR + CELL_PAR_LS + k] = 0.8 + 0.001 * c;
  }
  double *dx, *dpar, *out;
  CK(hipMalloc((void**)&dx, x.size() * 8));
  CK(hipMalloc((void**)&dpar, par.size() * 8));
  CK(hipMalloc((void**)&out, (size_t)cs * cells * 8));
  CK(hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dpar, par.data(), par.size() * 8, hipMemcpyHostToDevice));
  KmatArgs ka{dx, dx, nullptr, out, ld, n, n, d, np, np, 0.0, 0.0, 1, 1.0, nullptr, 0};
  ka.cell_par = dpar;
  ka.out_stride = cs;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) CK(launch_kmat(st, 0, ka, cells));
  CK(hipStreamSynchronize(st));
  float best = 1e9f, sum = 0.f;
  const int reps = 10;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, st));
    CK(launch_kmat(st, 0, ka, cells));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double bytes = 8.0 * 64 * 64 * (double)(np / 64) * (np / 64 + 1) / 2 * cells;
  // correctness sample: cell 3, a few hundred entries against the host's exp
  const int c = cells > 3 ? 3 : 0;
  std::vector<double> row(np);
  double worst = 0.0;
  for (int i : {0, 1, 63, 64, 1000 % n, n - 1}) {
    CK(hipMemcpy(row.data(), out + (size_t)c * cs + (size_t)i * ld, np * 8, hipMemcpyDeviceToHost));
    for (int j = 0; j <= i; j += 7) {
      double r2 = 0.0;
      const double l = par[(size_t)c * CELL_PAR + CELL_PAR_LS];
      for (int k = 0; k < d; ++k) {
        const double t = x[(size_t)i * d + k] * (1.0 / l) - x[(size_t)j * d + k] * (1.0 / l);
        r2 = std::fma(t, t, r2);
      }
      double want = par[(size_t)c * CELL_PAR] * std::exp(-0.5 * r2) + (i == j ? par[(size_t)c * CELL_PAR + 1] : 0.0);
      worst = std::fmax(worst, std::fabs(row[j] - want) / want);
    }
  }
  printf("cells %d n %d d %d: best %.3f ms avg %.3f ms  %.2f TB/s (best)  max rel err vs host %.2e\n", cells, n, d, best, sum / reps, bytes / (best * 1e-3) / 1e12,
         worst);
  return 0;
}
