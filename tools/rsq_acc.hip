#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* o0, double* o1, double* o2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = x[i];
  double y = __builtin_amdgcn_rsq(a);
  o0[i] = y;
  double h = 0.5 * a;
  y = y * __builtin_fma(-h * y, y, 1.5);
  o1[i] = y;
  y = y * __builtin_fma(-h * y, y, 1.5);
  o2[i] = y;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), r0(n), r1(n), r2(n);
  srand(1);
  for (int i = 0; i < n; ++i) x[i] = exp(((double)rand() / RAND_MAX - 0.5) * 40.0) * (1.0 + (double)rand() / RAND_MAX);
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    long double t = 1.0L / sqrtl((long double)x[i]);
    e0 = fmax(e0, (double)fabsl((r0[i] - t) / t)); e1 = fmax(e1, (double)fabsl((r1[i] - t) / t)); e2 = fmax(e2, (double)fabsl((r2[i] - t) / t));
  }
  printf("max rel err: seed %.3e (2^%.1f)  1 NR %.3e  2 NR %.3e\n", e0, log2(e0), e1, e2);
  return 0;
}
