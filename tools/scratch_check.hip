// Is per-wave scratch private when two 256-thread workgroups share a CU? (development aid, round 4 bug hunt)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k(int* bad, int iters, int* sink) {
  volatile int priv[8];
  const int me = blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < 8; ++i) priv[i] = me * 8 + i;
  int errs = 0, acc = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = 0; i < 8; ++i) {
      const int v = priv[(i + it) & 7];
      if (v != me * 8 + ((i + it) & 7)) ++errs;
      acc += v;
    }
    __builtin_amdgcn_s_sleep(8);
    priv[it & 7] = me * 8 + (it & 7);
  }
  if (errs) atomicAdd(bad, 1);
  if (acc == 123456789) *sink = acc;
}
int main() {
  int *bad, *sink, h = 0;
  hipMalloc(&bad, 4);
  hipMalloc(&sink, 4);
  hipMemset(bad, 0, 4);
  hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, bad, 20000, sink);
  hipDeviceSynchronize();
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("threads that read a wrong private value: %d\n", h);
  return 0;
}
