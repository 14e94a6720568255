// Read-modify-write of 512-byte row segments at a given row stride: is a power-of-two stride slower on MI355X? (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void rmw(double* a, long stride_elems, int rows, long cell_stride) {
  double* base = a + (long)blockIdx.y * cell_stride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // workgroup = 128 rows; a wave = 32 rows; lane covers 16 B of a 512-B segment per 2 rows ... simple: each wave iterates rows
  for (int i = 0; i < 32; ++i) {
    int row = blockIdx.x * 128 + wave * 32 + i;
    if (row < rows) {
      double* p = base + (long)row * stride_elems + lane;
      *p = *p * 1.0000001 + 1.0;
    }
  }
}
int main(int argc, char** argv) {
  const int rows = 2048, cells = 128;
  long strides[] = {4096, 4096 + 16, 4096 + 32, 4096 + 64, 4096 + 512, 4160};
  for (long s : strides) {
    long cell_stride = (long)(4096 + 64) * s + 4096 * 64;  // like the arena
    double* a;
    size_t bytes = sizeof(double) * cell_stride * cells;
    if (hipMalloc(&a, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(rmw, dim3(rows / 128, cells), dim3(256), 0, 0, a + 1024 * s + 512, s, rows, cell_stride);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double gb = 2.0 * rows * 512.0 * cells / 1e9;
    printf("stride %ld doubles: %.1f us, %.2f TB/s (r+w)\n", s, best * 1e3, gb / best);
    hipFree(a);
  }
  return 0;
}
