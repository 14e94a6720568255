// Which operand pattern lets v_mfma_f64_16x16x4_f64 issue fastest?  (development aid)
//   V=0: 4 accumulators, one (a, b) pair           V=1: 4 accumulators, 4 different (a, b) pairs
//   V=2: 16 accumulators as a GEMM wave has them: acc[i][j] += fa[i] * fb[j], i, j < 4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ __launch_bounds__(256) void k(double* out, int iters, double da) {
  d4 c[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) c[i][j] = d4{0, 0, 0, 0};
  double fa[4], fb[4];
  for (int i = 0; i < 4; ++i) {
    fa[i] = 1.0 + (threadIdx.x + 7 * i) * da;
    fb[i] = 1.0 - (threadIdx.x + 3 * i) * da;
  }
  for (int it = 0; it < iters; ++it) {
    if constexpr (V == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0], fb[0], c[0][j], 0, 0, 0);
    } else if constexpr (V == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[j], fb[j], c[0][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], c[i][j], 0, 0, 0);
    }
  }
  d4 s = d4{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s += c[i][j];
  if (s.x == 123.456) out[0] = s.y;
}
template <int V>
void run(const char* name, int blocks, double* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 10000;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, nullptr, out, iters, 1e-9);
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 16 * 2048.0;
    if (rep == 2) printf("%-44s %4d blocks: %8.3f ms %7.2f TFLOP/s\n", name, blocks, ms, flops / (ms * 1e-3) / 1e12);
  }
}
int main() {
  double* out;
  (void)hipMalloc(&out, 64);
  for (int blocks : {256, 512, 1024}) {
    run<0>("4 acc, same operands", blocks, out);
    run<1>("4 acc, 4 operand pairs", blocks, out);
    run<2>("16 acc, fa[i] x fb[j] (GEMM pattern)", blocks, out);
  }
  return 0;
}
