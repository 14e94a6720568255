// Do fp64 VALU instructions and fp64 MFMAs of one SIMD overlap on gfx950?  Three loops per wave: 16 independent v_fma_f64 chains
// (64 issue cycles per iteration), 2 independent v_mfma_f64_16x16x4 chains (128 MFMA cycles per iteration), or both in the same
// iteration.  If the two share a datapath the combined loop costs the sum, if not the maximum.
//   hipcc --offload-arch=gfx950 -O3 -o tools/f64_pipe_probe tools/f64_pipe_probe.hip && tools/f64_pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(double* out, int iters, double a, double b) {
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = a + i + threadIdx.x;
  d4 m0 = {a, b, a, b}, m1 = {b, a, b, a};
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE != 0) {
      m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m0, 0, 0, 0);
      m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m1, 0, 0, 0);
    }
    if constexpr (MODE != 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], a, b);
    }
  }
  double s = m0[0] + m0[1] + m0[2] + m0[3] + m1[0] + m1[1] + m1[2] + m1[3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.678) out[0] = s;
}
template <int MODE>
float run(double* out, int waves_per_simd, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * waves_per_simd;  // 256 CUs x (4 waves per workgroup = one per SIMD) x waves_per_simd workgroups per CU
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  double* out;
  hipMalloc(&out, 64);
  const int iters = 200000;
  for (int w : {1, 2, 4}) {
    const float valu = run<0>(out, w, iters), mfma = run<1>(out, w, iters), both = run<2>(out, w, iters);
    // per wave and iteration: cycles at 2.4 GHz nominal
    const double c = 2.4e6 / iters / w;  // ms -> cycles per iteration per resident wave slot
    printf("waves/SIMD %d: VALU only %.2f ms (%.0f cyc/iter/wave)  MFMA only %.2f ms (%.0f)  both %.2f ms (%.0f)  sum %.2f max %.2f\n", w, valu, valu * c,
           mfma, mfma * c, both, both * c, valu + mfma, valu > mfma ? valu : mfma);
  }
  return 0;
}
