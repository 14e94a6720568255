// Sustained v_mfma_f64_16x16x4_f64 rate on gfx950 under different occupancies and operand values
// (development aid: is the nominal 78.6 TFLOP/s reachable, or does the clock drop under fp64 matrix load?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0, double da, unsigned long long* cyc) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double a = a0 + threadIdx.x * da, b = b0 - threadIdx.x * da;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const d4 s = c0 + c1 + c2 + c3;
  if (s.x == 123.456) out[0] = s.y;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 40000;
  struct { int blocks; int threads; double a, b, da; const char* name; } cfg[] = {
      {256, 256, 1.0, 1.0, 1e-9, "1 wave/SIMD, varied operands"}, {1024, 256, 1.0, 1.0, 1e-9, "4 waves/SIMD, varied operands"},
      {1024, 256, 0.0, 0.0, 0.0, "4 waves/SIMD, zero operands"},  {256, 256, 0.0, 0.0, 0.0, "1 wave/SIMD, zero operands"},
      {128, 256, 1.0, 1.0, 1e-9, "half the CUs, 1 wave/SIMD"},
      {512, 256, 1.0, 1.0, 1e-9, "2 waves/SIMD, varied operands"},
      {768, 256, 1.0, 1.0, 1e-9, "3 waves/SIMD, varied operands"},
      {2048, 256, 1.0, 1.0, 1e-9, "8 waves/SIMD, varied operands"}};
  for (auto& c : cfg) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, nullptr);
      hipLaunchKernelGGL(k, dim3(c.blocks), dim3(c.threads), 0, nullptr, out, iters, c.a, c.b, c.da, cyc);
      hipEventRecord(e1, nullptr);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
      const double flops = (double)c.blocks * (c.threads / 64) * iters * 4 * 2048.0;
      printf("%-34s rep %d: %8.3f ms %7.2f TFLOP/s | wave 0: %llu ticks for %d MFMAs = %.1f ticks/MFMA, %.1f ticks/us\n", c.name, rep, ms,
             flops / (ms * 1e-3) / 1e12, hc, iters * 4, (double)hc / (iters * 4), hc / (ms * 1e3));
    }
  }
  return 0;
}
