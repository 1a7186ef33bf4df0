// Microbenchmark: throughput of v_exp_f32 alone, v_fma_f32 alone, and both interleaved (are they separate pipes?).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = a[i] * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0 || MODE == 2) a[i] = __builtin_amdgcn_exp2f(a[i]) * 0.0f + a[i];  // keeps value bounded; exp + fma
      if (MODE == 1 || MODE == 2) { f[i] = f[i] * 1.0001f + 0.5f; f[i] = f[i] * 0.9999f - 0.5f; f[i] = f[i] * 1.0001f + 0.25f; }
      if (MODE == 3) a[i] = __builtin_amdgcn_exp2f(a[i] * 1e-9f);   // exp + mul
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i] + f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int per_iter_exp, int per_iter_fma) {
  float* d; hipMalloc(&d, 256 * 8192 * 4);
  int iters = 2000;
  k<MODE><<<256 * 8, 256>>>(d, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<MODE><<<256 * 8, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double waves = 256.0 * 8 * 4, winst = waves * iters * 16;
  double cyc = ms * 1e-3 * 2.4e9 * 1024;   // SIMD-cycles at 2.4 GHz
  printf("%-28s %.3f ms: %.1f SIMD-cycles@2.4GHz per (exp x%d, fma x%d) wave-instr group\n", name, ms, cyc / winst, per_iter_exp, per_iter_fma);
  hipFree(d);
}
int main() {
  run<0>("exp + 1 fma", 1, 1);
  run<1>("3 fma", 0, 3);
  run<2>("exp + 4 fma", 1, 4);
  run<3>("exp + 1 mul", 1, 1);
  return 0;
}
