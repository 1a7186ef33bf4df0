// Microbenchmark: do v_exp_f32 (transcendental) and v_mfma_f32_32x32x16_bf16 overlap on one SIMD across waves?
// Per iteration: MODE 0 = 3 MFMAs, MODE 1 = 16 exps (+8 cvt), MODE 2 = both (attention's per-tile mix), 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc0 = (f32x16)(0.f), acc1 = (f32x16)(0.f);
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f); }
  float e[16];
  for (int i = 0; i < 16; ++i) e[i] = threadIdx.x * 1e-3f + i * 0.01f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) e[i] = __builtin_amdgcn_exp2f(e[i] * 1e-9f);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + e[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int blocks_per_cu) {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 4000, blocks = 256 * blocks_per_cu;
  k<MODE><<<blocks, 256>>>(d, 10); (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: blocks_per_cu waves, each does `iters` iterations
  printf("%-34s %d waves/SIMD: %.3f ms -> %.1f ns per iteration per SIMD (%.0f cycles @2.0 GHz)\n", name, blocks_per_cu, ms,
         ms * 1e6 / (iters * blocks_per_cu), ms * 1e6 / (iters * blocks_per_cu) * 2.0);
  (void)hipFree(d);
}
int main() {
  for (int w : {1, 4, 8}) {
    if (w == 1) { run<0>("3 MFMA 32x32x16", 1); run<1>("16 exp (+16 mul)", 1); run<2>("3 MFMA + 16 exp", 1); }
    if (w == 4) { run<0>("3 MFMA 32x32x16", 4); run<1>("16 exp (+16 mul)", 4); run<2>("3 MFMA + 16 exp", 4); }
    if (w == 8) { run<0>("3 MFMA 32x32x16", 8); run<1>("16 exp (+16 mul)", 8); run<2>("3 MFMA + 16 exp", 8); }
  }
  return 0;
}
