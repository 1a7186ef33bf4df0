// Microbenchmark (round 3): issue cost of the candidates for the d = 8 attention's exponentials and conversions on gfx950 --
// v_exp_f32 vs v_exp_f16 (is the 16-bit transcendental any cheaper?), v_cvt_pk_bf16_f32, v_cvt_pk_f16_f32 (v_cvt_pkrtz),
// v_pk_mul_f16, v_pk_fma_f32.  One wave per SIMD would hide nothing, so 4 waves per SIMD issue independent streams; the
// figure printed is SIMD-cycles (at 2.4 GHz nominal) per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exp_variants scripts/micro/exp_variants.hip && /tmp/exp_variants
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (MODE == 1) {
      REP16(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (MODE == 2) {
      REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
    } else if (MODE == 3) {
      REP16(asm volatile("v_cvt_pkrtz_f16_f32 %0, %4, %5\n v_cvt_pkrtz_f16_f32 %1, %5, %6\n v_cvt_pkrtz_f16_f32 %2, %6, %7\n v_cvt_pkrtz_f16_f32 %3, %7, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
    } else if (MODE == 4) {
      REP16(asm volatile("v_pk_mul_f16 %0, %0, %4\n v_pk_mul_f16 %1, %1, %5\n v_pk_mul_f16 %2, %2, %6\n v_pk_mul_f16 %3, %3, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
    } else if (MODE == 5) {
      REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %5, %6\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %7, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
    } else if (MODE == 6) {       // exp + cvt interleaved: do they share an issue port?
      REP16(asm volatile("v_exp_f32 %0, %0\n v_cvt_pk_bf16_f32 %2, %4, %5\n v_exp_f32 %1, %1\n v_cvt_pk_bf16_f32 %3, %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
    } else if (MODE == 7) {       // exp + fma interleaved
      REP16(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %2, %2, %4, %5\n v_exp_f32 %1, %1\n v_fma_f32 %3, %3, %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
    } else if (MODE == 8) {       // v_ldexp_f32 (scale by 2^n: the cheap half of a split exponential)
      REP16(asm volatile("v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %5\n v_ldexp_f32 %2, %2, %6\n v_ldexp_f32 %3, %3, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
    } else if (MODE == 10) {      // v_perm_b32: the upper halves of two fp32 words = a truncating bf16 pack
      REP16(asm volatile("v_perm_b32 %0, %4, %5, %8\n v_perm_b32 %1, %5, %6, %8\n v_perm_b32 %2, %6, %7, %8\n v_perm_b32 %3, %7, %4, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(0x07060302u));)
    } else if (MODE == 9) {       // v_pk_fma_f32
      double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5;
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %5, %4\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %5, %4"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d4), "v"(d5));)
      a0 += (float)d0; a1 += (float)d1; a2 += (float)d2; a3 += (float)d3;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int MODE> void run(const char* name) {
  float* d; (void)hipMalloc(&d, 256 * 4096 * 4);
  int iters = 1000;
  k<MODE><<<256 * 4, 256>>>(d, 10);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); k<MODE><<<256 * 4, 256>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // 1024 workgroups of 4 waves on 256 CUs = 4 workgroups per CU = 4 waves per SIMD; each wave issues iters * 64 instructions
  double per_simd_instr = 4.0 * iters * 64;
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-44s %.3f ms: %.2f SIMD-cycles@2.4GHz per wave-instruction\n", name, ms, cyc / per_simd_instr);
  (void)hipFree(d);
}
int main() {
  run<0>("v_exp_f32");
  run<1>("v_exp_f16");
  run<2>("v_cvt_pk_bf16_f32");
  run<3>("v_cvt_pkrtz_f16_f32");
  run<4>("v_pk_mul_f16");
  run<5>("v_fma_f32");
  run<6>("v_exp_f32 + v_cvt_pk_bf16_f32 (1:1)");
  run<7>("v_exp_f32 + v_fma_f32 (1:1)");
  run<8>("v_ldexp_f32");
  run<9>("v_pk_fma_f32");
  run<10>("v_perm_b32");
  return 0;
}
