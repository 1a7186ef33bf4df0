// Shared by the nn.Linear GEMM kernels (linear_gemm.hip: register- / DMA-staged forms; linear_p8.hip: the 256 x 256 eight-phase form).
#pragma once
#include "pd_common.h"

namespace pd {

struct LinP {
  long long M;
  int K, N, N_pad, x_stride;
  int t_tiles, c_tiles;
  unsigned xbytes;
  const void* x; const void* w; const float* bias; const void* residual; void* y;
  const float* scale; const float* shift;   // optional GroupNorm affine on x: [sample][K], sample = row / rows_per_sample
  int rows_per_sample;                      // multiple of 128 when scale is set (a token tile never straddles samples)
  int qkv_heads, B;                         // > 0: head-major q / k / v output [3][B][heads][rows_per_sample][8] (N = 3 * heads * 8)
  float* stats;                             // optional [B][rows_per_sample / 128][N][2]: per-tile channel (sum, sum of squares) of y
  float* kmax2;                             // optional (qkv_heads > 0, 16-bit) [B][heads]: atomic max of |k row|^2 as stored
  long long w_sstride;                      // linear_dma_kernel: bytes between the packed weights of consecutive samples (0: one set for all)
  int bias_sstride;                         // ... floats between their bias vectors
};

// gelu(g) = g/2 (1 + erf(g / sqrt 2)), F.gelu's default (exact) form.  The bf16 engine takes erf from Abramowitz-Stegun 7.1.26
// (|error| < 1.5e-7, far below bf16's 2^-9; one v_exp + one v_rcp + 7 FMAs), the fp32 parity mode calls erff.
template <int ES> __device__ __forceinline__ float gelu_f(float g) {
  if constexpr (ES == 4) return 0.5f * g * (1.0f + erff(g * 0.7071067811865476f));
  const float x = fabsf(g) * 0.7071067811865476f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-x * x * 1.4426950408889634f);     // erf(|g| / sqrt 2)
  return 0.5f * g + 0.5f * fabsf(g) * e;                                                     // g/2 (1 + sign(g) erf) 
}

// linear_p8.hip: the 256-token x 256-channel eight-phase kernel (plain / residual / fused-GEGLU layers of the 16-bit engines).
// `p.t_tiles` / `p.c_tiles` are in units of 256; returns PD_OK or PD_ERR_LAUNCH.
int launch_linear_p8(const LinP& p, int dtype, bool glu, hipStream_t st);

}  // namespace pd
