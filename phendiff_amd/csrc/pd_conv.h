// Launch parameters of the implicit-GEMM convolution kernel (conv_igemm.hip).
#pragma once
#include "pd_common.h"

namespace pd {

struct ConvP {
  int B, Hin, Win, Hout, Wout;
  int C0, C1, Cout, Cout_pad;
  int pad, upsample, silu, out_mode, heads;
  int tiles_x, tiles_y, tiles_x_shift, n_co_tiles, nchunks;
  unsigned bytes0, bytes1;
  const void* x0; const void* x1;
  const float* scale; const float* shift;
  const void* w;
  const float* bias;
  const float* temb; int temb_stride;
  const void* residual;
  void* y;
  float* stats;        // [B][tiles][Cout][2] per-tile channel (sum, sumsq) of the stored output, or null
  // fused 1x1 "tail": extra K chunks over [t0 | t1] (centre tap only, no GroupNorm transform) appended after the main
  // chunks -- ResnetBlock2D.conv_shortcut folded into conv2
  int n_main, n_tail, Ct0, Ct1;
  unsigned tbytes0, tbytes1;
  const void* t0; const void* t1;
  int im2col3;         // source is NCHW fp32 with C0r <= 3..4 real channels: 3x3 taps gathered into 32 virtual channels
  int C0r;
  // sub-pixel phase of an upsampling convolution (pd_conv_args.phase): column padding separate from the row padding, output pixel
  // (out_step oy + out_oy, out_step ox + out_ox) of a tensor out_step times as large, statistic tiles at stat_tile_base of stat_tiles
  int pad_x, out_step, out_oy, out_ox, stat_tile_base, stat_tiles;
  // input-side phase (pd_conv_args.phase_in): source pixel (in_step iy + in_oy, in_step ix + in_ox) of a tensor in_step times as large
  int in_step, in_oy, in_ox;
  // round 6: 1 = walk the grid output-channel-tile-major with a contiguous run of that list per XCD (weights larger than the input
  // activations -- the 8^2 / 16^2 levels of the latent-diffusion UNet: every XCD used to stream ALL weights through its L2)
  int co_major;
};

}  // namespace pd
