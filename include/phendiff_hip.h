/*
 * phendiff_hip.h -- C ABI of libphendiff_hip.so (MI355X / gfx950).
 *
 * The reference (thethomasboyer/PhenDiff) has no FFI of its own: its hot path is Python calling
 * diffusers-0.18.2 modules which dispatch to ATen/cuDNN kernels.  Each entry point below replaces
 * the device work of one such call site; the Python host (phendiff_amd/ *.py) mirrors the reference's
 * object protocol on top of this ABI and INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pd_<op>() returns 0 on success, a negative pd_status on failure; pd_last_error() gives
 *     the thread-local message.  No exceptions, no allocation, no host sync inside: all launches are
 *     asynchronous on `stream` (a hipStream_t passed as void*) and are hipGraph-capturable.
 *   - all pointers are caller-owned DEVICE pointers (the host takes them from live torch tensors).
 *   - activations are NHWC (channel-contiguous) in `dtype` (PD_F32 or PD_BF16); the model boundary
 *     tensors (UNet input sample / output prediction, scheduler state) are NCHW fp32 exactly as the
 *     reference's tensors are.
 *   - conv / linear weights are passed PRE-PACKED in the MFMA fragment order produced by
 *     phendiff_amd/packing.py (documented at pd_conv_args.w_packed).
 */
#ifndef PHENDIFF_HIP_H_
#define PHENDIFF_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an args struct grows or an entry point is added (a caller built against an older header passes shorter
 * structs): 1 = round 1; 2 = round 2's trailing fields (pd_gn_finalize_args.temb/temb_stride, pd_attn_args.kmax2,
 * pd_linear_args.kmax2_out), PD_F16 and pd_zero / pd_gn_apply / pd_attn_wide; 3 = round 3; 4 = round 4 (pd_comm_query, pd_linear_args.fold_ws /
 * fold_ws_bytes + pd_linear_fold_workspace,
 * pd_conv_args.phase); 5 = pd_conv_args.phase_in, pd_wgrad_args.phase; 6 = round 5: pd_attn_bwd_args.slab / slab_bytes +
 * pd_attn_d8_bwd_workspace (the one-pass backward); 7 = round 6: pd_gn_bwd_args.mod / mod_stride / dmod (scale_shift ResNet blocks train);
 * pd_resize_tf1, pd_conv_rect, pd_pool2d, pd_fc_f32 (the evaluation metrics' feature extractor); pd_pack_weight_args.dst2 / dst2_ct_stride;
 * 8 = pd_geglu_bwd_args.sums / sum_splits / B, pd_layernorm_bwd_args.dxsum (bias gradients without a pass over dY), pd_upsample_phase_weights,
 * pd_token_wgrad_args.stage / pd_wgrad_args.stage. */
#define PD_ABI_VERSION 8

typedef enum { PD_OK = 0, PD_ERR_ARG = -1, PD_ERR_SHAPE = -2, PD_ERR_LAUNCH = -3, PD_ERR_UNSUPPORTED = -4 } pd_status;
/* PD_F32: exact-fp32 MFMA (parity mode).  PD_BF16 / PD_F16: 16-bit storage + MFMA, fp32 accumulate / statistics / softmax.
 * PD_F16 is the reference's `--mixed_precision fp16` (args_parser.py:381-390): every inference entry point, and -- since round 5 -- the
 * pixel UNet's backward set (pd_conv_wgrad, pd_gn_silu_bwd, pd_attn_d8_bwd + pd_attn_d8's lse, pd_pool2x2_sum, pd_channel_sum, pd_im2col3,
 * pd_token_wgrad) and the latent-diffusion backward set (pd_attn_d64_bwd + pd_attn_d64's lse, pd_attn_wide_bwd, pd_layernorm_bwd,
 * pd_geglu_bwd, pd_token_embedding_grad) under a loss scale (pd_loss_args.grad_scale; phendiff_amd.training.LossScaler = accelerate's
 * GradScaler): UNetTrainer and SDUNetTrainer both train in fp16. */
typedef enum { PD_F32 = 0, PD_BF16 = 1, PD_F16 = 2 } pd_dtype;
typedef enum { PD_PRED_EPSILON = 0, PD_PRED_SAMPLE = 1, PD_PRED_V = 2 } pd_pred_type;

int pd_abi_version(void);
const char* pd_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * pd_temb: timestep + class embedding and every ResnetBlock2D's time_emb_proj in one launch.
 * Replaces cond_unet_2d.py:289-309 (Timesteps -> TimestepEmbedding -> +class embedding) and the
 * `time_emb_proj(SiLU(temb))` Linear of all ResnetBlock2D instances (diffusers resnet.py; reached via
 * cond_unet_2d.py:171,187,217).  One output row per (timestep, class) pair, so a whole sampling
 * trajectory (S steps x B images) is one call.
 *   emb[r]  = W2 . silu(W1 . sincos(t[r]) + b1) + b2 + (class_emb[r] | E[label[r]] | 0)
 *   proj[r] = Wp . silu(emb[r]) + bp            (Wp = all time_emb_proj stacked: [proj_dim][T])
 */
typedef struct {
  int rows;                 /* number of (timestep, class) rows */
  int c0;                   /* block_out_channels[0] (sinusoid width) */
  int tdim;                 /* time_embed_dim = 4*c0 */
  int proj_dim;             /* sum of all resnet out_channels */
  int flip_sin_to_cos;      /* cond_unet_2d.py:139 */
  float freq_shift;         /* downscale_freq_shift */
  int num_classes;          /* rows of class table (0: no class embedding) */
  const float* timesteps;   /* [rows] fp32 (integer-valued) */
  const int64_t* labels;    /* [rows] or NULL */
  const float* class_emb;   /* [rows][tdim] or NULL (zeros => unconditional, cond_unet_2d.py:306-309) */
  /* all Linear weights are passed TRANSPOSED ([in][out], i.e. weight.t().contiguous()) so thread `out` streams coalesced */
  const float* w1; const float* b1;   /* time_embedding.linear_1^T [c0][tdim], [tdim] */
  const float* w2; const float* b2;   /* time_embedding.linear_2^T [tdim][tdim], [tdim] */
  const float* class_table;           /* class_embedding.weight [num_classes][tdim] or NULL */
  const float* wp; const float* bp;   /* stacked time_emb_proj^T [tdim][proj_dim], [proj_dim] */
  float* emb;               /* out [rows][tdim] (may be NULL) */
  float* proj;              /* out [rows][proj_dim] */
  float* feat;              /* optional out [rows][c0]: the sinusoid features (kept for the backward), or NULL */
  float* z1;                /* optional out [rows][tdim]: linear_1's pre-activation (kept for the backward), or NULL */
} pd_temb_args;
int pd_temb(const pd_temb_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pd_conv_in: first 3x3 conv (cond_unet_2d.py:127-129,313): NCHW fp32 sample -> NHWC activations.
 */
typedef struct {
  int dtype;                /* output activation dtype */
  int B, H, W, Cin, Cout;   /* Cin <= 4, Cout % 32 == 0 */
  const float* x;           /* [B][Cin][H][W] fp32 */
  const float* w;           /* [Cout][Cin][3][3] fp32 (OIHW, unpacked) */
  const float* bias;        /* [Cout] */
  void* y;                  /* [B][H][W][Cout] */
} pd_conv_in_args;
int pd_conv_in(const pd_conv_in_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pd_gn_stats: GroupNorm statistics folded with the affine parameters into per-(sample, channel)
 * scale/shift, so that the consumer conv applies y = x*scale + shift (+SiLU) while staging its tile.
 * Replaces the statistics half of torch.nn.GroupNorm(32, C) as used by ResnetBlock2D.norm1/norm2,
 * Attention.group_norm and conv_norm_out (cond_unet_2d.py:175,195,221,236).  The input may be the
 * channel concatenation [x0 | x1] of two NHWC tensors (skip connections, cond_unet_2d.py:333-343),
 * (statistics are invariant under the nearest x2 upsample, so Upsample2D needs no flag here).
 * Two launches: partial sums (fp64 combine) and finalize.  Standalone form (one extra read of the tensor); the UNet
 * plan uses the fused form instead: pd_conv(stats_out) in the producer + pd_gn_finalize.
 */
typedef struct {
  int dtype;
  int B, HW;                /* pixels per sample */
  int C0, C1;               /* channels of source 0 / source 1 (C1 = 0: single source) */
  int groups;               /* 32 */
  float eps;
  const void* x0; const void* x1;
  const float* gamma; const float* beta;   /* [C0+C1] */
  double* partial;          /* workspace [B][splits][C0+C1][2] */
  int splits;               /* partial blocks per (sample, group); >= 1 */
  float* scale; float* shift;   /* out [B][C0+C1] */
} pd_gn_stats_args;
int pd_gn_stats(const pd_gn_stats_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pd_conv: implicit-GEMM convolution on MFMA (NHWC, LDS-staged halo tiles).
 * Replaces torch.nn.Conv2d 3x3 (ResnetBlock2D.conv1/conv2, Downsample2D.conv, Upsample2D.conv,
 * conv_out) and 1x1 / Linear (ResnetBlock2D.conv_shortcut, Attention.to_q/k/v/to_out) together with the
 * elementwise ops around them in diffusers' ResnetBlock2D.forward / AttnProcessor2_0:
 *   prologue : x <- silu?(x*scale + shift)    (GroupNorm apply + SiLU), zero padding AFTER the transform
 *              optional nearest x2 upsample of the source (Upsample2D: F.interpolate never materialised)
 *              optional channel concat of two sources (torch.cat([h, skip], 1) never materialised)
 *   epilogue : + bias[co] + temb[n][co] + residual[n][y][x][co]
 * w_packed layout (T = dtype element): [Cout/32][Cin/32][taps][2][64 lanes][8] where for lane l
 * (r = l & 31, h = l >> 5) element j is W[co = 32*ct + r][ci = 32*chunk + 16*s + 8*h + j][tap].
 */
typedef enum { PD_OUT_NHWC = 0, PD_OUT_NCHW_F32 = 1, PD_OUT_QKV_HEADS = 2 } pd_out_mode;
typedef struct {
  int dtype;
  int B, Hin, Win;          /* source spatial size (before upsample) */
  int Hout, Wout;           /* output spatial size (checked against the conv arithmetic) */
  int C0, C1;               /* source channels; Cin = C0 + C1, each % 32 == 0 */
  int Cout;                 /* logical output channels */
  int Cout_pad;             /* packed output channels, % 32 == 0 (>= Cout) */
  int ksize;                /* 1 or 3 */
  int stride;               /* 1 or 2 (3x3 only) */
  int pad;                  /* 1 for 3x3 pad 1; 0 for 1x1 or the asymmetric (0,1,0,1) downsample */
  int upsample;             /* 1: nearest x2 before the conv (Upsample2D); 2: zero-stuffed x2, samples at the even positions (input
                               gradient of a stride-2 pad-1 conv); 3: zero-stuffed x2, samples at the odd positions (input gradient of
                               Downsample2D(padding=0): pad (0,1,0,1) then an unpadded stride-2 conv) */
  int silu;                 /* 1: SiLU after the affine */
  int out_mode;             /* pd_out_mode */
  int heads;                /* PD_OUT_QKV_HEADS: number of heads (Cout = 3*heads*8) */
  const void* x0; const void* x1;
  const float* scale; const float* shift;   /* [B][Cin] or NULL (no GroupNorm) */
  const void* w_packed;
  const float* bias;        /* [Cout_pad] */
  const float* temb; int temb_stride;       /* temb[n*temb_stride + co] or NULL */
  const void* residual;     /* NHWC [B][Hout][Wout][Cout] or NULL */
  void* y;
  float* stats_out;         /* NULL, or [B][pd_conv_stat_tiles()][Cout][2]: per-tile per-channel (sum, sum of squares) of the
                               stored output -- the GroupNorm statistics of the CONSUMER, produced for free here */
  /* Fused 1x1 "tail" (ResnetBlock2D.conv_shortcut folded into conv2): y += Wt . [tail_x0 | tail_x1] on the same pixels
     (no GroupNorm / SiLU on the tail).  3x3 stride-1 convs only; the tail's weight fragments follow the main ones in
     w_packed: per 32-co tile [main: (C0+C1)/32 * 9 * 2 fragments][tail: (tail_C0+tail_C1)/32 * 2 fragments]; `bias` is
     the sum of both biases.  tail_x0 = NULL: no tail. */
  const void* tail_x0; const void* tail_x1; int tail_C0, tail_C1;
  int im2col3;              /* 0, or n <= 3: x0 is an NCHW fp32 tensor with n channels (the UNet input sample) and the op is
                               the 3x3 pad-1 conv_in run as a 1x1 conv over 32 virtual channels k = ci*9 + ky*3 + kx
                               (ksize must be 1, C0 = 32, weights packed accordingly) */
  int phase;                /* (ABI 4) 0: ordinary convolution.  1 + 2 a + b (a, b = 0 or 1): one PHASE of the sub-pixel form of Upsample2D
                               (F.interpolate(x, 2.0, "nearest") then conv 3x3 pad 1, diffusers resnet.py; cond_unet_2d.py:200-228): output
                               pixel (2 oy + a, 2 ox + b) of the y tensor [B][2 Hout][2 Wout][Cout] = sum over dy, dx = 0, 1 of
                               W_ab[dy][dx] . x[oy - (1 - a) + dy][ox - (1 - b) + dx] (zero outside the image), W_ab = the 3x3 taps that
                               fall on the same source pixel summed (4 of the 9 tap positions per output pixel: 4 / 9 of the FLOPs).
                               Requires ksize = 2, stride 1, no upsample / GroupNorm / tail / residual, NHWC output, Hout = Hin,
                               Wout = Win; `pad` is ignored.  stats_out is then [B][4 T][Cout][2], T = pd_conv_stat_tiles(Hout, Wout, 2, 1):
                               phase p's tiles fill slots (p - 1) T .. p T - 1 */
  int phase_in;             /* (ABI 5) with phase = 1 + 2 a + b: the phase selects INPUT pixels instead of output pixels -- the input
                               gradient of a sub-pixel phase: x0 is [B][2 Hin][2 Win][C0] (the gradient of the upsampled output), the
                               launch reads its pixels (2 iy + a, 2 ix + b) and writes the dense y [B][Hout][Wout][Cout], Hout = Hin,
                               Wout = Win: y[oy][ox] (+= residual) = sum over dy, dx = 0, 1 of W[dy][dx] . x0[2 (oy - a + dy) + a][2 (ox - b + dx) + b]
                               (W = the phase's 2x2 weights packed as input-gradient weights: transposed, taps flipped).  residual is
                               allowed (the four phases accumulate into one gradient tensor); no stats_out */
} pd_conv_args;
int pd_conv(const pd_conv_args* a, void* stream);
/* number of statistic tiles per sample pd_conv writes for this shape (depends on the kernel's tile choice) */
int pd_conv_stat_tiles(int Hout, int Wout, int ksize, int stride);

/* pd_gn_finalize: per-tile channel sums (pd_conv stats_out) of one or two tensors (channel concat [x0 | x1]) ->
 * GroupNorm scale/shift per (sample, channel):  scale = rstd*gamma, shift = beta - mean*rstd*gamma  (fp64 combine). */
typedef struct {
  int B, HW, groups; float eps;
  int C0, T0; const float* stats0;     /* [B][T0][C0][2] */
  int C1, T1; const float* stats1;     /* [B][T1][C1][2] or NULL (C1 = 0) */
  const float* gamma; const float* beta;
  float* scale; float* shift;          /* out [B][C0+C1] */
  float* mean; float* rstd;            /* optional out [B][groups] (kept for the backward, pd_gn_silu_bwd), or NULL */
  const float* temb; int temb_stride;  /* optional: ResnetBlock2D time_embedding_norm = "scale_shift" (cond_unet_2d.py:103,180):
                                          row n = temb + n*temb_stride holds [scale | shift] (2*(C0+C1) floats, this resnet's
                                          time_emb_proj(SiLU(emb))); the written affine becomes GN(x)*(1 + scale) + shift */
} pd_gn_finalize_args;
int pd_gn_finalize(const pd_gn_finalize_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pd_attn_d8: softmax(q k^T / sqrt(8)) v for head_dim 8 (attention_head_dim=8, cond_unet_2d.py:176-178),
 * streaming (flash-style) softmax, scores never materialised.  Replaces F.scaled_dot_product_attention in
 * AttnProcessor2_0.  Inputs come head-major from pd_conv(out_mode = PD_OUT_QKV_HEADS):
 *   q, k : [B][heads][N][8]     v : [B][heads][N][8]     out: NHWC [B][N][heads*8]
 */
typedef struct {
  int dtype;
  int B, heads, N;
  const void* q; const void* k; const void* v;
  void* out;
  float* lse;   /* optional out [B][heads][N]: log2-domain log-sum-exp of the scaled scores (kept for pd_attn_d8_bwd), or NULL */
  const float* kmax2;   /* optional [B][heads]: an upper bound of max_n |k[b][head][n]|^2 over the keys as stored (pd_linear
                           kmax2_out produces it in the q/k/v projection's epilogue), or NULL.  With it (bf16 / fp16, large
                           launches) the kernel needs no per-tile key norms: K / V tiles go global -> LDS by DMA
                           (global_load_lds), V^T fragments come from transposed LDS reads.  Results are the same function
                           of the inputs either way (softmax is invariant to the running reference maximum) PROVIDED the
                           bound holds: kmax2[b][head] >= max_n |k|^2 of THIS launch's keys is a hard precondition (a slot
                           that is too small lets p = 2^(s - m) grow without the rescale: inf / NaN in fp16).  A first
                           32-key score above the implied bound makes the wave fall back to the exact running-maximum
                           path -- a safety net for zero / stale slots, not a substitute for the precondition. */
} pd_attn_args;
int pd_attn_d8(const pd_attn_args* a, void* stream);

/* pd_attn_d8_bwd: gradient of pd_attn_d8 (autograd of F.scaled_dot_product_attention), P recomputed from lse.
 *   q, k, v: as the forward;  o: the forward's output, dout: gradient w.r.t. it (both NHWC [B][N][heads*8]);
 *   delta: workspace [B][heads][N];  dqkv: out NHWC [B][N][3*heads*8] = [dq | dk | dv] (channel = which*C + head*8 + d),
 *   the output-gradient layout of the fused q/k/v projection. */
typedef struct {
  int dtype;
  int B, heads, N;
  const void* q; const void* k; const void* v;
  const void* o; const void* dout;
  const float* lse; float* delta;
  void* dqkv;
  float* slab;              /* (ABI 6) NULL, or a device workspace of >= pd_attn_d8_bwd_workspace(a) bytes (when that is > 0): the backward then runs
                               in ONE pass -- P and dS formed once per tile, dK / dV lane-local, dS transposed through LDS for dQ, whose
                               partial sums per 512-key block land here [key blocks][B][heads][N][8] fp32 and are added in order by a
                               small reduce launch (bit-identical run to run).  16-bit engines, N >= 512; otherwise the two-kernel path */
  size_t slab_bytes;
} pd_attn_bwd_args;
int pd_attn_d8_bwd(const pd_attn_bwd_args* a, void* stream);
/* bytes of pd_attn_bwd_args.slab the one-pass backward needs for this shape; 0: not applicable (fp32 engine, N < 512) */
size_t pd_attn_d8_bwd_workspace(const pd_attn_bwd_args* a);

/* ------------------------------------------------------------------------------------------------
 * pd_ddim_step: one fused DDIM / inverse-DDIM update on NCHW fp32 tensors.
 * Replaces DDIMScheduler.step (pipeline_conditionial_ddim.py:340-347) and DDIMInverseScheduler.step
 * (utils_Img2Img.py:794-798): both are  x' = sqrt_ap * clamp(x0) + dir_coef * eps  with
 *   epsilon: x0 = (x - sqrt_b*out)/sqrt_a, eps = out        sample: x0 = out, eps = (x - sqrt_a*x0)/sqrt_b
 *   v:       x0 = sqrt_a*x - sqrt_b*out,   eps = sqrt_a*out + sqrt_b*x
 * The host passes the four coefficients (computed in fp32 exactly as the reference's 0-dim tensors).
 * Optional classifier-free-guidance combine (pipeline_conditionial_ddim.py:324-328):
 *   out = uncond + w*(out - uncond)  [imagen]   or   out + w*(out - uncond)  [CFG]
 */
typedef struct {
  int64_t numel;            /* B*C*H*W */
  int64_t per_sample;       /* C*H*W (for per-sample guidance weights) */
  int pred_type;            /* pd_pred_type */
  int clip;                 /* clip_sample */
  float clip_range;
  int use_clipped_model_output;
  float sqrt_a, sqrt_b;     /* sqrt(alpha_prod_t), sqrt(1 - alpha_prod_t) */
  float sqrt_ap, dir_coef;  /* sqrt(alpha_prod_t_prev), sqrt(1 - alpha_prod_t_prev - sigma^2) */
  const float* sample;      /* x_t */
  const float* model_out;   /* conditional prediction */
  const float* uncond_out;  /* NULL: no guidance */
  const float* w;           /* guidance weights: [B] or 1 value */
  int w_per_sample;         /* 1: w has one value per sample */
  int guidance_cfg;         /* 0: imagen eqn, 1: CFG eqn */
  float* prev_sample;       /* out (may alias sample) */
  float* pred_x0;           /* out or NULL */
} pd_ddim_step_args;
int pd_ddim_step(const pd_ddim_step_args* a, void* stream);

/* pd_add_noise: sa[n]*x + sb[n]*noise (DDIMScheduler.add_noise) or velocity sa*noise - sb*x. */
typedef struct {
  int64_t numel, per_sample;
  int velocity;
  const float* x; const float* noise;
  const float* sa; const float* sb;   /* [B] per-sample coefficients (device) */
  float* out;
} pd_add_noise_args;
int pd_add_noise(const pd_add_noise_args* a, void* stream);

/* pd_zero: hipMemsetAsync(ptr, 0, bytes) on the stream -- a launch plan's way to reset what later launches max / add into
 * (pd_linear kmax2_out); captured into a hipGraph like every other entry point. */
typedef struct { void* ptr; size_t bytes; } pd_zero_args;
int pd_zero(const pd_zero_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pd_postproc: (x/2 + 0.5).clamp(0,1), NCHW -> NHWC (pipeline_conditionial_ddim.py:349-350), optionally
 * also the uint8 quantisation round(255*x) of DiffusionPipeline.numpy_to_pil.
 */
typedef struct {
  int B, C, H, W;
  const float* x;           /* NCHW */
  float* out_f32;           /* NHWC or NULL */
  uint8_t* out_u8;          /* NHWC or NULL */
} pd_postproc_args;
int pd_postproc(const pd_postproc_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * UNet backward building blocks (autograd of the forward above, utils_training.py:436).
 * Input gradients of convolutions reuse pd_conv with W'[ci][co][ky][kx] = W[co][ci][K-1-ky][K-1-kx] (stride-2 convs: upsample = 2,
 * zero-stuffed dY; the conv fused with Upsample2D: pd_conv at 2x resolution followed by pd_pool2x2_sum).
 *
 * pd_gn_silu_bwd: gradient through z = silu?(GroupNorm(x)) of [x0 | x1]:  dz -> dx (per source, = or +=), dgamma += , dbeta += .
 *   dy = dz * silu'(gamma*xhat + beta);  dx = rstd * (gamma*dy - mean_g(gamma*dy) - xhat * mean_g(gamma*dy*xhat))
 */
typedef struct {
  int dtype;
  int B, HW, C0, C1, groups, silu;
  const void* x0; const void* x1;        /* forward inputs of the norm (NHWC) */
  const void* dz0; const void* dz1;      /* gradient w.r.t. the normalised (+SiLU) tensor, same split */
  const float* mean; const float* rstd;  /* [B][groups] from pd_gn_finalize */
  const float* gamma; const float* beta; /* [C0+C1] */
  double* partial; int splits;           /* workspace [B][splits][C0+C1][2] */
  float* coef;                           /* workspace [B][groups][2] */
  void* dx0; void* dx1;                  /* outputs (either may be NULL) */
  int accumulate0, accumulate1;          /* 1: dx += */
  float* dgamma; float* dbeta;           /* [C0+C1], accumulated (+=), or NULL */
  int dz_combined;                       /* 1: dz0 holds all C0+C1 channels (stride C0+C1), dz1 must be NULL */
  const void* res;                       /* optional [B][HW][C0+C1]: added to dx (gradient of the skip / shortcut around the block) */
  float* sum0; float* sum1;              /* optional out [B][splits][C0] / [B][splits][C1]: per-channel sums of the dx0 / dx1 values
                                            this call stores (feeds pd_channel_sum with x = NULL: the producer's bias /
                                            time-embedding gradients without another pass over dx) */
  /* (ABI 7) resnet_time_scale_shift = "scale_shift" (cond_unet_2d.py:180,191,225; diffusers ResnetBlock2D: h = norm2(h) * (1 + scale)
   * + shift with [scale | shift] = time_emb_proj(silu(temb))): the norm's affine is modulated per sample,
   *   gamma_n = gamma (1 + scale_n),  beta_n = beta (1 + scale_n) + shift_n,
   * mod = the forward's projection rows, row n at mod + n * mod_stride holding [scale (C) | shift (C)] (C1 must be 0).  dgamma / dbeta
   * then accumulate sum_n (1 + scale_n) x the per-sample sums, and dmod (same row layout, written "=") receives
   *   d scale_n = gamma * sum(dy xhat) + beta * sum(dy),   d shift_n = sum(dy). */
  const float* mod; int mod_stride; float* dmod;
} pd_gn_bwd_args;
int pd_gn_silu_bwd(const pd_gn_bwd_args* a, void* stream);

/* pd_pool2x2_sum: dx[n][y][x][c] (+)= sum of du[n][2y..2y+1][2x..2x+1][c]  (gradient of F.interpolate(scale_factor=2, "nearest")) */
typedef struct { int dtype; int B, H, W, C; const void* du; void* dx; int accumulate; } pd_pool2x2_args;
int pd_pool2x2_sum(const pd_pool2x2_args* a, void* stream);

/* pd_channel_sum: out[n*out_stride + c] (+)= sum over pixels of x[n][p][c]  (bias gradients after a sum over n; d temb_proj) */
typedef struct {
  int dtype; int B, HW, C; const void* x; float* out; int out_stride; int accumulate;
  float* total; int total_valid;   /* optional [total_valid <= C]: total[c] += sum over samples of this call's per-sample sums */
  float* workspace; int splits;    /* optional [B][splits][C] scratch: the pixels of a sample are split over `splits` workgroups;
                                      with x = NULL it already holds the per-split sums (pd_gn_silu_bwd sum0 / sum1) */
} pd_channel_sum_args;
int pd_channel_sum(const pd_channel_sum_args* a, void* stream);

/* pd_nchw_to_nhwc: out[n][p][c] = c < C ? x[n][c][p] : 0 for c < Cpad (the loss gradient w.r.t. the UNet output, NCHW fp32,
 * as conv_out's NHWC output gradient). */
typedef struct { int dtype; int B, C, HW, Cpad; const float* x; void* out; } pd_nchw_to_nhwc_args;
int pd_nchw_to_nhwc(const pd_nchw_to_nhwc_args* a, void* stream);

/* Backward of the fp32 Linear layers of the time-embedding path (TimestepEmbedding, class embedding, time_emb_proj):
 *   pd_linear_wgrad:  dw[o][i] += sum_r dy[r][o] * act(x[r][i]);  db[o] += sum_r dy[r][o]      (act = SiLU when x_silu)
 *   pd_linear_dgrad:  dx[r][i]  = (sum_o dy[r][o] * w[o][i]) * (pre ? silu'(pre[r][i]) : 1)       (w: nn.Linear layout [out][in])
 *   pd_embedding_grad: dtable[k][i] += sum over rows with labels[r] == k of d[r][i]                (deterministic order)  */
typedef struct { int rows, in_dim, out_dim, x_silu; const float* dy; const float* x; float* dw; float* db; } pd_linear_wgrad_args;
int pd_linear_wgrad(const pd_linear_wgrad_args* a, void* stream);
typedef struct { int rows, in_dim, out_dim; const float* dy; const float* w; const float* pre; float* dx; } pd_linear_dgrad_args;
int pd_linear_dgrad(const pd_linear_dgrad_args* a, void* stream);
typedef struct { int rows, dim, num_classes; const int64_t* labels; const float* d; float* dtable; } pd_embedding_grad_args;
int pd_embedding_grad(const pd_embedding_grad_args* a, void* stream);

/* pd_conv_wgrad: weight gradient of a convolution pd_conv ran in the forward:
 *   dw[co][ci][ky][kx] (+)= sum_{n,oy,ox} dy[n][oy][ox][co] * Z[n][oy*stride+ky-pad][ox*stride+kx-pad][ci],
 * Z = silu?(scale*[x0|x1] + shift) (optionally nearest-x2 upsampled) rebuilt on the fly exactly as pd_conv's staging does.
 * B/Hin/.../pad/upsample/silu/x0/x1/scale/shift have pd_conv's meaning (upsample: 0 or 1).  dy is NHWC with channel stride
 * Cout (multiple of 8); dw is the fp32 OIHW gradient [Cout_valid][Cin_valid][k][k] (Cout_valid/Cin_valid = 0: all channels).
 * slab: workspace of slab_bytes >= pd_conv_wgrad_workspace(a) (smaller is accepted down to one split, at lower occupancy);
 * the partial sums are reduced in a fixed order: results are bitwise reproducible. */
typedef struct {
  int dtype;
  int B, Hin, Win, Hout, Wout;
  int C0, C1, Cout;
  int ksize, stride, pad, upsample, silu;
  const void* x0; const void* x1;
  const float* scale; const float* shift;
  const void* dy;
  float* slab; size_t slab_bytes;
  float* dw; int Cout_valid, Cin_valid; int accumulate;
  int phase;                /* (ABI 5) 0: ordinary.  1 + 2 a + b: the weight gradient THROUGH sub-pixel phase (a, b) of an upsampler's 3x3 convolution
                               (pd_conv_args.phase): ksize = 2, x0 = the LOW-resolution input [B][Hin][Win][C0], dy = the gradient of the upsampled
                               output [B][2 Hout][2 Wout][Cout] (Hout = Hin, Wout = Win) read at its pixels (2 oy + a, 2 ox + b), dw = the 3x3
                               gradient [Cout][C0][3][3]: the phase kernel's tap gradients are added to the 3x3 taps they are sums of (phase 1
                               honours `accumulate`, phases 2-4 always add: run the four in order).  4 / 9 of the FLOPs of upsample = 1. */
  int stage;                /* (ABI 8) 0: both launches; 1: the GEMM only; 2: the fold of the slab into dw only (as pd_token_wgrad_args.stage) */
} pd_wgrad_args;
size_t pd_conv_wgrad_workspace(const pd_wgrad_args* a);
int pd_conv_wgrad(const pd_wgrad_args* a, void* stream);

/* pd_pack_weight: fp32 master weights (nn.Conv2d OIHW / nn.Linear [out][in]) -> pd_conv's packed MFMA fragment order
 * (phendiff_amd/packing.py: pack_conv_weight), on the device, after each optimizer step of a training run.
 *   dgrad = 0: packed[co][ci][tap] = src[co][ci][tap]                   (src rows of src_in input channels)
 *   dgrad = 1: packed[co][ci][tap] = src[ci][co][taps-1-tap]            (input-gradient weights, packing.dgrad_weight)
 * cout/cin: valid packed rows / columns (zero beyond, up to cout_pad / cin_pad); dst_ct_stride: elements between consecutive
 * 32-row tiles of dst (> the tile size when another conv's fragments follow each tile: the fused conv_shortcut). */
typedef struct {
  int dtype;
  int cout, cin, cout_pad, cin_pad, ksize, src_in, dgrad;
  const float* src; void* dst; long long dst_ct_stride;
  /* (ABI 7) optional, dgrad = 0 only: the SAME source blocks also written as the input-gradient packing -- what a second call with
   * dgrad = 1, cout / cin (and their pads) swapped and dst = dst2 would write -- from one read of the fp32 master weights
   * (an optimizer step re-packs every weight both ways: half the re-pack's HBM reads).  dst2_ct_stride: elements between
   * consecutive 32-row (input-channel) tiles of dst2, >= (cout_pad/32) * ksize^2 * 1024. */
  void* dst2; long long dst2_ct_stride;
} pd_pack_weight_args;
int pd_pack_weight(const pd_pack_weight_args* a, void* stream);

/* pd_pack_weight_batch: n pd_pack_weight jobs of ONE dtype as a single launch (an optimizer step re-packs ~130 weights of the
 * pixel UNet, ~1400 of the SD UNet: 13 us of launch latency each).  jobs: DEVICE array of n descriptors (validated by the caller
 * as pd_pack_weight would); starts: DEVICE int[n + 1], starts[j] = sum over i < j of (cout_pad/32)*(cin_pad/32), starts[n] =
 * total_blocks; max_ksize: largest ksize among the jobs (sizes the LDS tile). */
typedef struct {
  int dtype; int n;
  const pd_pack_weight_args* jobs; const int* starts;
  int total_blocks; int max_ksize;
} pd_pack_weight_batch_args;
int pd_pack_weight_batch(const pd_pack_weight_batch_args* a, void* stream);

/* pd_upsample_phase_weights (ABI 8): the four 2x2 sub-pixel phase kernels of "nearest x2, then 3x3 pad-1 convolution" (diffusers Upsample2D,
 * cond_unet_2d.py via UpBlock2D; phendiff_amd.packing.upsample_phase_weights) from the OIHW fp32 master weight, stacked:
 * out[2 a + b][o][i][u][v] = sum_{y in Y_a(u), x in Y_b(v)} w[o][i][y][x],  Y_0 = ({0}, {1, 2}),  Y_1 = ({0, 1}, {2})  -- rows first, then columns,
 * in fp32.  The fine-tuning step refreshes them after every optimizer step (before pd_pack_weight_batch packs them). */
typedef struct { int cout, cin; const float* w; float* out; } pd_upsample_phase_weights_args;
int pd_upsample_phase_weights(const pd_upsample_phase_weights_args* a, void* stream);

/* pd_im2col3: out[n][y][x][ci*9+ky*3+kx] = x[n][ci][y+ky-1][x+kx-1] (zero padded; 27 of 32 channels used): the input of
 * conv_in seen as a 1x1 convolution, for its weight gradient (cond_unet_2d.py:127-129). x: NCHW fp32, C <= 3; out: NHWC dtype. */
typedef struct { int dtype; int B, H, W, C; const float* x; void* out; } pd_im2col3_args;
int pd_im2col3(const pd_im2col3_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training-step building blocks: the fused bandwidth-bound passes around the UNet
 * forward/backward, on flat fp32 buffers.
 *
 * pd_diffusion_loss (utils_training.py:415-433): loss = mean(w_n (out - target)^2) and d loss / d out, with
 *   epsilon: target = noise, w = 1      sample: target = clean, w_n = alpha_t/(1-alpha_t) (SNR weights, `weight`)
 *   v_prediction: target = sa_n*noise - sb_n*clean  (DDIMScheduler.get_velocity)
 * Deterministic two-stage reduction (fp64 partials).
 */
typedef struct {
  int64_t numel, per_sample;
  int pred_type;
  const float* model_out; const float* noise; const float* clean;
  const float* weight;            /* [B] SNR weights (sample prediction) or NULL */
  const float* sa; const float* sb;   /* [B] sqrt(alpha_bar_t), sqrt(1-alpha_bar_t) (v_prediction) or NULL */
  float grad_scale;               /* multiplies d loss / d out (1.0; loss scaling for fp16) */
  float* grad_out;                /* [numel] or NULL */
  double* partial;                /* workspace [1024] */
  float* loss_out;                /* [1] */
} pd_loss_args;
int pd_diffusion_loss(const pd_loss_args* a, void* stream);

/* pd_grad_norm (utils_training.py:438-440, torch.nn.utils.clip_grad_norm_): global L2 norm of a flat gradient buffer and the
 * clip coefficient min(1, max_norm / (norm + 1e-6)), both left on the device (no host sync). partial: workspace [1024]. */
int pd_grad_norm(const float* grad, int64_t numel, double* partial, float max_norm, float* norm_out, float* clip_coef_out, void* stream);

/* pd_adamw_ema (utils_training.py:452-454 optimizer.step/zero_grad, :553-556 EMAModel.step): one pass over flat fp32 buffers:
 *   g *= clip_coef;  torch.optim.AdamW single-tensor update;  ema -= one_minus_decay * (ema - param);  optional grad zeroing.
 * Host passes step_size = lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t), one_minus_decay = 1 - EMA decay_t. */
typedef struct {
  int64_t numel;
  float lr, beta1, beta2, eps, weight_decay, step_size, bias_correction2_sqrt, one_minus_decay;
  int zero_grad;
  const float* clip_coef;         /* device scalar from pd_grad_norm, or NULL */
  float* param; float* grad; float* exp_avg; float* exp_avg_sq;
  float* ema;                     /* EMA shadow parameters or NULL */
  int ema_only;                   /* 1: no AdamW update (torch skips parameters whose .grad is None, e.g. the CustomEmbedding on an
                                     unconditional step, utils_training.py:465-471); EMA and grad zeroing still run */
} pd_adamw_ema_args;
int pd_adamw_ema(const pd_adamw_ema_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stable-Diffusion tier (diffusers UNet2DConditionModel as driven by custom_pipeline_stable_diffusion_img2img.py:680-686 and
 * _SD_prediction_wrapper, utils_training.py:459-496).  ResnetBlock2D / GroupNorm / every nn.Linear (as a 1x1 conv over NHWC
 * tokens) / sampling convs reuse pd_conv; the blocks below are what Transformer2DModel adds.
 *
 * pd_attn_d64: softmax(q k^T / sqrt(64)) v, head_dim 64 (BasicTransformerBlock.attn1: self attention; .attn2: cross
 *   attention over the 77 encoder_hidden_states tokens).  Token-major operands with explicit strides, so q/k/v may be
 *   slices of one fused projection output:  q[(b*Nq+i)*q_stride + head*64 + d],  k|v[(b*Nkv+j)*kv_stride + head*64 + d],
 *   out[(b*Nq+i)*out_stride + head*64 + d].
 */
typedef struct {
  int dtype;
  int B, heads, Nq, Nkv;
  const void* q; int q_stride;
  const void* k; const void* v; int kv_stride;
  void* out; int out_stride;
  float* lse;   /* optional out [B][heads][Nq]: log2-domain log-sum-exp of the scaled scores (kept for pd_attn_d64_bwd), or NULL */
} pd_attn_d64_args;
int pd_attn_d64(const pd_attn_d64_args* a, void* stream);

/* pd_token_wgrad: dw[n][k] (+)= sum_m dy[m][n] * x[m][k] -- weight gradient of nn.Linear over M tokens (the Linear layers of
 * BasicTransformerBlock trained through utils_training.py:436), MFMA GEMM with the reduction over tokens; dw is the fp32
 * [N][K] gradient of the nn.Linear weight.  slab: workspace of slab_bytes >= pd_token_wgrad_workspace(a) (smaller is accepted
 * down to one split); partial sums are reduced in a fixed order (bitwise reproducible). */
typedef struct {
  int dtype;
  long long M; int K, N;
  const void* x; int x_stride;
  const void* dy; int dy_stride;
  float* dw; int accumulate;
  float* slab; size_t slab_bytes;
  int stage;                /* (ABI 8) 0: both launches.  1: the GEMM only (partial tiles -> slab).  2: the ordered fold of the slab into dw only -- a caller
                               with one slab per call may run the folds (bandwidth-bound, a few CUs) on a second stream under the next layers' GEMMs */
} pd_token_wgrad_args;
int pd_token_wgrad(const pd_token_wgrad_args* a, void* stream);
size_t pd_token_wgrad_workspace(const pd_token_wgrad_args* a);

/* pd_gn_apply: y = silu?(GroupNorm([x0 | x1])) materialised once (scale / shift from pd_gn_finalize): the GroupNorm-apply +
 * SiLU of ResnetBlock2D.norm1/norm2 (diffusers resnet.py; cond_unet_2d.py:171,187,217 and the SD UNet / VAE blocks) for layers
 * whose convolution has many 64-channel output tiles, where pd_conv's in-flight transform would be repeated per tile. */
typedef struct {
  int dtype; int B, HW, C0, C1; int silu;
  const void* x0; const void* x1;
  const float* scale; const float* shift;   /* [B][C0+C1] */
  void* y;                                  /* NHWC [B][HW][C0+C1] */
} pd_gn_apply_args;
int pd_gn_apply(const pd_gn_apply_args* a, void* stream);

/* pd_linear: y[m][n] = sum_k x[m][k] * W[n][k] + bias[n] (+ residual[m][n]) -- nn.Linear over M tokens as a dedicated MFMA GEMM
 * (128 x 128 workgroup tiles, 64-channel K chunks).  Replaces the Linear layers of BasicTransformerBlock (attn1/attn2 to_q/k/v,
 * to_out.0, FeedForward; reached from custom_pipeline_stable_diffusion_img2img.py:680-686 and utils_training.py:486-494) and,
 * with the transposed weights (pd_pack_weight dgrad = 1), their input gradients.  w_packed: pd_conv's layout for a 1x1 kernel,
 * [N_pad/32][K/32][1][2][64 lanes][8].  x rows may be strided (x_stride elements, >= K); y / residual are dense [M][N]. */
typedef struct {
  int dtype;
  long long M;              /* rows (tokens) */
  int K;                    /* input features, % 32 == 0 */
  int N;                    /* output features, % 8 == 0 */
  int N_pad;                /* packed output features, % 32 == 0 */
  const void* x; int x_stride;
  const void* w_packed;
  const float* bias;        /* [N_pad] */
  const void* residual;     /* [M][N] or NULL */
  void* y;                  /* [M][N] */
  /* optional fusions for the q/k/v projection of diffusers' Attention (cond_unet_2d.py:176-178; AttnProcessor2_0):
     GroupNorm apply on x while staging, and the head-major layout pd_attn_d8 reads */
  const float* scale; const float* shift;   /* [M / rows_per_sample][K] or NULL: x <- x*scale + shift (pd_gn_finalize's output) */
  int rows_per_sample;      /* tokens per sample (multiple of 128 with scale; required with qkv_heads) */
  int qkv_heads;            /* 0: dense y; > 0: y = [3][B][heads][rows_per_sample][8] (N = 3*heads*8, as pd_conv PD_OUT_QKV_HEADS) */
  float* stats_out;         /* NULL, or [M / rows_per_sample][rows_per_sample / 128][N][2]: per-128-token-tile channel (sum, sum of
                               squares) of the stored y -- the consumer's GroupNorm statistics, folded by pd_gn_finalize (T = rows_per_sample/128) */
  int glu;                  /* 1: fused GEGLU (diffusers FeedForward.net[0] = GEGLU: proj -> chunk(2) -> value * gelu(gate)):
                               y = [M][N/2]; w_packed holds the projection's value rows (0 .. N/2) in the even and its gate rows
                               (N/2 .. N) in the odd 32-channel tiles (two pd_pack_weight calls with dst_ct_stride = two tiles);
                               bias stays in module order [N].  N % 64 == 0, no residual / statistics / head-major output */
  float* kmax2_out;         /* NULL, or (qkv_heads > 0, bf16 / fp16) [B][heads] fp32, ZEROED by the caller before the launch:
                               atomically maxed with |k[b][head][n]|^2 of every stored key row -- pd_attn_d8's kmax2 */
  void* fold_ws;            /* (ABI 4) NULL, or a device workspace of >= pd_linear_fold_workspace(a) bytes: the q/k/v projection behind a
                               GroupNorm (scale / shift + qkv_heads) then runs as the DMA-staged GEMM over per-sample weights
                               W diag(scale_n) and biases b + W shift_n written there by a small fold launch (same stream) */
  size_t fold_ws_bytes;
} pd_linear_args;
int pd_linear(const pd_linear_args* a, void* stream);
/* bytes of `fold_ws` that make pd_linear take the folded route for these arguments, 0 when the route does not apply */
size_t pd_linear_fold_workspace(const pd_linear_args* a);

/* pd_attn_wide: softmax(q k^T * scale) v with ONE wide head per D channels, D in {128, 256, 512} -- the mid-block attention of
 * the SD VAE (diffusers AutoencoderKL: Encoder/Decoder.mid_block.attentions[0], a single head over all 512 channels;
 * vae.encode / vae.decode at custom_pipeline_stable_diffusion_img2img.py:431,709-711).  Operand addressing as pd_attn_d64
 * (token-major with strides; channel = head*D + d).  scale = D^-1/2 for diffusers' Attention. */
typedef struct {
  int dtype;
  int B, heads, D, Nq, Nkv;
  float scale;
  const void* q; int q_stride;
  const void* k; const void* v; int kv_stride;
  void* out; int out_stride;
  float* lse;   /* optional out [B][heads][Nq]: log2-domain log-sum-exp of the scaled scores (kept for pd_attn_wide_bwd), or NULL */
} pd_attn_wide_args;
int pd_attn_wide(const pd_attn_wide_args* a, void* stream);

/* pd_attn_wide_bwd: gradient of pd_attn_wide (autograd of F.scaled_dot_product_attention, one wide head per D channels): what
 * accelerator.backward(loss) (utils_training.py:436) runs for the attention blocks of orig_google_ddpm_model_denoiser.json
 * (attention_head_dim null -> one 512-channel head, cond_unet_2d.py:176-197).  P is recomputed from the forward's lse.
 *   o / dout: the forward's output and the gradient w.r.t. it, [B][Nq][o_stride];  delta: workspace [B][heads][Nq] (written by the
 *   dQ pass, read by the dK / dV pass);  dq: [B][Nq][dq_stride];  dk, dv: [B][Nkv][dkv_stride].  bf16 / fp32 (fp16: inference only). */
typedef struct {
  int dtype, B, heads, D, Nq, Nkv; float scale;
  const void* q; int q_stride;
  const void* k; const void* v; int kv_stride;
  const void* o; const void* dout; int o_stride;
  const float* lse; float* delta;
  void* dq; int dq_stride;
  void* dk; void* dv; int dkv_stride;
} pd_attn_wide_bwd_args;
int pd_attn_wide_bwd(const pd_attn_wide_bwd_args* a, void* stream);

/* pd_latent_sample: AutoencoderKL.encode(x).latent_dist.sample(generator) (or .mode() when noise = NULL) times the
 * pipeline's scaling_factor (custom_pipeline_stable_diffusion_img2img.py:431-433, utils_Img2Img.py:833-836):
 *   out = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise),  moments = [B][2C][HW] fp32 = [mean | logvar]. */
typedef struct { int B, C, HW; float scale; const float* moments; const float* noise; float* out; } pd_latent_sample_args;
int pd_latent_sample(const pd_latent_sample_args* a, void* stream);

/* pd_layernorm: y[r][c] = (x[r][c] - mean_r) * rstd_r * gamma[c] + beta[c]  (nn.LayerNorm(C), biased variance) */
typedef struct { int dtype; long long rows; int C; float eps; const void* x; const float* gamma; const float* beta; void* y; } pd_layernorm_args;
int pd_layernorm(const pd_layernorm_args* a, void* stream);

/* pd_geglu: y[r][i] = x[r][i] * gelu(x[r][inner + i])   (diffusers GEGLU after its Linear(C, 2*inner); exact erf GELU) */
typedef struct { int dtype; long long rows; int inner; const void* x; void* y; } pd_geglu_args;
int pd_geglu(const pd_geglu_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the Transformer2DModel blocks: what autograd runs under accelerator.backward(loss) (utils_training.py:436)
 * when _SD_prediction_wrapper (utils_training.py:459-496) trains the SD UNet.  Linear layers reuse pd_conv (input gradient,
 * transposed weights) and pd_conv_wgrad (ksize 1).
 *
 * pd_attn_d64_bwd: gradient of pd_attn_d64; P is recomputed from the forward's lse.  o / dout: the forward's output and the
 * gradient w.r.t. it, [B][Nq][o_stride]; delta: workspace [B][heads][Nq]; dq: [B][Nq][dq_stride], dk / dv: [B][Nkv][dkv_stride]
 * (channel = head*64 + d, so they may be slices of one fused projection's output gradient). */
typedef struct {
  int dtype;
  int B, heads, Nq, Nkv;
  const void* q; int q_stride;
  const void* k; const void* v; int kv_stride;
  const void* o; const void* dout; int o_stride;
  const float* lse; float* delta;
  void* dq; int dq_stride;
  void* dk; void* dv; int dkv_stride;
} pd_attn_d64_bwd_args;
int pd_attn_d64_bwd(const pd_attn_d64_bwd_args* a, void* stream);

/* pd_layernorm_bwd: dx = LayerNorm'(x; gamma)(dy) [+ res: the gradient arriving over the skip connection around the
 * normalised branch]; dgamma += sum_rows dy*xhat, dbeta += sum_rows dy (both or neither; `partial` is a workspace of
 * pd_layernorm_bwd_blocks(rows) * 2 * C floats, summed in a fixed order). */
typedef struct {
  int dtype; long long rows; int C; float eps;
  const void* x; const void* dy; const float* gamma; const void* res;
  void* dx; float* dgamma; float* dbeta; float* partial;
  /* ABI 8 (optional, needs `partial` sized for 3 * C floats per block): dxsum[c] += sum_rows dx[row][c] of the STORED dx -- the bias gradient of the
   * Linear layer whose output gradient dx is (the residual stream), without another pass over it */
  float* dxsum;
} pd_layernorm_bwd_args;
int pd_layernorm_bwd(const pd_layernorm_bwd_args* a, void* stream);
int pd_layernorm_bwd_blocks(long long rows);

/* pd_token_embedding_grad: gradient of CustomEmbedding (custom_embedding.py:36-47) from the encoder_hidden_states gradient:
 * dtable[labels[n]][c] += d[n*row_stride + c] (row n = token 0 of sample n; the 76 padding tokens are constants).
 * labels = NULL (unconditional step, utils_training.py:465-471): no-op. */
typedef struct { int dtype; int rows, dim, num_classes; long long row_stride; const int64_t* labels; const void* d; float* dtable; } pd_token_embedding_grad_args;
int pd_token_embedding_grad(const pd_token_embedding_grad_args* a, void* stream);

/* pd_geglu_bwd: x = [h | g] (the forward's input, [rows][2*inner]), dy [rows][inner] -> dx = [dy*gelu(g) | dy*h*gelu'(g)] */
typedef struct {
  int dtype; long long rows; int inner; const void* x; const void* dy; void* dx;
  /* ABI 8 (optional): per-split column sums of dx AS STORED -- the bias gradient of the projection that feeds the gate without another pass over
   * dx: sums[(n * sum_splits + sp) * 2 * inner + c] for the B samples of rows / B rows each (the workspace layout of pd_channel_sum with x = NULL,
   * which folds them).  Needs inner % 256 == 0 and rows % B == 0; sums = NULL: not computed. */
  float* sums; int sum_splits; int B;
} pd_geglu_bwd_args;
int pd_geglu_bwd(const pd_geglu_bwd_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient-guided transfer (_custom_guided_generation, utils_Img2Img.py:699-760).
 * pd_lp_guidance: losses[n] = || x0_n - target_n ||_p (Lp_loss, :245-270) with x0 = DDIMScheduler.step(...).pred_original_sample
 * (prediction type, clipping as pd_ddim_step), and its gradient split the way the chain rule needs it:
 *   d_model_out = dL/dx0 * dx0/d(model_out)   (fed to the UNet backward)      d_sample_direct = dL/dx0 * dx0/d(sample)
 * pd_guidance_apply: out = x - scale * (g_direct + g_unet)   (:747-751)
 */
typedef struct {
  int64_t numel, per_sample;
  int pred_type, clip; float clip_range, sqrt_a, sqrt_b;
  float p;                                  /* finite, >= 1 */
  const float* sample; const float* model_out; const float* target;
  double* partial; int splits;              /* workspace [B][splits] */
  float* d_model_out; float* d_sample_direct;
  float* losses;                            /* out [B] or NULL */
} pd_lp_guidance_args;
int pd_lp_guidance(const pd_lp_guidance_args* a, void* stream);
typedef struct { int64_t numel; float scale; const float* x; const float* g_direct; const float* g_unet; float* out; } pd_guidance_apply_args;
int pd_guidance_apply(const pd_guidance_apply_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stream capture helpers (hipGraph): the S-step sampling loop is captured once and replayed.
 */
int pd_graph_begin(void* stream);
int pd_graph_end(void* stream, void** graph_exec_out);
int pd_graph_launch(void* graph_exec, void* stream);
int pd_graph_destroy(void* graph_exec);

/* Timing helpers: HIP events on the launch stream (bench.py roofline leg). */
int pd_event_create(void** ev);
int pd_event_record(void* ev, void* stream);
int pd_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */
int pd_event_destroy(void* ev);

/* ------------------------------------------------------------------------------------------------
 * (ABI 7) The evaluation metrics' feature extractor: FID / IS / KID as torch_fidelity.calculate_metrics computes them for
 * utils_training.py:948-1001 (per class at evaluation time) and utils_Img2Img.py:462-563 (after a class-transfer experiment) run the
 * "inception-v3-compat" network (TF-Slim InceptionV3 of the original FID code) over uint8 images.  csrc/metric_kernels.hip; host side
 * phendiff_amd/metrics.py (BatchNorm folded into weights / bias at pack time; statistics in fp64 on the host).
 *   pd_resize_tf1: x uint8 NHWC [N][H][W][3] -> y NHWC [N][OH][OW][32] in `dtype` (channels 3..31 zero):
 *       v = bilinear(x) with source coordinate = destination index * scale (scale_y = H / OH, scale_x = W / OW as fp32; no half-pixel
 *       centres, neighbours floor / min(floor + 1, size - 1): interpolate_bilinear_2d_like_tensorflow1x), y = (v - sub) / div.
 *   pd_conv_rect: y[b][oy][ox][y_co + co] = relu?(bias[co] + sum W[co][ci][ky][kx] x[b][oy stride + ky - pad_h][ox stride + kx - pad_w][ci])
 *       for co < Cout_pad; x NHWC with channel stride x_cs (>= Cin, Cin % 32 == 0), y NHWC with channel stride y_cs (the output is a
 *       channel SLICE of a wider tensor: the concatenation of the Inception branches); w_packed = pd_conv's fragment order with
 *       taps = KH * KW (tap = ky * KW + kx), [Cout_pad/32][Cin/32][taps][2][64][8]; bias fp32 [Cout_pad].
 *   pd_pool2d: mode 0 max / 1 average with count_include_pad = False over k x k windows (stride, pad; padding never wins a max),
 *       NHWC -> a channel slice of y; mode 2: global average over Hin x Win -> y fp32 [B][C].
 *   pd_fc_f32: y[r][o] = sum_k x[r][k] wt[k][o] (+ bias[o]), fp32 (wt = the Linear weight transposed, [in][out]). */
typedef struct { int dtype; int N, H, W, OH, OW; float scale_y, scale_x, sub, div; const unsigned char* x; void* y; } pd_resize_tf1_args;
int pd_resize_tf1(const pd_resize_tf1_args* a, void* stream);
typedef struct {
  int dtype; int B, Hin, Win, Cin, Hout, Wout, Cout_pad, KH, KW, stride, pad_h, pad_w, relu;
  const void* x; int x_cs; const void* w_packed; const float* bias; void* y; int y_cs, y_co;
} pd_conv_rect_args;
int pd_conv_rect(const pd_conv_rect_args* a, void* stream);
typedef struct { int dtype; int B, Hin, Win, C, Hout, Wout, k, stride, pad, mode; const void* x; int x_cs; void* y; int y_cs, y_co; } pd_pool2d_args;
int pd_pool2d(const pd_pool2d_args* a, void* stream);
typedef struct { int rows, in_dim, out_dim; const float* x; const float* wt; const float* bias; float* y; } pd_fc_f32_args;
int pd_fc_f32(const pd_fc_f32_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pd_comm_*: the data-parallel gradient exchange -- DistributedDataParallel's bucketed all-reduce under accelerator.backward(loss)
 * (train.py:311-326, utils_training.py:436) -- directly on RCCL (loaded at run time; PD_ERR_UNSUPPORTED when librccl.so is absent).
 * One communicator per process / GPU, created on the current device.  Rank 0 draws the id (pd_comm_unique_id); the host program ships
 * the 128 bytes to the other ranks; every rank calls pd_comm_init(id, rank, world, &comm).
 * pd_allreduce_bucket: in-place fp32 sum (mean != 0: then divided by world) of one contiguous bucket on `stream`;
 *   algo 0 = ncclAllReduce, algo 1 = reduce-scatter + all-gather over count / world shards (count % world == 0, else algo 0 is used). */
typedef struct { char bytes[128]; } pd_comm_id;
int pd_comm_unique_id(pd_comm_id* out);
int pd_comm_init(const pd_comm_id* id, int rank, int world, void** comm_out);
int pd_allreduce_bucket(void* comm, float* buf, size_t count, int mean, int algo, void* stream);
/* rank and size as the communicator itself reports them (ncclCommUserRank / ncclCommCount) -- what bench.py prints as `rccl_world_size` */
int pd_comm_query(void* comm, int* rank_out, int* world_out);
int pd_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* PHENDIFF_HIP_H_ */
