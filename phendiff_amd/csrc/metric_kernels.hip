// Kernels of the evaluation metrics' feature extractor (FID / IS / KID: utils_training.py:948-1001, utils_Img2Img.py:462-563 call
// torch_fidelity.calculate_metrics, whose "inception-v3-compat" extractor is the TF-Slim InceptionV3 of the original FID code).  The network
// is 94 convolutions of many shapes (3x3 stride 2 pad 0, 1x7 / 7x1, 1x3 / 3x1, 5x5, 1x1; channel counts 32 .. 2048) with BatchNorm (eval:
// folded into weights and bias at pack time) + ReLU, max / average pools and channel concatenations -- an evaluation-time side computation
// (5.7 GMAC per image, a few thousand images per evaluation), not the sampling hot path: ONE generic implicit-GEMM kernel instead of
// pd_conv's tuned variants.
//   pd_resize_tf1     uint8 NHWC image -> TF1-style bilinear resize (source = destination index * in / out, no half-pixel centres:
//                     torch-fidelity interpolate_bilinear_2d_like_tensorflow1x) -> (v - 128) / 128 -> NHWC, 32 channels (3 used)
//   pd_conv_rect      y[b][oy][ox][co_off + co] = relu?(bias[co] + sum_{ky,kx,ci} W[co][ci][ky][kx] x[b][oy s + ky - ph][ox s + kx - pw][ci]):
//                     wave = 32 output pixels x 64 channels on the 32x32 MFMA (exact-fp32 chain in PD_F32), operands straight from
//                     global / L2 (weights in pd_conv's packed fragment order with taps = KH * KW: 1 KiB coalesced per fragment; the
//                     pixel fragment is 16 contiguous bytes per lane), the output written INTO a channel slice of a wider tensor
//                     (torch.cat of the Inception branches is never materialised)
//   pd_pool2d         3x3 max / average (count_include_pad = False) pooling, stride 1 or 2, NHWC, into a channel slice; mode 2 =
//                     global average pool -> fp32 [B][C] (the 2048-d pool3 features)
//   pd_fc_f32         logits[b][o] = sum_k f[b][k] Wt[k][o] (+ bias[o]) in fp32 (fc 2048 -> 1008 on the pooled features)
#include "pd_common.h"
#include "pd_stage.h"

namespace pd {

// ---- resize ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void resize_tf1_kernel(const pd_resize_tf1_args a) {
#pragma clang fp contract(off)      // the oracle's lerp is mul, then add: keep the fp32 engine bit-comparable
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)a.N * a.OH * a.OW;
  if (idx >= total) return;
  const int ox = (int)(idx % a.OW), oy = (int)((idx / a.OW) % a.OH), n = (int)(idx / ((long long)a.OW * a.OH));
  const float gy = (float)oy * a.scale_y, gx = (float)ox * a.scale_x;
  const int y0 = (int)gy, x0 = (int)gx;
  const int y1 = min(y0 + 1, a.H - 1), x1 = min(x0 + 1, a.W - 1);
  const float dy = gy - (float)y0, dx = gx - (float)x0;
  const unsigned char* img = a.x + (size_t)n * a.H * a.W * 3;
  float o[8];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float i00 = (float)img[((size_t)y0 * a.W + x0) * 3 + c], i01 = (float)img[((size_t)y0 * a.W + x1) * 3 + c];
    const float i10 = (float)img[((size_t)y1 * a.W + x0) * 3 + c], i11 = (float)img[((size_t)y1 * a.W + x1) * 3 + c];
    const float top = i00 + (i01 - i00) * dx, bot = i10 + (i11 - i10) * dx;
    o[c] = ((top + (bot - top) * dy) - a.sub) / a.div;
  }
#pragma unroll
  for (int c = 3; c < 8; ++c) o[c] = 0.f;
  T* dst = (T*)a.y + (size_t)idx * 32;
  using E = Elem<T>;
  E::store(dst, E::pack(o));
  float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 1; q < 4; ++q) E::store(dst + 8 * q, E::pack(z));
}

// ---- generic convolution ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_rect_kernel(const pd_conv_rect_args a) {
  using E = Elem<T>;
  using Frag = typename E::Frag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const long long P = (long long)a.B * a.Hout * a.Wout;
  const long long pix = (long long)blockIdx.x * 128 + wave * 32 + r;
  const bool pvalid = pix < P;
  const long long pc = pvalid ? pix : P - 1;
  const int ox = (int)(pc % a.Wout), oy = (int)((pc / a.Wout) % a.Hout), b = (int)(pc / ((long long)a.Wout * a.Hout));
  const int ntile = a.Cout_pad / 32;
  int ct[2];
  ct[0] = min(blockIdx.y * 2, ntile - 1); ct[1] = min(blockIdx.y * 2 + 1, ntile - 1);
  const int taps = a.KH * a.KW, chunks = a.Cin / 32;
  const T* wbase = (const T*)a.w_packed;
  f32x16 acc[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *(const f32x4*)(a.bias + ct[c] * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[c][4 * g + i] = bv[i];
    }
  }
  const T* xb = (const T*)a.x + (size_t)b * a.Hin * a.Win * a.x_cs;
  for (int tap = 0; tap < taps; ++tap) {
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    const int iy = oy * a.stride + ky - a.pad_h, ix = ox * a.stride + kx - a.pad_w;
    const bool ok = pvalid && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
    const T* px = xb + ((size_t)(ok ? iy : 0) * a.Win + (ok ? ix : 0)) * a.x_cs + 8 * h;
    for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frag bf = E::load(px + ch * 32 + 16 * s);
        if (!ok) bf = E::zero();
        const size_t fo = (((size_t)ch * taps + tap) * 2 + s) * 512 + lane * 8;
        const Frag a0 = E::load(wbase + (size_t)ct[0] * chunks * taps * 1024 + fo);
        const Frag a1 = E::load(wbase + (size_t)ct[1] * chunks * taps * 1024 + fo);
        acc[0] = E::mma(a0, bf, acc[0]);
        acc[1] = E::mma(a1, bf, acc[1]);
      }
    }
  }
  if (!pvalid) return;
  T* yp = (T*)a.y + (size_t)pix * a.y_cs + a.y_co;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c == 1 && (int)(blockIdx.y * 2 + 1) >= ntile) break;       // odd number of 32-channel tiles: the clamped second tile is not stored
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = acc[c][4 * g + i]; if (a.relu) v[i] = fmaxf(v[i], 0.f); }
      store4(yp + ct[c] * 32 + 8 * g + 4 * h, v[0], v[1], v[2], v[3]);
    }
  }
}

// ---- pooling -----------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pool2d_kernel(const pd_pool2d_args a) {
  using E = Elem<T>;
  const int C8 = a.C / 8;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (a.mode == 2) {            // global average: one thread per (sample, 8 channels), fp32 output [B][C]
    if (idx >= (long long)a.B * C8) return;
    const int c8 = (int)(idx % C8), b = (int)(idx / C8);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const T* xp = (const T*)a.x + (size_t)b * a.Hin * a.Win * a.x_cs + c8 * 8;
    for (int p = 0; p < a.Hin * a.Win; ++p) {
      float v[8];
      E::unpack(E::load(xp + (size_t)p * a.x_cs), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += v[j];
    }
    const float inv = 1.0f / (float)(a.Hin * a.Win);
#pragma unroll
    for (int j = 0; j < 8; ++j) ((float*)a.y)[(size_t)b * a.C + c8 * 8 + j] = s[j] * inv;
    return;
  }
  const long long total = (long long)a.B * a.Hout * a.Wout * C8;
  if (idx >= total) return;
  const int c8 = (int)(idx % C8);
  const long long pix = idx / C8;
  const int ox = (int)(pix % a.Wout), oy = (int)((pix / a.Wout) % a.Hout), b = (int)(pix / ((long long)a.Wout * a.Hout));
  float m[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = a.mode == 0 ? -3.0e38f : 0.f;
  int cnt = 0;
  for (int ky = 0; ky < a.k; ++ky) {
    const int iy = oy * a.stride + ky - a.pad;
    if (iy < 0 || iy >= a.Hin) continue;
    for (int kx = 0; kx < a.k; ++kx) {
      const int ix = ox * a.stride + kx - a.pad;
      if (ix < 0 || ix >= a.Win) continue;
      float v[8];
      E::unpack(E::load((const T*)a.x + (((size_t)b * a.Hin + iy) * a.Win + ix) * a.x_cs + c8 * 8), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = a.mode == 0 ? fmaxf(m[j], v[j]) : m[j] + v[j];
      ++cnt;
    }
  }
  if (a.mode == 1) {           // F.avg_pool2d(count_include_pad = False): the divisor counts the pixels inside the image
    const float inv = 1.0f / (float)cnt;
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] *= inv;
  }
  E::store((T*)a.y + (size_t)pix * a.y_cs + a.y_co + c8 * 8, E::pack(m));
}

__global__ __launch_bounds__(256) void fc_f32_kernel(const pd_fc_f32_args a) {
  const int o = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (o >= a.out_dim) return;
  const float* f = a.x + (size_t)b * a.in_dim;
  float s = 0.f;
  for (int k = 0; k < a.in_dim; ++k) s = fmaf(f[k], a.wt[(size_t)k * a.out_dim + o], s);
  a.y[(size_t)b * a.out_dim + o] = a.bias ? s + a.bias[o] : s;
}

}  // namespace pd

using namespace pd;

extern "C" int pd_resize_tf1(const pd_resize_tf1_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->x && a->y, PD_ERR_ARG, "pd_resize_tf1: null pointer");
  PD_CHECK(a->N > 0 && a->H > 0 && a->W > 0 && a->OH > 0 && a->OW > 0 && a->div != 0.f, PD_ERR_SHAPE, "pd_resize_tf1: bad shape");
  const long long total = (long long)a->N * a->OH * a->OW;
  PD_CHECK(total < (1ll << 31) * 256, PD_ERR_SHAPE, "pd_resize_tf1: too many pixels");
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == PD_F32) hipLaunchKernelGGL(resize_tf1_kernel<float>, grid, dim3(256), 0, st, *a);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(resize_tf1_kernel<bf16_t>, grid, dim3(256), 0, st, *a);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(resize_tf1_kernel<half_t>, grid, dim3(256), 0, st, *a);
  else { set_error("pd_resize_tf1: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_conv_rect(const pd_conv_rect_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->x && a->w_packed && a->bias && a->y, PD_ERR_ARG, "pd_conv_rect: null pointer");
  PD_CHECK(a->B > 0 && a->Hin > 0 && a->Win > 0 && a->Hout > 0 && a->Wout > 0 && a->Cin > 0 && a->Cin % 32 == 0 && a->Cout_pad > 0 &&
               a->Cout_pad % 32 == 0 && a->KH >= 1 && a->KW >= 1 && a->KH * a->KW <= 49 && a->stride >= 1 && a->pad_h >= 0 && a->pad_w >= 0,
           PD_ERR_SHAPE, "pd_conv_rect: Cin and Cout_pad must be multiples of 32, kernel <= 7 x 7");
  PD_CHECK(a->x_cs >= a->Cin && a->x_cs % 8 == 0 && a->y_cs >= a->y_co + a->Cout_pad && a->y_cs % 4 == 0 && a->y_co % 4 == 0, PD_ERR_SHAPE,
           "pd_conv_rect: channel strides must cover the slice (x_cs %% 8, y_cs / y_co %% 4)");
  PD_CHECK((a->Hin + 2 * a->pad_h - a->KH) / a->stride + 1 == a->Hout && (a->Win + 2 * a->pad_w - a->KW) / a->stride + 1 == a->Wout, PD_ERR_SHAPE,
           "pd_conv_rect: output size does not match floor((in + 2 pad - k) / stride) + 1");
  const long long P = (long long)a->B * a->Hout * a->Wout;
  PD_CHECK((P + 127) / 128 < (1ll << 31), PD_ERR_SHAPE, "pd_conv_rect: grid too large");
  const dim3 grid((unsigned)((P + 127) / 128), (unsigned)((a->Cout_pad / 32 + 1) / 2));
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == PD_F32) hipLaunchKernelGGL(conv_rect_kernel<float>, grid, dim3(256), 0, st, *a);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(conv_rect_kernel<bf16_t>, grid, dim3(256), 0, st, *a);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(conv_rect_kernel<half_t>, grid, dim3(256), 0, st, *a);
  else { set_error("pd_conv_rect: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_pool2d(const pd_pool2d_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->x && a->y, PD_ERR_ARG, "pd_pool2d: null pointer");
  PD_CHECK(a->B > 0 && a->Hin > 0 && a->Win > 0 && a->C > 0 && a->C % 8 == 0 && a->x_cs >= a->C && a->x_cs % 8 == 0 && a->mode >= 0 && a->mode <= 2,
           PD_ERR_SHAPE, "pd_pool2d: C and the channel stride must be multiples of 8, mode 0 | 1 | 2");
  long long total;
  if (a->mode == 2) total = (long long)a->B * (a->C / 8);
  else {
    PD_CHECK(a->k >= 1 && a->stride >= 1 && a->pad >= 0 && 2 * a->pad < a->k + 1 && (a->Hin + 2 * a->pad - a->k) / a->stride + 1 == a->Hout &&
                 (a->Win + 2 * a->pad - a->k) / a->stride + 1 == a->Wout && a->y_cs >= a->y_co + a->C && a->y_cs % 8 == 0 && a->y_co % 8 == 0,
             PD_ERR_SHAPE, "pd_pool2d: window / output size / output slice mismatch");
    total = (long long)a->B * a->Hout * a->Wout * (a->C / 8);
  }
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == PD_F32) hipLaunchKernelGGL(pool2d_kernel<float>, grid, dim3(256), 0, st, *a);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(pool2d_kernel<bf16_t>, grid, dim3(256), 0, st, *a);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(pool2d_kernel<half_t>, grid, dim3(256), 0, st, *a);
  else { set_error("pd_pool2d: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_fc_f32(const pd_fc_f32_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->x && a->wt && a->y && a->rows > 0 && a->in_dim > 0 && a->out_dim > 0 && a->rows < 65536, PD_ERR_ARG, "pd_fc_f32: bad args");
  hipLaunchKernelGGL(fc_f32_kernel, dim3((unsigned)((a->out_dim + 255) / 256), (unsigned)a->rows), dim3(256), 0, (hipStream_t)stream, *a);
  PD_LAUNCH_CHECK();
  return PD_OK;
}
