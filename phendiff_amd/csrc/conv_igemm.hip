// pd_conv: implicit-GEMM convolution on gfx950 MFMA, NHWC, LDS-staged halo tiles.
//
// GEMM view:  D[co][pixel] = sum_k W[co][k] * X[k][pixel],  k = (tap, ci).
//   A operand = packed weights   (lane: row = co,    8 consecutive ci)   -> straight from global/L2, 1 KiB coalesced per wave
//   B operand = activation tile  (lane: col = pixel, 8 consecutive ci)   -> ds_read_b128 from the LDS halo tile
//   D         : lane owns one pixel and 16 output channels (4 runs of 4 consecutive co) -> 8/16-byte NHWC stores
// Workgroup = 256 threads = 4 waves = 2 (pixel halves) x 2 (co halves of 32); tile = TH*TW pixels x 64 co.
// K loop: chunks of 32 input channels; per chunk the halo tile is staged global -> regs -> (GroupNorm affine,
// SiLU, zero padding) -> LDS once and reused by all KS*KS taps (9x LDS reuse instead of 9x global re-reads).
// The next chunk's global loads are issued before the MFMA phase and written to LDS after it (latency hidden).
// LDS pixel pitch = 32 ch + 16 B pad: an odd number of 16-B slots, so a wave's 32 consecutive pixels hit 16
// distinct slots per ds_read_b128 lane group (conflict-free for TW = 32).
#include "pd_common.h"

namespace pd {

struct ConvP {
  int B, Hin, Win, Hout, Wout;
  int C0, C1, Cout, Cout_pad;
  int pad, upsample, silu, out_mode, heads;
  int tiles_x, tiles_y, n_pix_tiles, n_co_tiles, nchunks;
  const void* x0; const void* x1;
  const float* scale; const float* shift;
  const void* w;
  const float* bias;
  const float* temb; int temb_stride;
  const void* residual;
  void* y;
};

template <typename T, int KS, int STRIDE, int TH, int TW>
__global__ __launch_bounds__(256) void conv_kernel(const ConvP p) {
  using E = Elem<T>;
  using Frag = typename E::Frag;
  constexpr int TP = TH * TW;              // pixels per workgroup tile
  constexpr int NF = TP / 64;              // 32-pixel fragments per wave
  constexpr int RPF = 32 / TW;             // tile rows per fragment (TW == 32 -> 1)
  constexpr int IN_TH = (TH - 1) * STRIDE + KS;
  constexpr int IN_TW = (TW - 1) * STRIDE + KS;
  constexpr int NPIX = IN_TH * IN_TW;
  constexpr int PITCH = 32 * E::BYTES + 16;
  constexpr int NIT = (NPIX * 4 + 255) / 256;
  constexpr int TAPS = KS * KS;
  static_assert(RPF >= 1 && TW * RPF == 32, "TW must divide 32");

  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  // ---- block -> (pixel tile, co tile); blocks b and b+8 share an XCD (L2): co tiles of one pixel tile go there
  const int bid = blockIdx.x;
  const int xcd = bid & 7, jj = bid >> 3;
  const int co_t = jj % p.n_co_tiles;
  const int pt = (jj / p.n_co_tiles) * 8 + xcd;
  if (pt >= p.n_pix_tiles) return;
  const int tx = pt % p.tiles_x;
  const int ty = (pt / p.tiles_x) % p.tiles_y;
  const int n = pt / (p.tiles_x * p.tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int ct32 = co_t * 2 + wc;                      // this wave's 32-co tile
  const bool wave_active = (ct32 * 32) < p.Cout_pad;   // wave-uniform

  // ---- staging bookkeeping: this thread's pieces (pixel, 8-channel sub-block)
  const int sub = tid & 3;
  const int Hc = p.upsample ? p.Hin * 2 : p.Hin;
  const int Wc = p.upsample ? p.Win * 2 : p.Win;
  int spix[NIT];   // linear source pixel index (n, sy, sx) or -1 (zero padding) / -2 (no such piece)
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + 256 * i;
    const int pix = idx >> 2;
    int v = -2;
    if (pix < NPIX) {
      const int u = pix / IN_TW, vv = pix - u * IN_TW;
      const int iy = y0 * STRIDE - p.pad + u, ix = x0 * STRIDE - p.pad + vv;
      v = -1;
      if (iy >= 0 && iy < Hc && ix >= 0 && ix < Wc) {
        const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
        v = (n * p.Hin + sy) * p.Win + sx;
      }
    }
    spix[i] = v;
  }

  Frag stage[NIT];
  auto issue_loads = [&](int chunk) {
    const int cch = chunk * 32;
    const T* src; int cs, coff;
    if (cch < p.C0) { src = (const T*)p.x0; cs = p.C0; coff = cch; }
    else            { src = (const T*)p.x1; cs = p.C1; coff = cch - p.C0; }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      if (spix[i] >= 0) stage[i] = E::load(src + (size_t)spix[i] * cs + coff + sub * 8);
      else stage[i] = E::zero();
    }
  };
  auto write_lds = [&](int chunk) {
    float sc[8], sh[8];
    const bool affine = p.scale != nullptr;
    if (affine) {
      const int cin = p.C0 + p.C1;
      const float* ps = p.scale + (size_t)n * cin + chunk * 32 + sub * 8;
      const float* pb = p.shift + (size_t)n * cin + chunk * 32 + sub * 8;
      f32x4 a0 = *(const f32x4*)ps, a1 = *((const f32x4*)ps + 1);
      f32x4 b0 = *(const f32x4*)pb, b1 = *((const f32x4*)pb + 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) { sc[j] = a0[j]; sc[4 + j] = a1[j]; sh[j] = b0[j]; sh[4 + j] = b1[j]; }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      if (spix[i] == -2) continue;
      Frag f = stage[i];
      if (spix[i] >= 0 && (affine || p.silu)) {
        float v[8];
        E::unpack(f, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = v[j];
          if (affine) t = t * sc[j] + sh[j];
          if (p.silu) t = silu_f(t);
          v[j] = t;
        }
        f = E::pack(v);
      }
      const int pix = (tid + 256 * i) >> 2;
      E::store(lds + pix * PITCH + sub * 8 * E::BYTES, f);
    }
  };

  // ---- per-lane LDS read bases for the B (activation) fragments
  int rbase[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int fi = wp * NF + f;
    const int py = fi * RPF + r / TW, px = r % TW;
    rbase[f] = ((py * STRIDE) * IN_TW + px * STRIDE) * PITCH + h * 8 * E::BYTES;
  }

  f32x16 acc[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[f] = (f32x16)(0.f);

  const T* wbase = (const T*)p.w + (size_t)ct32 * p.nchunks * (TAPS * 2 * 512) + lane * 8;

  issue_loads(0);
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    if (chunk > 0) __syncthreads();
    write_lds(chunk);
    __syncthreads();
    if (chunk + 1 < p.nchunks) issue_loads(chunk + 1);
    if (wave_active) {
      const T* wc_ptr = wbase + (size_t)chunk * (TAPS * 2 * 512);
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int ky = tap / KS, kx = tap % KS;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const Frag a = E::load(wc_ptr + (tap * 2 + s) * 512);
          const int toff = (ky * IN_TW + kx) * PITCH + s * 16 * E::BYTES;
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const Frag b = E::load(lds + rbase[f] + toff);
            acc[f] = E::mma(a, b, acc[f]);
          }
        }
      }
    }
  }

  if (!wave_active) return;
  // ---- epilogue: + bias + temb + residual, store
  const int co_w = ct32 * 32;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int fi = wp * NF + f;
    const int py = fi * RPF + r / TW, px = r % TW;
    const int oy = y0 + py, ox = x0 + px;
    if (oy >= p.Hout || ox >= p.Wout) continue;
    const size_t opix = ((size_t)n * p.Hout + oy) * p.Wout + ox;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co_w + 8 * g + 4 * h;
      float v[4];
      const f32x4 bb = *(const f32x4*)(p.bias + co);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = acc[f][4 * g + i] + bb[i];
      if (p.temb) {
        const float* tp = p.temb + (size_t)n * p.temb_stride + co;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (co + i < p.Cout) v[i] += tp[i];
      }
      if (p.out_mode == PD_OUT_NHWC) {
        if (co >= p.Cout) continue;
        if (p.residual) {
          float rr[4];
          load4((const T*)p.residual + opix * p.Cout + co, rr);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += rr[i];
        }
        store4((T*)p.y + opix * p.Cout + co, v[0], v[1], v[2], v[3]);
      } else if (p.out_mode == PD_OUT_NCHW_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (co + i < p.Cout) ((float*)p.y)[(((size_t)n * p.Cout + co + i) * p.Hout + oy) * p.Wout + ox] = v[i];
      } else {  // PD_OUT_QKV_HEADS: [which][B][heads][N][8]
        if (co >= p.Cout) continue;
        const int C = p.heads * 8;
        const int which = co / C, cc = co - which * C;
        const int head = cc >> 3, d = cc & 7;
        const size_t N = (size_t)p.Hout * p.Wout;
        const size_t tok = (size_t)oy * p.Wout + ox;
        T* dst = (T*)p.y + ((((size_t)which * p.B + n) * p.heads + head) * N + tok) * 8 + d;
        store4(dst, v[0], v[1], v[2], v[3]);
      }
    }
  }
}

template <typename T, int KS, int STRIDE, int TH, int TW>
static int launch_conv(const ConvP& p, hipStream_t st) {
  constexpr int IN_TH = (TH - 1) * STRIDE + KS, IN_TW = (TW - 1) * STRIDE + KS;
  constexpr int PITCH = 32 * Elem<T>::BYTES + 16;
  constexpr int LDS_BYTES = IN_TH * IN_TW * PITCH;
  static_assert(LDS_BYTES <= 160 * 1024, "tile too large");
  auto kern = conv_kernel<T, KS, STRIDE, TH, TW>;
  if (LDS_BYTES > 64 * 1024) {
    static bool attr_set = false;   // per instantiation
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PD_ERR_LAUNCH; }
      attr_set = true;
    }
  }
  ConvP q = p;
  q.tiles_x = (p.Wout + TW - 1) / TW;
  q.tiles_y = (p.Hout + TH - 1) / TH;
  q.n_pix_tiles = p.B * q.tiles_x * q.tiles_y;
  q.n_co_tiles = (p.Cout_pad + 63) / 64;
  const int grid = ((q.n_pix_tiles + 7) / 8) * 8 * q.n_co_tiles;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, st, q);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

template <typename T>
static int dispatch_conv(const ConvP& p, int ksize, int stride, hipStream_t st) {
  // tile shape by output width: 32-wide rows when possible (bank-conflict-free), else squarer tiles
  const int w = p.Wout;
  if (ksize == 3 && stride == 1) {
    if (w >= 32) return launch_conv<T, 3, 1, 8, 32>(p, st);
    if (w >= 16) return launch_conv<T, 3, 1, 16, 16>(p, st);
    return launch_conv<T, 3, 1, 32, 8>(p, st);
  }
  if (ksize == 3 && stride == 2) {
    if (w >= 32) return launch_conv<T, 3, 2, 4, 32>(p, st);
    if (w >= 16) return launch_conv<T, 3, 2, 8, 16>(p, st);
    return launch_conv<T, 3, 2, 16, 8>(p, st);
  }
  if (ksize == 1 && stride == 1) {
    if (w >= 32) return launch_conv<T, 1, 1, 8, 32>(p, st);
    if (w >= 16) return launch_conv<T, 1, 1, 16, 16>(p, st);
    return launch_conv<T, 1, 1, 32, 8>(p, st);
  }
  set_error("pd_conv: unsupported ksize=%d stride=%d", ksize, stride);
  return PD_ERR_UNSUPPORTED;
}

}  // namespace pd

extern "C" int pd_conv(const pd_conv_args* a, void* stream) {
  using namespace pd;
  PD_CHECK(a != nullptr, PD_ERR_ARG, "pd_conv: null args");
  PD_CHECK(a->dtype == PD_F32 || a->dtype == PD_BF16, PD_ERR_ARG, "pd_conv: bad dtype %d", a->dtype);
  PD_CHECK(a->B > 0 && a->Hin > 0 && a->Win > 0 && a->Hout > 0 && a->Wout > 0, PD_ERR_SHAPE, "pd_conv: bad spatial shape");
  PD_CHECK(a->C0 > 0 && a->C0 % 32 == 0 && a->C1 >= 0 && a->C1 % 32 == 0, PD_ERR_SHAPE,
           "pd_conv: C0=%d C1=%d must be multiples of 32", a->C0, a->C1);
  PD_CHECK(a->Cout > 0 && a->Cout_pad >= a->Cout && a->Cout_pad % 32 == 0, PD_ERR_SHAPE, "pd_conv: bad Cout/Cout_pad");
  PD_CHECK(a->Cout % 4 == 0 || a->out_mode == PD_OUT_NCHW_F32, PD_ERR_SHAPE, "pd_conv: NHWC output needs Cout %% 4 == 0");
  PD_CHECK(a->x0 && a->w_packed && a->bias && a->y, PD_ERR_ARG, "pd_conv: null pointer");
  PD_CHECK((a->C1 == 0) == (a->x1 == nullptr), PD_ERR_ARG, "pd_conv: x1/C1 mismatch");
  PD_CHECK((a->scale == nullptr) == (a->shift == nullptr), PD_ERR_ARG, "pd_conv: scale/shift mismatch");
  PD_CHECK(a->ksize == 1 || a->ksize == 3, PD_ERR_UNSUPPORTED, "pd_conv: ksize %d", a->ksize);
  PD_CHECK(a->stride == 1 || (a->stride == 2 && a->ksize == 3), PD_ERR_UNSUPPORTED, "pd_conv: stride %d", a->stride);
  PD_CHECK(!(a->upsample && a->stride != 1), PD_ERR_UNSUPPORTED, "pd_conv: upsample with stride");
  {
    const int hc = a->upsample ? 2 * a->Hin : a->Hin, wc = a->upsample ? 2 * a->Win : a->Win;
    const int extra = (a->ksize == 3 && a->pad == 0) ? 1 : 0;   // asymmetric (0,1,0,1) zero pad of Downsample2D(padding=0)
    const int ho = (hc + 2 * a->pad + extra - a->ksize) / a->stride + 1;
    const int wo = (wc + 2 * a->pad + extra - a->ksize) / a->stride + 1;
    PD_CHECK(ho == a->Hout && wo == a->Wout, PD_ERR_SHAPE, "pd_conv: Hout/Wout (%d,%d) != expected (%d,%d)", a->Hout, a->Wout, ho, wo);
  }
  if (a->out_mode == PD_OUT_QKV_HEADS)
    PD_CHECK(a->heads > 0 && a->Cout == 3 * a->heads * 8, PD_ERR_SHAPE, "pd_conv: QKV mode needs Cout == 3*heads*8");
  PD_CHECK(a->out_mode == PD_OUT_NHWC || a->residual == nullptr, PD_ERR_UNSUPPORTED, "pd_conv: residual needs NHWC output");
  ConvP p{};
  p.B = a->B; p.Hin = a->Hin; p.Win = a->Win; p.Hout = a->Hout; p.Wout = a->Wout;
  p.C0 = a->C0; p.C1 = a->C1; p.Cout = a->Cout; p.Cout_pad = a->Cout_pad;
  p.pad = a->pad; p.upsample = a->upsample; p.silu = a->silu; p.out_mode = a->out_mode; p.heads = a->heads;
  p.nchunks = (a->C0 + a->C1) / 32;
  p.x0 = a->x0; p.x1 = a->x1; p.scale = a->scale; p.shift = a->shift; p.w = a->w_packed; p.bias = a->bias;
  p.temb = a->temb; p.temb_stride = a->temb_stride; p.residual = a->residual; p.y = a->y;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == PD_F32) return dispatch_conv<float>(p, a->ksize, a->stride, st);
  return dispatch_conv<bf16_t>(p, a->ksize, a->stride, st);
}
