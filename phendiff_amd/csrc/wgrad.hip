// pd_conv_wgrad: weight gradient of the convolutions pd_conv runs in the forward (autograd of nn.Conv2d / nn.Linear at the
// call sites listed in include/phendiff_hip.h), as an MFMA GEMM whose K dimension is the pixel index:
//     dW[co][ci][ky][kx] = sum over (n, oy, ox) of dY[n][oy][ox][co] * Z[n][oy*s + ky - pad][ox*s + kx - pad][ci]
// Z is the tensor the forward convolution consumed, i.e. silu?(scale*x + shift) of [x0 | x1] (optionally nearest-x2
// upsampled); it is rebuilt in the staging pass exactly as pd_conv builds it, never materialised in HBM.
//
// Workgroup = 4 waves = a 64(co) x 64(ci) x taps weight tile; wave (cf, cif) owns the 32x32xtaps sub-tile in
// 16*taps accumulator registers (144 for 3x3).  The workgroup walks a contiguous range of 16-pixel-wide output tiles
// (its "split" of the K dimension): per tile dY [TH*16 px][64 co] and the Z halo tile [(TH-1)s+k][(16-1)s+k][64 ci] are
// staged NHWC into LDS as two 32-channel planes with 64-byte pixels (conflict-free for the transposed reads), and both
// MFMA operands -- which need 8 consecutive PIXELS per lane for a fixed channel -- are fetched with ds_read_b64_tr_b16:
// the lanes supply per-pixel row addresses, so a filter tap is just an immediate offset on the Z address and the 9 taps
// reuse one dY fragment.  fp32 validation mode reads the same image with ds_read_b32 and runs the exact-fp32 MFMA.
// Each workgroup leaves its partial tile in a slab [split][tap][co][ci]; pd_wgrad_reduce sums the splits in a fixed order
// (bitwise reproducible, no atomics) into the OIHW fp32 gradient buffer.
#include "pd_common.h"
#include "pd_stage.h"

namespace pd {

struct WgradP {
  int B, Hin, Win, Hout, Wout, C0, C1, Cout, pad, upsample, silu;
  int pad_x, dy_step, dy_oy, dy_ox;      // sub-pixel phase (pd_wgrad_args.phase): column padding, dy read at (dy_step oy + dy_oy, dy_step ox + dy_ox) of a tensor dy_step times as large
  int tiles_x, tiles_y, ntiles, tiles_per_split, n_ci_t, n_co_t, splits, nwork;
  int COP, CIP;                    // slab dims (multiples of 64)
  unsigned bytes0, bytes1, bytesdy;
  const void* x0; const void* x1;
  const float* scale; const float* shift;
  const void* dy;
  float* slab;
};

template <typename T, int KS, int STRIDE, int TH>
struct WgradCfg {
  static constexpr int TW = 16, TP = TH * TW, TAPS = KS * KS;
  static constexpr int HH = (TH - 1) * STRIDE + KS, HWD = (TW - 1) * STRIDE + KS, NPIX = HH * HWD;
  static constexpr int PXB = 32 * Elem<T>::BYTES;                       // one pixel of a 32-channel plane
  static constexpr int ZP = PXB + (STRIDE == 2 ? PXB / 2 : 0);         // stride 2: 96-B pixels keep every other pixel conflict-free
  static constexpr int DY_PLANE = TP * PXB;
  static constexpr int Z_PLANE = ((NPIX * ZP + 15) / 16) * 16;
  static constexpr int DY_BASE = 0, Z_BASE = 2 * DY_PLANE, SCSH_BASE = Z_BASE + 2 * Z_PLANE;
  static constexpr int LDS_BYTES = SCSH_BASE + 128 * 4;
  static constexpr int NITD = (TP * 8 + 255) / 256;
  static constexpr int NITZ = (NPIX * 4 + 255) / 256;
};

template <typename T, int KS, int STRIDE, int TH>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradP p) {
  using E = Elem<T>;
  using Frag = typename E::Frag;
  using SR = typename Stage<T>::R;
  using Cf = WgradCfg<T, KS, STRIDE, TH>;
  constexpr int TW = Cf::TW, TAPS = Cf::TAPS, HWD = Cf::HWD, NPIX = Cf::NPIX, PXB = Cf::PXB, ZP = Cf::ZP;
  constexpr int NITD = Cf::NITD, NITZ = Cf::NITZ;

  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  // ---- block -> work item: consecutive work items (the weight tiles of one pixel split) share an XCD / L2
  const unsigned nb = gridDim.x;                 // multiple of 8
  const int L = (int)((blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3));
  if (L >= p.nwork) return;
  const int ncombo = p.n_co_t * p.n_ci_t;
  const int split = L / ncombo, combo = L - split * ncombo;
  const int ct = combo / p.n_ci_t, cc = combo - ct * p.n_ci_t;
  const int co0 = ct * 64, ci0 = cc * 64;
  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(p.ntiles, t_begin + p.tiles_per_split);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cf = wave & 1, cif = wave >> 1;
  const int cin = p.C0 + p.C1;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x0, 0, p.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x1 ? p.x1 : p.x0), 0, p.x1 ? p.bytes1 : p.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.bytesdy, 0x00020000);

  // per-plane source of the 64 input channels of this weight tile: 0 = x0, 1 = x1, 2 = beyond the input (zeros)
  int psrc[2], pch[2], pcs[2];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    const int c = ci0 + 32 * pl;
    psrc[pl] = c < p.C0 ? 0 : (c < cin ? 1 : 2);
    pch[pl] = c < p.C0 ? c : c - p.C0;
    pcs[pl] = c < p.C0 ? p.C0 : p.C1;
  }
  const bool affine = p.scale != nullptr;
  const bool do_silu = p.silu != 0;
  const int Hc = p.upsample ? p.Hin * 2 : p.Hin, Wc = p.upsample ? p.Win * 2 : p.Win;
  const int sub4 = tid & 3;

  SR sdy[NITD], sz[2][NITZ];
  unsigned zvalid = 0;      // bit (pl*NITZ + i): the piece is an in-image pixel
  int n_staged = -1;        // sample whose scale/shift sit in LDS
  int n_next = -1;
  float scsh_reg = 0.f;

  auto decode = [&](int t, int& n, int& oy0, int& ox0) {
    const int per = p.tiles_x * p.tiles_y;
    n = t / per;
    const int rem = t - n * per;
    const int ty = rem / p.tiles_x;
    oy0 = ty * TH; ox0 = (rem - ty * p.tiles_x) * TW;
  };

  auto issue = [&](int t) {
    int n, oy0, ox0;
    decode(t, n, oy0, ox0);
    n_next = n;
#pragma unroll
    for (int i = 0; i < NITD; ++i) {
      const int q = tid + 256 * i, pix = q >> 3, sub = q & 7;
      const int oy = oy0 + (pix >> 4), ox = ox0 + (pix & 15);
      const bool ok = pix < Cf::TP && oy < p.Hout && ox < p.Wout && (co0 + sub * 8) < p.Cout;
      const unsigned off = ok ? (unsigned)((((n * p.Hout * p.dy_step + oy * p.dy_step + p.dy_oy) * (p.Wout * p.dy_step) + ox * p.dy_step + p.dy_ox) * p.Cout + co0 + sub * 8) * E::BYTES) : OOB_OFF;
      sdy[i] = Stage<T>::load(rsd, off);
    }
    zvalid = 0;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
      for (int i = 0; i < NITZ; ++i) {
        const int q = tid + 256 * i, pix = q >> 2;
        unsigned off = OOB_OFF;
        if (pix < NPIX && psrc[pl] < 2) {
          const int u = pix / HWD, v = pix - u * HWD;
          const int iy = oy0 * STRIDE - p.pad + u, ix = ox0 * STRIDE - p.pad_x + v;
          if (iy >= 0 && iy < Hc && ix >= 0 && ix < Wc) {
            const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
            off = (unsigned)((((n * p.Hin + sy) * p.Win + sx) * pcs[pl] + pch[pl] + sub4 * 8) * E::BYTES);
            zvalid |= 1u << (pl * NITZ + i);
          }
        }
        sz[pl][i] = psrc[pl] == 0 ? Stage<T>::load(rs0, off) : Stage<T>::load(rs1, off);
      }
    }
    if (affine && n != n_staged && tid < 128) {
      const int c = ci0 + (tid & 63);
      scsh_reg = c < cin ? (tid < 64 ? p.scale : p.shift)[(size_t)n * cin + c] : 0.f;
    }
  };

  auto write_tile = [&]() {
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < NITD; ++i) {
      const int q = tid + 256 * i, pix = q >> 3, sub = q & 7;
      if (pix < Cf::TP)
        Stage<T>::xform_store(lds + Cf::DY_BASE + (sub >> 2) * Cf::DY_PLANE + pix * PXB + (sub & 3) * 8 * E::BYTES, sdy[i], sc, sh,
                              false, false, true);
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      if (affine) {
        const float* ps = (const float*)(lds + Cf::SCSH_BASE) + pl * 32 + sub4 * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = ps[j]; sh[j] = ps[64 + j]; }
      }
#pragma unroll
      for (int i = 0; i < NITZ; ++i) {
        const int q = tid + 256 * i, pix = q >> 2;
        if (pix < NPIX)
          Stage<T>::xform_store(lds + Cf::Z_BASE + pl * Cf::Z_PLANE + pix * ZP + sub4 * 8 * E::BYTES, sz[pl][i], sc, sh, affine,
                                do_silu, (zvalid >> (pl * NITZ + i)) & 1u);
      }
    }
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int k = 0; k < TAPS; ++k) acc[k] = (f32x16)(0.f);

  const unsigned char* a_base = lds + Cf::DY_BASE + cf * Cf::DY_PLANE + FragLd<T>::lane_off(lane, PXB);
  const unsigned char* b_base = lds + Cf::Z_BASE + cif * Cf::Z_PLANE + FragLd<T>::lane_off(lane, STRIDE * ZP);

  auto compute = [&]() {
#pragma unroll
    for (int y = 0; y < TH; ++y) {
      const Frag a = FragLd<T>::template load<PXB>(a_base + y * TW * PXB);
#pragma unroll
      for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const Frag b = FragLd<T>::template load<STRIDE * ZP>(b_base + ((y * STRIDE + ky) * HWD + kx) * ZP);
          acc[ky * KS + kx] = E::mma(a, b, acc[ky * KS + kx]);
        }
    }
  };

  auto publish_scsh = [&]() {     // workgroup-uniform: the next tile belongs to another sample
    if (affine && n_next != n_staged) {
      if (tid < 128) ((float*)(lds + Cf::SCSH_BASE))[tid] = scsh_reg;
      n_staged = n_next;
      __syncthreads();
    }
  };

  if (t_begin < t_end) {
    issue(t_begin);
    publish_scsh();
    write_tile();
    __syncthreads();
  }
  for (int t = t_begin; t < t_end; ++t) {
    const bool have_next = t + 1 < t_end;
    if (have_next) issue(t + 1);
    __builtin_amdgcn_s_setprio(1);
    compute();
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    if (have_next) { publish_scsh(); write_tile(); }
    __syncthreads();
  }

  // ---- partial tile -> slab[split][tap][co][ci]
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int k = 0; k < TAPS; ++k) {
    float* out = p.slab + (((size_t)split * TAPS + k) * p.COP + co0 + cf * 32) * p.CIP + ci0 + cif * 32 + r;
#pragma unroll
    for (int g = 0; g < 16; ++g) out[(size_t)((g & 3) + 8 * (g >> 2) + 4 * h) * p.CIP] = acc[k][g];
  }
}

// dw[co][ci][tap] (+)= sum over splits of slab[split][tap][co][ci], splits added in a fixed order (bitwise reproducible).
// Workgroup = one co x 64 consecutive ci x all taps: the slab rows are read coalesced (ci fastest), transposed through LDS
// and written as one contiguous run of 64*taps floats of dw (for a 3x3 conv the taps are the fastest dw index).
// Sub-pixel phase (a, b) of an upsampler's convolution (pd_wgrad_args.phase = 1 + 2 a + b): the slab holds the gradient of the phase's 2x2
// kernel, whose tap (u, v) is the SUM of the 3x3 taps {rows R_a(u)} x {columns R_b(v)}, R_0 = {0 | 1, 2}, R_1 = {0, 1 | 2}
// (packing.upsample_phase_weights) -- so its gradient is ADDED to each of those 3x3 taps.  Within one phase the four (u, v) cover the nine
// taps exactly once (no two threads touch the same element); the four phases are four launches on one stream.
__device__ __forceinline__ void phase_scatter(float* out9, int phase, int k, float v, bool add) {
  const int a = (phase - 1) >> 1, b = (phase - 1) & 1, u = k >> 1, w = k & 1;
  const int y0 = a == 0 ? (u == 0 ? 0 : 1) : (u == 0 ? 0 : 2), y1 = a == 0 ? (u == 0 ? 0 : 2) : (u == 0 ? 1 : 2);
  const int x0 = b == 0 ? (w == 0 ? 0 : 1) : (w == 0 ? 0 : 2), x1 = b == 0 ? (w == 0 ? 0 : 2) : (w == 0 ? 1 : 2);
  for (int ky = y0; ky <= y1; ++ky)
    for (int kx = x0; kx <= x1; ++kx) out9[ky * 3 + kx] = add ? out9[ky * 3 + kx] + v : v;
}

template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits,
                                                            int COP, int CIP, int cout_valid, int cin_valid, int accumulate, int phase = 0) {
  __shared__ float tile[64 * TAPS];
  const int tid = threadIdx.x;
  const int ci_blocks = CIP / 64;
  const int co = blockIdx.x / ci_blocks, ci0 = (blockIdx.x - co * ci_blocks) * 64;
  if (co >= cout_valid) return;                        // workgroup-uniform
  const size_t per = (size_t)TAPS * COP * CIP;
  for (int e = tid; e < 64 * TAPS; e += 256) {
    const int k = e >> 6, ci = e & 63;
    const float* src = slab + ((size_t)k * COP + co) * CIP + ci0 + ci;
    float s0 = 0.f, s1 = 0.f;                          // two chains: the loads of consecutive splits overlap
    int sp = 0;
    for (; sp + 2 <= splits; sp += 2) { s0 += src[(size_t)sp * per]; s1 += src[(size_t)(sp + 1) * per]; }
    if (sp < splits) s0 += src[(size_t)sp * per];
    tile[ci * TAPS + k] = s0 + s1;
  }
  __syncthreads();
  const int nci = min(64, cin_valid - ci0);            // may be <= 0 for padded input channels
  if (TAPS == 4 && phase) {
    float* out3 = dw + ((size_t)co * cin_valid + ci0) * 9;
    for (int e = tid; e < nci * TAPS; e += 256) phase_scatter(out3 + (e >> 2) * 9, phase, e & 3, tile[e], accumulate || phase > 1);
    return;
  }
  float* out = dw + ((size_t)co * cin_valid + ci0) * TAPS;
  for (int e = tid; e < nci * TAPS; e += 256) out[e] = accumulate ? out[e] + tile[e] : tile[e];
}

// Many splits (small weights, many pixels): block = 64 consecutive slab elements x 4 split groups, combined in a fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_split_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits, int taps,
                                                                  int COP, int CIP, int cout_valid, int cin_valid, int accumulate, int phase = 0) {
  __shared__ float red[256];
  const int tid = threadIdx.x, sg = tid >> 6;
  const size_t per = (size_t)taps * COP * CIP;
  const size_t e = (size_t)blockIdx.x * 64 + (tid & 63);       // (tap, co, ci) in slab order; per is a multiple of 64
  float s = 0.f;
  for (int sp = sg; sp < splits; sp += 4) s += slab[(size_t)sp * per + e];
  red[tid] = s;
  __syncthreads();
  if (sg == 0) {
    s = ((red[tid] + red[tid + 64]) + red[tid + 128]) + red[tid + 192];
    const int ci = (int)(e % CIP);
    const int co = (int)((e / CIP) % COP), k = (int)(e / ((size_t)CIP * COP));
    if (co < cout_valid && ci < cin_valid) {
      if (phase) phase_scatter(dw + ((size_t)co * cin_valid + ci) * 9, phase, k, s, accumulate || phase > 1);
      else {
        float* o = dw + ((size_t)co * cin_valid + ci) * taps + k;
        *o = accumulate ? *o + s : s;
      }
    }
  }
}

// 3x3 taps of a <=3-channel fp32 NCHW image gathered into 32 channels k = ci*9 + ky*3 + kx (NHWC): the input the forward's
// conv_in im2col mode builds in LDS, materialised for the weight gradient of conv_in (cond_unet_2d.py:127-129)
template <typename T>
__global__ __launch_bounds__(256) void im2col3_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int H, int W, int Cr) {
  const size_t total = (size_t)B * H * W * 4;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int sub = (int)(idx & 3);
    size_t pix = idx >> 2;
    const int xx = (int)(pix % W); pix /= W;
    const int yy = (int)(pix % H); const int n = (int)(pix / H);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = sub * 8 + j, ci = k / 9, tap = k - ci * 9, ky = tap / 3, kx = tap - ky * 3;
      const int iy = yy + ky - 1, ix = xx + kx - 1;
      v[j] = (ci < Cr && iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[(((size_t)n * Cr + ci) * H + iy) * W + ix] : 0.f;
    }
    Elem<T>::store(out + (idx >> 2) * 32 + sub * 8, Elem<T>::pack(v));
  }
}

// OIHW fp32 master weights -> pd_conv's packed fragment order [ct][chunk][tap][s][lane][j] (packing.py), optionally as the
// input-gradient weights W'[ci][co][K-1-ky][K-1-kx]; thread = one 8-element lane fragment
template <typename T>
__device__ __forceinline__ void pack_weight_block(const pd_pack_weight_args& a, int block, float* tile) {
  // workgroup = one (32-co tile, 32-ci chunk) of the packed matrix, all taps: the 32 x 32 x taps source block is read as 32
  // contiguous runs (coalesced), staged in LDS, and every lane fragment is gathered from there
  // tile: [row o][col i * taps + t], row pitch 32*taps + 1
  const int taps = a.ksize * a.ksize;
  const int chunks = a.cin_pad / 32;
  const int ct = block / chunks, chunk = block - ct * chunks;
  const int tid = threadIdx.x;
  const int run = 32 * taps, pitch = run + 1;
  // source block: forward rows o = co (ct), cols i = ci (chunk); input-gradient rows o = ci (chunk), cols i = co (ct)
  const int o0 = a.dgrad ? chunk * 32 : ct * 32, i0 = a.dgrad ? ct * 32 : chunk * 32;
  const int n_o = a.dgrad ? a.cin : a.cout, n_i = a.dgrad ? a.cout : a.cin;       // valid source rows / cols
  if (taps == 1 && o0 + 32 <= n_o && i0 + 32 <= n_i && (a.src_in & 3) == 0 && (((size_t)a.src) & 15) == 0) {
    // 1x1 / Linear weights, whole block valid (workgroup-uniform): the 32 x 32 block is 256 float4 = ONE 16-byte load per thread
    // (round 3: four 4-byte loads per thread and a workgroup's worth of launch overhead per KiB made the re-pack of the SD UNet's
    // Linear layers run at a quarter of the memory rate)
    const int ro = tid >> 3, c4 = (tid & 7) * 4;
    const f32x4 x = *(const f32x4*)(a.src + (size_t)(o0 + ro) * a.src_in + i0 + c4);
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[ro * pitch + c4 + j] = x[j];
  } else {
    for (int e = tid; e < 32 * run; e += 256) {
      const int ro = e / run, c = e - ro * run;          // c = i_local * taps + t
      const int i = i0 + c / taps;
      float x = 0.f;
      if (o0 + ro < n_o && i < n_i) x = a.src[((size_t)(o0 + ro) * a.src_in + i0) * taps + c];
      tile[ro * pitch + c] = x;
    }
  }
  __syncthreads();
  for (int f = tid; f < taps * 2 * 64; f += 256) {
    const int lane = f & 63, sidx = (f >> 6) & 1, tap = f >> 7;
    const int r = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cl = sidx * 16 + h * 8 + j;            // ci within the chunk; co within the tile = r
      v[j] = a.dgrad ? tile[cl * pitch + r * taps + (taps - 1 - tap)] : tile[r * pitch + cl * taps + tap];
    }
    T* dst = (T*)a.dst + (size_t)ct * a.dst_ct_stride + ((((size_t)chunk * taps + tap) * 2 + sidx) * 64 + lane) * 8;
    Elem<T>::store(dst, Elem<T>::pack(v));
  }
  if (!a.dgrad && a.dst2) {
    // round 6: the input-gradient packing of the same 32 x 32 source block (its tile (ci chunk) x chunk (co tile) transposed, taps flipped)
    // from the LDS tile already loaded: the separate dgrad job read every master weight a second time
    for (int f = tid; f < taps * 2 * 64; f += 256) {
      const int lane = f & 63, sidx = (f >> 6) & 1, tap = f >> 7;
      const int r = lane & 31, h = lane >> 5;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int cl = sidx * 16 + h * 8 + j;            // co within this block = the dgrad matrix's k index; its row = ci = r
        v[j] = tile[cl * pitch + r * taps + (taps - 1 - tap)];
      }
      T* dst = (T*)a.dst2 + (size_t)chunk * a.dst2_ct_stride + ((((size_t)ct * taps + tap) * 2 + sidx) * 64 + lane) * 8;
      Elem<T>::store(dst, Elem<T>::pack(v));
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const pd_pack_weight_args a) {
  extern __shared__ float tile[];
  pack_weight_block<T>(a, blockIdx.x, tile);
}
// every re-pack of an optimizer step as ONE launch: block b belongs to job j with starts[j] <= b < starts[j+1] (binary search
// over <= a few hundred jobs), its descriptor is read from device memory
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const pd_pack_weight_args* jobs, const int* starts, int n) {
  extern __shared__ float tile[];
  const int b = blockIdx.x;
  int lo = 0, hi = n;                                  // invariant: starts[lo] <= b < starts[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (starts[mid] <= b) lo = mid; else hi = mid;
  }
  const pd_pack_weight_args a = jobs[lo];
  pack_weight_block<T>(a, b - starts[lo], tile);
}

static int pick_splits(int ntiles, int ncombo) {
  // fill the 256 CUs x 2 resident workgroups ONCE, from below: rounding up (rounds 1-3) made grids of 513-528 workgroups for
  // 3, 6, 12, 24 or 25 channel-tile combinations (the up path's concatenated inputs; every 320-wide SD layer) = a second,
  // nearly empty round of workgroups
  int want = 512 / ncombo;
  int amort = ntiles / 8 > 1 ? ntiles / 8 : 1;      // >= 8 pixel tiles per slab written
  int s = want < amort ? want : amort;
  if (s > ntiles) s = ntiles;
  return s < 1 ? 1 : s;
}

template <typename T, int KS, int STRIDE, int TH>
static int launch_wgrad(const pd_wgrad_args* a, hipStream_t st) {
  using Cf = WgradCfg<T, KS, STRIDE, TH>;
  static_assert(Cf::LDS_BYTES <= 80 * 1024, "two workgroups per CU");
  WgradP p;
  p.B = a->B; p.Hin = a->Hin; p.Win = a->Win; p.Hout = a->Hout; p.Wout = a->Wout; p.C0 = a->C0; p.C1 = a->C1; p.Cout = a->Cout;
  p.pad = a->pad; p.upsample = a->upsample; p.silu = a->silu;
  p.pad_x = a->pad; p.dy_step = 1; p.dy_oy = 0; p.dy_ox = 0;
  if (a->phase) {      // rows start at oy - (1 - a), columns at ox - (1 - b); dy = the phase's pixels of the upsampled output's gradient
    const int pa = (a->phase - 1) >> 1, pb = (a->phase - 1) & 1;
    p.pad = 1 - pa; p.pad_x = 1 - pb; p.dy_step = 2; p.dy_oy = pa; p.dy_ox = pb;
  }
  p.tiles_x = (a->Wout + Cf::TW - 1) / Cf::TW; p.tiles_y = (a->Hout + TH - 1) / TH;
  p.ntiles = a->B * p.tiles_x * p.tiles_y;
  const int cin = a->C0 + a->C1;
  p.n_co_t = (a->Cout + 63) / 64; p.n_ci_t = (cin + 63) / 64;
  p.COP = p.n_co_t * 64; p.CIP = p.n_ci_t * 64;
  const int ncombo = p.n_co_t * p.n_ci_t;
  const size_t per_split = (size_t)Cf::TAPS * p.COP * p.CIP * sizeof(float);
  int splits = pick_splits(p.ntiles, ncombo);
  if ((size_t)splits * per_split > a->slab_bytes) splits = (int)(a->slab_bytes / per_split);
  PD_CHECK(splits >= 1, PD_ERR_ARG, "pd_conv_wgrad: slab of %zu bytes cannot hold one split (%zu bytes)", a->slab_bytes, per_split);
  p.splits = splits;
  p.tiles_per_split = (p.ntiles + splits - 1) / splits;
  p.nwork = splits * ncombo;
  const size_t es = sizeof(T);
  p.bytes0 = (unsigned)((size_t)a->B * a->Hin * a->Win * a->C0 * es);
  p.bytes1 = (unsigned)((size_t)a->B * a->Hin * a->Win * a->C1 * es);
  p.bytesdy = (unsigned)((size_t)a->B * a->Hout * a->Wout * a->Cout * es * (a->phase ? 4 : 1));
  p.x0 = a->x0; p.x1 = a->x1; p.scale = a->scale; p.shift = a->shift; p.dy = a->dy; p.slab = a->slab;
  auto kern = wgrad_kernel<T, KS, STRIDE, TH>;
  static LdsAttr attr;
  if (!ensure_lds(attr, kern, Cf::LDS_BYTES)) {
    set_error("pd_conv_wgrad: cannot reserve %d bytes of LDS", Cf::LDS_BYTES);
    return PD_ERR_LAUNCH;
  }
  const unsigned grid = (unsigned)((p.nwork + 7) / 8 * 8);
  if (a->stage != 2) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cf::LDS_BYTES, st, p);
    PD_LAUNCH_CHECK();
  }
  if (a->stage == 1) return PD_OK;
  const int cout_v = a->Cout_valid > 0 ? a->Cout_valid : a->Cout, cin_v = a->Cin_valid > 0 ? a->Cin_valid : cin;
  if (splits > 8)     // small weights spread over many pixel splits: parallelise over the splits
    hipLaunchKernelGGL(wgrad_reduce_split_kernel, dim3((unsigned)((size_t)Cf::TAPS * p.COP * p.CIP / 64)), dim3(256), 0, st,
                       (const float*)a->slab, a->dw, splits, Cf::TAPS, p.COP, p.CIP, cout_v, cin_v, a->accumulate, a->phase);
  else                // large weights (few splits): coalesced transposing copy
    hipLaunchKernelGGL(wgrad_reduce_kernel<Cf::TAPS>, dim3((unsigned)((size_t)p.COP * (p.CIP / 64))), dim3(256), 0, st, (const float*)a->slab,
                       a->dw, splits, p.COP, p.CIP, cout_v, cin_v, a->accumulate, a->phase);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

}  // namespace pd

using namespace pd;

extern "C" size_t pd_conv_wgrad_workspace(const pd_wgrad_args* a) {
  if (!a || a->ksize < 1 || a->B < 1) return 0;
  const int th = (a->dtype != PD_F32 ? 8 : 4) / (a->stride == 2 ? 2 : 1);
  const int ntiles = a->B * ((a->Wout + 15) / 16) * ((a->Hout + th - 1) / th);
  const int nco = (a->Cout + 63) / 64, nci = (a->C0 + a->C1 + 63) / 64;
  return (size_t)pick_splits(ntiles, nco * nci) * a->ksize * a->ksize * nco * 64 * nci * 64 * sizeof(float);
}

extern "C" int pd_conv_wgrad(const pd_wgrad_args* a, void* stream) {
  PD_CHECK(a != nullptr, PD_ERR_ARG, "pd_conv_wgrad: null args");
  PD_CHECK(a->dtype == PD_F32 || a->dtype == PD_BF16 || a->dtype == PD_F16, PD_ERR_ARG, "pd_conv_wgrad: bad dtype");
  PD_CHECK(a->B > 0 && a->Hin > 0 && a->Win > 0 && a->Hout > 0 && a->Wout > 0, PD_ERR_SHAPE, "pd_conv_wgrad: bad shape");
  PD_CHECK(a->stage >= 0 && a->stage <= 2, PD_ERR_ARG, "pd_conv_wgrad: stage %d", a->stage);
  PD_CHECK(a->C0 > 0 && a->C0 % 32 == 0 && a->C1 >= 0 && a->C1 % 32 == 0, PD_ERR_SHAPE, "pd_conv_wgrad: C0=%d C1=%d must be multiples of 32", a->C0, a->C1);
  PD_CHECK(a->Cout > 0 && a->Cout % 8 == 0, PD_ERR_SHAPE, "pd_conv_wgrad: Cout=%d (channel stride of dy) must be a multiple of 8", a->Cout);
  PD_CHECK(a->phase >= 0 && a->phase <= 4, PD_ERR_ARG, "pd_conv_wgrad: phase %d", a->phase);
  if (a->phase) {
    PD_CHECK(a->ksize == 2 && a->stride == 1 && !a->upsample && !a->scale && a->C1 == 0 && a->Hout == a->Hin && a->Wout == a->Win && !a->Cout_valid && !a->Cin_valid,
             PD_ERR_UNSUPPORTED, "pd_conv_wgrad: a sub-pixel phase is a plain 2x2 weight gradient over one source with Hout = Hin, Wout = Win");
    PD_CHECK(a->x0 && a->dy && a->slab && a->dw, PD_ERR_ARG, "pd_conv_wgrad: null pointer");
    PD_CHECK((size_t)a->B * a->Hout * a->Wout * 4 * a->Cout * (a->dtype == PD_F32 ? 4 : 2) < ((size_t)1 << 31) && (size_t)a->B * a->Hin * a->Win * a->C0 * (a->dtype == PD_F32 ? 4 : 2) < ((size_t)1 << 31),
             PD_ERR_SHAPE, "pd_conv_wgrad: tensors must be < 2 GiB (32-bit buffer offsets)");
    hipStream_t stp = (hipStream_t)stream;
    if (a->dtype == PD_F16) return launch_wgrad<half_t, 2, 1, 8>(a, stp);
    return a->dtype == PD_BF16 ? launch_wgrad<bf16_t, 2, 1, 8>(a, stp) : launch_wgrad<float, 2, 1, 4>(a, stp);
  }
  PD_CHECK((a->ksize == 3 && (a->stride == 1 || a->stride == 2)) || (a->ksize == 1 && a->stride == 1 && a->pad == 0), PD_ERR_SHAPE,
           "pd_conv_wgrad: unsupported ksize=%d stride=%d", a->ksize, a->stride);
  PD_CHECK(!(a->upsample && a->stride != 1) && (a->upsample == 0 || a->upsample == 1), PD_ERR_SHAPE, "pd_conv_wgrad: bad upsample");
  const int hc = a->upsample ? 2 * a->Hin : a->Hin, wc = a->upsample ? 2 * a->Win : a->Win;
  const int extra = (a->ksize == 3 && a->pad == 0) ? 1 : 0;
  PD_CHECK(a->pad == 0 || a->pad == 1, PD_ERR_SHAPE, "pd_conv_wgrad: pad=%d", a->pad);
  PD_CHECK(a->Hout == (hc + 2 * a->pad + extra - a->ksize) / a->stride + 1 && a->Wout == (wc + 2 * a->pad + extra - a->ksize) / a->stride + 1,
           PD_ERR_SHAPE, "pd_conv_wgrad: Hout/Wout inconsistent with the forward convolution");
  PD_CHECK(a->x0 && a->dy && a->slab && a->dw, PD_ERR_ARG, "pd_conv_wgrad: null pointer");
  PD_CHECK((a->C1 == 0) == (a->x1 == nullptr), PD_ERR_ARG, "pd_conv_wgrad: x1/C1 mismatch");
  PD_CHECK((a->scale == nullptr) == (a->shift == nullptr), PD_ERR_ARG, "pd_conv_wgrad: scale/shift must be given together");
  const size_t es = a->dtype == PD_F32 ? 4 : 2;
  const size_t lim = (size_t)1 << 31;
  PD_CHECK((size_t)a->B * a->Hin * a->Win * (a->C0 > a->C1 ? a->C0 : a->C1) * es < lim && (size_t)a->B * a->Hout * a->Wout * a->Cout * es < lim,
           PD_ERR_SHAPE, "pd_conv_wgrad: tensors must be < 2 GiB (32-bit buffer offsets)");
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == PD_BF16) {
    if (a->ksize == 1) return launch_wgrad<bf16_t, 1, 1, 8>(a, st);
    if (a->stride == 1) return launch_wgrad<bf16_t, 3, 1, 8>(a, st);
    return launch_wgrad<bf16_t, 3, 2, 4>(a, st);
  }
  if (a->dtype == PD_F16) {      // fp16 training (round 5: --mixed_precision fp16, launch_script_DDIM.sh:56): the same kernel on the f16 MFMA
    if (a->ksize == 1) return launch_wgrad<half_t, 1, 1, 8>(a, st);
    if (a->stride == 1) return launch_wgrad<half_t, 3, 1, 8>(a, st);
    return launch_wgrad<half_t, 3, 2, 4>(a, st);
  }
  if (a->ksize == 1) return launch_wgrad<float, 1, 1, 4>(a, st);
  if (a->stride == 1) return launch_wgrad<float, 3, 1, 4>(a, st);
  return launch_wgrad<float, 3, 2, 2>(a, st);
}

extern "C" int pd_im2col3(const pd_im2col3_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->B > 0 && a->H > 0 && a->W > 0 && a->C >= 1 && a->C <= 3 && a->x && a->out, PD_ERR_ARG, "pd_im2col3: bad args");
  const size_t total = (size_t)a->B * a->H * a->W * 4;
  const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (a->dtype == PD_F32) hipLaunchKernelGGL(im2col3_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a->x, (float*)a->out, a->B, a->H, a->W, a->C);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(im2col3_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a->x, (bf16_t*)a->out, a->B, a->H, a->W, a->C);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(im2col3_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a->x, (half_t*)a->out, a->B, a->H, a->W, a->C);
  else { set_error("pd_im2col3: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}

// pd_upsample_phase_weights: thread = one (o, i) pair: 9 taps in, 4 x 4 phase taps out (rows combined first, then columns: the order of the
// host-side contraction R_a w R_b^T it replaces -- torch.einsum ran it as two hipBLASLt GEMMs + copies, 1.8 ms per fine-tuning step)
__global__ __launch_bounds__(256) void upsample_phase_weights_kernel(const pd_upsample_phase_weights_args a) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)a.cout * a.cin;
  if (idx >= total) return;
  float w[3][3];
#pragma unroll
  for (int y = 0; y < 3; ++y)
#pragma unroll
    for (int x = 0; x < 3; ++x) w[y][x] = a.w[idx * 9 + y * 3 + x];
#pragma unroll
  for (int pa = 0; pa < 2; ++pa) {
    float t[2][3];                                   // rows: phase a = 0: {0}, {1 + 2};  a = 1: {0 + 1}, {2}
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      t[0][x] = pa == 0 ? w[0][x] : w[0][x] + w[1][x];
      t[1][x] = pa == 0 ? w[1][x] + w[2][x] : w[2][x];
    }
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      f32x4 o;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        o[2 * u] = pb == 0 ? t[u][0] : t[u][0] + t[u][1];
        o[2 * u + 1] = pb == 0 ? t[u][1] + t[u][2] : t[u][2];
      }
      *(f32x4*)(a.out + ((size_t)(2 * pa + pb) * total + idx) * 4) = o;
    }
  }
}

extern "C" int pd_upsample_phase_weights(const pd_upsample_phase_weights_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->cout > 0 && a->cin > 0 && a->w && a->out, PD_ERR_ARG, "pd_upsample_phase_weights: bad args");
  PD_CHECK(((size_t)a->out & 15) == 0, PD_ERR_ARG, "pd_upsample_phase_weights: out must be 16-byte aligned");
  const size_t total = (size_t)a->cout * a->cin;
  hipLaunchKernelGGL(upsample_phase_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_pack_weight_batch(const pd_pack_weight_batch_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->jobs && a->starts && a->n > 0 && a->total_blocks > 0 && a->max_ksize >= 1, PD_ERR_ARG, "pd_pack_weight_batch: bad args");
  const size_t lds = (size_t)32 * (32 * a->max_ksize * a->max_ksize + 1) * sizeof(float);
  PD_CHECK(lds <= 64 * 1024, PD_ERR_SHAPE, "pd_pack_weight_batch: max_ksize %d", a->max_ksize);
  const dim3 grid((unsigned)a->total_blocks);
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == PD_F32) hipLaunchKernelGGL(pack_weight_batch_kernel<float>, grid, dim3(256), lds, st, a->jobs, a->starts, a->n);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(pack_weight_batch_kernel<bf16_t>, grid, dim3(256), lds, st, a->jobs, a->starts, a->n);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(pack_weight_batch_kernel<half_t>, grid, dim3(256), lds, st, a->jobs, a->starts, a->n);
  else { set_error("pd_pack_weight_batch: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_pack_weight(const pd_pack_weight_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->src && a->dst && a->cout > 0 && a->cin > 0 && a->ksize >= 1 && a->src_in > 0, PD_ERR_ARG, "pd_pack_weight: bad args");
  PD_CHECK(a->cout_pad % 32 == 0 && a->cin_pad % 32 == 0 && a->cout_pad >= a->cout && a->cin_pad >= a->cin, PD_ERR_SHAPE,
           "pd_pack_weight: padded sizes must be multiples of 32 (cout %d/%d, cin %d/%d)", a->cout, a->cout_pad, a->cin, a->cin_pad);
  const size_t per_ct = (size_t)(a->cin_pad / 32) * a->ksize * a->ksize * 2 * 64 * 8;
  PD_CHECK((size_t)a->dst_ct_stride >= per_ct, PD_ERR_ARG, "pd_pack_weight: dst_ct_stride too small");
  PD_CHECK(a->dst2 == nullptr || (a->dgrad == 0 && (size_t)a->dst2_ct_stride >= (size_t)(a->cout_pad / 32) * a->ksize * a->ksize * 2 * 64 * 8), PD_ERR_ARG,
           "pd_pack_weight: dst2 needs dgrad = 0 and dst2_ct_stride >= (cout_pad/32) * ksize^2 * 1024");
  const unsigned grid = (unsigned)((a->cout_pad / 32) * (a->cin_pad / 32));
  const size_t lds = (size_t)32 * (32 * a->ksize * a->ksize + 1) * sizeof(float);
  if (a->dtype == PD_F32) hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), lds, (hipStream_t)stream, *a);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(grid), dim3(256), lds, (hipStream_t)stream, *a);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(pack_weight_kernel<half_t>, dim3(grid), dim3(256), lds, (hipStream_t)stream, *a);
  else { set_error("pd_pack_weight: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}
