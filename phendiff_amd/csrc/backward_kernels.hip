// Backward building blocks of the UNet, bandwidth-bound ones:
//   pd_gn_silu_bwd : gradient through  z = silu?(GroupNorm(x))  (ResnetBlock2D.norm1/norm2 + nonlinearity, Attention.group_norm,
//                    conv_norm_out): dz -> dx (per source of a channel concat), dgamma, dbeta
//   pd_pool2x2_sum : gradient of the nearest x2 upsample fused into Upsample2D's conv (sum over each 2x2 block)
//   pd_channel_sum : per-(sample, channel) sums of an NHWC tensor (bias gradients, time-embedding projection gradients)
// The MFMA pieces live elsewhere: input gradients reuse pd_conv with flipped/transposed weights (packing.dgrad_weight),
// weight gradients are pd_conv_wgrad (wgrad.hip).
#include "pd_common.h"

namespace pd {

__device__ __forceinline__ float dsilu(float y) {          // d/dy [y * sigmoid(y)]
  const float s = 1.0f / (1.0f + __expf(-y));
  return s * (1.0f + y * (1.0f - s));
}
// bf16 path: v_exp_f32 + v_rcp_f32 (as the forward's silu_fast); fp32 validation path: accurate division
template <typename T> __device__ __forceinline__ float dsilu_t(float y);
template <> __device__ __forceinline__ float dsilu_t<float>(float y) { return dsilu(y); }
template <> __device__ __forceinline__ float dsilu_t<bf16_t>(float y) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.4426950408889634f));
  return s * (1.0f + y * (1.0f - s));
}
template <> __device__ __forceinline__ float dsilu_t<half_t>(float y) { return dsilu_t<bf16_t>(y); }   // fp16 training (round 5): the 16-bit form

// Thread -> (8-channel piece, pixel row) of one (sample, pixel split); the per-channel constants of the piece sit in
// registers for the whole pixel loop.
constexpr int GN_CHUNK = 1024;
constexpr int GN_MAXC = 3072;

template <typename T>
struct GnPiece {
  int c8, cs, coff, dcs, dcoff;
  bool first, active;
  const T* xs; const T* ds;
  float mu[8], rs[8], sc[8], sh[8];
  // channels are walked in chunks of GN_CHUNK (blockIdx.y): cb = first channel of this workgroup's chunk, PP = its pieces
  __device__ __forceinline__ void init(const pd_gn_bwd_args& a, int n, int tid, int cb, int PP, int& prow, int& ppi) {
    const int C = a.C0 + a.C1, gs = C / a.groups;
    ppi = 256 / PP;
    active = tid < ppi * PP;
    const int piece = tid % PP;
    prow = tid / PP;
    c8 = cb + piece * 8;
    first = c8 < a.C0;
    xs = (const T*)(first ? a.x0 : a.x1);
    ds = (const T*)((first || a.dz_combined) ? a.dz0 : a.dz1);
    cs = first ? a.C0 : a.C1; coff = first ? c8 : c8 - a.C0;
    dcs = a.dz_combined ? C : cs; dcoff = a.dz_combined ? c8 : coff;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = (c8 + j) / gs;
      mu[j] = a.mean[n * a.groups + g]; rs[j] = a.rstd[n * a.groups + g];
      float gm = a.gamma[c8 + j], bt = a.beta[c8 + j];
      if (a.mod) {            // scale_shift ResNet: the affine of THIS sample, gamma (1 + scale_n), beta (1 + scale_n) + shift_n
        const float* row = a.mod + (size_t)n * a.mod_stride;
        const float s1p = 1.0f + row[c8 + j];
        gm *= s1p; bt = bt * s1p + row[C + c8 + j];
      }
      sc[j] = rs[j] * gm; sh[j] = bt - mu[j] * sc[j];     // y = sc*x + sh (the forward's affine)
    }
  }
};

// (1) per-(sample, split, channel) partial sums of dy and dy * xhat, coalesced 8-channel pieces
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const pd_gn_bwd_args a) {
  using E = Elem<T>;
  __shared__ float red[256 * 16];
  const int C = a.C0 + a.C1, cb = blockIdx.y * GN_CHUNK, Cc = min(C - cb, GN_CHUNK), PP = Cc / 8;
  const int n = blockIdx.x / a.splits, split = blockIdx.x % a.splits;
  const int per = (a.HW + a.splits - 1) / a.splits;
  const int p0 = split * per, p1 = min(a.HW, p0 + per);
  const int tid = threadIdx.x;
  GnPiece<T> g;
  int prow, ppi;
  g.init(a, n, tid, cb, PP, prow, ppi);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  if (g.active) {
    // two pixels per iteration: four 16-byte loads in flight per lane
    for (int p = p0 + prow; p < p1; p += 2 * ppi) {
      const bool two = p + ppi < p1;
      const int pb = two ? p + ppi : p;
      typename E::Frag fx0 = E::load(g.xs + ((size_t)n * a.HW + p) * g.cs + g.coff);
      typename E::Frag fd0 = E::load(g.ds + ((size_t)n * a.HW + p) * g.dcs + g.dcoff);
      typename E::Frag fx1 = E::load(g.xs + ((size_t)n * a.HW + pb) * g.cs + g.coff);
      typename E::Frag fd1 = E::load(g.ds + ((size_t)n * a.HW + pb) * g.dcs + g.dcoff);
      float xv[8], dv[8], xw[8], dw[8];
      E::unpack(fx0, xv); E::unpack(fd0, dv); E::unpack(fx1, xw); E::unpack(fd1, dw);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dy = a.silu ? dv[j] * dsilu_t<T>(g.sc[j] * xv[j] + g.sh[j]) : dv[j];
        s1[j] += dy; s2[j] += dy * ((xv[j] - g.mu[j]) * g.rs[j]);
        float dy2 = a.silu ? dw[j] * dsilu_t<T>(g.sc[j] * xw[j] + g.sh[j]) : dw[j];
        dy2 = two ? dy2 : 0.f;
        s1[j] += dy2; s2[j] += dy2 * ((xw[j] - g.mu[j]) * g.rs[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = s1[j]; red[tid * 16 + 8 + j] = s2[j]; }
  __syncthreads();
  for (int c = tid; c < Cc; c += 256) {
    const int piece = c >> 3, j = c & 7;
    double d1 = 0.0, d2 = 0.0;
    for (int k = 0; k < ppi; ++k) { d1 += (double)red[(k * PP + piece) * 16 + j]; d2 += (double)red[(k * PP + piece) * 16 + 8 + j]; }
    double* out = a.partial + (((size_t)n * a.splits + split) * C + cb + c) * 2;
    out[0] = d1; out[1] = d2;
  }
}

// (2) per-(sample, group) coefficients rstd*A/M, rstd*B/M and the parameter gradients dgamma, dbeta (+=)
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const pd_gn_bwd_args a) {
  __shared__ double s1[GN_MAXC], s2[GN_MAXC];
  const int C = a.C0 + a.C1, gs = C / a.groups, tid = threadIdx.x;
  if (blockIdx.x < (unsigned)a.B) {
    const int n = blockIdx.x;
    for (int c = tid; c < C; c += 256) {
      double d1 = 0.0, d2 = 0.0;
      for (int sp = 0; sp < a.splits; ++sp) {
        const double* in = a.partial + (((size_t)n * a.splits + sp) * C + c) * 2;
        d1 += in[0]; d2 += in[1];
      }
      double gm = (double)a.gamma[c];
      if (a.mod) {
        const float* row = a.mod + (size_t)n * a.mod_stride;
        if (a.dmod) {        // d scale_n = gamma sum(dy xhat) + beta sum(dy), d shift_n = sum(dy)
          float* drow = a.dmod + (size_t)n * a.mod_stride;
          drow[c] = (float)(gm * d2 + (double)a.beta[c] * d1);
          drow[C + c] = (float)d1;
        }
        gm *= 1.0 + (double)row[c];
      }
      s1[c] = d1 * gm; s2[c] = d2 * gm;
    }
    __syncthreads();
    if (tid < a.groups) {
      double A = 0.0, Bq = 0.0;
      for (int c = tid * gs; c < (tid + 1) * gs; ++c) { A += s1[c]; Bq += s2[c]; }
      const double M = (double)gs * (double)a.HW, r = (double)a.rstd[n * a.groups + tid];
      a.coef[(n * a.groups + tid) * 2] = (float)(r * A / M);
      a.coef[(n * a.groups + tid) * 2 + 1] = (float)(r * Bq / M);
    }
  } else {   // parameter gradients: FOUR channels per block (round 6: one channel per block read 16 bytes of every 64-byte sector of the partial --
             // 16 us per launch for a few hundred KB, 61 launches per SD-2.1 step), 64 term lanes each, combined in a fixed order
    const int cl = tid & 3, tl = tid >> 2;
    const int c = (blockIdx.x - a.B) * 4 + cl;
    double d1 = 0.0, d2 = 0.0;
    const int terms = a.B * a.splits;
    if (c < C) {
      for (int t = tl; t < terms; t += 64) {
        const double* in = a.partial + ((size_t)t * C + c) * 2;
        const double wgt = a.mod ? 1.0 + (double)a.mod[(size_t)(t / a.splits) * a.mod_stride + c] : 1.0;     // (1 + scale_n) of the term's sample
        d1 += wgt * in[0]; d2 += wgt * in[1];
      }
    }
#pragma unroll
    for (int msk = 4; msk < 64; msk <<= 1) { d1 += __shfl_xor(d1, msk); d2 += __shfl_xor(d2, msk); }     // the 16 term lanes of this wave that share channel cl
    if ((tid & 63) < 4) { s1[(tid >> 6) * 4 + cl] = d1; s2[(tid >> 6) * 4 + cl] = d2; }
    __syncthreads();
    if (tid < 4 && c < C) {
      if (a.dbeta) a.dbeta[c] += (float)(((s1[cl] + s1[4 + cl]) + s1[8 + cl]) + s1[12 + cl]);
      if (a.dgamma) a.dgamma[c] += (float)(((s2[cl] + s2[4 + cl]) + s2[8 + cl]) + s2[12 + cl]);
    }
  }
}

// (3) dx = sc*dy - rstd*A/M - xhat*rstd*B/M   (+ dx when accumulate, + res)
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const pd_gn_bwd_args a) {
  using E = Elem<T>;
  const int C = a.C0 + a.C1, gs = C / a.groups;
  const int n = blockIdx.x / a.splits, split = blockIdx.x % a.splits;
  const int per = (a.HW + a.splits - 1) / a.splits;
  const int p0 = split * per, p1 = min(a.HW, p0 + per);
  __shared__ float red[256 * 8];
  const int cb = blockIdx.y * GN_CHUNK, Cc = min(C - cb, GN_CHUNK);
  GnPiece<T> g;
  int prow, ppi;
  g.init(a, n, threadIdx.x, cb, Cc / 8, prow, ppi);
  T* dxp = (T*)(g.first ? a.dx0 : a.dx1);
  float* sump = g.first ? a.sum0 : a.sum1;         // optional per-(sample, split, channel) sums of the dx written here
  float cs_[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs_[j] = 0.f;
  if (g.active && dxp) {
  float ka[8], kb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int grp = (g.c8 + j) / gs;
    ka[j] = a.coef[(n * a.groups + grp) * 2]; kb[j] = a.coef[(n * a.groups + grp) * 2 + 1];
  }
  const bool accum = g.first ? a.accumulate0 : a.accumulate1;
  const bool has_res = a.res != nullptr;
  typedef typename E::Frag Frag;
  for (int p = p0 + prow; p < p1; p += 2 * ppi) {      // two pixels per iteration: up to eight 16-byte loads in flight
    const bool two = p + ppi < p1;
    size_t pixl[2] = {(size_t)n * a.HW + p, (size_t)n * a.HW + (two ? p + ppi : p)};
    Frag fx[2], fd[2], fa[2], fr[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t off = pixl[u] * g.cs + g.coff;
      fx[u] = E::load(g.xs + off);
      fd[u] = E::load(g.ds + pixl[u] * g.dcs + g.dcoff);
      if (accum) fa[u] = E::load(dxp + off);
      if (has_res) fr[u] = E::load((const T*)a.res + pixl[u] * C + g.c8);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      float xv[8], dv[8], acc[8], rv[8];
      E::unpack(fx[u], xv); E::unpack(fd[u], dv);
      if (accum) E::unpack(fa[u], acc);
      if (has_res) E::unpack(fr[u], rv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dy = a.silu ? dv[j] * dsilu_t<T>(g.sc[j] * xv[j] + g.sh[j]) : dv[j];
        const float dx = g.sc[j] * dy - ka[j] - ((xv[j] - g.mu[j]) * g.rs[j]) * kb[j];
        acc[j] = (accum ? acc[j] + dx : dx) + (has_res ? rv[j] : 0.f);
      }
      const Frag packed = E::pack(acc);
      E::store(dxp + pixl[u] * g.cs + g.coff, packed);
      if (sump) {                      // sums of the values as stored (rounded), what a later pass over dx would read
        E::unpack(packed, acc);
#pragma unroll
        for (int j = 0; j < 8; ++j) cs_[j] += acc[j];
      }
    }
  }
  }
  if (a.sum0 || a.sum1) {              // workgroup-uniform
    const int PP = Cc / 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = cs_[j];
    __syncthreads();
    for (int cl = threadIdx.x; cl < Cc; cl += 256) {
      const int c = cb + cl;
      const bool first = c < a.C0;
      float* sp = first ? a.sum0 : a.sum1;
      if (!sp) continue;
      float t = 0.f;
      for (int k = 0; k < ppi; ++k) t += red[(k * PP + (cl >> 3)) * 8 + (cl & 7)];
      const int cs = first ? a.C0 : a.C1, cc = first ? c : c - a.C0;
      sp[((size_t)n * a.splits + split) * cs + cc] = t;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pool2x2_kernel(const pd_pool2x2_args a) {
  using E = Elem<T>;
  const int PP = a.C / 8;
  const size_t total = (size_t)a.B * a.H * a.W * PP;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int piece = (int)(idx % PP);
    size_t pix = idx / PP;
    const int sx = (int)(pix % a.W); pix /= a.W;
    const int sy = (int)(pix % a.H); const int n = (int)(pix / a.H);
    float acc[8];
    const size_t o = (((size_t)n * a.H + sy) * a.W + sx) * a.C + piece * 8;
    if (a.accumulate) E::unpack(E::load((const T*)a.dx + o), acc);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dxx = 0; dxx < 2; ++dxx) {
        float v[8];
        E::unpack(E::load((const T*)a.du + ((((size_t)n * 2 * a.H + 2 * sy + dy) * 2 * a.W) + 2 * sx + dxx) * a.C + piece * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
    E::store((T*)a.dx + o, E::pack(acc));
  }
}

// per-(sample, channel) sum over pixels of an NHWC tensor; out[n][c] (+=)
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const pd_channel_sum_args a) {
  using E = Elem<T>;
  __shared__ float red[256 * 8];
  const int c0 = blockIdx.y * 2048, Cc = min(a.C - c0, 2048);        // channels are walked in chunks of 2048 (blockIdx.y)
  const int PP = Cc / 8, ppi = 256 / PP, nthr = ppi * PP;
  const int nsp = a.workspace ? a.splits : 1;
  const int n = blockIdx.x / nsp, sp = blockIdx.x - n * nsp, tid = threadIdx.x;
  const int per = (a.HW + nsp - 1) / nsp;
  const int p_lo = sp * per, p_hi = min(a.HW, p_lo + per);
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  if (tid < nthr) {
    const int piece = tid % PP, prow = tid / PP;
    for (int p = p_lo + prow; p < p_hi; p += 4 * ppi) {          // four 16-byte loads in flight per lane
      typename E::Frag f[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pu = p + u * ppi;
        f[u] = pu < p_hi ? E::load((const T*)a.x + ((size_t)n * a.HW + pu) * a.C + c0 + piece * 8) : E::zero();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8];
        E::unpack(f[u], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += v[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid * 8 + j] = s[j];
  __syncthreads();
  for (int c = tid; c < Cc; c += 256) {
    double d = 0.0;
    for (int k = 0; k < ppi; ++k) d += (double)red[(k * PP + (c >> 3)) * 8 + (c & 7)];
    if (a.workspace) { a.workspace[((size_t)n * nsp + sp) * a.C + c0 + c] = (float)d; continue; }
    float* o = a.out + (size_t)n * a.out_stride + c0 + c;
    *o = a.accumulate ? *o + (float)d : (float)d;
  }
}

// out[n][c] (+)= sum over splits of workspace[n][split][c];  total[c] += sum over n of those (when total != NULL).
// block = 4 channels x 64 sample groups, combined in a fixed order (round 4: 16 x 16 left C / 16 = 4 ... 24 workgroups to walk
// B x splits x C floats, 448 loads per thread at B = 112 -- 14.6 us per launch, 63 launches per training step)
__global__ __launch_bounds__(256) void channel_sum_combine_kernel(const pd_channel_sum_args a) {
  __shared__ float red[256];
  const int cl = threadIdx.x & 3, ng = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  float tot = 0.f;
  if (c < a.C) {
    for (int n = ng; n < a.B; n += 64) {
      const float* w = a.workspace + (size_t)n * a.splits * a.C + c;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // independent chains: the loads overlap
      int sp = 0;
      for (; sp + 4 <= a.splits; sp += 4) {
        s0 += w[(size_t)sp * a.C]; s1 += w[(size_t)(sp + 1) * a.C]; s2 += w[(size_t)(sp + 2) * a.C]; s3 += w[(size_t)(sp + 3) * a.C];
      }
      for (; sp < a.splits; ++sp) s0 += w[(size_t)sp * a.C];
      const float s = (s0 + s1) + (s2 + s3);
      float* o = a.out + (size_t)n * a.out_stride + c;
      *o = a.accumulate ? *o + s : s;
      tot += s;
    }
  }
  red[threadIdx.x] = tot;
  __syncthreads();
  if (ng == 0 && c < a.total_valid && a.total) {
    float t = 0.f;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) t += red[k * 4 + cl];
    a.total[c] += t;
  }
}

// total[c] += sum_n per[n*stride + c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ per, int B, int stride, int C, float* __restrict__ total) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int n = 0; n < B; ++n) s += per[(size_t)n * stride + c];
  total[c] += s;
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const pd_nchw_to_nhwc_args a) {
  const size_t total = (size_t)a.B * a.HW * (a.Cpad / 8);
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int sub = (int)(idx % (a.Cpad / 8));
    const size_t pix = idx / (a.Cpad / 8);
    const int n = (int)(pix / a.HW), p = (int)(pix % a.HW);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = sub * 8 + j;
      v[j] = c < a.C ? a.x[((size_t)n * a.C + c) * a.HW + p] : 0.f;
    }
    Elem<T>::store((T*)a.out + pix * a.Cpad + sub * 8, Elem<T>::pack(v));
  }
}

__global__ __launch_bounds__(256) void linear_wgrad_kernel(const pd_linear_wgrad_args a) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)a.out_dim * a.in_dim) return;
  const int o = (int)(idx / a.in_dim), i = (int)(idx % a.in_dim);
  float s = 0.f, sb = 0.f;
  for (int r = 0; r < a.rows; ++r) {
    const float d = a.dy[(size_t)r * a.out_dim + o];
    float x = a.x[(size_t)r * a.in_dim + i];
    if (a.x_silu) x = silu_f(x);
    s += d * x; sb += d;
  }
  a.dw[idx] += s;
  if (i == 0 && a.db) a.db[o] += sb;
}

// block = (row, 32 inputs) x 8 groups of outputs, combined in a fixed order
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const pd_linear_dgrad_args a) {
  __shared__ float red[256];
  const int chunks = (a.in_dim + 31) / 32;
  const int r = blockIdx.x / chunks, i = (blockIdx.x - r * chunks) * 32 + (threadIdx.x & 31), og = threadIdx.x >> 5;
  float s = 0.f;
  if (i < a.in_dim)
    for (int o = og; o < a.out_dim; o += 8) s += a.dy[(size_t)r * a.out_dim + o] * a.w[(size_t)o * a.in_dim + i];
  red[threadIdx.x] = s;
  __syncthreads();
  if (og == 0 && i < a.in_dim) {
    s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k * 32 + (threadIdx.x & 31)];
    const size_t idx = (size_t)r * a.in_dim + i;
    if (a.pre) s *= dsilu(a.pre[idx]);
    a.dx[idx] = s;
  }
}

__global__ __launch_bounds__(256) void embedding_grad_kernel(const pd_embedding_grad_args a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.num_classes * a.dim) return;
  const int k = idx / a.dim, i = idx % a.dim;
  float s = 0.f;
  for (int r = 0; r < a.rows; ++r)
    if (a.labels[r] == k) s += a.d[(size_t)r * a.dim + i];
  a.dtable[idx] += s;
}

}  // namespace pd

using namespace pd;

extern "C" int pd_gn_silu_bwd(const pd_gn_bwd_args* a, void* stream) {
  PD_CHECK(a != nullptr, PD_ERR_ARG, "pd_gn_silu_bwd: null args");
  const int C = a->C0 + a->C1;
  PD_CHECK(a->B > 0 && a->HW > 0 && a->C0 > 0 && a->C0 % 32 == 0 && a->C1 >= 0 && a->C1 % 32 == 0 && C <= GN_MAXC, PD_ERR_SHAPE, "pd_gn_silu_bwd: bad shape (C <= 3072)");
  PD_CHECK(a->groups > 0 && a->groups <= 64 && C % a->groups == 0, PD_ERR_SHAPE, "pd_gn_silu_bwd: groups=%d C=%d", a->groups, C);
  PD_CHECK(a->x0 && a->dz0 && a->mean && a->rstd && a->gamma && a->beta && a->partial && a->coef && a->splits >= 1, PD_ERR_ARG, "pd_gn_silu_bwd: null pointer");
  PD_CHECK((a->C1 == 0) == (a->x1 == nullptr) && (a->C1 == 0 || a->dz_combined) == (a->dz1 == nullptr), PD_ERR_ARG, "pd_gn_silu_bwd: source 1 mismatch");
  PD_CHECK(a->dx0 || a->dx1, PD_ERR_ARG, "pd_gn_silu_bwd: no output");
  PD_CHECK(a->mod == nullptr || (a->C1 == 0 && a->mod_stride >= 2 * a->C0), PD_ERR_SHAPE,
           "pd_gn_silu_bwd: the scale_shift modulation needs one source and rows of [scale | shift] (mod_stride >= 2 C)");
  PD_CHECK(a->dmod == nullptr || a->mod != nullptr, PD_ERR_ARG, "pd_gn_silu_bwd: dmod without mod");
  hipStream_t st = (hipStream_t)stream;
  const dim3 agrid((unsigned)(a->B * a->splits), (unsigned)((C + GN_CHUNK - 1) / GN_CHUNK));
  if (a->dtype == PD_F32) {
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, agrid, dim3(256), 0, st, *a);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(a->B + (C + 3) / 4), dim3(256), 0, st, *a);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, agrid, dim3(256), 0, st, *a);
  } else if (a->dtype == PD_BF16) {
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<bf16_t>, agrid, dim3(256), 0, st, *a);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(a->B + (C + 3) / 4), dim3(256), 0, st, *a);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, agrid, dim3(256), 0, st, *a);
  } else if (a->dtype == PD_F16) {
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<half_t>, agrid, dim3(256), 0, st, *a);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(a->B + (C + 3) / 4), dim3(256), 0, st, *a);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<half_t>, agrid, dim3(256), 0, st, *a);
  } else { set_error("pd_gn_silu_bwd: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_pool2x2_sum(const pd_pool2x2_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->B > 0 && a->H > 0 && a->W > 0 && a->C > 0 && a->C % 8 == 0 && a->du && a->dx, PD_ERR_ARG, "pd_pool2x2_sum: bad args");
  const size_t total = (size_t)a->B * a->H * a->W * (a->C / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (a->dtype == PD_F32) hipLaunchKernelGGL(pool2x2_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(pool2x2_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(pool2x2_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else { set_error("pd_pool2x2_sum: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_channel_sum(const pd_channel_sum_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->B > 0 && a->HW > 0 && a->C > 0 && a->C % 8 == 0 && a->out, PD_ERR_ARG, "pd_channel_sum: bad args");
  PD_CHECK(a->x || a->workspace, PD_ERR_ARG, "pd_channel_sum: x = NULL needs a pre-filled workspace (pd_gn_silu_bwd sum0/sum1)");
  PD_CHECK(!a->workspace || a->splits >= 1, PD_ERR_ARG, "pd_channel_sum: workspace without splits");
  const int nblk = a->B * (a->workspace ? a->splits : 1);
  if (!a->x) { /* per-split sums already in the workspace */ }
  else if (a->dtype == PD_F32) hipLaunchKernelGGL(channel_sum_kernel<float>, dim3(nblk, (a->C + 2047) / 2048), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, dim3(nblk, (a->C + 2047) / 2048), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(channel_sum_kernel<half_t>, dim3(nblk, (a->C + 2047) / 2048), dim3(256), 0, (hipStream_t)stream, *a);
  else { set_error("pd_channel_sum: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  if (a->total) PD_CHECK(a->total_valid > 0 && a->total_valid <= a->C, PD_ERR_ARG, "pd_channel_sum: total_valid=%d", a->total_valid);
  if (a->workspace) {
    hipLaunchKernelGGL(channel_sum_combine_kernel, dim3((a->C + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
    PD_LAUNCH_CHECK();
  } else if (a->total) {
    PD_CHECK(!a->accumulate, PD_ERR_ARG, "pd_channel_sum: total needs this call's own per-sample sums (accumulate = 0)");
    hipLaunchKernelGGL(colsum_kernel, dim3((a->total_valid + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)a->out, a->B,
                       a->out_stride, a->total_valid, a->total);
    PD_LAUNCH_CHECK();
  }
  return PD_OK;
}

extern "C" int pd_nchw_to_nhwc(const pd_nchw_to_nhwc_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->B > 0 && a->C > 0 && a->HW > 0 && a->Cpad >= a->C && a->Cpad % 8 == 0 && a->x && a->out, PD_ERR_ARG, "pd_nchw_to_nhwc: bad args");
  const size_t total = (size_t)a->B * a->HW * (a->Cpad / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (a->dtype == PD_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else { set_error("pd_nchw_to_nhwc: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_linear_wgrad(const pd_linear_wgrad_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->rows > 0 && a->in_dim > 0 && a->out_dim > 0 && a->dy && a->x && a->dw, PD_ERR_ARG, "pd_linear_wgrad: bad args");
  const size_t total = (size_t)a->out_dim * a->in_dim;
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_linear_dgrad(const pd_linear_dgrad_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->rows > 0 && a->in_dim > 0 && a->out_dim > 0 && a->dy && a->w && a->dx, PD_ERR_ARG, "pd_linear_dgrad: bad args");
  hipLaunchKernelGGL(linear_dgrad_kernel, dim3((unsigned)(a->rows * ((a->in_dim + 31) / 32))), dim3(256), 0, (hipStream_t)stream, *a);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_embedding_grad(const pd_embedding_grad_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->rows > 0 && a->dim > 0 && a->num_classes > 0 && a->labels && a->d && a->dtable, PD_ERR_ARG, "pd_embedding_grad: bad args");
  hipLaunchKernelGGL(embedding_grad_kernel, dim3((a->num_classes * a->dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
  PD_LAUNCH_CHECK();
  return PD_OK;
}
