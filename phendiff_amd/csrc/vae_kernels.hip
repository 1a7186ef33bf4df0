// pd_attn_wide: softmax(q k^T / sqrt(D)) v for ONE wide head per group of D channels (D = 128 / 256 / 512) -- the mid-block
// attention of the Stable-Diffusion VAE (diffusers AutoencoderKL: UNetMidBlock2D with attention_head_dim = 512 channels, one
// head; reached from custom_pipeline_stable_diffusion_img2img.py:431,709-711).
//
// The contraction is dense (K = D for the scores, D output rows for P.V), so both products are MFMA tiles.  A wave cannot
// hold a 32-query x 512-d output tile and the matching Q fragments, so the head dimension is SPLIT ACROSS THE 4 WAVES of a
// workgroup, which all work on the same 32 queries:
//   wave w owns d in [w*D/4, (w+1)*D/4):
//     partial S^T[key][query] = K[32 keys x D/4] . Q^T[D/4 x 32]          D/64 MFMA k-steps
//     partials of the 4 waves are summed through LDS in a fixed order  ->  every wave holds the same full S^T tile
//     online softmax, lane-local (identical in the 4 waves; 16 exps per lane per tile)
//     O^T[D/4][query] += V^T[D/4 x 32 keys] . P^T                         D/128 row tiles x 2 k-steps
// so neither product is computed twice.  K / V tiles of 32 keys are staged row-major in LDS; the A operand V^T is a
// transposed LDS read (ds_read_b64_tr_b16 for bf16; scalar gathers in the exact-fp32 parity mode).
#include "pd_common.h"
#include "pd_stage.h"

namespace pd {

template <typename T, int PITCH> struct WideX {      // primary: 16-bit element types (bf16, fp16); fp32 below
  typedef typename Elem<T>::Frag Frag;
  static __device__ __forceinline__ int vt_lane_off(int lane) {      // block row q <-> key 4h + q, columns 16*cg + 4*pp of a 32-d row tile
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    return (4 * (g >> 1) + q) * PITCH + (16 * (g & 1) + 4 * pp) * 2;
  }
  // A fragment of V^T for k-step s (16 keys): element j <-> key 16s + 8(j>>2) + 4h + (j&3)   (P's register order)
  static __device__ __forceinline__ Frag load_vt(const unsigned char* base) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s* lp;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + 8 * PITCH));
    Frag f; f.v = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
  }
  static __device__ __forceinline__ Frag pack_p(const f32x16& p, int s) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = Pack16<T>::pack(p[8 * s + 2 * j], p[8 * s + 2 * j + 1]);
    Frag f; f.v = __builtin_bit_cast(s16x8, (u32x4){w[0], w[1], w[2], w[3]});
    return f;
  }
};
template <int PITCH> struct WideX<float, PITCH> {
  typedef Elem<float>::Frag Frag;
  static __device__ __forceinline__ int vt_lane_off(int lane) { return (4 * (lane >> 5)) * PITCH + (lane & 31) * 4; }   // key 4h, d = r
  static __device__ __forceinline__ Frag load_vt(const unsigned char* base) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.lo[j] = *(const float*)(base + j * PITCH); f.hi[j] = *(const float*)(base + (8 + j) * PITCH); }
    return f;
  }
  static __device__ __forceinline__ Frag pack_p(const f32x16& p, int s) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.lo[j] = p[8 * s + j]; f.hi[j] = p[8 * s + 4 + j]; }
    return f;
  }
};

template <typename T, int D> struct WideCfg {
  static constexpr int ES = Elem<T>::BYTES;
  static constexpr int PITCH = D * ES + 16;            // K and V rows
  static constexpr int TILE = 32 * PITCH;
  static constexpr int XCH = 4 * 16 * 64 * 4;          // partial-score exchange [wave][register][lane] fp32
  static constexpr int LDS = 2 * TILE + XCH;
};

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_wide_kernel(const pd_attn_wide_args a) {
  using E = Elem<T>;
  using Frag = typename E::Frag;
  using Cfg = WideCfg<T, D>;
  using X = WideX<T, Cfg::PITCH>;
  constexpr int ES = Cfg::ES, PITCH = Cfg::PITCH, TILE = Cfg::TILE;
  constexpr int SL = D / 4, KS = SL / 16, RT = SL / 32;
  constexpr int PIECES = 32 * D / 8 / 256;             // 8-element pieces per thread per tensor
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* kb = lds;
  unsigned char* vb = lds + TILE;
  float* xch = (float*)(lds + 2 * TILE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nqb = (a.Nq + 31) / 32;
  const int qb = blockIdx.x % nqb, head = (blockIdx.x / nqb) % a.heads, b = blockIdx.x / (nqb * a.heads);
  const T* qp = (const T*)a.q + (size_t)b * a.Nq * a.q_stride + head * D;
  const T* kp = (const T*)a.k + (size_t)b * a.Nkv * a.kv_stride + head * D;
  const T* vp = (const T*)a.v + (size_t)b * a.Nkv * a.kv_stride + head * D;

  // Q^T fragments of this wave's d-slice (B operand): lane (query r, h), k-step ks: d = wave*SL + 16 ks + 8 h + j
  const int query = qb * 32 + r, qc = min(query, a.Nq - 1);
  const float qscale = a.scale * 1.4426950408889634f;
  Frag qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float v[8];
    E::unpack(E::load(qp + (size_t)qc * a.q_stride + wave * SL + 16 * ks + 8 * h), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= qscale;
    qf[ks] = E::pack(v);
  }
  f32x16 o[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) o[rt] = (f32x16)(0.f);
  float m = -INFINITY, l = 0.f;                        // l: this lane half's share of the row sum

  const int k_lane = r * PITCH + (wave * SL + 8 * h) * ES;
  const int v_lane = X::vt_lane_off(lane) + wave * SL * ES;

  for (int k0 = 0; k0 < a.Nkv; k0 += 32) {
    __syncthreads();                                   // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int pc = tid + 256 * i, row = pc / (D / 8), sub = pc % (D / 8), key = k0 + row;
      Frag fk = E::zero(), fv = E::zero();
      if (key < a.Nkv) {
        fk = E::load(kp + (size_t)key * a.kv_stride + sub * 8);
        fv = E::load(vp + (size_t)key * a.kv_stride + sub * 8);
      }
      E::store(kb + row * PITCH + sub * 8 * ES, fk);
      E::store(vb + row * PITCH + sub * 8 * ES, fv);
    }
    __syncthreads();
    f32x16 s = (f32x16)(0.f);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) s = E::mma(E::load(kb + k_lane + ks * 16 * ES), qf[ks], s);
#pragma unroll
    for (int i = 0; i < 16; ++i) xch[(wave * 16 + i) * 64 + lane] = s[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i)                        // same order in every wave: bit-identical S in all four
      s[i] = ((xch[(0 * 16 + i) * 64 + lane] + xch[(1 * 16 + i) * 64 + lane]) + xch[(2 * 16 + i) * 64 + lane]) + xch[(3 * 16 + i) * 64 + lane];
    if (k0 + 32 > a.Nkv) {                             // keys beyond the sequence
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (k0 + (i & 3) + 8 * (i >> 2) + 4 * h >= a.Nkv) s[i] = -INFINITY;
    }
    float tmax = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, s[i]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));          // finite: every tile visited holds at least one real key
    const float mn = fmaxf(m, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = __builtin_amdgcn_exp2f(s[i] - mn); ps += s[i]; }
    l = l * alpha + ps;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[rt][i] *= alpha;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const Frag pf = X::pack_p(s, st);
      const unsigned char* vs = vb + v_lane + 16 * st * PITCH;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) o[rt] = E::mma(X::load_vt(vs + 32 * rt * ES), pf, o[rt]);
    }
  }
  l += __shfl_xor(l, 32);
  if (query < a.Nq) {
    const float inv = 1.0f / l;
    // log2-domain log-sum-exp of the scaled scores (kept for pd_attn_wide_bwd): identical in the four waves
    if (a.lse && wave == 0 && h == 0) a.lse[((size_t)b * a.heads + head) * a.Nq + query] = m + __builtin_amdgcn_logf(l);
    T* dst = (T*)a.out + ((size_t)b * a.Nq + query) * a.out_stride + head * D + wave * SL + 4 * h;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g)                      // register 4g + i <-> d = 8g + 4h + i of row tile rt
        store4(dst + 32 * rt + 8 * g, o[rt][4 * g] * inv, o[rt][4 * g + 1] * inv, o[rt][4 * g + 2] * inv, o[rt][4 * g + 3] * inv);
  }
}

template <typename T, int D>
static int launch_attn_wide(const pd_attn_wide_args* a, hipStream_t st) {
  constexpr int LDS = WideCfg<T, D>::LDS;
  auto kern = attn_wide_kernel<T, D>;
  static LdsAttr attr;
  if (!ensure_lds(attr, kern, LDS)) {
    set_error("pd_attn_wide: cannot reserve %d bytes of LDS", LDS);
    return PD_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(((a->Nq + 31) / 32) * a->heads * a->B), dim3(256), LDS, st, *a);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

template <typename T>
static int dispatch_attn_wide(const pd_attn_wide_args* a, hipStream_t st) {
  switch (a->D) {
    case 128: return launch_attn_wide<T, 128>(a, st);
    case 256: return launch_attn_wide<T, 256>(a, st);
    case 512: return launch_attn_wide<T, 512>(a, st);
  }
  set_error("pd_attn_wide: head dimension %d not built (128, 256, 512)", a->D);
  return PD_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------------------------
// pd_attn_wide_bwd: gradient of pd_attn_wide (autograd of F.scaled_dot_product_attention with one wide head per D channels) --
// what `accelerator.backward(loss)` runs for the attention blocks of models_configs/denoiser/orig_google_ddpm_model_denoiser.json
// (attention_head_dim null: one 512-channel head, cond_unet_2d.py:176-197; utils_training.py:436).  Same decomposition as the
// forward: the head dimension is split over the 4 waves of a workgroup, partial 32 x 32 score tiles are summed through LDS in a
// fixed order, P is recomputed from the forward's log-sum-exp.  Two passes so that every reduction stays lane-local (no atomics):
//   DQ pass   (lane = query, loop over key tiles):   S^T = K.Q^T, dP^T = V.dO^T, dS = P (dP - delta), dQ^T += K^T . dS^T;
//             also emits delta[query] = sum_d O dO for the second pass
//   DKV pass  (lane = key, loop over query tiles):   S = Q.K^T, dP = dO.V^T (rows = queries in registers),
//             dV^T += dO^T . P,  dK^T += Q^T . dS
// Sequence lengths here are small (N <= 1024: the attention sits at 1/16 .. 1/32 resolution), so the kernel is built for
// correctness and reasonable MFMA use, not tuned: 2 % of a training step of that configuration.
template <typename T, int D, bool DKV>
__global__ __launch_bounds__(256) void attn_wide_bwd_kernel(const pd_attn_wide_bwd_args a) {
  using E = Elem<T>;
  using Frag = typename E::Frag;
  using Cfg = WideCfg<T, D>;
  using X = WideX<T, Cfg::PITCH>;
  constexpr int ES = Cfg::ES, PITCH = Cfg::PITCH, TILE = Cfg::TILE;
  constexpr int SL = D / 4, KS = SL / 16, RT = SL / 32;
  constexpr int PIECES = 32 * D / 8 / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* t0 = lds;                 // DQ: K tile      DKV: Q tile
  unsigned char* t1 = lds + TILE;          // DQ: V tile      DKV: dO tile
  float* xch = (float*)(lds + 2 * TILE);   // [wave][register][lane] partial tiles; DKV: the tile's lse / delta rows behind it
  float* rows = xch + 4 * 16 * 64;         // [2][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int Nl = DKV ? a.Nkv : a.Nq;       // sequence the lanes index
  const int No = DKV ? a.Nq : a.Nkv;       // sequence the loop walks
  const int nlb = (Nl + 31) / 32;
  const int lb = blockIdx.x % nlb, head = (blockIdx.x / nlb) % a.heads, b = blockIdx.x / (nlb * a.heads);
  const size_t qo = (size_t)b * a.Nq, ko = (size_t)b * a.Nkv;
  const T* qp = (const T*)a.q + qo * a.q_stride + head * D;
  const T* kp = (const T*)a.k + ko * a.kv_stride + head * D;
  const T* vp = (const T*)a.v + ko * a.kv_stride + head * D;
  const T* op = (const T*)a.o + qo * a.o_stride + head * D;
  const T* dop = (const T*)a.dout + qo * a.o_stride + head * D;
  const size_t bh = ((size_t)b * a.heads + head) * a.Nq;
  const float qscale = a.scale * 1.4426950408889634f;

  // this lane's row of the "lane" sequence: fragments of its d-slice in registers (B operands)
  const int row = lb * 32 + r, rc = min(row, Nl - 1);
  Frag fa[KS], fb[KS];                     // DQ: Q * scale * log2e, dO          DKV: K * scale * log2e, V
  float dpart = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int dcol = wave * SL + 16 * ks + 8 * h;
    float v[8];
    E::unpack(E::load((DKV ? kp + (size_t)rc * a.kv_stride : qp + (size_t)rc * a.q_stride) + dcol), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= qscale;
    fa[ks] = E::pack(v);
    fb[ks] = E::load((DKV ? vp + (size_t)rc * a.kv_stride : dop + (size_t)rc * a.o_stride) + dcol);
    if (!DKV) {                            // delta = sum_d O dO over this lane's share of the head dimension
      float ov[8], dv[8];
      E::unpack(E::load(op + (size_t)rc * a.o_stride + dcol), ov);
      E::unpack(fb[ks], dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) dpart += ov[j] * dv[j];
    }
  }
  float lse_l = 0.f, delta_l = 0.f;
  if (!DKV) {
    dpart += __shfl_xor(dpart, 32);
    if (h == 0) xch[wave * 64 + r] = dpart;
    __syncthreads();
    delta_l = ((xch[0 * 64 + r] + xch[1 * 64 + r]) + xch[2 * 64 + r]) + xch[3 * 64 + r];
    lse_l = a.lse[bh + rc];
    if (wave == 0 && h == 0 && row < a.Nq) a.delta[bh + row] = delta_l;
  }
  f32x16 acc0[RT], acc1[RT];               // DQ: dQ^T (acc0)        DKV: dK^T (acc0), dV^T (acc1)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) { acc0[rt] = (f32x16)(0.f); acc1[rt] = (f32x16)(0.f); }

  const int a_lane = r * PITCH + (wave * SL + 8 * h) * ES;          // row-major A fragments of a staged tile
  const int t_lane = X::vt_lane_off(lane) + wave * SL * ES;         // transposed A fragments of a staged tile

  for (int o0 = 0; o0 < No; o0 += 32) {
    __syncthreads();                       // previous tile fully consumed (and the delta exchange above)
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int pc = tid + 256 * i, trow = pc / (D / 8), sub = pc % (D / 8), orow = o0 + trow;
      Frag f0 = E::zero(), f1 = E::zero();
      if (orow < No) {
        f0 = E::load((DKV ? qp + (size_t)orow * a.q_stride : kp + (size_t)orow * a.kv_stride) + sub * 8);
        f1 = E::load((DKV ? dop + (size_t)orow * a.o_stride : vp + (size_t)orow * a.kv_stride) + sub * 8);
      }
      E::store(t0 + trow * PITCH + sub * 8 * ES, f0);
      E::store(t1 + trow * PITCH + sub * 8 * ES, f1);
    }
    if (DKV && tid < 64) {                 // the tile's per-query rows: lse | delta
      const int orow = o0 + (tid & 31);
      rows[tid] = orow < No ? (tid < 32 ? a.lse[bh + orow] : a.delta[bh + orow]) : 0.f;
    }
    __syncthreads();
    // partial tiles over this wave's d-slice: register i <-> loop row (i&3) + 8(i>>2) + 4h, lane r <-> lane row
    f32x16 s = (f32x16)(0.f), dp = (f32x16)(0.f);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s = E::mma(E::load(t0 + a_lane + ks * 16 * ES), fa[ks], s);
      dp = E::mma(E::load(t1 + a_lane + ks * 16 * ES), fb[ks], dp);
    }
    // DKV: the scores need Q.K^T with K scaled -- fa holds K * scale * log2e, t0 the raw Q tile: the same product
#pragma unroll
    for (int i = 0; i < 16; ++i) xch[(wave * 16 + i) * 64 + lane] = s[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i)
      s[i] = ((xch[(0 * 16 + i) * 64 + lane] + xch[(1 * 16 + i) * 64 + lane]) + xch[(2 * 16 + i) * 64 + lane]) + xch[(3 * 16 + i) * 64 + lane];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) xch[(wave * 16 + i) * 64 + lane] = dp[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i)
      dp[i] = ((xch[(0 * 16 + i) * 64 + lane] + xch[(1 * 16 + i) * 64 + lane]) + xch[(2 * 16 + i) * 64 + lane]) + xch[(3 * 16 + i) * 64 + lane];
    // P = 2^(S - lse), dS = P (dP - delta); rows of the loop sequence beyond its end contribute nothing
    f32x16 pm, ds;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int lr = (i & 3) + 8 * (i >> 2) + 4 * h;
      const float lse_i = DKV ? rows[lr] : lse_l, del_i = DKV ? rows[32 + lr] : delta_l;
      float pv = __builtin_amdgcn_exp2f(s[i] - lse_i);
      if (o0 + lr >= No) pv = 0.f;
      pm[i] = pv;
      ds[i] = pv * (dp[i] - del_i);
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const Frag dsf = X::pack_p(ds, st);
      const unsigned char* ts0 = t0 + t_lane + 16 * st * PITCH;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc0[rt] = E::mma(X::load_vt(ts0 + 32 * rt * ES), dsf, acc0[rt]);      // DQ: K^T.dS^T   DKV: Q^T.dS
      if (DKV) {
        const Frag pf = X::pack_p(pm, st);
        const unsigned char* ts1 = t1 + t_lane + 16 * st * PITCH;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc1[rt] = E::mma(X::load_vt(ts1 + 32 * rt * ES), pf, acc1[rt]);   // dO^T.P
      }
    }
  }
  if (row < Nl) {
    // dS carries no softmax scale yet: d(scale q.k) -> dQ, dK take it here
    T* d0 = (DKV ? (T*)a.dk + (ko + row) * a.dkv_stride : (T*)a.dq + (qo + row) * a.dq_stride) + head * D + wave * SL + 4 * h;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        store4(d0 + 32 * rt + 8 * g, acc0[rt][4 * g] * a.scale, acc0[rt][4 * g + 1] * a.scale, acc0[rt][4 * g + 2] * a.scale, acc0[rt][4 * g + 3] * a.scale);
    if (DKV) {
      T* d1 = (T*)a.dv + (ko + row) * a.dkv_stride + head * D + wave * SL + 4 * h;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          store4(d1 + 32 * rt + 8 * g, acc1[rt][4 * g], acc1[rt][4 * g + 1], acc1[rt][4 * g + 2], acc1[rt][4 * g + 3]);
    }
  }
}

template <typename T, int D>
static int launch_attn_wide_bwd(const pd_attn_wide_bwd_args* a, hipStream_t st) {
  constexpr int LDS = WideCfg<T, D>::LDS + 64 * 4;
  auto kq = attn_wide_bwd_kernel<T, D, false>;
  auto kkv = attn_wide_bwd_kernel<T, D, true>;
  static LdsAttr attr_q, attr_kv;
  if (!ensure_lds(attr_q, kq, LDS) || !ensure_lds(attr_kv, kkv, LDS)) {
    set_error("pd_attn_wide_bwd: cannot reserve %d bytes of LDS", LDS);
    return PD_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kq, dim3(((a->Nq + 31) / 32) * a->heads * a->B), dim3(256), LDS, st, *a);
  PD_LAUNCH_CHECK();
  hipLaunchKernelGGL(kkv, dim3(((a->Nkv + 31) / 32) * a->heads * a->B), dim3(256), LDS, st, *a);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

template <typename T>
static int dispatch_attn_wide_bwd(const pd_attn_wide_bwd_args* a, hipStream_t st) {
  switch (a->D) {
    case 128: return launch_attn_wide_bwd<T, 128>(a, st);
    case 256: return launch_attn_wide_bwd<T, 256>(a, st);
    case 512: return launch_attn_wide_bwd<T, 512>(a, st);
  }
  set_error("pd_attn_wide_bwd: head dimension %d not built (128, 256, 512)", a->D);
  return PD_ERR_UNSUPPORTED;
}

// DiagonalGaussianDistribution.sample of AutoencoderKL.encode, with the pipeline's latent scaling folded in:
//   z = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise),  moments = [mean | logvar] along the channel axis (NCHW fp32)
__global__ __launch_bounds__(256) void latent_sample_kernel(const pd_latent_sample_args a) {
  const int64_t per = (int64_t)a.C * a.HW, total = (int64_t)a.B * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / per, rem = i - n * per;
    const float mean = a.moments[n * 2 * per + rem];
    float z = mean;
    if (a.noise) {
      const float logvar = fminf(fmaxf(a.moments[n * 2 * per + per + rem], -30.0f), 20.0f);
      z = mean + expf(0.5f * logvar) * a.noise[i];
    }
    a.out[i] = a.scale * z;
  }
}

// GroupNorm apply (+ SiLU) as its own pass: y[n][p][c] = silu?(x[n][p][c] * scale[n][c] + shift[n][c]) over the channel concat
// [x0 | x1].  pd_conv applies this transform while staging, once per 64-channel OUTPUT tile: with Cout = 1280 the same halo
// tile is transformed 20 times and the exp/rcp issue slots, not the MFMAs, pace the convolution.  Wide layers therefore read
// a tensor normalised once here (one extra bandwidth-bound pass) and run pd_conv without a prologue.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const pd_gn_apply_args a) {
  using E = Elem<T>;
  const int C = a.C0 + a.C1, PP = C / 8;
  const size_t total = (size_t)a.B * a.HW * PP;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int piece = (int)(idx % PP);
    const size_t pix = idx / PP;                       // n * HW + p
    const int n = (int)(pix / a.HW);
    const int c8 = piece * 8;
    const bool first = c8 < a.C0;
    const T* src = first ? (const T*)a.x0 + pix * a.C0 + c8 : (const T*)a.x1 + pix * a.C1 + (c8 - a.C0);
    typename Stage<T>::R r;
    if constexpr (E::BYTES == 2) r.v = *(const u32x4*)src;
    else { r.a = *(const u32x4*)src; r.b = *((const u32x4*)src + 1); }
    float sc[8], sh[8];
    const float* ps = a.scale + (size_t)n * C + c8;
    const float* pb = a.shift + (size_t)n * C + c8;
    const f32x4 s0 = *(const f32x4*)ps, s1 = *((const f32x4*)ps + 1), b0 = *(const f32x4*)pb, b1 = *((const f32x4*)pb + 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[j] = s0[j]; sc[4 + j] = s1[j]; sh[j] = b0[j]; sh[4 + j] = b1[j]; }
    Stage<T>::xform_store((unsigned char*)((T*)a.y + pix * C + c8), r, sc, sh, true, a.silu != 0, true);   // the conv's own transform
  }
}

}  // namespace pd

using namespace pd;

extern "C" int pd_gn_apply(const pd_gn_apply_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->B > 0 && a->HW > 0 && a->C0 > 0 && a->C0 % 8 == 0 && a->C1 >= 0 && a->C1 % 8 == 0, PD_ERR_SHAPE,
           "pd_gn_apply: bad shape");
  PD_CHECK(a->x0 && a->scale && a->shift && a->y && ((a->C1 == 0) == (a->x1 == nullptr)), PD_ERR_ARG, "pd_gn_apply: null pointer / x1 mismatch");
  const size_t total = (size_t)a->B * a->HW * ((a->C0 + a->C1) / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (a->dtype == PD_F32) hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(gn_apply_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else { set_error("pd_gn_apply: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_latent_sample(const pd_latent_sample_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->B > 0 && a->C > 0 && a->HW > 0 && a->moments && a->out, PD_ERR_ARG, "pd_latent_sample: bad args");
  const int64_t total = (int64_t)a->B * a->C * a->HW;
  const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(latent_sample_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_attn_wide(const pd_attn_wide_args* a, void* stream) {
  PD_CHECK(a != nullptr, PD_ERR_ARG, "pd_attn_wide: null args");
  PD_CHECK(a->B > 0 && a->heads > 0 && a->Nq > 0 && a->Nkv > 0, PD_ERR_SHAPE, "pd_attn_wide: bad shape");
  PD_CHECK(a->q && a->k && a->v && a->out, PD_ERR_ARG, "pd_attn_wide: null pointer");
  PD_CHECK(a->D > 0 && a->q_stride >= a->heads * a->D && a->kv_stride >= a->heads * a->D && a->out_stride >= a->heads * a->D &&
               a->q_stride % 8 == 0 && a->kv_stride % 8 == 0 && a->out_stride % 8 == 0, PD_ERR_SHAPE,
           "pd_attn_wide: strides must cover heads*D channels and be multiples of 8");
  PD_CHECK((long long)((a->Nq + 31) / 32) * a->heads * a->B < (1ll << 31), PD_ERR_SHAPE, "pd_attn_wide: grid too large");
  if (a->dtype == PD_F32) return dispatch_attn_wide<float>(a, (hipStream_t)stream);
  if (a->dtype == PD_BF16) return dispatch_attn_wide<bf16_t>(a, (hipStream_t)stream);
  if (a->dtype == PD_F16) return dispatch_attn_wide<half_t>(a, (hipStream_t)stream);
  set_error("pd_attn_wide: bad dtype");
  return PD_ERR_ARG;
}

extern "C" int pd_attn_wide_bwd(const pd_attn_wide_bwd_args* a, void* stream) {
  PD_CHECK(a != nullptr, PD_ERR_ARG, "pd_attn_wide_bwd: null args");
  PD_CHECK(a->B > 0 && a->heads > 0 && a->Nq > 0 && a->Nkv > 0, PD_ERR_SHAPE, "pd_attn_wide_bwd: bad shape");
  PD_CHECK(a->q && a->k && a->v && a->o && a->dout && a->lse && a->delta && a->dq && a->dk && a->dv, PD_ERR_ARG, "pd_attn_wide_bwd: null pointer");
  const int hd = a->heads * a->D;
  PD_CHECK(a->D > 0 && a->q_stride >= hd && a->kv_stride >= hd && a->o_stride >= hd && a->dq_stride >= hd && a->dkv_stride >= hd &&
               a->q_stride % 8 == 0 && a->kv_stride % 8 == 0 && a->o_stride % 8 == 0 && a->dq_stride % 8 == 0 && a->dkv_stride % 8 == 0,
           PD_ERR_SHAPE, "pd_attn_wide_bwd: strides must cover heads*D channels and be multiples of 8");
  PD_CHECK((long long)((a->Nq + 31) / 32) * a->heads * a->B < (1ll << 31) && (long long)((a->Nkv + 31) / 32) * a->heads * a->B < (1ll << 31),
           PD_ERR_SHAPE, "pd_attn_wide_bwd: grid too large");
  if (a->dtype == PD_F32) return dispatch_attn_wide_bwd<float>(a, (hipStream_t)stream);
  if (a->dtype == PD_BF16) return dispatch_attn_wide_bwd<bf16_t>(a, (hipStream_t)stream);
  if (a->dtype == PD_F16) return dispatch_attn_wide_bwd<half_t>(a, (hipStream_t)stream);      // fp16 training (round 5)
  set_error("pd_attn_wide_bwd: bad dtype");
  return PD_ERR_ARG;
}
