// pd_linear: Y[m][n] = sum_k X[m][k] W[n][k] + bias[n] (+ R[m][n]) -- nn.Linear over NHWC tokens as a dedicated MFMA GEMM.
// Replaces the Linear layers of diffusers' BasicTransformerBlock inside UNet2DConditionModel (attn1 / attn2 projections and
// to_out, FeedForward; custom_pipeline_stable_diffusion_img2img.py:680-686, utils_training.py:486-494) and, with the transposed
// weights, their input gradients.  pd_conv runs a 1x1 convolution with its 3x3 machinery (32-channel chunks, one barrier per
// 8 MFMAs and a 64-channel output tile per staged activation tile); here the tile is shaped for a plain GEMM:
//   workgroup = 4 waves = 128 tokens x 128 output channels, wave = 64 x 64 (2 x 2 MFMA tiles: every A and every B fragment
//   feeds two MFMAs), K in chunks of 64 channels staged global -> registers -> LDS (double-buffered, one barrier per 16
//   MFMAs per wave), weights in pd_conv's packed fragment order straight from L2 (1 KiB coalesced per fragment).
// Operand conventions are pd_conv's: B fragment = lane (token r, half h) holds channels 16 s + 8 h + (0..7) of a k-step,
// D: lane owns a token, register i <-> channel 8 (i>>2) + 4 h + (i&3) of a 32-channel tile.
#include <stdlib.h>
#include "pd_common.h"
#include <type_traits>
#include "pd_stage.h"
#include "pd_linear.h"

namespace pd {

// GLU = true (NC = 2): the weights are packed with the VALUE and GATE rows of FeedForward.net[0].proj (GEGLU) interleaved per
// 32-channel tile -- packed tile 2u = value channels 32u.., tile 2u+1 = gate channels N/2 + 32u.. -- so a wave's two channel
// tiles are a value / gate pair and the epilogue stores value * gelu(gate): N/2 output channels, the 2N-wide projection
// never reaches HBM.
template <typename T, int NC, bool GLU = false>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && NC == 2) ? 3 : 2) void linear_kernel(const LinP p) {   // bf16 wide tiles: 3 workgroups per CU (<= 168 registers)
  using E = Elem<T>;
  using Frag = typename E::Frag;
  using SR = typename Stage<T>::R;
  constexpr int ES = E::BYTES;
  constexpr int TM = 128, TN = 64 * NC, CK = 64;
  constexpr int PITCH = CK * ES + 16;                  // odd number of 16-B slots: conflict-free ds_read_b128 over 32 tokens
  constexpr int XTILE = TM * PITCH;
  constexpr int NIT = TM * CK / 8 / 256;               // 8-channel pieces per thread per chunk (4)
  constexpr int TNO = GLU ? TN / 2 : TN;               // output channels per workgroup tile
  constexpr int EP_PITCH = TN * ES + 16;
  static_assert(!GLU || NC == 2 || NC == 4, "the fused GEGLU epilogue pairs the (value, gate) channel tiles of a wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [2][XTILE] | epilogue [TM][EP_PITCH]

  // block -> (token tile, channel tile): the channel tiles of one token tile are 8 blocks apart in dispatch order, i.e. on
  // the same XCD / L2 (the X tile is fetched from HBM once) whenever the number of token tiles is a multiple of 8
  int tt, ct;
  {
    const int b = blockIdx.x;
    if ((p.t_tiles & 7) == 0) { tt = (b & 7) + 8 * (b / (8 * p.c_tiles)); ct = (b >> 3) % p.c_tiles; }
    else { ct = b % p.c_tiles; tt = b / p.c_tiles; }
  }
  const long long m0 = (long long)tt * TM;
  const int n0 = ct * TN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int ksteps = p.K / 16;
  const int last_ct32 = p.N_pad / 32 - 1;
  // channel tiles beyond N_pad do not exist in w_packed: clamp (the garbage is never stored)
  int ct32[NC];
  const T* wb[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ct32[c] = min(n0 / 32 + wc * NC + c, last_ct32);
    wb[c] = (const T*)p.w + (size_t)ct32[c] * ksteps * 512 + lane * 8;
  }

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
  unsigned soff[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int pc = tid + 256 * i, tok = pc >> 3, sub = pc & 7;
    const long long m = m0 + tok;
    soff[i] = m < p.M ? (unsigned)(((size_t)m * p.x_stride + sub * 8) * ES) : OOB_OFF;
  }
  SR stage[NIT];
  // GroupNorm apply fused into the staging (Attention.group_norm in front of the q/k/v projection): this thread's 8 channels
  // (sub-block tid & 7 of the chunk) of the tile's sample
  const bool affine = p.scale != nullptr;
  const size_t aff_row = affine ? (size_t)(m0 / p.rows_per_sample) * p.K + (tid & 7) * 8 : 0;
  float sc[8], sh[8];
  auto issue = [&](int chunk) {
    const unsigned cb = (unsigned)(chunk * CK * ES);
#pragma unroll
#ifdef PD_LIN_ABL_X          /* diagnostic build: no activation traffic (prices the X stream) */
    for (int i = 0; i < NIT; ++i) stage[i] = Stage<T>::load(rsx, OOB_OFF + 0 * cb);
#else
    for (int i = 0; i < NIT; ++i) stage[i] = Stage<T>::load(rsx, soff[i] == OOB_OFF ? OOB_OFF : soff[i] + cb);
#endif
    if (affine) {
      const int kc = min(chunk * CK, p.K - 8 - (tid & 7) * 8);       // a trailing half chunk's upper sub-blocks are never read
      const float* ps = p.scale + aff_row + kc;
      const float* pb = p.shift + aff_row + kc;
      const f32x4 a0 = *(const f32x4*)ps, a1 = *((const f32x4*)ps + 1), b0 = *(const f32x4*)pb, b1 = *((const f32x4*)pb + 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) { sc[j] = a0[j]; sc[4 + j] = a1[j]; sh[j] = b0[j]; sh[4 + j] = b1[j]; }
    }
  };
  auto commit = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int pc = tid + 256 * i, tok = pc >> 3, sub = pc & 7;
      Stage<T>::xform_store(buf + tok * PITCH + sub * 8 * ES, stage[i], sc, sh, affine, false, soff[i] != OOB_OFF);
    }
  };

  // accumulators start at the bias: acc[c][f] = channel tile c (0/1) x token fragment f (0/1)
  f32x16 acc[NC][2];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    f32x16 init;
    const int cob = GLU ? ((ct32[c] & 1) ? p.N / 2 : 0) + 32 * (ct32[c] >> 1) : ct32[c] * 32;   // bias stays in module order
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *(const f32x4*)(p.bias + cob + 8 * g + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) init[4 * g + i] = bv[i];
    }
    acc[c][0] = init; acc[c][1] = init;
  }

  const int nchunks = p.K / CK, tail_ksteps = (p.K % CK) / 16;     // K % 32 == 0: a trailing half chunk of 2 k-steps is possible
  const int total_chunks = nchunks + (tail_ksteps ? 1 : 0);
  const int b_lane = (wp * 64 + r) * PITCH + 8 * h * ES;

  if constexpr (NC >= 2) {
    // wide tiles (big grids): weight fragments two k-steps ahead in registers, activation fragments read at the top of the
    // k-step; the kernel is held at 3 workgroups per CU (__launch_bounds__: 164-166 registers, accumulators in VGPRs) --
    // prefetching the activation fragments as well spills there (measured), the other resident waves cover that latency
#ifdef PD_LIN_AD                                       // diagnostic builds: same-box A/B of the prefetch depth
    constexpr int AD = PD_LIN_AD, AR = 4;
#else
    constexpr int AD = NC == 4 ? 1 : 2, AR = NC == 4 ? 2 : 4;   // weight fragments 2 k-steps ahead in a ring of 4 (static indices: 4 k-steps / chunk);
                                                                // NC = 4 (256-channel tiles, 128 accumulator registers): 1 ahead in a ring of 2
#endif
    Frag aring[AR][NC];
    const int last_kstep = ksteps - 1;
#pragma unroll
    for (int i = 0; i < AD; ++i)
#pragma unroll
      for (int c = 0; c < NC; ++c) aring[i][c] = E::load(wb[c] + (size_t)min(i, last_kstep) * 512);
    issue(0);
    commit(lds);
    if (total_chunks > 1) issue(1);
    __syncthreads();
    for (int chunk = 0; chunk < total_chunks; ++chunk) {
      const unsigned char* buf = lds + (chunk & 1) * XTILE;
      const int nks = (chunk < nchunks) ? CK / 16 : tail_ksteps;
      const int g0 = chunk * (CK / 16);
#pragma unroll
      for (int ks = 0; ks < CK / 16; ++ks) {
        if (ks < nks) {
          const int gn = min(g0 + ks + AD, last_kstep);   // clamped at the end
#pragma unroll
#ifdef PD_LIN_ABL_W          /* diagnostic build: every weight fragment load hits fragment 0 (prices the weight stream) */
          for (int c = 0; c < NC; ++c) aring[(ks + AD) % AR][c] = E::load(wb[c] + (size_t)(gn & 0) * 512);
#else
          for (int c = 0; c < NC; ++c) aring[(ks + AD) % AR][c] = E::load(wb[c] + (size_t)gn * 512);
#endif
          const Frag b0 = E::load(buf + b_lane + ks * 16 * ES), b1 = E::load(buf + b_lane + 32 * PITCH + ks * 16 * ES);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            acc[c][0] = E::mma(aring[ks % AR][c], b0, acc[c][0]);
            acc[c][1] = E::mma(aring[ks % AR][c], b1, acc[c][1]);
          }
          __builtin_amdgcn_s_setprio(0);
        }
      }
      if (chunk + 1 < total_chunks) {
        commit(lds + ((chunk + 1) & 1) * XTILE);
        if (chunk + 2 < total_chunks) issue(chunk + 2);
      }
      __syncthreads();
    }
  } else {
    // narrow tiles (small grids, long K per workgroup): the per-workgroup latency chain matters -- weight fragments in a
    // register ring AD k-steps ahead, continuously across chunk boundaries, activation fragments one k-step ahead
    constexpr int AD = 2, AR = 4;                      // AR divides the 4 k-steps of a chunk: static ring indices
    Frag aring[AR][NC];
    const int last_kstep = ksteps - 1;
    auto compute = [&](int chunk, const unsigned char* buf, auto nks_c) {
      constexpr int NKS = decltype(nks_c)::value;
      const int g0 = chunk * (CK / 16);
      Frag bc[2], bn[2];
      bc[0] = E::load(buf + b_lane); bc[1] = E::load(buf + b_lane + 32 * PITCH);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int gp = min(g0 + ks + AD, last_kstep);
#pragma unroll
        for (int c = 0; c < NC; ++c) aring[(ks + AD) % AR][c] = E::load(wb[c] + (size_t)gp * 512);
        if (ks + 1 < NKS) {
          bn[0] = E::load(buf + b_lane + (ks + 1) * 16 * ES); bn[1] = E::load(buf + b_lane + 32 * PITCH + (ks + 1) * 16 * ES);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          acc[c][0] = E::mma(aring[ks % AR][c], bc[0], acc[c][0]);
          acc[c][1] = E::mma(aring[ks % AR][c], bc[1], acc[c][1]);
        }
        __builtin_amdgcn_s_setprio(0);
        bc[0] = bn[0]; bc[1] = bn[1];
      }
    };
    issue(0);
#pragma unroll
    for (int i = 0; i < AD; ++i)
#pragma unroll
      for (int c = 0; c < NC; ++c) aring[i][c] = E::load(wb[c] + (size_t)min(i, last_kstep) * 512);
    commit(lds);
    if (total_chunks > 1) issue(1);
    __syncthreads();
    for (int chunk = 0; chunk < total_chunks; ++chunk) {
      const unsigned char* buf = lds + (chunk & 1) * XTILE;
      if (chunk < nchunks) compute(chunk, buf, std::integral_constant<int, CK / 16>{});
      else compute(chunk, buf, std::integral_constant<int, 2>{});         // trailing half chunk (K % 64 == 32)
      if (chunk + 1 < total_chunks) {
        commit(lds + ((chunk + 1) & 1) * XTILE);
        if (chunk + 2 < total_chunks) issue(chunk + 2);
      }
      __syncthreads();
    }
  }

  // ---- epilogue: [token][128 channels] through LDS, then coalesced 16-byte residual loads / stores
  if constexpr (GLU) {
#pragma unroll
    for (int u = 0; u < NC / 2; ++u)           // this wave's (value, gate) tile pairs: packed tiles 2u, 2u + 1
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int tok = wp * 64 + f * 32 + r;
      f32x16 o;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] = acc[2 * u][f][i] * gelu_f<ES>(acc[2 * u + 1][f][i]);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        store4((T*)(lds + tok * EP_PITCH) + (wc * (NC / 2) + u) * 32 + 8 * g + 4 * h, o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]);
    }
  } else
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int tok = wp * 64 + f * 32 + r;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        store4((T*)(lds + tok * EP_PITCH) + (wc * NC + c) * 32 + 8 * g + 4 * h, acc[c][f][4 * g], acc[c][f][4 * g + 1], acc[c][f][4 * g + 2],
               acc[c][f][4 * g + 3]);
    }
  __shared__ unsigned kmax_s[TN / 8];                  // p.kmax2: per-head maxima of this tile (one sample per tile), bit patterns
  if (tid < TN / 8) kmax_s[tid] = 0u;
  __syncthreads();
  constexpr int EPC = 16 / ES;                         // channels per 16-byte piece
  constexpr int PPT = TNO / EPC;                       // pieces per token
  constexpr int TPI = 256 / PPT;                       // tokens per iteration
  // dense output: a token's pieces on consecutive lanes (each token row is one contiguous run).  Head-major q/k/v output:
  // consecutive TOKENS on consecutive lanes -- a (head, token) piece is 16 bytes and a head's tokens are contiguous, so 16
  // lanes write one 256-byte run instead of 16 lanes writing 16 runs of 64 bytes
  const bool tokmajor = p.qkv_heads > 0;               // kernel-uniform
  const int piece = tokmajor ? tid / TPI : tid % PPT, trow = tokmajor ? tid % TPI : tid / PPT;
  const int NO = GLU ? p.N / 2 : p.N;                  // output row length
  const int n0o = GLU ? ct * TNO : n0;
  const int co = n0o + piece * EPC;
  float ssum[EPC], ssq[EPC];           // GroupNorm statistics of what is stored (the consumer's norm input), as pd_conv emits them
#pragma unroll
  for (int j = 0; j < EPC; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  long long kmax_nn = -1; float kmax_run = 0.f;   // running max |k|^2 of this thread's key rows (p.kmax2)
  if (co < NO) {
    // residual pieces of all iterations up front, unconditionally, from clamped (valid) rows: a load under the per-iteration
    // guards is waited for before the next one is issued
    u32x4 rres[TM / TPI];
    if (p.residual) {
#pragma unroll
      for (int it = 0; it < TM / TPI; ++it) {
        const long long mc = min(m0 + it * TPI + trow, p.M - 1);
        rres[it] = *(const u32x4*)((const T*)p.residual + (size_t)mc * NO + co);
      }
    }
#pragma unroll
    for (int it = 0; it < TM / TPI; ++it) {
      const int tok = it * TPI + trow;
      const long long m = m0 + tok;
      if (m >= p.M) continue;
      u32x4 v = *(const u32x4*)(lds + tok * EP_PITCH + piece * 16);
      if (p.residual) {
        const u32x4 rr = rres[it];
        if constexpr (ES == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float lo, hi, rl, rh;
            Pack16<T>::unpack(v[j], lo, hi); Pack16<T>::unpack(rr[j], rl, rh);
            v[j] = Pack16<T>::pack(lo + rl, hi + rh);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = __float_as_uint(__uint_as_float(v[j]) + __uint_as_float(rr[j]));
        }
      }
      if (p.stats) {
        if constexpr (ES == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float lo, hi;
            Pack16<T>::unpack(v[j], lo, hi);
            ssum[2 * j] += lo; ssq[2 * j] += lo * lo; ssum[2 * j + 1] += hi; ssq[2 * j + 1] += hi * hi;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float xv = __uint_as_float(v[j]); ssum[j] += xv; ssq[j] += xv * xv; }
        }
      }
      if (p.qkv_heads > 0) {           // [which][B][heads][token][8]; a 16-byte piece never crosses a head
        const int Cq = p.qkv_heads * 8, which = co / Cq, cc = co - which * Cq;
        const long long nn = m / p.rows_per_sample, tok_s = m - nn * p.rows_per_sample;
        *(u32x4*)((T*)p.y + ((((size_t)which * p.B + nn) * p.qkv_heads + (cc >> 3)) * p.rows_per_sample + tok_s) * 8 + (cc & 7)) = v;
        if constexpr (ES == 2) {
          if (p.kmax2 && which == 1) {   // this piece IS one key row (8 values of one head): |k|^2 of the values as stored
            float n2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { float lo, hi; Pack16<T>::unpack(v[j], lo, hi); n2 = fmaf(lo, lo, fmaf(hi, hi, n2)); }
            if (nn != kmax_nn) {           // a token tile may straddle samples (rows_per_sample % TM != 0): flush per sample
              if (kmax_nn >= 0) atomicMax((unsigned*)p.kmax2 + kmax_nn * p.qkv_heads + (cc >> 3), __float_as_uint(kmax_run));
              kmax_nn = nn; kmax_run = 0.f;
            }
            kmax_run = fmaxf(kmax_run, n2);
          }
        }
      } else {
        *(u32x4*)((T*)p.y + (size_t)m * NO + co) = v;
      }
    }
  }
  if constexpr (ES == 2) {
    // non-negative floats order like their bit patterns: unsigned atomic max.  Tiles inside one sample (the usual case) fold
    // their 256 threads through LDS first -- one global atomic per head and tile instead of one per thread (a per-thread
    // form cost the q/k/v projection +60 us at 131 072 tokens: 256 atomics per address)
    if (p.kmax2) {                                      // kernel-uniform
      const bool one_sample = p.rows_per_sample % TM == 0;
      if (!one_sample) {
        if (kmax_nn >= 0) atomicMax((unsigned*)p.kmax2 + kmax_nn * p.qkv_heads + ((co % (p.qkv_heads * 8)) >> 3), __float_as_uint(kmax_run));
      } else {
        if (kmax_nn >= 0) atomicMax(&kmax_s[piece], __float_as_uint(kmax_run));
        __syncthreads();
        const int Cq = p.qkv_heads * 8, c0 = n0o + tid * 8;
        if (tid < TNO / 8 && c0 < NO && c0 / Cq == 1 && kmax_s[tid] != 0u)
          atomicMax((unsigned*)p.kmax2 + (m0 / p.rows_per_sample) * p.qkv_heads + ((c0 - Cq) >> 3), kmax_s[tid]);
      }
    }
  }
  if (p.stats) {       // kernel-uniform: per-thread partials -> LDS behind the output tile; (channel, sum | sumsq) threads add the
                       // TPI token rows in a fixed order (deterministic)
    float* red = (float*)(lds + TM * EP_PITCH);
#pragma unroll
    for (int j = 0; j < EPC; ++j) { red[tid * (2 * EPC) + j] = ssum[j]; red[tid * (2 * EPC) + EPC + j] = ssq[j]; }
    __syncthreads();
    if (tid < 2 * TN) {
      const int c = tid >> 1, which = tid & 1;
      const int pc = c / EPC, j = c % EPC;
      float tot = 0.f;
#pragma unroll 8
      for (int pr = 0; pr < TPI; ++pr) tot += red[(pr * PPT + pc) * (2 * EPC) + which * EPC + j];
      if (n0 + c < p.N) {
        const long long nn = m0 / p.rows_per_sample;
        const int tile = (int)((m0 - nn * p.rows_per_sample) / TM), T_ = p.rows_per_sample / TM;
        p.stats[((nn * T_ + tile) * (size_t)p.N + n0 + c) * 2 + which] = tot;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// pd_token_wgrad: dW[n][k] (+)= sum_m dY[m][n] X[m][k] -- the weight gradient of nn.Linear over M tokens, an MFMA GEMM whose
// reduction runs over the tokens.  Both operands need 8 consecutive TOKENS per lane for a fixed channel, i.e. transposed
// reads of the token-major tiles (ds_read_b64_tr_b16; 32-channel planes with 64-byte token rows, conflict-free), as in
// pd_conv_wgrad -- but shaped for a plain GEMM: workgroup tile 128 (n) x 128 (k), wave 64 x 64 (2 x 2 MFMA tiles: 8
// transposed reads per 4 MFMAs instead of 4 per 1), 64-token stages double-buffered in LDS (one barrier per 16 MFMAs per
// wave).  The token range is split over workgroups; partial tiles go to a slab [split][n][k] that pd_conv_wgrad's ordered
// reduce folds into the fp32 gradient (bitwise reproducible, no atomics).
struct TwP {
  long long M;
  int K, N, x_stride, dy_stride, NP, KP, n_tiles, k_tiles, splits, chunks_per_split, nchunks;
  int xcd_order;                                                      // 1: XCD-aware work order (PD_TW_XCD=0: diagnostic override, same-box A/B)
  unsigned xbytes, dybytes;
  const void* x; const void* dy; float* slab;
};

template <typename T>
__global__ __launch_bounds__(256) void token_wgrad_kernel(const TwP p) {
  using E = Elem<T>;
  using Frag = typename E::Frag;
  using SR = typename Stage<T>::R;
  constexpr int ES = E::BYTES, TM = 64, PXB = 32 * ES;                 // one token row of a 32-channel plane
  constexpr int PLANE = TM * PXB, OPER = 4 * PLANE, BUF = 2 * OPER;    // [dY planes 0..3 | X planes 0..3]
  constexpr int NIT = TM * 128 / 8 / 256;                              // 8-channel pieces per thread per operand per stage (4)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [2][BUF]

  const int ncombo = p.n_tiles * p.k_tiles;
  // round 6: the (n, k) tiles of one token split read the SAME token rows of dY and X: one XCD runs a contiguous run of the (split-major)
  // work list, so those rows come out of its L2 once instead of out of eight (PMC: 4.3 x the algorithmic bytes left L2 before)
  const int widx = p.xcd_order ? xcd_chunk_index(blockIdx.x, p.splits * ncombo) : (int)blockIdx.x;
  const int split = widx / ncombo, combo = widx - split * ncombo;
  const int nt = combo / p.k_tiles, kt = combo - nt * p.k_tiles;
  const int n0 = nt * 128, k0 = kt * 128;
  const int c_begin = split * p.chunks_per_split, c_end = min(p.nchunks, c_begin + p.chunks_per_split);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wk = wave >> 1;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dybytes, 0x00020000);

  // this thread's pieces: token = q >> 4, 8-channel sub-block = q & 15 of the 128-channel tile
  SR sdy[NIT], sx[NIT];
  auto issue = [&](int chunk) {
    const long long m0 = (long long)chunk * TM;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int q = tid + 256 * i, tok = q >> 4, sub = q & 15;
      const long long m = m0 + tok;
      const bool okd = m < p.M && n0 + sub * 8 < p.N, okx = m < p.M && k0 + sub * 8 < p.K;
      sdy[i] = Stage<T>::load(rsd, okd ? (unsigned)(((size_t)m * p.dy_stride + n0 + sub * 8) * ES) : OOB_OFF);
      sx[i] = Stage<T>::load(rsx, okx ? (unsigned)(((size_t)m * p.x_stride + k0 + sub * 8) * ES) : OOB_OFF);
    }
  };
  auto commit = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int q = tid + 256 * i, tok = q >> 4, sub = q & 15;
      unsigned char* d = buf + (sub >> 2) * PLANE + tok * PXB + (sub & 3) * 8 * ES;
      if constexpr (ES == 2) { *(u32x4*)d = sdy[i].v; *(u32x4*)(d + OPER) = sx[i].v; }
      else { *(u32x4*)d = sdy[i].a; *(u32x4*)(d + 16) = sdy[i].b; *(u32x4*)(d + OPER) = sx[i].a; *(u32x4*)(d + OPER + 16) = sx[i].b; }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x16)(0.f);
  const unsigned lo = FragLd<T>::lane_off(lane, PXB);
  const int a_off = (wn * 2) * PLANE + lo, b_off = OPER + (wk * 2) * PLANE + lo;

  if (c_begin < c_end) {
    issue(c_begin);
    commit(lds);
    if (c_begin + 1 < c_end) issue(c_begin + 1);
    __syncthreads();
  }
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const unsigned char* buf = lds + ((chunk - c_begin) & 1) * BUF;
#pragma unroll
    for (int ks = 0; ks < TM / 16; ++ks) {
      const Frag a0 = FragLd<T>::template load<PXB>(buf + a_off + ks * 16 * PXB);
      const Frag a1 = FragLd<T>::template load<PXB>(buf + a_off + PLANE + ks * 16 * PXB);
      const Frag b0 = FragLd<T>::template load<PXB>(buf + b_off + ks * 16 * PXB);
      const Frag b1 = FragLd<T>::template load<PXB>(buf + b_off + PLANE + ks * 16 * PXB);
      __builtin_amdgcn_s_setprio(1);
      acc[0][0] = E::mma(a0, b0, acc[0][0]);
      acc[0][1] = E::mma(a0, b1, acc[0][1]);
      acc[1][0] = E::mma(a1, b0, acc[1][0]);
      acc[1][1] = E::mma(a1, b1, acc[1][1]);
      __builtin_amdgcn_s_setprio(0);
    }
    if (chunk + 1 < c_end) {
      commit(lds + ((chunk + 1 - c_begin) & 1) * BUF);
      if (chunk + 2 < c_end) issue(chunk + 2);
    }
    __syncthreads();
  }

  // partial tile -> slab[split][n][k]: lane = column k, register g <-> row n = 8 (g>>2) + 4 h + (g&3)
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float* out = p.slab + ((size_t)split * p.NP + n0 + wn * 64 + a * 32) * p.KP + k0 + wk * 64 + b * 32 + r;
#pragma unroll
      for (int g = 0; g < 16; ++g) out[(size_t)((g & 3) + 8 * (g >> 2) + 4 * h) * p.KP] = acc[a][b][g];
    }
}

// token_wgrad_dma_kernel (round 3): the same GEMM with both operands staged by DMA (global_load_lds: no staging registers, no
// ds_write, no VALU) and a larger register tile.  token_wgrad_kernel reads 1 KB of LDS per MFMA (4 fragments per 4 MFMAs) and its
// 128 x 128 tiles waste 20 % of the MFMAs per 320-multiple dimension of the latent-diffusion UNet (320 -> 384).  Here a wave owns
// FN x FK 32 x 32 tiles (5 x 2 or 2 x 5: 7 fragments per 10 MFMAs = 0.7 KB per MFMA, 160 accumulator registers), the four waves
// 2 x 2 of them: workgroup tile 320 x 128 or 128 x 320 -- whichever wastes less of N x K (every width of the SD UNet is a multiple
// of 320 = 5 x 64 on one side and of 128 on the other, except the 320 x 320 layers: 1.2 instead of 1.44).  Same LDS layout as above
// (32-channel planes of 64-byte token rows, transposed fragment reads), one stage = TM = 32 tokens = 2 (FN + FK) planes of 2 KiB =
// 28 one-KiB DMA pieces (16 token rows of 64 bytes: 4 lanes per row), 7 per wave.  NBUF buffers, DMA NBUF - 1 stages ahead, per
// stage one counted `s_waitcnt vmcnt` + one barrier (as linear_dma_kernel).  Requires M % 32 == 0 (a partial stage would have to
// be zero-filled: the reduction runs over the tokens) and N, K multiples of 32 (a tile's planes past N / K re-read the last valid
// plane; those rows / columns of the slab are never read).
template <typename T, int FN, int FK, int NBUF>
__global__ __launch_bounds__(256, 2) void token_wgrad_dma_kernel(const TwP p) {
  static_assert(sizeof(T) == 2, "16-bit element types");
  using E = Elem<T>;
  using Frag = typename E::Frag;
  constexpr int ES = 2, TM = 32, PXB = 32 * ES;
  constexpr int PLANE = TM * PXB;                        // 2 KiB
  constexpr int NPL = 2 * (FN + FK);                     // planes per stage: [dY: 2 FN | X: 2 FK]
  constexpr int BUF = NPL * PLANE;
  constexpr int UPP = PLANE / 1024;                      // DMA pieces per plane (2)
  constexpr int U = NPL * UPP;
  static_assert(U % 4 == 0, "pieces must divide evenly over the four waves (one wait count for all)");
  constexpr int UW = U / 4;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // [NBUF][BUF]

  const int ncombo = p.n_tiles * p.k_tiles;
  const int widx = p.xcd_order ? xcd_chunk_index(blockIdx.x, p.splits * ncombo) : (int)blockIdx.x;      // (as token_wgrad_kernel: split-major runs per XCD)
  const int split = widx / ncombo, combo = widx - split * ncombo;
  const int nt = combo / p.k_tiles, kt = combo - nt * p.k_tiles;
  const int n0 = nt * (64 * FN), k0 = kt * (64 * FK);
  const int c_begin = split * p.chunks_per_split, c_end = min(p.nchunks, c_begin + p.chunks_per_split);
  if (c_begin >= c_end) return;                          // (cannot happen: splits = ceil(nchunks / chunks_per_split))

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wk = wave >> 1;

  // ---- DMA sources: piece u = 4 j + wave -> plane u / UPP, 16-token block u % UPP; lane -> token row lane / 4, 16-byte slot lane % 4
  const unsigned char* src[UW];
  long long adv[UW];                                      // bytes per stage (wave-uniform)
#pragma unroll
  for (int j = 0; j < UW; ++j) {
    const int u = 4 * j + wave;
    const int plane = u / UPP, tb = u % UPP;
    const long long m = (long long)c_begin * TM + tb * 16 + (lane >> 2);
    if (plane < 2 * FN) {
      const int ch = min(n0 + plane * 32, p.N - 32) + (lane & 3) * 8;
      src[j] = (const unsigned char*)p.dy + ((size_t)m * p.dy_stride + ch) * ES;
      adv[j] = (long long)TM * p.dy_stride * ES;
    } else {
      const int ch = min(k0 + (plane - 2 * FN) * 32, p.K - 32) + (lane & 3) * 8;
      src[j] = (const unsigned char*)p.x + ((size_t)m * p.x_stride + ch) * ES;
      adv[j] = (long long)TM * p.x_stride * ES;
    }
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto dma = [&](const unsigned char* s, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(s), "s"(dst) : "memory");
  };
  auto stage = [&](int b2) {                              // issues the NEXT stage of this workgroup's token range into buffer b2
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_base + b2 * BUF);
#pragma unroll
    for (int j = 0; j < UW; ++j) {
      dma(src[j], base + (4 * j + wave) * 1024);
      src[j] += adv[j];
    }
  };

  f32x16 acc[FN][FK];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FK; ++b) acc[a][b] = (f32x16)(0.f);
  const unsigned lo = FragLd<T>::lane_off(lane, PXB);
  const int a_off = (wn * FN) * PLANE + lo, b_off = (2 * FN + wk * FK) * PLANE + lo;

  const int nst = c_end - c_begin;
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i)
    if (i < nst) stage(i);
  for (int st = 0; st < nst; ++st) {
    // this stage's pieces have landed when at most min(NBUF - 2, stages left after this one) younger stages are in flight
    const int younger = min(NBUF - 2, nst - 1 - st);
    if (NBUF >= 4 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * UW) : "memory");
    else if (NBUF >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(UW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // everybody's pieces landed AND everybody is done with the buffer re-filled next
    if (st + NBUF - 1 < nst) stage((st + NBUF - 1) % NBUF);
    const unsigned char* buf = lds + (st % NBUF) * BUF;
#pragma unroll
    for (int ks = 0; ks < TM / 16; ++ks) {
      Frag fa[FN], fb[FK];
#pragma unroll
      for (int a = 0; a < FN; ++a) fa[a] = FragLd<T>::template load<PXB>(buf + a_off + a * PLANE + ks * 16 * PXB);
#pragma unroll
      for (int b = 0; b < FK; ++b) fb[b] = FragLd<T>::template load<PXB>(buf + b_off + b * PLANE + ks * 16 * PXB);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FK; ++b) acc[a][b] = E::mma(fa[a], fb[b], acc[a][b]);
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // partial tile -> slab[split][n][k]: lane = column k, register g <-> row n = 8 (g>>2) + 4 h + (g&3)
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FK; ++b) {
      float* out = p.slab + ((size_t)split * p.NP + n0 + (wn * FN + a) * 32) * p.KP + k0 + (wk * FK + b) * 32 + r;
#pragma unroll
      for (int g = 0; g < 16; ++g) out[(size_t)((g & 3) + 8 * (g >> 2) + 4 * h) * p.KP] = acc[a][b][g];
    }
}

// dw[n][k] (+)= sum over splits of slab[split][n][k] (splits added in order: bitwise reproducible); coalesced rows
__global__ __launch_bounds__(256) void token_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits, int NP, int KP,
                                                                  int N, int K, int accumulate) {
  const int n = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const size_t per = (size_t)NP * KP;
  const float* src = slab + (size_t)n * KP + k;
  float s0 = 0.f, s1 = 0.f;
  int sp = 0;
  for (; sp + 2 <= splits; sp += 2) { s0 += src[(size_t)sp * per]; s1 += src[(size_t)(sp + 1) * per]; }
  if (sp < splits) s0 += src[(size_t)sp * per];
  float* o = dw + (size_t)n * K + k;
  *o = accumulate ? *o + (s0 + s1) : (s0 + s1);
}

// Token splits: the grid (splits x tiles) should fill the resident workgroup slots ONCE -- a grid of 1.1 x the slots runs two rounds,
// the second nearly empty (round 3: ceil() here cost up to 40 % on the 80-tile layers) -- unless the token range is long enough
// for several full rounds to amortise the tail; each split writes one fp32 tile set to the slab, so splits are also bounded by
// `min_stages` stages per split.
static int token_wgrad_splits(int nchunks, int combos, int slots, int min_stages) {
  int s = slots / combos;                                // one round, filled from below
  if (s < 1) s = 1;
  const int amort = nchunks / min_stages > 1 ? nchunks / min_stages : 1;
  return s < amort ? s : amort;
}
static void token_wgrad_plan(long long M, int K, int N, int esz, int* n_tiles, int* k_tiles, int* splits, int* cps, int* nchunks) {
  *n_tiles = (N + 127) / 128; *k_tiles = (K + 127) / 128;
  *nchunks = (int)((M + 63) / 64);
  // resident workgroups: 64 KiB of LDS each in bf16 = 2 per CU, 128 KiB in fp32 = 1 per CU
  const int s = token_wgrad_splits(*nchunks, *n_tiles * *k_tiles, esz == 2 ? 512 : 256, 4);
  *cps = (*nchunks + s - 1) / s;
  *splits = (*nchunks + *cps - 1) / *cps;
}
// the DMA form: tiles of tn x tk channels, 32-token stages, two workgroups per CU
static void token_wgrad_plan_dma(long long M, int K, int N, int tn, int tk, int* n_tiles, int* k_tiles, int* splits, int* cps, int* nchunks) {
  *n_tiles = (N + tn - 1) / tn; *k_tiles = (K + tk - 1) / tk;
  *nchunks = (int)(M / 32);
  const int s = token_wgrad_splits(*nchunks, *n_tiles * *k_tiles, diag_env("PD_TW_SLOTS", 512), 16);   // >= 16 stages (448 KB staged) per slab tile written (160 KB)
  *cps = (*nchunks + s - 1) / s;
  *splits = (*nchunks + *cps - 1) / *cps;
}
// which form: 0 = token_wgrad_kernel (128 x 128), 1 = DMA 320 (n) x 128 (k), 2 = DMA 128 x 320, 3 = DMA 128 x 128.  PD_TW_DMA overrides
// (diagnostic).  The DMA forms need whole 32-token stages and 32-channel planes.
static int token_wgrad_variant(const pd_token_wgrad_args* a) {
  if (a->dtype != PD_BF16 || a->M % 32 != 0 || a->N % 32 != 0 || a->K % 32 != 0 || a->N < 32 || a->K < 32) return 0;
  static int forced = -2;
  if (forced == -2) { const char* e = getenv("PD_TW_DMA"); forced = e ? atoi(e) : -1; }
  if (forced >= 0 && forced <= 3) return forced;
  auto padded = [&](int tn, int tk) { return (double)((a->N + tn - 1) / tn * tn) * (double)((a->K + tk - 1) / tk * tk); };
  const double w0 = padded(128, 128), w1 = padded(320, 128), w2 = padded(128, 320);
  if (a->M < 2048) return 0;                           // a handful of stages per workgroup: the DMA pipeline never fills
  if (w1 <= w2 && w1 <= w0) return 1;
  if (w2 <= w0) return 2;
  return 3;
}
static void token_wgrad_tile(int variant, int* tn, int* tk) {
  *tn = variant == 1 ? 320 : 128; *tk = variant == 2 ? 320 : 128;
}

template <typename T, int FN, int FK, int NBUF>
static int launch_token_wgrad_dma(const TwP& p, hipStream_t st) {
  constexpr int LDS = NBUF * 2 * (FN + FK) * 32 * 64;
  auto kern = token_wgrad_dma_kernel<T, FN, FK, NBUF>;
  if (LDS > 64 * 1024) {
    static LdsAttr attr;
    if (!ensure_lds(attr, kern, LDS)) {
      set_error("pd_token_wgrad: cannot reserve %d bytes of LDS", LDS);
      return PD_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.splits * p.n_tiles * p.k_tiles)), dim3(256), LDS, st, p);
  return PD_OK;
}

template <typename T, int NC, bool GLU = false>
static int launch_linear(const LinP& p, hipStream_t st) {
  constexpr int ES = Elem<T>::BYTES;
  constexpr int XT = 128 * (64 * ES + 16), EPI = 128 * (64 * NC * ES + 16);
  constexpr int EPI_ST = EPI + 256 * (2 * 16 / ES) * 4;        // + per-thread statistics partials
  // NC = 4 is launched without statistics (no partials behind the output tile): 67.6 KB, two workgroups per CU
  constexpr int LDS = NC == 4 ? (2 * XT > EPI ? 2 * XT : EPI) : (2 * XT > EPI_ST ? 2 * XT : EPI_ST);
  auto kern = linear_kernel<T, NC, GLU>;
  if (LDS > 64 * 1024) {
    static LdsAttr attr;
    if (!ensure_lds(attr, kern, LDS)) {
      set_error("pd_linear: cannot reserve %d bytes of LDS", LDS);
      return PD_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.t_tiles * p.c_tiles)), dim3(256), LDS, st, p);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// linear_dma_kernel (round 3): the plain nn.Linear GEMM with BOTH operands staged by DMA (global_load_lds) and the weight chunk
// SHARED by the workgroup through LDS.  linear_kernel moves 0.75 KB through the vector L1 per MFMA (16 KB of activations + 2 x 16 KB
// of per-wave weight fragments per 64 MFMAs) = ~96 B/clk/CU at matrix-pipe peak against the L1's 64: the diagnosis of round 2.
// Here a workgroup is 8 waves = 256 tokens x (64 NC) channels; per 64-channel K chunk it stages
//   X: 256 tokens x 64 k            = 32 KB  as 32 one-KiB pieces of 8 full 128-byte rows (one per wave-instruction, no registers,
//                                      no VALU, no ds_write); rows are XOR-swizzled through the SOURCE address (16-byte piece q of
//                                      row r sits in slot q ^ (r & 7)), so the 32 consecutive tokens of a B fragment read
//                                      (ds_read_b128) cover all 64 banks exactly once,
//   W: (2 NC) x 4 packed fragments  = 8 NC KB: pd_conv's fragment order is already lane-linear (1 KiB per fragment),
// i.e. 0.375 KB (NC = 2) / 0.25 KB (NC = 4) per MFMA.  Two LDS buffers: the DMA of chunk c + 1 is issued before chunk c's MFMAs,
// retired by `s_waitcnt vmcnt(0)` + the chunk's one barrier (inline asm: a compiler-visible global_load_lds would order every
// later ds_read behind it).  Plain layers only (no GroupNorm prologue, dense output, no statistics): the transformer blocks'
// q/k/v, to_out, GEGLU and FeedForward projections and their input gradients -- 90 % of the SD UNet's Linear FLOPs.
template <typename T, int NC, bool GLU, int CK, bool QKV = false>
__global__ __launch_bounds__(512, (CK == 32 && NC == 2) ? 4 : 2) void linear_dma_kernel(const LinP p) {
  static_assert(sizeof(T) == 2, "16-bit element types");
  static_assert(!(QKV && GLU), "head-major output: plain projection");
  using E = Elem<T>;
  using Frag = typename E::Frag;
  static_assert(CK == 64 || CK == 32, "K chunk");
  static_assert(NC != 5 || (!GLU && !QKV && CK == 64), "the 320-channel tile: plain projections, 64-channel K chunks");
  constexpr int ES = 2, TM = 256, TN = 64 * NC;
  constexpr int KSC = CK / 16;                           // k-steps per chunk
  constexpr int RPP = 1024 / (CK * ES);                  // token rows per 1-KiB DMA piece (8 / 16)
  constexpr int SPR = CK * ES / 16;                      // 16-byte slots per row (8 / 4)
  constexpr int XP = TM / RPP / 8;                       // X pieces per wave per chunk (4 / 2)
  constexpr int WP = 2 * NC * KSC / 8;                   // W fragments per wave per chunk
  constexpr int XB = TM * CK * ES;                       // 32 KB: one X chunk, [token][64 k] unpadded, slot-swizzled
  constexpr int WB = (2 * NC) * KSC * 1024;              // 2 NC channel tiles x KSC k-steps x 1 KiB fragments
  constexpr int BUF = XB + WB;
  constexpr int NBUF = 3 * BUF <= 160 * 1024 ? 3 : 2;    // three buffers (DMA two chunks ahead) where they fit; CK = 64, NC = 4: two of 64 KB
  constexpr int TNO = GLU ? TN / 2 : TN;
  constexpr int EP_PITCH = TN * ES + 16;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // [NBUF][X | W]; epilogue [TM][EP_PITCH]

  int tt, ct;
  {
    const int b = blockIdx.x;
    if ((p.t_tiles & 7) == 0) { tt = (b & 7) + 8 * (b / (8 * p.c_tiles)); ct = (b >> 3) % p.c_tiles; }
    else { ct = b % p.c_tiles; tt = b / p.c_tiles; }
  }
  const long long m0 = (long long)tt * TM;
  const int n0 = ct * TN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wt = wave >> 1, wc = wave & 1;               // token quarter (64 tokens), channel half (32 NC channels)
  const int r = lane & 31, h = lane >> 5;
  const int ksteps = p.K / 16, nchunks = p.K / CK;
  const int last_ct32 = p.N_pad / 32 - 1;

  // ---- DMA sources.  X piece pc (8 rows) = wave * 4 + j: lane -> row pc * 8 + lane / 8, LDS slot lane % 8 holds source piece slot ^ (row & 7)
  // CK = 64: 8 rows of 128 B per piece, slot s of row r holds source piece s ^ (r & 7).  CK = 32: 16 rows of 64 B, slot s holds
  // source piece s ^ ((r >> 2) & 3) -- either way the 16 lanes of a ds_read_b128 group cover the 64 banks once.
  const unsigned char* xsrc[XP];
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int row = (wave * XP + j) * RPP + lane / SPR;
    const int slot = lane % SPR;
    const int srcp = CK == 64 ? (slot ^ (row & 7)) : (slot ^ ((row >> 2) & 3));
    const long long m = m0 + row < p.M ? m0 + row : p.M - 1;      // rows past M re-read the last row (never stored)
    xsrc[j] = (const unsigned char*)p.x + ((size_t)m * p.x_stride + srcp * 8) * ES;
  }
  // W fragment f = wave * WP + j of the chunk's 2 NC x KSC: channel tile f / KSC, k-step f % KSC
  // per-sample weights / bias (QKV: the GroupNorm affine of the projection's input is folded into them, linear_fold_gn_kernel):
  // a 256-token tile lies inside one sample (rows_per_sample % 256 == 0, checked by pd_linear)
  const long long smp = p.w_sstride ? m0 / p.rows_per_sample : 0;
  const unsigned char* wbase = (const unsigned char*)p.w + smp * p.w_sstride;
  const float* bias = p.bias + smp * p.bias_sstride;
  const unsigned char* wsrc[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int f = wave * WP + j;
    const int c32 = min(n0 / 32 + f / KSC, last_ct32);              // tiles beyond N_pad do not exist: clamp (never stored)
    wsrc[j] = wbase + (((size_t)c32 * ksteps + (f % KSC)) * 512 + lane * 8) * ES;
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto dma = [&](const unsigned char* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  auto stage = [&](int chunk, int b2) {
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_base + b2 * BUF);
#pragma unroll
    for (int j = 0; j < XP; ++j) dma(xsrc[j] + (size_t)chunk * CK * ES, base + (wave * XP + j) * 1024);
#pragma unroll
    for (int j = 0; j < WP; ++j) dma(wsrc[j] + (size_t)chunk * KSC * 512 * ES, base + XB + (wave * WP + j) * 1024);
  };

  // accumulators start at the bias: acc[c][f] = channel tile c of this wave's half x token fragment f
  f32x16 acc[NC][2];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int c32 = min(n0 / 32 + wc * NC + c, last_ct32);
    const int cob = GLU ? ((c32 & 1) ? p.N / 2 : 0) + 32 * (c32 >> 1) : c32 * 32;
    f32x16 init;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *(const f32x4*)(bias + cob + 8 * g + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) init[4 * g + i] = bv[i];
    }
    acc[c][0] = init; acc[c][1] = init;
  }
  // fragment read offsets inside a buffer: B (tokens): row = wt * 64 + f * 32 + r, piece 2 ks + h in slot (2 ks + h) ^ (row & 7);
  // A (weights): fragment (wc * NC + c) * 4 + ks, lane-linear
  int brow[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) brow[f] = (wt * 64 + f * 32 + r) * (CK * ES);
  const int bsw = CK == 64 ? (r & 7) : ((r >> 2) & 3);   // the row's swizzle key: the other terms of the row index are multiples of 32
  const int a_lane = XB + (wc * NC * KSC) * 1024 + lane * 16;

  // NBUF buffers, DMA running NBUF - 1 chunks ahead; per chunk: wait until this chunk's pieces have landed (each wave counts its own
  // PER DMAs: the younger chunk may stay in flight), ONE barrier (everybody's pieces landed AND everybody finished the previous
  // chunk, whose buffer the next DMA overwrites), issue the DMA of chunk + NBUF - 1, multiply.
  constexpr int PER = XP + WP;
  stage(0, 0);
  if (NBUF == 3 && nchunks > 1) stage(1, 1);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (NBUF == 3 && chunk + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (chunk + NBUF - 1 < nchunks) stage(chunk + NBUF - 1, (chunk + NBUF - 1) % NBUF);
    const unsigned char* buf = lds + (chunk % NBUF) * BUF;
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) {
      Frag a[NC], b[2];
#pragma unroll
      for (int c = 0; c < NC; ++c) a[c] = E::load(buf + a_lane + (c * KSC + ks) * 1024);
#pragma unroll
      for (int f = 0; f < 2; ++f) b[f] = E::load(buf + brow[f] + (((2 * ks + h) ^ bsw) * 16));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        acc[c][0] = E::mma(a[c], b[0], acc[c][0]);
        acc[c][1] = E::mma(a[c], b[1], acc[c][1]);
      }
      __builtin_amdgcn_s_setprio(0);
    }
  }
  __syncthreads();                                       // every wave is done with the last buffer: the epilogue reuses the LDS

  // ---- epilogue: [token][TN channels] through LDS, then coalesced 16-byte residual loads / stores
  if constexpr (NC == 5) {
    // 320-channel tile (round 4: the SD UNet's narrowest level, N = 320 -- the 128- / 256-channel tiles computed 384 / 512 of them and
    // re-staged the token tile per column tile): the [256][320] output tile is 164 KB, more than the LDS holds, so the two token halves
    // go through the staging area one after the other; pieces are dealt to the 512 threads by their flat index (40 pieces per token)
    constexpr int PPT5 = TN / 8, NIT5 = (TM / 2) * PPT5 / 512;
    static_assert((TM / 2) * PPT5 % 512 == 0, "piece map");
    const unsigned ybytes5 = (unsigned)min((unsigned long long)p.M * p.N * ES, 0xffffffffull);
    const __amdgpu_buffer_rsrc_t ry5 = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, ybytes5, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr5 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual ? p.residual : p.y), 0, p.residual ? ybytes5 : 0u, 0x00020000);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if ((wt >> 1) == pass) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const int tok = (wt & 1) * 64 + f * 32 + r;
#pragma unroll
            for (int g = 0; g < 4; ++g)
              store4((T*)(lds + tok * EP_PITCH) + (wc * NC + c) * 32 + 8 * g + 4 * h, acc[c][f][4 * g], acc[c][f][4 * g + 1], acc[c][f][4 * g + 2],
                     acc[c][f][4 * g + 3]);
          }
      }
      __syncthreads();
      u32x4 res5[NIT5];
      unsigned off5[NIT5];
#pragma unroll
      for (int it = 0; it < NIT5; ++it) {
        const int q = it * 512 + tid, tok = q / PPT5, piece = q - tok * PPT5;
        const long long m = m0 + pass * (TM / 2) + tok;
        const int co = n0 + piece * 8;
        off5[it] = (m < p.M && co < p.N) ? (unsigned)(((size_t)m * p.N + co) * ES) : OOB_OFF;
        res5[it] = __builtin_amdgcn_raw_buffer_load_b128(rr5, off5[it], 0, 0);
      }
#pragma unroll
      for (int it = 0; it < NIT5; ++it) {
        const int q = it * 512 + tid, tok = q / PPT5, piece = q - tok * PPT5;
        u32x4 v = *(const u32x4*)(lds + tok * EP_PITCH + piece * 16);
        if (p.residual) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float lo, hi, rl, rh;
            Pack16<T>::unpack(v[j], lo, hi); Pack16<T>::unpack(res5[it][j], rl, rh);
            v[j] = Pack16<T>::pack(lo + rl, hi + rh);
          }
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, ry5, off5[it], 0, 0);
      }
      if (pass == 0) __syncthreads();                    // the second half overwrites the staging area
    }
    return;
  }
  if constexpr (GLU) {
#pragma unroll
    for (int u = 0; u < NC / 2; ++u)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int tok = wt * 64 + f * 32 + r;
        f32x16 o;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = acc[2 * u][f][i] * gelu_f<ES>(acc[2 * u + 1][f][i]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
          store4((T*)(lds + tok * EP_PITCH) + (wc * (NC / 2) + u) * 32 + 8 * g + 4 * h, o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]);
      }
  } else {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int tok = wt * 64 + f * 32 + r;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          store4((T*)(lds + tok * EP_PITCH) + (wc * NC + c) * 32 + 8 * g + 4 * h, acc[c][f][4 * g], acc[c][f][4 * g + 1], acc[c][f][4 * g + 2],
                 acc[c][f][4 * g + 3]);
      }
  }
  __syncthreads();
  constexpr int EPC = 8;                                 // channels per 16-byte piece
  constexpr int PPT = TNO / EPC;                         // pieces per token
  constexpr int TPI = 512 / PPT;                         // tokens per iteration
  constexpr int NIT = TM / TPI;
  if constexpr (QKV) {
    // head-major q / k / v output [which][B][heads][token][8] (what pd_attn_d8 reads): consecutive TOKENS on consecutive lanes -- a
    // (head, token) piece is 16 bytes and a head's tokens are contiguous, so TPI lanes write one run of 16 TPI bytes -- and the
    // per-(sample, head) max |k|^2 of the key rows as stored (pd_attn_d8's score bound), folded through LDS to one global atomic
    // per head and tile (non-negative floats order like their bit patterns)
    unsigned* kmax_s = (unsigned*)(lds + TM * EP_PITCH);
    if (tid < TN / 8) kmax_s[tid] = 0u;
    __syncthreads();
    const int pc = tid / TPI, tr = tid % TPI;
    const int coq = n0 + pc * EPC;
    const int Cq = p.qkv_heads * 8, which = coq / Cq, cc = coq - which * Cq;
    const long long nn = m0 / p.rows_per_sample, tokb = m0 - nn * p.rows_per_sample;      // the tile lies inside sample nn
    float kmax_run = 0.f;
    if (coq < p.N) {
      T* dst = (T*)p.y + ((((size_t)which * p.B + nn) * p.qkv_heads + (cc >> 3)) * p.rows_per_sample + tokb) * 8 + (cc & 7);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int tok = it * TPI + tr;
        if (m0 + tok >= p.M) continue;
        const u32x4 v = *(const u32x4*)(lds + tok * EP_PITCH + pc * 16);
        *(u32x4*)(dst + (size_t)tok * 8) = v;
        if (p.kmax2 && which == 1) {
          float n2 = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) { float lo, hi; Pack16<T>::unpack(v[j], lo, hi); n2 = fmaf(lo, lo, fmaf(hi, hi, n2)); }
          kmax_run = fmaxf(kmax_run, n2);
        }
      }
      if (p.kmax2 && which == 1) atomicMax(&kmax_s[pc], __float_as_uint(kmax_run));
    }
    __syncthreads();
    if (p.kmax2 && tid < TN / 8) {
      const int c0 = n0 + tid * 8;
      if (c0 < p.N && c0 / Cq == 1 && kmax_s[tid] != 0u)
        atomicMax((unsigned*)p.kmax2 + nn * p.qkv_heads + ((c0 - Cq) >> 3), kmax_s[tid]);
    }
    return;
  }
  const int piece = tid % PPT, trow = tid / PPT;
  const int NO = GLU ? p.N / 2 : p.N;
  const int co = (GLU ? ct * TNO : n0) + piece * EPC;
  const unsigned ybytes = (unsigned)min((unsigned long long)p.M * NO * ES, 0xffffffffull);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual ? p.residual : p.y), 0, p.residual ? ybytes : 0u, 0x00020000);
  u32x4 res[NIT];
  unsigned off[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const long long m = m0 + it * TPI + trow;
    off[it] = (m < p.M && co < NO) ? (unsigned)(((size_t)m * NO + co) * ES) : OOB_OFF;
    res[it] = __builtin_amdgcn_raw_buffer_load_b128(rr, off[it], 0, 0);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    u32x4 v = *(const u32x4*)(lds + (it * TPI + trow) * EP_PITCH + piece * 16);
    if (p.residual) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float lo, hi, rl, rh;
        Pack16<T>::unpack(v[j], lo, hi); Pack16<T>::unpack(res[it][j], rl, rh);
        v[j] = Pack16<T>::pack(lo + rl, hi + rh);
      }
    }
    __builtin_amdgcn_raw_buffer_store_b128(v, ry, off[it], 0, 0);
  }
}

template <typename T, int NC, bool GLU, int CK, bool QKV = false>
static int launch_linear_dma(const LinP& p, hipStream_t st) {
  constexpr int BUF = 256 * CK * 2 + 2 * NC * (CK / 16) * 1024, NBUF = 3 * BUF <= 160 * 1024 ? 3 : 2;
  constexpr int MAIN = NBUF * BUF, EPI = (NC == 5 ? 128 : 256) * (64 * NC * 2 + 16) + (QKV ? 64 * NC / 8 * 4 : 0);     // QKV: + the per-head key maxima; NC = 5: two token halves in turn
  constexpr int LDS = MAIN > EPI ? MAIN : EPI;
  auto kern = linear_dma_kernel<T, NC, GLU, CK, QKV>;
  static LdsAttr attr;
  if (!ensure_lds(attr, kern, LDS)) {
    set_error("pd_linear: cannot reserve %d bytes of LDS", LDS);
    return PD_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.t_tiles * p.c_tiles)), dim3(512), LDS, st, p);
  PD_LAUNCH_CHECK();
  return PD_OK;
}
// The GroupNorm affine in front of a projection folded into per-sample weights: (x a_n + s_n) . W^T + b = x . (W diag(a_n))^T + (b + W s_n).
// One workgroup per (32-channel tile, sample): reads the packed tile (lane (r, h) of k-step ks holds W[32 ct + r][16 ks + 8 h + j]),
// writes it scaled by a_n in the same packed order, and reduces the bias term over the tile's K (fixed order: deterministic).
template <typename T>
__global__ __launch_bounds__(256) void linear_fold_gn_kernel(const T* w, const float* bias, const float* scale, const float* shift, int K, int N_pad,
                                                            T* wn, float* bn) {
  using E = Elem<T>;
  __shared__ float red[8][32];
  const int ct = blockIdx.x, n = blockIdx.y, ksteps = K / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const size_t tile = (size_t)ct * ksteps * 512;
  T* dstn = wn + (size_t)n * (N_pad / 32) * ksteps * 512;
  float acc = 0.f;
  for (int ks = wave; ks < ksteps; ks += 4) {
    const typename E::Frag f = E::load(w + tile + (size_t)ks * 512 + lane * 8);
    float v[8];
    E::unpack(f, v);
    const float* a8 = scale + (size_t)n * K + ks * 16 + h * 8;
    const float* s8 = shift + (size_t)n * K + ks * 16 + h * 8;
    const f32x4 a0 = *(const f32x4*)a8, a1 = *(const f32x4*)(a8 + 4), s0 = *(const f32x4*)s8, s1 = *(const f32x4*)(s8 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc = fmaf(v[j], s0[j], acc); acc = fmaf(v[4 + j], s1[j], acc); v[j] *= a0[j]; v[4 + j] *= a1[j]; }
    E::store(dstn + tile + (size_t)ks * 512 + lane * 8, E::pack(v));
  }
  red[wave * 2 + h][r] = acc;
  __syncthreads();
  if (tid < 32) {
    float t = bias[ct * 32 + tid];
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][tid];
    bn[(size_t)n * N_pad + ct * 32 + tid] = t;
  }
}

template <typename T>
static int dispatch_linear_dma(const LinP& p, int variant, bool glu, hipStream_t st) {   // variant: 2 = (NC 2, CK 64), 3 = (NC 2, CK 32), 4 = (NC 4, CK 64)
  if (variant == 5) return launch_linear_dma<T, 5, false, 64>(p, st);      // (never with the fused GEGLU: chosen for plain N_pad = 320 k only)
  if (variant == 4) return glu ? launch_linear_dma<T, 4, true, 64>(p, st) : launch_linear_dma<T, 4, false, 64>(p, st);
  if (variant == 3) return glu ? launch_linear_dma<T, 2, true, 32>(p, st) : launch_linear_dma<T, 2, false, 32>(p, st);
  return glu ? launch_linear_dma<T, 2, true, 64>(p, st) : launch_linear_dma<T, 2, false, 64>(p, st);
}

}  // namespace pd

using namespace pd;

extern "C" size_t pd_linear_fold_workspace(const pd_linear_args* a) {
  // > 0 iff pd_linear can take the folded route for these arguments: 16-bit engine, GroupNorm prologue + head-major q/k/v output, no
  // residual / statistics / GEGLU, K a multiple of 64, whole 256-token tiles inside a sample, a 16-byte-aligned row stride
  if (!a || a->dtype == PD_F32 || !a->scale || !a->shift || a->qkv_heads <= 0 || a->residual || a->stats_out || a->glu) return 0;
  if (a->rows_per_sample <= 0 || a->rows_per_sample % 256 != 0 || a->M % a->rows_per_sample != 0 || a->K % 64 != 0 || a->x_stride % 8 != 0) return 0;
  if (a->N_pad % 32 != 0 || a->N != 3 * a->qkv_heads * 8) return 0;
  const size_t B = (size_t)(a->M / a->rows_per_sample);
  return B * ((size_t)(a->N_pad / 32) * (a->K / 16) * 512 * 2 + (size_t)a->N_pad * 4);
}

extern "C" int pd_linear(const pd_linear_args* a, void* stream) {
  PD_CHECK(a != nullptr, PD_ERR_ARG, "pd_linear: null args");
  PD_CHECK(a->dtype == PD_F32 || a->dtype == PD_BF16 || a->dtype == PD_F16, PD_ERR_ARG, "pd_linear: bad dtype %d", a->dtype);
  PD_CHECK(a->M > 0 && a->K > 0 && a->K % 32 == 0 && a->N > 0 && a->N % 8 == 0 && a->N_pad >= a->N && a->N_pad % 32 == 0, PD_ERR_SHAPE,
           "pd_linear: M=%lld K=%d (multiple of 32) N=%d (multiple of 8) N_pad=%d (multiple of 32)", a->M, a->K, a->N, a->N_pad);
  PD_CHECK(a->x_stride >= a->K && a->x_stride % 8 == 0, PD_ERR_SHAPE, "pd_linear: x_stride must cover K and be a multiple of 8");
  PD_CHECK(a->x && a->w_packed && a->bias && a->y, PD_ERR_ARG, "pd_linear: null pointer");
  const size_t esz = a->dtype == PD_F32 ? 4 : 2;
  const size_t xbytes = ((size_t)(a->M - 1) * a->x_stride + a->K) * esz;
  PD_CHECK(xbytes < 0x80000000ull, PD_ERR_SHAPE, "pd_linear: input exceeds 2 GiB (32-bit buffer offsets); split the rows");
  LinP p{};
  p.M = a->M; p.K = a->K; p.N = a->N; p.N_pad = a->N_pad; p.x_stride = a->x_stride;
  p.t_tiles = (int)((a->M + 127) / 128); p.c_tiles = (a->N_pad + 127) / 128;
  PD_CHECK((long long)p.t_tiles * p.c_tiles < (1ll << 30), PD_ERR_SHAPE, "pd_linear: grid too large");
  // 128-channel tiles halve the LDS reads per MFMA; 64-channel tiles when those would leave most of the 256 CUs (x 3
  // resident workgroups) idle
  const bool glu = a->glu != 0;
  PD_CHECK(!glu || (a->N % 64 == 0 && a->N_pad == a->N && !a->residual && !a->stats_out && a->qkv_heads == 0), PD_ERR_SHAPE,
           "pd_linear: glu needs N %% 64 == 0 (value | gate halves of whole 32-channel tiles), dense output, no residual / statistics");
  const bool narrow = !glu && (long long)p.t_tiles * p.c_tiles < 512;
  if (narrow) p.c_tiles = (a->N_pad + 63) / 64;
  p.xbytes = (unsigned)xbytes;
  p.x = a->x; p.w = a->w_packed; p.bias = a->bias; p.residual = a->residual; p.y = a->y;
  PD_CHECK((a->scale == nullptr) == (a->shift == nullptr), PD_ERR_ARG, "pd_linear: scale/shift must be given together");
  PD_CHECK(a->scale == nullptr || (a->rows_per_sample > 0 && a->rows_per_sample % 128 == 0 && a->M % a->rows_per_sample == 0 && a->K % 64 == 0),
           PD_ERR_SHAPE, "pd_linear: the GroupNorm prologue needs rows_per_sample %% 128 == 0 and K %% 64 == 0");
  PD_CHECK(a->qkv_heads == 0 || (a->N == 3 * a->qkv_heads * 8 && a->rows_per_sample > 0 && a->M % a->rows_per_sample == 0 && !a->residual),
           PD_ERR_SHAPE, "pd_linear: head-major output needs N == 3*heads*8, rows_per_sample and no residual");
  p.scale = a->scale; p.shift = a->shift; p.rows_per_sample = a->rows_per_sample > 0 ? a->rows_per_sample : 1;
  p.qkv_heads = a->qkv_heads; p.B = a->rows_per_sample > 0 ? (int)(a->M / a->rows_per_sample) : 0;
  PD_CHECK(a->stats_out == nullptr || (a->rows_per_sample > 0 && a->rows_per_sample % 128 == 0 && a->M % a->rows_per_sample == 0 && a->qkv_heads == 0),
           PD_ERR_SHAPE, "pd_linear: stats_out needs rows_per_sample %% 128 == 0 and dense output");
  p.stats = a->stats_out;
  PD_CHECK(a->kmax2_out == nullptr || (a->qkv_heads > 0 && a->dtype != PD_F32), PD_ERR_SHAPE,
           "pd_linear: kmax2_out needs the head-major q/k/v output and a 16-bit dtype");
  p.kmax2 = a->kmax2_out;
  // Round 4: the attention's fused q/k/v projection behind a GroupNorm (cond_unet_2d.py:176-178; diffusers Attention.group_norm ->
  // to_q / to_k / to_v) through the DMA-staged GEMM: the affine is folded into per-sample weights and biases (linear_fold_gn_kernel into
  // the caller's workspace: B x (packed weights + bias), 12.6 MB at B = 32, C = 256), so the token tiles are staged as they are (no
  // register staging, no per-element affine) and the head-major + key-bound epilogue is the register-staged kernel's.
  // PD_LIN_FOLD=0: diagnostic override (same-box A/B); 2 / 4: force the 128- / 256-channel tile.
  {
    const int fold_env = diag_env("PD_LIN_FOLD", -1);
    const size_t need = pd_linear_fold_workspace(a);
    if (fold_env != 0 && need > 0 && a->fold_ws != nullptr && a->fold_ws_bytes >= need) {
      hipStream_t st = (hipStream_t)stream;
      const int B = (int)(a->M / a->rows_per_sample);
      const size_t wbytes = (size_t)(a->N_pad / 32) * (a->K / 16) * 512 * 2;
      unsigned char* wn = (unsigned char*)a->fold_ws;
      float* bn = (float*)(wn + (size_t)B * wbytes);
      if (a->dtype == PD_F16)
        hipLaunchKernelGGL(linear_fold_gn_kernel<half_t>, dim3(a->N_pad / 32, B), dim3(256), 0, st, (const half_t*)a->w_packed, a->bias, a->scale,
                           a->shift, a->K, a->N_pad, (half_t*)wn, bn);
      else
        hipLaunchKernelGGL(linear_fold_gn_kernel<bf16_t>, dim3(a->N_pad / 32, B), dim3(256), 0, st, (const bf16_t*)a->w_packed, a->bias, a->scale,
                           a->shift, a->K, a->N_pad, (bf16_t*)wn, bn);
      PD_LAUNCH_CHECK();
      LinP q = p;
      q.scale = nullptr; q.shift = nullptr;
      q.w = wn; q.bias = bn; q.w_sstride = (long long)wbytes; q.bias_sstride = a->N_pad;
      q.t_tiles = (int)((a->M + 255) / 256);
      const int c4 = (a->N_pad + 255) / 256, c2 = (a->N_pad + 127) / 128;
      const bool wide = fold_env == 4 || (fold_env != 2 && (long long)c4 * 256 * 10 <= (long long)a->N_pad * 11);
      q.c_tiles = wide ? c4 : c2;
      if (a->dtype == PD_F16) return wide ? launch_linear_dma<half_t, 4, false, 64, true>(q, st) : launch_linear_dma<half_t, 2, false, 32, true>(q, st);
      return wide ? launch_linear_dma<bf16_t, 4, false, 64, true>(q, st) : launch_linear_dma<bf16_t, 2, false, 32, true>(q, st);
    }
  }
  // DMA-staged 256-token kernel (round 3) for the plain layers with enough tiles to fill the chip: no GroupNorm prologue, dense
  // output, no statistics, K a multiple of 64, a 16-byte-aligned row stride.  PD_LIN_DMA=0 / 2 / 3 / 4: diagnostic override (off / variant).
  {
    const int dma_env = diag_env("PD_LIN_DMA", -1);
    const bool plain = a->dtype != PD_F32 && !a->scale && a->qkv_heads == 0 && !a->stats_out && !a->kmax2_out && a->K % 64 == 0 &&
                       a->x_stride % 8 == 0 && ((size_t)a->M * (glu ? a->N / 2 : a->N) * 2) < 0xC0000000ull;
    if (plain && dma_env != 0) {
      const int t256 = (int)((a->M + 255) / 256);
      const int c4 = (a->N_pad + 255) / 256, c2 = (a->N_pad + 127) / 128;
      const bool waste4_ok = (long long)c4 * 256 * 10 <= (long long)a->N_pad * 11;
      int nc = 0;           // variant: 2 = (NC 2, CK 64), 3 = (NC 2, CK 32: two workgroups per CU), 4 = (NC 4, CK 64)
      // measured on the SD-2.1 transformer shapes at B = 32 (scripts/bench_linear.py, same box, ms: linear_kernel / variant 2 / 3 / 4):
      //   64^2 qkv 320->960 .156/.176/.152/.140   out 320->320 .056/.057/.047/.061   ff1+glu 320->2560 .330/.351/.305/.311   ff2 1280->320 .195/.171/.182/.180
      //   32^2 qkv 640->1920 .111/.129/.100/.109  out .045/.045/.039/.048            ff1+glu .261/.306/.255/.269            ff2 2560->640 .145/.144/.132/.151
      //   16^2 qkv 1280->3840 .100/.114/.100/.093 out .042/.051/.043/.044            ff1+glu .241/.263/.232/.223            ff2 5120->1280 .143/.173/.148/.142
      // -> 256-channel tiles for wide outputs (N >= 960; behind the fused GEGLU only at K >= 1280), else the two-workgroups-per-CU form
      // round 4, 320-channel tiles (NC = 5; every SD-2.1 width is a multiple of 320): same box, B = 32, ms (variant 3 / 4 / 5):
      //   64^2 qkv .153/.140/.130  out .047/.064/.045  ff2 1280->320 .157/.182/.121 (888 TF/s)     32^2 qkv .102/.103/.099  out .039/.049/.036  ff2 2560->640 .124/.144/.107
      //   16^2 qkv .094/.095/.104  out .044/.044/.051  ff2 .149/.142/.159 -> from 32 768 tokens up, with at least one workgroup per CU (not behind the fused GEGLU:
      //   its (value, gate) tiles pair up inside a wave).  PD_LIN_NC5=0: diagnostic override (same-box A/B)
      // round 6: the 256 x 256 eight-phase kernel (linear_p8.hip) for the deep / wide layers.  PD_LIN_P8=0 / 1: diagnostic override (off / wherever eligible)
      {
        const int p8_env = diag_env("PD_LIN_P8", -1);
        const long long tiles8 = (long long)t256 * c4;
        // measured (scripts/bench_linear_p8.py, B = 32, same process, x the round-5 choice): 16^2 level (M = 8192) 1.21 .. 1.46 on every
        // layer and input gradient, also at 160 tiles on 256 CUs; 32^2 level K = 640 -> N = 5120 / 2560 / 1920: 1.11 / 1.12 / 1.04;
        // slower for K = 320 (five K tiles: the prologue and the epilogue are a third of the workgroup's life) and for N = 320 / 640
        // (a 256-wide tile wastes 37 / 17 % of them): those stay on the 320- / 128-channel tiles
        const bool p8 = p8_env >= 0 ? p8_env == 1 : (a->K >= 640 && a->N_pad >= 1280 && waste4_ok && tiles8 >= 128);
        if (p8 && dma_env < 2) {
          p.t_tiles = t256; p.c_tiles = c4;
          return launch_linear_p8(p, a->dtype, glu, (hipStream_t)stream);
        }
      }
      const bool nc5_off = diag_env("PD_LIN_NC5", 1) == 0;
      const bool can5 = !glu && a->N_pad % 320 == 0 && a->N == a->N_pad;
      if (dma_env >= 2 && dma_env <= 4) nc = dma_env;
      else if (dma_env == 5 ? can5 : (can5 && !nc5_off && t256 >= 128 && (long long)t256 * (a->N_pad / 320) >= 256)) nc = 5;
      else if ((long long)t256 * c4 >= 256 && waste4_ok && a->N_pad >= 960 && (!glu || a->K >= 1280)) nc = 4;
      else if ((long long)t256 * c2 >= 256) nc = 3;
      if (nc) {
        p.t_tiles = t256; p.c_tiles = nc == 5 ? a->N_pad / 320 : (nc == 4 ? c4 : c2);
        p.xbytes = (unsigned)xbytes; p.x = a->x; p.w = a->w_packed; p.bias = a->bias; p.residual = a->residual; p.y = a->y;
        hipStream_t st = (hipStream_t)stream;
        return a->dtype == PD_F16 ? dispatch_linear_dma<half_t>(p, nc, glu, st) : dispatch_linear_dma<bf16_t>(p, nc, glu, st);
      }
    }
  }
  // 256-channel tiles (NC = 4): every staged token tile feeds twice the MFMAs (the activation stream is this kernel's larger cost,
  // DESIGN.md 9); two workgroups per CU.  PD_LIN_NC4=0/1: diagnostic override (same-box A/B).
  const int nc4_env = diag_env("PD_LIN_NC4", -1);
  // a ragged last tile (N not a multiple of 256) repeats clamped weight tiles whose results are never stored: allowed up to 10 % waste
  const int c_tiles4 = (a->N_pad + 255) / 256;
  const bool nc4 = !narrow && !a->stats_out && (long long)c_tiles4 * 256 * 10 <= (long long)a->N_pad * 11 &&
                   (nc4_env >= 0 ? nc4_env == 1 : (long long)p.t_tiles * c_tiles4 >= 512);
  if (nc4 && a->dtype != PD_F32) {
    p.c_tiles = c_tiles4;
    if (glu) return a->dtype == PD_F16 ? launch_linear<half_t, 4, true>(p, (hipStream_t)stream) : launch_linear<bf16_t, 4, true>(p, (hipStream_t)stream);
    return a->dtype == PD_F16 ? launch_linear<half_t, 4>(p, (hipStream_t)stream) : launch_linear<bf16_t, 4>(p, (hipStream_t)stream);
  }
  if (a->dtype == PD_F16) {
    if (glu) return launch_linear<half_t, 2, true>(p, (hipStream_t)stream);
    return narrow ? launch_linear<half_t, 1>(p, (hipStream_t)stream) : launch_linear<half_t, 2>(p, (hipStream_t)stream);
  }
  if (glu) return a->dtype == PD_F32 ? launch_linear<float, 2, true>(p, (hipStream_t)stream) : launch_linear<bf16_t, 2, true>(p, (hipStream_t)stream);
  if (a->dtype == PD_F32) return narrow ? launch_linear<float, 1>(p, (hipStream_t)stream) : launch_linear<float, 2>(p, (hipStream_t)stream);
  return narrow ? launch_linear<bf16_t, 1>(p, (hipStream_t)stream) : launch_linear<bf16_t, 2>(p, (hipStream_t)stream);
}

extern "C" size_t pd_token_wgrad_workspace(const pd_token_wgrad_args* a) {
  if (!a || a->M < 1 || a->K < 1 || a->N < 1) return 0;
  int nt, kt, splits, cps, nch;
  const int v = token_wgrad_variant(a);
  if (v == 0) {
    token_wgrad_plan(a->M, a->K, a->N, a->dtype == PD_F32 ? 4 : 2, &nt, &kt, &splits, &cps, &nch);
    return (size_t)splits * nt * 128 * kt * 128 * sizeof(float);
  }
  int tn, tk;
  token_wgrad_tile(v, &tn, &tk);
  token_wgrad_plan_dma(a->M, a->K, a->N, tn, tk, &nt, &kt, &splits, &cps, &nch);
  return (size_t)splits * nt * tn * kt * tk * sizeof(float);
}

extern "C" int pd_token_wgrad(const pd_token_wgrad_args* a, void* stream) {
  PD_CHECK(a != nullptr, PD_ERR_ARG, "pd_token_wgrad: null args");
  PD_CHECK(a->dtype == PD_F32 || a->dtype == PD_BF16 || a->dtype == PD_F16, PD_ERR_ARG, "pd_token_wgrad: bad dtype %d", a->dtype);
  PD_CHECK(a->M > 0 && a->K > 0 && a->K % 8 == 0 && a->N > 0 && a->N % 8 == 0 && a->x_stride >= a->K && a->dy_stride >= a->N &&
               a->x_stride % 8 == 0 && a->dy_stride % 8 == 0, PD_ERR_SHAPE, "pd_token_wgrad: K, N and the row strides must be multiples of 8");
  PD_CHECK(a->x && a->dy && a->dw && a->slab, PD_ERR_ARG, "pd_token_wgrad: null pointer");
  const size_t esz = a->dtype == PD_F32 ? 4 : 2;
  const size_t xbytes = ((size_t)(a->M - 1) * a->x_stride + a->K) * esz, dybytes = ((size_t)(a->M - 1) * a->dy_stride + a->N) * esz;
  PD_CHECK(xbytes < 0x80000000ull && dybytes < 0x80000000ull, PD_ERR_SHAPE, "pd_token_wgrad: operands must be < 2 GiB (32-bit buffer offsets)");
  TwP p{};
  p.M = a->M; p.K = a->K; p.N = a->N; p.x_stride = a->x_stride; p.dy_stride = a->dy_stride;
  const int variant = token_wgrad_variant(a);
  int tn = 128, tk = 128;
  if (variant == 0) token_wgrad_plan(a->M, a->K, a->N, (int)esz, &p.n_tiles, &p.k_tiles, &p.splits, &p.chunks_per_split, &p.nchunks);
  else {
    token_wgrad_tile(variant, &tn, &tk);
    token_wgrad_plan_dma(a->M, a->K, a->N, tn, tk, &p.n_tiles, &p.k_tiles, &p.splits, &p.chunks_per_split, &p.nchunks);
  }
  p.NP = p.n_tiles * tn; p.KP = p.k_tiles * tk;
  const size_t per_split = (size_t)p.NP * p.KP * sizeof(float);
  if ((size_t)p.splits * per_split > a->slab_bytes) {
    const int s = (int)(a->slab_bytes / per_split);
    PD_CHECK(s >= 1, PD_ERR_ARG, "pd_token_wgrad: slab of %zu bytes cannot hold one split (%zu bytes)", a->slab_bytes, per_split);
    p.chunks_per_split = (p.nchunks + s - 1) / s;
    p.splits = (p.nchunks + p.chunks_per_split - 1) / p.chunks_per_split;
  }
  p.xbytes = (unsigned)xbytes; p.dybytes = (unsigned)dybytes;
  p.x = a->x; p.dy = a->dy; p.slab = a->slab;
  p.xcd_order = diag_env("PD_TW_XCD", 1) != 0;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)(p.splits * p.n_tiles * p.k_tiles);
  PD_CHECK(a->stage >= 0 && a->stage <= 2, PD_ERR_ARG, "pd_token_wgrad: stage %d", a->stage);
  if (a->stage == 2) goto fold;
  if (variant != 0) {
    const int rc = variant == 1 ? launch_token_wgrad_dma<bf16_t, 5, 2, 2>(p, st)
                 : variant == 2 ? launch_token_wgrad_dma<bf16_t, 2, 5, 2>(p, st) : launch_token_wgrad_dma<bf16_t, 2, 2, 3>(p, st);
    if (rc != PD_OK) return rc;
  } else if (a->dtype == PD_BF16) {
    constexpr int LDS = 2 * 2 * 4 * 64 * 64;            // 64 KiB
    hipLaunchKernelGGL(token_wgrad_kernel<bf16_t>, dim3(grid), dim3(256), LDS, st, p);
  } else if (a->dtype == PD_F16) {                     // fp16 training (round 5): the register-staged form on the f16 MFMA
    constexpr int LDS = 2 * 2 * 4 * 64 * 64;
    hipLaunchKernelGGL(token_wgrad_kernel<half_t>, dim3(grid), dim3(256), LDS, st, p);
  } else {
    constexpr int LDS = 2 * 2 * 4 * 64 * 128;           // 128 KiB
    static LdsAttr attr;
    if (!ensure_lds(attr, token_wgrad_kernel<float>, LDS)) {
      set_error("pd_token_wgrad: cannot reserve %d bytes of LDS", LDS);
      return PD_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(token_wgrad_kernel<float>, dim3(grid), dim3(256), LDS, st, p);
  }
  PD_LAUNCH_CHECK();
  if (a->stage == 1) return PD_OK;
fold:
  hipLaunchKernelGGL(token_wgrad_reduce_kernel, dim3((unsigned)((a->K + 255) / 256), (unsigned)a->N), dim3(256), 0, st, (const float*)a->slab, a->dw,
                     p.splits, p.NP, p.KP, a->N, a->K, a->accumulate);
  PD_LAUNCH_CHECK();
  return PD_OK;
}
