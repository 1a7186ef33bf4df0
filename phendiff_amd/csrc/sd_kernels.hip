// Kernels of the Stable-Diffusion tier (diffusers UNet2DConditionModel / Transformer2DModel blocks):
//   pd_attn_d64  : softmax(q k^T / 8) v for head_dim 64 -- BasicTransformerBlock.attn1 (self, N up to 4096) and .attn2 (cross,
//                  77 context tokens); the contraction is genuinely dense here, so both products run as full MFMA tiles
//   pd_layernorm : nn.LayerNorm(C) over the channels of each token (norm1/2/3 of BasicTransformerBlock)
//   pd_geglu     : GEGLU gate  h * gelu(g)  of FeedForward (exact erf form, F.gelu default)
// Everything else of the SD UNet (ResnetBlock2D, GroupNorm, every Linear as a 1x1 conv on NHWC tokens, down/up-sampling)
// reuses pd_conv / pd_gn_finalize / pd_temb.
#include <stdlib.h>
#include "pd_common.h"
#include "pd_stage.h"
#include "pd_d64.h"

namespace pd {

#ifndef PD_ATTN64_PIPE      // attn_d64_kernel, two-fragment form.  0: sub-tile by sub-tile; 1: the two query fragments as a two-stage pipeline; 2 (diagnostic builds): + both sub-tiles' scores up front, Q parked in LDS -- parity-green, 912 vs 928 TF/s: the vector issue port binds, not the overlap
#define PD_ATTN64_PIPE 1
#endif
#ifndef PD_ATTN64_PARKQ     // 1: the Q fragments are parked in LDS (lane-private slots) and re-read per MFMA: 32 registers
#define PD_ATTN64_PARKQ 0
#endif
#ifndef PD_ATTN64_QK_SPLIT
#define PD_ATTN64_QK_SPLIT 0
#endif
#ifndef PD_ATTN64_LAZYMAX   // 1: the pipelined form takes the row maximum only when the lane sums of the probabilities say it has to (0: every sub-tile)
#define PD_ATTN64_LAZYMAX 1
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Attention, head_dim 64.  Workgroup = 4 waves = 128 queries of one (batch, head); keys/values stream through LDS in
// double-buffered tiles of 64.  Per wave and 32-key sub-tile:
//   S^T[key][query] = K[32 x 64] . Q^T[64 x 32]        4 MFMA k-steps (Q fragments live in registers, pre-scaled by
//                                                      64^-1/2 * log2 e); D layout: query on the lane, 16 keys in registers
//   online softmax, lane-local (one cross-half exchange for the row max); P stays in registers ...
//   O^T[d][query]  += V^T[64 x 32 keys] . P^T          ... as the B operand: 2 row tiles x 2 k-steps.  V sits row-major
//                                                      [key][d] in LDS; the A operand V^T is a TRANSPOSED read
//                                                      (ds_read_b64_tr_b16, rows chosen per lane to match P's key order).
// QB = 32-query fragments per wave (workgroup = 4 waves = 128 QB queries).  QB = 2: every K and V^T fragment read from LDS
// feeds two MFMAs, halving the LDS traffic per MFMA (with one fragment the LDS pipe -- 12 KB-equivalents per 8 MFMAs, the
// transposed reads at half rate -- is busier than the matrix pipe); costs a workgroup of occupancy (2 per CU).
template <typename T, int QB>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && QB == 1) ? 3 : 2) void attn_d64_kernel(const pd_attn_d64_args a) {
  using E = Elem<T>;
  using Frag = typename E::Frag;
  using X = D64<T>;
  constexpr int KT = 64, KP = X::KP, VP = X::VP, ES = E::BYTES;
  constexpr int KBYTES = KT * KP, VBYTES = KT * VP;
  constexpr int PIECES = KT * 64 / 8 / 256;            // 8-element pieces per thread per tensor (2)
  constexpr int QPW = 32 * QB, QPB = 4 * QPW;          // queries per wave / workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [2][K tile | V tile]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nqb = (a.Nq + QPB - 1) / QPB;
  const int total = nqb * a.heads * a.B;
  int item = blockIdx.x;
  if ((total & 7) == 0) item = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);     // query blocks of one head share an XCD / L2
  const int qb = item % nqb, head = (item / nqb) % a.heads, b = item / (nqb * a.heads);
  const T* qp = (const T*)a.q + (size_t)b * a.Nq * a.q_stride + head * 64;
  const T* kp = (const T*)a.k + (size_t)b * a.Nkv * a.kv_stride + head * 64;
  const T* vp = (const T*)a.v + (size_t)b * a.Nkv * a.kv_stride + head * 64;

  // Q^T fragments (B operand): lane (query, h), k-step ks: d = 16 ks + 8 h + j, pre-scaled
  const float qscale = 0.125f * 1.4426950408889634f;
  int query[QB];
  Frag qf[QB][4];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    query[j] = qb * QPB + wave * QPW + j * 32 + r;
    const int qc = min(query[j], a.Nq - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float v[8];
      E::unpack(E::load(qp + (size_t)qc * a.q_stride + 16 * ks + 8 * h), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= qscale;
      qf[j][ks] = E::pack(v);
    }
  }
  // PIPE2: the Q fragments live in LDS, each lane reading back exactly the 16-byte slots it wrote ([fragment][k-step][thread]: consecutive lanes,
  // consecutive slots -- no conflict, no barrier): 32 registers for the second sub-tile's scores (see the key loop)
  constexpr bool PIPE2 = PD_ATTN64_PIPE >= 2 && QB == 2 && sizeof(T) == 2;
  constexpr bool PARKQ = (PIPE2 || PD_ATTN64_PARKQ) && QB == 2 && sizeof(T) == 2;
  unsigned char* qpark = lds + 2 * (KBYTES + VBYTES) + tid * 16;
  if constexpr (PARKQ) {
#pragma unroll
    for (int j = 0; j < QB; ++j)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) E::store(qpark + (j * 4 + ks) * 4096, qf[j][ks]);
  }
  auto qfrag = [&](int j, int ks) __attribute__((always_inline)) -> Frag {
    if constexpr (PARKQ) return E::load(qpark + (j * 4 + ks) * 4096);
    else return qf[j][ks];
  };
  // Deferred-rescale online softmax (as pd_attn_d8): `m` is a REFERENCE maximum (log2 domain) shared by both lane halves of a
  // query, p = exp2(s - m).  It is only raised when some score of the sub-tile exceeds m + RESCALE_THR (p <= 2^THR
  // otherwise: harmless in fp32 / bf16), so the accumulator rescale (32 multiplies), the cross-half exchange and the second
  // exp2 leave the common path; the first sub-tile always takes the exact maximum.  The decision is wave-uniform.
  // -m rides in the C operand of the first QK^T MFMA (a register tile that only changes on a rescale), so the scores leave
  // the matrix pipe ready for exp2.
  constexpr float RESCALE_THR = 16.0f;
  f32x16 o0[QB], o1[QB], negm[QB];                   // O^T row tiles d 0..31, 32..63
  float m[QB], l[QB];                                // l: this lane half's share of the row sum
#pragma unroll
  for (int j = 0; j < QB; ++j) { o0[j] = (f32x16)(0.f); o1[j] = (f32x16)(0.f); negm[j] = (f32x16)(0.f); m[j] = 0.f; l[j] = 0.f; }
  bool first = true;

  Frag stk[PIECES], stv[PIECES];
  // K / V tiles through buffer resources (round 6): per-lane offset fixed, the tile's offset scalar -- no 64-bit vector address arithmetic and no
  // bounds test per piece (a key past Nkv lies beyond the resource: zeros), ~20 vector instructions per 64-key tile of a kernel bound by that port
  const unsigned kv_bytes = (unsigned)(((size_t)(a.Nkv - 1) * a.kv_stride + 64) * ES);      // this (batch, head)'s slice ends with its last key's 64 channels
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, kv_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, kv_bytes, 0x00020000);
  unsigned kvoff[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) { const int pc = tid + 256 * i; kvoff[i] = (unsigned)(((pc >> 3) * a.kv_stride + (pc & 7) * 8) * ES); }
  auto issue = [&](int k0) {
    const unsigned so = (unsigned)k0 * (unsigned)a.kv_stride * ES;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      stk[i] = E::load_buf(rk, kvoff[i], so);
      stv[i] = E::load_buf(rv, kvoff[i], so);
    }
  };
  auto commit = [&](int buf) {
    unsigned char* kb = lds + buf * (KBYTES + VBYTES);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int pc = tid + 256 * i, row = pc >> 3, sub = pc & 7;
      E::store(kb + row * KP + sub * 8 * ES, stk[i]);
      E::store(kb + KBYTES + row * VP + sub * 8 * ES, stv[i]);
    }
  };
  const int k_lane = r * KP + 8 * h * ES;             // K row fragment: key r, d = 16 ks + 8 h + (0..7)
  const int v_lane = X::vt_lane_off(lane);

  issue(0);
  commit(0);
  if (KT < a.Nkv) issue(KT);
  __syncthreads();
  for (int k0 = 0, cur = 0; k0 < a.Nkv; k0 += KT, cur ^= 1) {
    const unsigned char* kb = lds + cur * (KBYTES + VBYTES);
    const unsigned char* vb = kb + KBYTES;
    // the three stages of a 32-key sub-tile
    auto qk = [&](int sub, f32x16 (&s)[QB]) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const Frag kf = E::load(kb + k_lane + sub * 32 * KP + ks * 16 * ES);
#pragma unroll
        for (int j = 0; j < QB; ++j) s[j] = E::mma(kf, qfrag(j, ks), ks == 0 ? negm[j] : s[j]);     // s = S - m
      }
    };
    // (`ahead`: the scores of the NEXT sub-tile, already computed against the old reference -- a rescale moves them too)
    auto soft = [&](f32x16 (&s)[QB], f32x16 (*ahead)[QB]) __attribute__((always_inline)) {
      float tmax[QB];
      bool need = first;
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        float t = fmaxf(fmaxf(s[j][0], s[j][1]), s[j][2]);   // v_max3_f32 chain
#pragma unroll
        for (int i = 3; i < 15; i += 2) t = fmaxf(fmaxf(t, s[j][i]), s[j][i + 1]);
        tmax[j] = fmaxf(t, s[j][15]);
        need = need || tmax[j] > RESCALE_THR;
      }
      if (__builtin_amdgcn_ballot_w64(need) != 0) {
#pragma unroll
        for (int j = 0; j < QB; ++j) {
          const float tq = fmaxf(tmax[j], __shfl_xor(tmax[j], 32));   // finite: every sub-tile visited holds at least one real key
          const float delta = first ? tq : fmaxf(0.f, tq);      // how far the reference moves up (first: to the exact maximum)
          const float alpha = __builtin_amdgcn_exp2f(-delta);   // first: l = o = 0, any finite factor will do
          m[j] += delta;
          l[j] *= alpha;
#pragma unroll
          for (int i = 0; i < 16; ++i) { o0[j][i] *= alpha; o1[j][i] *= alpha; s[j][i] -= delta; }
          if (ahead) {
#pragma unroll
            for (int i = 0; i < 16; ++i) (*ahead)[j][i] -= delta;
          }
          negm[j] = (f32x16)(-m[j]);
        }
        first = false;
      }
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        f32x2 acc2 = (f32x2)(0.f);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {              // packed fp32 add: one VALU slot per score pair
          f32x2 d;
          d.x = __builtin_amdgcn_exp2f(s[j][i]); d.y = __builtin_amdgcn_exp2f(s[j][i + 1]);
          s[j][i] = d.x; s[j][i + 1] = d.y;
          acc2 += d;
        }
        l[j] += acc2.x + acc2.y;
      }
    };
    auto pv = [&](int sub, f32x16 (&s)[QB]) __attribute__((always_inline)) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const unsigned char* vs = vb + v_lane + (sub * 32 + 16 * st) * VP;
        const Frag v0 = X::load_vt(vs), v1 = X::load_vt(vs + 32 * ES);
#pragma unroll
        for (int j = 0; j < QB; ++j) {
          const Frag pf = X::pack_p(s[j], st);
          o0[j] = E::mma(v0, pf, o0[j]);
          o1[j] = E::mma(v1, pf, o1[j]);
        }
      }
    };
    if constexpr (PD_ATTN64_PIPE && QB == 2) {      // (dispatch: QB = 2 only with Nkv % 64 == 0)
      // Round 6, whole 64-key tiles of the two-fragment form: the wave's two query fragments are walked as a two-stage pipeline -- scores of
      // fragment 0, scores of fragment 1 (4 + 4 MFMAs from the same K fragments), then softmax(0) while the matrix pipe still works on the
      // scores of 1, PV(0), softmax(1) under PV(0), PV(1).  In the old order (both score chains interleaved, both softmaxes, both PVs) a wave's
      // own MFMAs and vector work never overlapped -- only the other wave of the SIMD filled the gaps (matrix pipe 0.41 busy, LAB_r6 section 3).
      // No extra registers: the V^T fragments are read per query fragment (the LDS pipe was 8.5 % busy).
      // PIPE2: the scores of BOTH sub-tiles are issued up front (8 + 8 MFMAs; the second sub-tile's run under the first one's softmaxes)
      f32x16 sall[PIPE2 ? 2 : 1][QB];
      if constexpr (PIPE2) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const Frag kf1 = E::load(kb + k_lane + sub * 32 * KP + ks * 16 * ES);
#pragma unroll
            for (int j = 0; j < QB; ++j) sall[sub][j] = E::mma(kf1, qfrag(j, ks), ks == 0 ? negm[j] : sall[sub][j]);
          }
      }
#pragma unroll
      for (int sub = 0; sub < KT / 32; ++sub) {
        if constexpr (!PIPE2) {
#if PD_ATTN64_QK_SPLIT      // diagnostic: the two fragments' score chains one after the other (K fragments held in 16 registers)
          Frag kf[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) kf[ks] = E::load(kb + k_lane + sub * 32 * KP + ks * 16 * ES);
#pragma unroll
          for (int j = 0; j < QB; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) sall[0][j] = E::mma(kf[ks], qfrag(j, ks), ks == 0 ? negm[j] : sall[0][j]);
#else
          qk(sub, sall[0]);
#endif
        }
        const bool was_first = first;
#pragma unroll
        for (int j = 0; j < QB; ++j) {
          f32x16& sc = sall[PIPE2 ? sub : 0][j];
          // Common path WITHOUT the row maximum (16 v_max3 + compares of ~130 vector instructions per sub-tile; the vector issue port is what binds
          // this kernel): exponentiate against the current reference and look at the lane's SUM of the 16 probabilities, which the row sum needs
          // anyway -- sum <= 2^15 proves every p <= 2^15 (the deferred-rescale invariant p <= 2^THR); a larger or non-finite sum on any lane sends
          // the wave to the exact path below, which recomputes the four score MFMAs (K fragments re-read from LDS) and takes the maximum as before.
          bool exact = was_first || !PD_ATTN64_LAZYMAX;     // the first sub-tile sets the reference to the exact maximum
          f32x16 pe;                                          // the probabilities (the scores stay intact for the exact path)
          if (!exact) {
            f32x2 acc2 = (f32x2)(0.f);
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
              f32x2 d;
              d.x = __builtin_amdgcn_exp2f(sc[i]); d.y = __builtin_amdgcn_exp2f(sc[i + 1]);
              pe[i] = d.x; pe[i + 1] = d.y;
              acc2 += d;
            }
            const float a1 = acc2.x + acc2.y;
            exact = __builtin_amdgcn_ballot_w64(!(a1 <= 32768.0f)) != 0;       // (NaN / inf fail the comparison too)
            if (!exact) l[j] += a1;
          }
          if (exact) {
            float t = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) t = fmaxf(fmaxf(t, sc[i]), sc[i + 1]);
            t = fmaxf(t, sc[15]);
            float delta = 0.f;
            if (__builtin_amdgcn_ballot_w64(was_first || t > RESCALE_THR) != 0) {
              const float tq = fmaxf(t, __shfl_xor(t, 32));
              delta = was_first ? tq : fmaxf(0.f, tq);
              const float alpha = __builtin_amdgcn_exp2f(-delta);
              m[j] += delta;
              l[j] *= alpha;
#pragma unroll
              for (int i = 0; i < 16; ++i) { o0[j][i] *= alpha; o1[j][i] *= alpha; }
              if constexpr (PIPE2) {
                if (sub == 0) {                    // the second sub-tile's scores were computed against the old reference
#pragma unroll
                  for (int i = 0; i < 16; ++i) sall[1][j][i] -= delta;
                }
              }
              negm[j] = (f32x16)(-m[j]);
            }
            f32x2 acc2 = (f32x2)(0.f);
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
              f32x2 d;
              d.x = __builtin_amdgcn_exp2f(sc[i] - delta); d.y = __builtin_amdgcn_exp2f(sc[i + 1] - delta);
              pe[i] = d.x; pe[i + 1] = d.y;
              acc2 += d;
            }
            l[j] += acc2.x + acc2.y;
          }
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            const unsigned char* vs = vb + v_lane + (sub * 32 + 16 * st) * VP;
            const Frag v0 = X::load_vt(vs), v1 = X::load_vt(vs + 32 * ES);
            const Frag pf = X::pack_p(pe, st);
            o0[j] = E::mma(v0, pf, o0[j]);
            o1[j] = E::mma(v1, pf, o1[j]);
          }
        }
        first = false;
      }
    } else {
#pragma unroll
    for (int sub = 0; sub < KT / 32; ++sub) {
      if (k0 + sub * 32 < a.Nkv) {                    // workgroup-uniform
        f32x16 s[QB];
        qk(sub, s);
        if (k0 + sub * 32 + 32 > a.Nkv) {              // keys beyond the context length
#pragma unroll
          for (int j = 0; j < QB; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i)
              if (k0 + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * h >= a.Nkv) s[j][i] = -INFINITY;
        }
        soft(s, nullptr);
        pv(sub, s);
      }
    }
    }
    if (k0 + KT < a.Nkv) {
      commit(cur ^ 1);
      if (k0 + 2 * KT < a.Nkv) issue(k0 + 2 * KT);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    l[j] += __shfl_xor(l[j], 32);
    if (query[j] < a.Nq) {
      const float inv = 1.0f / l[j];
      if (a.lse && h == 0) a.lse[((size_t)b * a.heads + head) * a.Nq + query[j]] = m[j] + __log2f(l[j]);     // log2 domain, scale included
      T* dst = (T*)a.out + ((size_t)b * a.Nq + query[j]) * a.out_stride + head * 64 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {                      // register 4g + i <-> d = 8g + 4h + i (+32 for the second row tile)
        store4(dst + 8 * g, o0[j][4 * g] * inv, o0[j][4 * g + 1] * inv, o0[j][4 * g + 2] * inv, o0[j][4 * g + 3] * inv);
        store4(dst + 32 + 8 * g, o1[j][4 * g] * inv, o1[j][4 * g + 1] * inv, o1[j][4 * g + 2] * inv, o1[j][4 * g + 3] * inv);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm over C per token, 8-element pieces: a wave normalises TPW = 4 / MAXP tokens at once (MAXP = pieces per lane: 1 for C <= 512,
// 2 for C <= 1024, 4 for C <= 2048) with every load issued before the first reduction -- one token per wave left a single 16-byte load
// per lane in flight (3.4 TB/s at C = 320, round 3)
template <typename T, int MAXP>
__global__ __launch_bounds__(256) void layernorm_kernel(const pd_layernorm_args a) {
  using E = Elem<T>;
  constexpr int TPW = 4 / MAXP;
  const int lane = threadIdx.x & 63;
  const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * TPW;
  if (row0 >= a.rows) return;
  const int pieces = a.C / 8;
  float v[TPW][MAXP][8];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const long long row = row0 + t < a.rows ? row0 + t : a.rows - 1;      // (a partial last group re-reads the last row; not stored)
    const T* x = (const T*)a.x + row * a.C;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int pc = lane + 64 * i;
      if (pc < pieces) E::unpack(E::load(x + pc * 8), v[t][i]);
    }
  }
  float gm[MAXP][8], bt[MAXP][8];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pc = lane + 64 * i;
    if (pc < pieces) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { gm[i][j] = a.gamma[pc * 8 + j]; bt[i][j] = a.beta[pc * 8 + j]; }
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
      if (lane + 64 * i < pieces) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[t][i][j];
      }
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) s += __shfl_xor(s, msk);
    const float mean = s / (float)a.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
      if (lane + 64 * i < pieces) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[t][i][j] - mean; q += d * d; }
      }
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) q += __shfl_xor(q, msk);
    const float rstd = 1.0f / sqrtf(q / (float)a.C + a.eps);
    if (row0 + t < a.rows) {
      T* y = (T*)a.y + (row0 + t) * a.C;
#pragma unroll
      for (int i = 0; i < MAXP; ++i) {
        const int pc = lane + 64 * i;
        if (pc < pieces) {
          float o[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (v[t][i][j] - mean) * rstd * gm[i][j] + bt[i][j];
          E::store(y + pc * 8, E::pack(o));
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void geglu_kernel(const pd_geglu_args a) {
  using E = Elem<T>;
  const int pieces = a.inner / 8;
  const size_t total = (size_t)a.rows * pieces;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t row = idx / pieces;
    const int pc = (int)(idx - row * pieces);
    float hv[8], gv[8], o[8];
    E::unpack(E::load((const T*)a.x + row * 2 * a.inner + pc * 8), hv);
    E::unpack(E::load((const T*)a.x + row * 2 * a.inner + a.inner + pc * 8), gv);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = hv[j] * (0.5f * gv[j] * (1.0f + erff(gv[j] * 0.7071067811865476f)));
    E::store((T*)a.y + row * a.inner + pc * 8, E::pack(o));
  }
}

template <typename T, int QB>
static int launch_attn_d64(const pd_attn_d64_args* a, hipStream_t st) {
  constexpr int LDS = 2 * 64 * (D64<T>::KP + D64<T>::VP) + (((PD_ATTN64_PIPE >= 2 || PD_ATTN64_PARKQ) && QB == 2 && sizeof(T) == 2) ? QB * 4 * 4096 : 0);
  auto kern = attn_d64_kernel<T, QB>;
  static LdsAttr attr;
  if (!ensure_lds(attr, kern, LDS)) {
    set_error("pd_attn_d64: cannot reserve %d bytes of LDS", LDS);
    return PD_ERR_LAUNCH;
  }
  constexpr int QPB = 128 * QB;
  hipLaunchKernelGGL(kern, dim3(((a->Nq + QPB - 1) / QPB) * a->heads * a->B), dim3(256), LDS, st, *a);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

}  // namespace pd

using namespace pd;

extern "C" int pd_attn_d64(const pd_attn_d64_args* a, void* stream) {
  PD_CHECK(a != nullptr, PD_ERR_ARG, "pd_attn_d64: null args");
  PD_CHECK(a->B > 0 && a->heads > 0 && a->Nq > 0 && a->Nkv > 0, PD_ERR_SHAPE, "pd_attn_d64: bad shape");
  PD_CHECK(a->q && a->k && a->v && a->out, PD_ERR_ARG, "pd_attn_d64: null pointer");
  PD_CHECK(a->q_stride >= a->heads * 64 && a->kv_stride >= a->heads * 64 && a->out_stride >= a->heads * 64 && a->q_stride % 8 == 0 &&
               a->kv_stride % 8 == 0 && a->out_stride % 8 == 0, PD_ERR_SHAPE, "pd_attn_d64: strides must cover heads*64 channels and be multiples of 8");
  PD_CHECK((long long)((a->Nq + 127) / 128) * a->heads * a->B < (1ll << 31), PD_ERR_SHAPE, "pd_attn_d64: grid too large");
  PD_CHECK((unsigned long long)a->Nkv * (unsigned long long)a->kv_stride * 4ull < (1ull << 32), PD_ERR_SHAPE, "pd_attn_d64: one sample's K / V rows must span < 4 GiB (32-bit buffer offsets)");
  if (a->dtype == PD_F32) return launch_attn_d64<float, 1>(a, (hipStream_t)stream);
  if (a->dtype == PD_BF16) {
    // two query fragments per wave once that still fills the chip (256 CUs x 2 resident workgroups) and the key sequence is
    // long enough for the LDS traffic to matter
    const bool qb1_only = diag_env("PD_ATTN64_QB1", 0) != 0;      // diagnostic: same-box A/B
    const bool wide = !qb1_only && a->Nkv >= 512 && (!PD_ATTN64_PIPE || a->Nkv % 64 == 0) && (long long)(a->Nq / 256) * a->heads * a->B >= 1024;
    return wide ? launch_attn_d64<bf16_t, 2>(a, (hipStream_t)stream) : launch_attn_d64<bf16_t, 1>(a, (hipStream_t)stream);
  }
  if (a->dtype == PD_F16) {
    const bool wide = a->Nkv >= 512 && (!PD_ATTN64_PIPE || a->Nkv % 64 == 0) && (long long)(a->Nq / 256) * a->heads * a->B >= 1024;
    return wide ? launch_attn_d64<half_t, 2>(a, (hipStream_t)stream) : launch_attn_d64<half_t, 1>(a, (hipStream_t)stream);
  }
  set_error("pd_attn_d64: bad dtype");
  return PD_ERR_ARG;
}

extern "C" int pd_layernorm(const pd_layernorm_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->rows > 0 && a->C > 0 && a->C % 8 == 0 && a->C <= 2048 && a->x && a->y && a->gamma && a->beta, PD_ERR_ARG,
           "pd_layernorm: bad args (C must be a multiple of 8, <= 2048)");
  const int maxp = a->C <= 512 ? 1 : (a->C <= 1024 ? 2 : 4);
  const int rows_per_wg = 4 * (4 / maxp);
  const unsigned grid = (unsigned)((a->rows + rows_per_wg - 1) / rows_per_wg);
  hipStream_t st = (hipStream_t)stream;
#define PD_LN_LAUNCH(T)                                                                                             \
  do {                                                                                                              \
    if (maxp == 1) hipLaunchKernelGGL((layernorm_kernel<T, 1>), dim3(grid), dim3(256), 0, st, *a);                  \
    else if (maxp == 2) hipLaunchKernelGGL((layernorm_kernel<T, 2>), dim3(grid), dim3(256), 0, st, *a);             \
    else hipLaunchKernelGGL((layernorm_kernel<T, 4>), dim3(grid), dim3(256), 0, st, *a);                            \
  } while (0)
  if (a->dtype == PD_F32) PD_LN_LAUNCH(float);
  else if (a->dtype == PD_BF16) PD_LN_LAUNCH(bf16_t);
  else if (a->dtype == PD_F16) PD_LN_LAUNCH(half_t);
  else { set_error("pd_layernorm: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_geglu(const pd_geglu_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->rows > 0 && a->inner > 0 && a->inner % 8 == 0 && a->x && a->y, PD_ERR_ARG, "pd_geglu: bad args");
  const size_t total = (size_t)a->rows * (a->inner / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (a->dtype == PD_F32) hipLaunchKernelGGL(geglu_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(geglu_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(geglu_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else { set_error("pd_geglu: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}
