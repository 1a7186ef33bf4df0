// Backward kernels of the Stable-Diffusion tier (what autograd runs under `accelerator.backward(loss)`, utils_training.py:436,
// over diffusers' Transformer2DModel blocks when _SD_prediction_wrapper trains the SD UNet, utils_training.py:459-496):
//   pd_attn_d64_bwd  : gradient of softmax(q k^T / 8) v, head_dim 64 (self attention and the 77-token cross attention)
//   pd_layernorm_bwd : gradient of nn.LayerNorm(C) (+ the skip-connection gradient folded into dx)
//   pd_geglu_bwd     : gradient of the GEGLU gate h * gelu(g)
#include "pd_common.h"
#include "pd_stage.h"
#include "pd_d64.h"

namespace pd {

// ---------------------------------------------------------------------------------------------------------------------
// Attention backward, head_dim 64.  P is recomputed from the forward's log-sum-exp (log2 domain); two kernels so that every
// reduction is lane-local and no atomics are needed (same split as pd_attn_d8_bwd):
//   dQ kernel    : query on the lane.  Per 32-key sub-tile  S^T = K.Q^T,  dP^T = V.dO^T  (K / V rows from LDS, Q^T / dO^T
//                  fragments in registers),  dS^T = P^T o (dP^T - delta)  lane-local,  dQ^T += K^T . dS^T  with K^T a
//                  transposed LDS read of the same K tile and dS^T packed straight from the accumulator registers.
//   dK/dV kernel : key on the lane.  Per 32-query sub-tile  S = Q.K^T,  dP = dO.V^T  (Q / dO rows from LDS, K^T / V^T
//                  fragments in registers),  dV^T += dO^T . P,  dK^T += Q^T . dS  (transposed reads of the dO / Q tiles).
// delta[i] = sum_d o[i][d] * do[i][d] comes from a small pre-pass.
template <typename T>
__global__ __launch_bounds__(256) void attn_d64_delta_kernel(const pd_attn_d64_bwd_args a) {
  using E = Elem<T>;
  const size_t total = (size_t)a.B * a.Nq * a.heads * 8;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  size_t row = 0; int head = 0;
  if (idx < total) {
    const int sub = (int)(idx & 7);
    head = (int)((idx >> 3) % a.heads);
    row = idx / ((size_t)a.heads * 8);                 // b * Nq + i
    float o[8], d[8];
    E::unpack(E::load((const T*)a.o + row * a.o_stride + head * 64 + sub * 8), o);
    E::unpack(E::load((const T*)a.dout + row * a.o_stride + head * 64 + sub * 8), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += o[j] * d[j];
  }
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
  if (idx < total && (idx & 7) == 0) {
    const size_t b = row / a.Nq, i = row - b * a.Nq;
    a.delta[(b * a.heads + head) * a.Nq + i] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_d64_dq_kernel(const pd_attn_d64_bwd_args a, const int xcd_order) {
  using E = Elem<T>;
  using Frag = typename E::Frag;
  using X = D64<T>;
  using XB = D64B<T>;                                  // tile layout: pd_d64.h (16-bit: 128-byte rows, XOR-swizzled slots)
  constexpr int KT = 64, VP = XB::P, ES = E::BYTES;
  constexpr int TB = KT * VP;                          // bytes of one [64][64] tile
  constexpr int PIECES = KT * 64 / 8 / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [2][K tile | V tile]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nqb = (a.Nq + 127) / 128;
  // round 6: the query blocks of one (sample, head) stream the SAME K / V: one XCD takes a contiguous run of the block list (the forward
  // kernel always did; PMC before: 3.4 x the algorithmic bytes left L2).  PD_ATTN64_BWD_XCD=0: diagnostic override (same-box A/B)
  const int item = xcd_order ? xcd_chunk_index((int)blockIdx.x, nqb * a.heads * a.B) : (int)blockIdx.x;
  const int qb = item % nqb, head = (item / nqb) % a.heads, b = item / (nqb * a.heads);
  const T* qp = (const T*)a.q + (size_t)b * a.Nq * a.q_stride + head * 64;
  const T* kp = (const T*)a.k + (size_t)b * a.Nkv * a.kv_stride + head * 64;
  const T* vp = (const T*)a.v + (size_t)b * a.Nkv * a.kv_stride + head * 64;
  const T* dop = (const T*)a.dout + (size_t)b * a.Nq * a.o_stride + head * 64;

  const int query = qb * 128 + wave * 32 + r, qc = min(query, a.Nq - 1);
  const float qscale = 0.125f * 1.4426950408889634f;
  Frag qf[4], dof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    float v[8];
    E::unpack(E::load(qp + (size_t)qc * a.q_stride + 16 * ks + 8 * h), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= qscale;
    qf[ks] = E::pack(v);
    dof[ks] = E::load(dop + (size_t)qc * a.o_stride + 16 * ks + 8 * h);
  }
  const size_t stat = ((size_t)b * a.heads + head) * a.Nq + qc;
  const float lse = a.lse[stat], delta = a.delta[stat];
  f32x16 dq0 = (f32x16)(0.f), dq1 = (f32x16)(0.f);

  Frag stk[PIECES], stv[PIECES];
  // (round 6, as pd_attn_d64: tiles through buffer resources -- fixed per-lane offset, scalar tile offset, keys past Nkv read zeros)
  const unsigned kv_bytes = (unsigned)(((size_t)(a.Nkv - 1) * a.kv_stride + 64) * ES);
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
    unsigned char* kb = lds + buf * 2 * TB;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int pc = tid + 256 * i, row = pc >> 3, sub = pc & 7;
      E::store(kb + XB::store_off(row, sub), stk[i]);
      E::store(kb + TB + XB::store_off(row, sub), stv[i]);
    }
  };
  const int row_lane0 = XB::row_base(r, h);            // row fragment: row r, d = 16 ks + 8 h + (0..7)
  const int row_ks[4] = {XB::row_off(row_lane0, 0), XB::row_off(row_lane0, 1), XB::row_off(row_lane0, 2), XB::row_off(row_lane0, 3)};
  const typename XB::VtOff t_lane = XB::vt_off(lane);

  issue(0);
  commit(0);
  if (KT < a.Nkv) issue(KT);
  __syncthreads();
  for (int k0 = 0, cur = 0; k0 < a.Nkv; k0 += KT, cur ^= 1) {
    const unsigned char* kb = lds + cur * 2 * TB;
    const unsigned char* vb = kb + TB;
#pragma unroll
    for (int sub = 0; sub < KT / 32; ++sub) {
      if (k0 + sub * 32 < a.Nkv) {                     // workgroup-uniform
        f32x16 s = (f32x16)(0.f), dp = (f32x16)(0.f);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = E::mma(E::load(kb + sub * 32 * VP + row_ks[ks]), qf[ks], s);
          dp = E::mma(E::load(vb + sub * 32 * VP + row_ks[ks]), dof[ks], dp);
        }
        if (k0 + sub * 32 + 32 > a.Nkv) {               // keys beyond the context length (workgroup-uniform): P = 0
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (k0 + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * h >= a.Nkv) s[i] = -INFINITY;
        }
        {
          const f32x2 lse2 = (f32x2)(lse), delta2 = (f32x2)(delta);
#pragma unroll
          for (int i = 0; i < 16; i += 2) {             // packed fp32 math: one VALU slot per score pair
            const f32x2 e = (f32x2){s[i], s[i + 1]} - lse2;
            f32x2 p; p.x = __builtin_amdgcn_exp2f(e.x); p.y = __builtin_amdgcn_exp2f(e.y);
            const f32x2 ds = p * ((f32x2){dp[i], dp[i + 1]} - delta2);   // dS^T
            s[i] = ds.x; s[i + 1] = ds.y;
          }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const Frag df = X::pack_p(s, st);
          const unsigned char* ks_ = kb + (sub * 32 + 16 * st) * VP;
          dq0 = E::mma(XB::load_vt(ks_, t_lane, 0), df, dq0);
          dq1 = E::mma(XB::load_vt(ks_, t_lane, 1), df, dq1);
        }
      }
    }
    if (k0 + KT < a.Nkv) {
      commit(cur ^ 1);
      if (k0 + 2 * KT < a.Nkv) issue(k0 + 2 * KT);
    }
    __syncthreads();
  }
  if (query < a.Nq) {
    T* dst = (T*)a.dq + ((size_t)b * a.Nq + query) * a.dq_stride + head * 64 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      store4(dst + 8 * g, dq0[4 * g] * 0.125f, dq0[4 * g + 1] * 0.125f, dq0[4 * g + 2] * 0.125f, dq0[4 * g + 3] * 0.125f);
      store4(dst + 32 + 8 * g, dq1[4 * g] * 0.125f, dq1[4 * g + 1] * 0.125f, dq1[4 * g + 2] * 0.125f, dq1[4 * g + 3] * 0.125f);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_d64_dkv_kernel(const pd_attn_d64_bwd_args a, const int xcd_order) {
  using E = Elem<T>;
  using Frag = typename E::Frag;
  using X = D64<T>;
  using XB = D64B<T>;
  constexpr int QT = 64, VP = XB::P, ES = E::BYTES;
  constexpr int TB = QT * VP;
  constexpr int BUF = 2 * TB + 2 * QT * 4;             // Q tile | dO tile | lse[64] | delta[64]
  constexpr int PIECES = QT * 64 / 8 / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [2][BUF]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nkb = (a.Nkv + 127) / 128;
  const int item = xcd_order ? xcd_chunk_index((int)blockIdx.x, nkb * a.heads * a.B) : (int)blockIdx.x;      // (as the dQ kernel: a head's key blocks share Q / dO)
  const int kblk = item % nkb, head = (item / nkb) % a.heads, b = item / (nkb * a.heads);
  const T* qp = (const T*)a.q + (size_t)b * a.Nq * a.q_stride + head * 64;
  const T* kp = (const T*)a.k + (size_t)b * a.Nkv * a.kv_stride + head * 64;
  const T* vp = (const T*)a.v + (size_t)b * a.Nkv * a.kv_stride + head * 64;
  const T* dop = (const T*)a.dout + (size_t)b * a.Nq * a.o_stride + head * 64;
  const float* lsep = a.lse + ((size_t)b * a.heads + head) * a.Nq;
  const float* delp = a.delta + ((size_t)b * a.heads + head) * a.Nq;

  // K^T / V^T fragments (B operands): lane (key r, h), k-step ks: d = 16 ks + 8 h + j
  const int key = kblk * 128 + wave * 32 + r;
  const bool kvalid = key < a.Nkv;
  const float kscale = 0.125f * 1.4426950408889634f;
  Frag kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (kvalid) {
      float v[8];
      E::unpack(E::load(kp + (size_t)key * a.kv_stride + 16 * ks + 8 * h), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= kscale;
      kf[ks] = E::pack(v);
      vf[ks] = E::load(vp + (size_t)key * a.kv_stride + 16 * ks + 8 * h);
    } else { kf[ks] = E::zero(); vf[ks] = E::zero(); }
  }
  f32x16 dk0 = (f32x16)(0.f), dk1 = (f32x16)(0.f), dv0 = (f32x16)(0.f), dv1 = (f32x16)(0.f);

  Frag stq[PIECES], std_[PIECES];
  // (round 6: Q / dO tiles through buffer resources, queries past Nq read zeros)
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)qp, 0, (unsigned)(((size_t)(a.Nq - 1) * a.q_stride + 64) * ES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc((void*)dop, 0, (unsigned)(((size_t)(a.Nq - 1) * a.o_stride + 64) * ES), 0x00020000);
  unsigned qoff[PIECES], dooff[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int pc = tid + 256 * i;
    qoff[i] = (unsigned)(((pc >> 3) * a.q_stride + (pc & 7) * 8) * ES);
    dooff[i] = (unsigned)(((pc >> 3) * a.o_stride + (pc & 7) * 8) * ES);
  }
  float st_stat = 0.f;                                 // thread t < 64: lse of query t; 64 <= t < 128: delta of query t - 64
  auto issue = [&](int q0) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      stq[i] = E::load_buf(rq, qoff[i], (unsigned)q0 * (unsigned)a.q_stride * ES);
      std_[i] = E::load_buf(rdo, dooff[i], (unsigned)q0 * (unsigned)a.o_stride * ES);
    }
    if (tid < 128) {
      const int qi = q0 + (tid & 63);
      st_stat = tid < 64 ? (qi < a.Nq ? lsep[qi] : INFINITY) : (qi < a.Nq ? delp[qi] : 0.f);   // lse = +inf: P = 0 on the padding
    }
  };
  auto commit = [&](int buf) {
    unsigned char* qb_ = lds + buf * BUF;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int pc = tid + 256 * i, row = pc >> 3, sub = pc & 7;
      E::store(qb_ + XB::store_off(row, sub), stq[i]);
      E::store(qb_ + TB + XB::store_off(row, sub), std_[i]);
    }
    if (tid < 128) ((float*)(qb_ + 2 * TB))[tid] = st_stat;
  };
  const int row_lane0 = XB::row_base(r, h);
  const int row_ks[4] = {XB::row_off(row_lane0, 0), XB::row_off(row_lane0, 1), XB::row_off(row_lane0, 2), XB::row_off(row_lane0, 3)};
  const typename XB::VtOff t_lane = XB::vt_off(lane);

  issue(0);
  commit(0);
  if (QT < a.Nq) issue(QT);
  __syncthreads();
  for (int q0 = 0, cur = 0; q0 < a.Nq; q0 += QT, cur ^= 1) {
    const unsigned char* qb_ = lds + cur * BUF;
    const unsigned char* db = qb_ + TB;
    const float* lse_t = (const float*)(qb_ + 2 * TB);
    const float* del_t = lse_t + QT;
#pragma unroll
    for (int sub = 0; sub < QT / 32; ++sub) {
      if (q0 + sub * 32 < a.Nq) {                      // workgroup-uniform
        f32x16 s = (f32x16)(0.f), dp = (f32x16)(0.f);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = E::mma(E::load(qb_ + sub * 32 * VP + row_ks[ks]), kf[ks], s);     // S[query][key]
          dp = E::mma(E::load(db + sub * 32 * VP + row_ks[ks]), vf[ks], dp);    // dP[query][key]
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {                   // registers 4g..4g+3 <-> queries sub*32 + 8g + 4h + (0..3)
          const f32x4 l4 = *(const f32x4*)(lse_t + sub * 32 + 8 * g + 4 * h), d4 = *(const f32x4*)(del_t + sub * 32 + 8 * g + 4 * h);
#pragma unroll
          for (int j = 0; j < 4; j += 2) {              // packed fp32 math: one VALU slot per score pair
            const int i = 4 * g + j;
            const f32x2 e = (f32x2){s[i], s[i + 1]} - (f32x2){l4[j], l4[j + 1]};
            f32x2 p; p.x = __builtin_amdgcn_exp2f(e.x); p.y = __builtin_amdgcn_exp2f(e.y);
            const f32x2 ds = p * ((f32x2){dp[i], dp[i + 1]} - (f32x2){d4[j], d4[j + 1]});   // dS
            s[i] = p.x; s[i + 1] = p.y;
            dp[i] = ds.x; dp[i + 1] = ds.y;
          }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const Frag pf = X::pack_p(s, st), df = X::pack_p(dp, st);
          const unsigned char* dro = db + (sub * 32 + 16 * st) * VP;
          const unsigned char* qro = qb_ + (sub * 32 + 16 * st) * VP;
          dv0 = E::mma(XB::load_vt(dro, t_lane, 0), pf, dv0);
          dv1 = E::mma(XB::load_vt(dro, t_lane, 1), pf, dv1);
          dk0 = E::mma(XB::load_vt(qro, t_lane, 0), df, dk0);
          dk1 = E::mma(XB::load_vt(qro, t_lane, 1), df, dk1);
        }
      }
    }
    if (q0 + QT < a.Nq) {
      commit(cur ^ 1);
      if (q0 + 2 * QT < a.Nq) issue(q0 + 2 * QT);
    }
    __syncthreads();
  }
  if (kvalid) {
    T* dkd = (T*)a.dk + ((size_t)b * a.Nkv + key) * a.dkv_stride + head * 64 + 4 * h;
    T* dvd = (T*)a.dv + ((size_t)b * a.Nkv + key) * a.dkv_stride + head * 64 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      store4(dkd + 8 * g, dk0[4 * g] * 0.125f, dk0[4 * g + 1] * 0.125f, dk0[4 * g + 2] * 0.125f, dk0[4 * g + 3] * 0.125f);
      store4(dkd + 32 + 8 * g, dk1[4 * g] * 0.125f, dk1[4 * g + 1] * 0.125f, dk1[4 * g + 2] * 0.125f, dk1[4 * g + 3] * 0.125f);
      store4(dvd + 8 * g, dv0[4 * g], dv0[4 * g + 1], dv0[4 * g + 2], dv0[4 * g + 3]);
      store4(dvd + 32 + 8 * g, dv1[4 * g], dv1[4 * g + 1], dv1[4 * g + 2], dv1[4 * g + 3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm backward.  One wave per token (statistics recomputed from x, as cheap as loading them):
//   xhat = (x - mean) rstd,  g = dy gamma,  dx = rstd (g - mean(g) - xhat mean(g xhat)) [+ res]
// The grid is fixed-size; every wave walks a strided set of tokens and keeps its share of dgamma = sum dy xhat and
// dbeta = sum dy in registers; per-workgroup partials go to a workspace and are summed in a fixed order by a second kernel.
[[maybe_unused]] constexpr int LN_MAXP = 4;                             // C <= 64 lanes * 8 * 4 = 2048

// NP = 8-element pieces per lane (C <= 512 * NP): narrow rows keep few registers, so more waves hide the three dependent
// reductions per token
// DXS (round 6, pd_layernorm_bwd_args.dxsum): a third column sum -- of dx as stored, the bias gradient of the Linear layer that wrote the residual
// stream this dx is the gradient of (attn1 / attn2 to_out, proj_in): pd_channel_sum no longer reads dx back for it
template <typename T, int NP, bool DXS>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const pd_layernorm_bwd_args a) {
  using E = Elem<T>;
  constexpr int NS = DXS ? 3 : 2;
  __shared__ float red[3][NS][NP][8][64];         // waves 1..3: [dgamma | dbeta (| dx sum)] shares
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pieces = a.C / 8;
  float gam[NP][8], dg[NP][8], db[NP][8], ds[DXS ? NP : 1][8];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pc = lane + 64 * i;
#pragma unroll
    for (int j = 0; j < 8; ++j) { gam[i][j] = pc < pieces ? a.gamma[pc * 8 + j] : 0.f; dg[i][j] = 0.f; db[i][j] = 0.f; if (DXS) ds[i][j] = 0.f; }
  }
  const float invC = 1.0f / (float)a.C;
  // Round 6: the NEXT row's x / dy / res pieces are requested before this row's three dependent reductions (a wave walks its rows one after the
  // other: with the loads issued at the top of the row, two or three 16-byte loads per lane were all a wave had in flight)
  typename E::Frag fx[NP], fdy[NP], fres[NP];
  const long long stride = (long long)gridDim.x * 4;
  auto fetch = [&](long long row) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int pc = lane + 64 * i;
      if (pc < pieces) {
        fx[i] = E::load((const T*)a.x + row * a.C + pc * 8);
        fdy[i] = E::load((const T*)a.dy + row * a.C + pc * 8);
        if (a.res) fres[i] = E::load((const T*)a.res + row * a.C + pc * 8);
      }
    }
  };
  long long row = (long long)blockIdx.x * 4 + wave;
  if (row < a.rows) fetch(row);
  for (; row < a.rows; row += stride) {
    float v[NP][8], g[NP][8], rv[NP][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int pc = lane + 64 * i;
      if (pc < pieces) {
        E::unpack(fx[i], v[i]);
        E::unpack(fdy[i], g[i]);
        if (a.res) E::unpack(fres[i], rv[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[i][j];
      }
    }
    if (row + stride < a.rows) fetch(row + stride);
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) s += __shfl_xor(s, msk);
    const float mean = s * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (lane + 64 * i < pieces) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
      }
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) q += __shfl_xor(q, msk);
    const float rstd = 1.0f / sqrtf(q * invC + a.eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (lane + 64 * i < pieces) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (v[i][j] - mean) * rstd, dyv = g[i][j];
          dg[i][j] += dyv * xh;
          db[i][j] += dyv;
          const float gg = dyv * gam[i][j];
          v[i][j] = xh; g[i][j] = gg;
          sg += gg; sgx += gg * xh;
        }
      }
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) { sg += __shfl_xor(sg, msk); sgx += __shfl_xor(sgx, msk); }
    const float mg = sg * invC, mgx = sgx * invC;
    T* dx = (T*)a.dx + row * a.C;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int pc = lane + 64 * i;
      if (pc < pieces) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (g[i][j] - mg - v[i][j] * mgx) + (a.res ? rv[i][j] : 0.f);
        const typename E::Frag fo = E::pack(o);
        E::store(dx + pc * 8, fo);
        if constexpr (DXS) {
          E::unpack(fo, o);                      // the sum is over the stored values (what the weight gradient of that layer reads)
#pragma unroll
          for (int j = 0; j < 8; ++j) ds[i][j] += o[j];
        }
      }
    }
  }
  if (!a.partial) return;
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[wave - 1][0][i][j][lane] = dg[i][j]; red[wave - 1][1][i][j][lane] = db[i][j];
        if constexpr (DXS) red[wave - 1][2][i][j][lane] = ds[i][j];
      }
  }
  __syncthreads();
  if (wave == 0) {
    float* pg = a.partial + (size_t)blockIdx.x * NS * a.C;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int pc = lane + 64 * i;
      if (pc < pieces) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pg[pc * 8 + j] = ((dg[i][j] + red[0][0][i][j][lane]) + red[1][0][i][j][lane]) + red[2][0][i][j][lane];
          pg[a.C + pc * 8 + j] = ((db[i][j] + red[0][1][i][j][lane]) + red[1][1][i][j][lane]) + red[2][1][i][j][lane];
          if constexpr (DXS) pg[2 * a.C + pc * 8 + j] = ((ds[i][j] + red[0][2][i][j][lane]) + red[1][2][i][j][lane]) + red[2][2][i][j][lane];
        }
      }
    }
  }
}

// dgamma[c] += sum over workgroups of partial[wg][0][c], dbeta likewise (fixed order: bitwise reproducible).
// Workgroup = 16 columns x 16 row groups: row group r sums workgroups r, r+16, ...; the 16 shares are added in order.
// (ns = 3: a third column block, the sums of dx -> dxsum)
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* partial, int nblocks, int C, float* dgamma, float* dbeta, int ns, float* dxsum) {
  __shared__ float red[16][17];
  const int col = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + col;
  float s = 0.f;
  if (idx < ns * C)
    for (int b = rg; b < nblocks; b += 16) s += partial[(size_t)b * ns * C + idx];
  red[rg][col] = s;
  __syncthreads();
  if (rg == 0 && idx < ns * C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][col];
    float* dst = idx < C ? dgamma + idx : (idx < 2 * C ? dbeta + (idx - C) : dxsum + (idx - 2 * C));
    *dst += t;
  }
}

// GEGLU backward: y = h * gelu(g)  ->  dh = dy * gelu(g),  dg = dy * h * (Phi(g) + g * phi(g));  dx = [dh | dg]
template <typename T>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const pd_geglu_bwd_args a) {
  using E = Elem<T>;
  const int pieces = a.inner / 8;
  const size_t total = (size_t)a.rows * pieces;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t row = idx / pieces;
    const int pc = (int)(idx - row * pieces);
    float hv[8], gv[8], dyv[8], dh[8], dgv[8];
    E::unpack(E::load((const T*)a.x + row * 2 * a.inner + pc * 8), hv);
    E::unpack(E::load((const T*)a.x + row * 2 * a.inner + a.inner + pc * 8), gv);
    E::unpack(E::load((const T*)a.dy + row * a.inner + pc * 8), dyv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float cdf = 0.5f * (1.0f + erff(gv[j] * 0.7071067811865476f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * gv[j] * gv[j]);
      dh[j] = dyv[j] * gv[j] * cdf;
      dgv[j] = dyv[j] * hv[j] * (cdf + gv[j] * pdf);
    }
    E::store((T*)a.dx + row * 2 * a.inner + pc * 8, E::pack(dh));
    E::store((T*)a.dx + row * 2 * a.inner + a.inner + pc * 8, E::pack(dgv));
  }
}

// The same with the column sums of dx as stored (round 6; pd_geglu_bwd_args.sums): the bias gradient of ff.net.0.proj is the column sum of dx = [dh | dg],
// the widest gradient tensor of a transformer block (8 x the hidden width: 671 MB at the 64 x 64 level) -- read back by pd_channel_sum it was 60 % of all the
// bytes the bias gradients of a fine-tuning step fetched.  Workgroup = 32 piece columns (256 channels of h and of g: 512-byte row segments) x 8 row
// phases over one sample's row range; a thread owns its 16 channels, the 8 phases are folded in a fixed order (fp64, as channel_sum_kernel).
template <typename T>
__global__ __launch_bounds__(256) void geglu_bwd_sums_kernel(const pd_geglu_bwd_args a) {
  using E = Elem<T>;
  __shared__ float red[8][32][16];
  const int chunks = a.inner / 256;
  const int chunk = blockIdx.x % chunks, sp = blockIdx.x / chunks;          // chunk fastest: neighbouring workgroups read neighbouring columns of the same rows
  const int n = sp / a.sum_splits, sq = sp - n * a.sum_splits;
  const long long N = a.rows / a.B, per = (N + a.sum_splits - 1) / a.sum_splits;
  const long long r_lo = n * N + sq * per, r_hi = min(n * N + N, r_lo + per);
  const int pcol = threadIdx.x & 31, phase = threadIdx.x >> 5;
  const int pc = chunk * 32 + pcol;
  float sh[8], sg[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sh[j] = 0.f; sg[j] = 0.f; }
  for (long long row = r_lo + phase; row < r_hi; row += 8) {
    float hv[8], gv[8], dyv[8], dh[8], dgv[8];
    E::unpack(E::load((const T*)a.x + row * 2 * a.inner + pc * 8), hv);
    E::unpack(E::load((const T*)a.x + row * 2 * a.inner + a.inner + pc * 8), gv);
    E::unpack(E::load((const T*)a.dy + row * a.inner + pc * 8), dyv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float cdf = 0.5f * (1.0f + erff(gv[j] * 0.7071067811865476f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * gv[j] * gv[j]);
      dh[j] = dyv[j] * gv[j] * cdf;
      dgv[j] = dyv[j] * hv[j] * (cdf + gv[j] * pdf);
    }
    const typename E::Frag fh = E::pack(dh), fg = E::pack(dgv);
    E::store((T*)a.dx + row * 2 * a.inner + pc * 8, fh);
    E::store((T*)a.dx + row * 2 * a.inner + a.inner + pc * 8, fg);
    E::unpack(fh, dh); E::unpack(fg, dgv);                                 // the sums are over what the weight gradient reads: the stored values
#pragma unroll
    for (int j = 0; j < 8; ++j) { sh[j] += dh[j]; sg[j] += dgv[j]; }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[phase][pcol][j] = sh[j]; red[phase][pcol][8 + j] = sg[j]; }
  __syncthreads();
  for (int o = threadIdx.x; o < 512; o += 256) {
    const int pq = o >> 4, j = o & 15;
    double d = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) d += (double)red[k][pq][j];
    const int c = (j < 8 ? 0 : a.inner) + (chunk * 32 + pq) * 8 + (j & 7);
    a.sums[(size_t)sp * 2 * a.inner + c] = (float)d;
  }
}

// CustomEmbedding gradient from the encoder_hidden_states gradient: only token 0 of the 77 carries the class embedding
// (utils_training.py:479-484), so dtable[labels[n]][c] += d[n][token 0][c].  One thread per column, samples in order.
template <typename T>
__global__ __launch_bounds__(256) void token_embedding_grad_kernel(const pd_token_embedding_grad_args a) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= a.dim) return;
  for (int n = 0; n < a.rows; ++n) {
    const long long k = a.labels[n];
    if (k < 0 || k >= a.num_classes) continue;
    a.dtable[(size_t)k * a.dim + c] += Elem<T>::to_f(((const T*)a.d)[(size_t)n * a.row_stride + c]);
  }
}

template <typename T>
static int launch_attn_d64_bwd(const pd_attn_d64_bwd_args* a, hipStream_t st) {
  constexpr int TB = 64 * D64B<T>::P;
  constexpr int LDS_DQ = 2 * 2 * TB, LDS_DKV = 2 * (2 * TB + 2 * 64 * 4);
  auto kq = attn_d64_dq_kernel<T>;
  auto kkv = attn_d64_dkv_kernel<T>;
  static LdsAttr attr_q, attr_kv;
  if (!ensure_lds(attr_q, kq, LDS_DQ) || !ensure_lds(attr_kv, kkv, LDS_DKV)) {
    set_error("pd_attn_d64_bwd: cannot reserve %d / %d bytes of LDS", LDS_DQ, LDS_DKV);
    return PD_ERR_LAUNCH;
  }
  const int xcd_order = diag_env("PD_ATTN64_BWD_XCD", 1) != 0;
  const size_t nd = (size_t)a->B * a->Nq * a->heads * 8;
  hipLaunchKernelGGL(attn_d64_delta_kernel<T>, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, *a);
  PD_LAUNCH_CHECK();
  hipLaunchKernelGGL(kq, dim3(((a->Nq + 127) / 128) * a->heads * a->B), dim3(256), LDS_DQ, st, *a, xcd_order);
  PD_LAUNCH_CHECK();
  hipLaunchKernelGGL(kkv, dim3(((a->Nkv + 127) / 128) * a->heads * a->B), dim3(256), LDS_DKV, st, *a, xcd_order);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

}  // namespace pd

using namespace pd;

extern "C" int pd_attn_d64_bwd(const pd_attn_d64_bwd_args* a, void* stream) {
  PD_CHECK(a != nullptr, PD_ERR_ARG, "pd_attn_d64_bwd: null args");
  PD_CHECK(a->B > 0 && a->heads > 0 && a->Nq > 0 && a->Nkv > 0, PD_ERR_SHAPE, "pd_attn_d64_bwd: bad shape");
  PD_CHECK(a->q && a->k && a->v && a->o && a->dout && a->lse && a->delta && a->dq && a->dk && a->dv, PD_ERR_ARG, "pd_attn_d64_bwd: null pointer");
  const int c = a->heads * 64;
  PD_CHECK(a->q_stride >= c && a->kv_stride >= c && a->o_stride >= c && a->dq_stride >= c && a->dkv_stride >= c && a->q_stride % 8 == 0 &&
               a->kv_stride % 8 == 0 && a->o_stride % 8 == 0 && a->dq_stride % 8 == 0 && a->dkv_stride % 8 == 0, PD_ERR_SHAPE,
           "pd_attn_d64_bwd: strides must cover heads*64 channels and be multiples of 8");
  PD_CHECK((long long)((a->Nq + 127) / 128) * a->heads * a->B < (1ll << 31) && (long long)a->B * a->Nq * a->heads * 8 / 256 < (1ll << 31),
           PD_ERR_SHAPE, "pd_attn_d64_bwd: grid too large");
  PD_CHECK((unsigned long long)a->Nkv * (unsigned long long)a->kv_stride * 4ull < (1ull << 32) && (unsigned long long)a->Nq * (unsigned long long)(a->q_stride > a->o_stride ? a->q_stride : a->o_stride) * 4ull < (1ull << 32),
           PD_ERR_SHAPE, "pd_attn_d64_bwd: one sample's rows must span < 4 GiB (32-bit buffer offsets)");
  if (a->dtype == PD_F32) return launch_attn_d64_bwd<float>(a, (hipStream_t)stream);
  if (a->dtype == PD_BF16) return launch_attn_d64_bwd<bf16_t>(a, (hipStream_t)stream);
  if (a->dtype == PD_F16) return launch_attn_d64_bwd<half_t>(a, (hipStream_t)stream);      // fp16 training (round 5): under the trainer's loss scale
  set_error("pd_attn_d64_bwd: bad dtype");
  return PD_ERR_ARG;
}

// workspace bound: `partial` holds this many rows of 2 (3 with dxsum) * C floats
extern "C" int pd_layernorm_bwd_blocks(long long rows) {
  const long long nb = (rows + 3) / 4;
  return (int)(nb < 2048 ? nb : 2048);
}
// Workgroups actually launched (<= the bound above): every workgroup leaves one row of the partial, and the fold of those rows was as long as
// the kernel itself at 2 048 of them (C = 1 280: 94 us for kernel + fold, 39 us with 512; C = 640: 90 -> 61; C = 320: 132 -> 108 with 1 024,
// 149 with 512 -- scripts/experiments/bench_ln_bwd.py).  PD_LN_BWD_BLOCKS: diagnostic override.
static int ln_bwd_grid(long long rows, int C) {
  const long long nb = (rows + 3) / 4;
  const int cap = diag_env("PD_LN_BWD_BLOCKS", C <= 384 ? 1024 : 512);
  const int bound = pd_layernorm_bwd_blocks(rows);
  const long long g = nb < cap ? nb : cap;
  return (int)(g < bound ? g : bound);
}

extern "C" int pd_layernorm_bwd(const pd_layernorm_bwd_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->rows > 0 && a->C > 0 && a->C % 8 == 0 && a->C <= 2048 && a->x && a->dy && a->dx && a->gamma, PD_ERR_ARG,
           "pd_layernorm_bwd: bad args (C must be a multiple of 8, <= 2048)");
  PD_CHECK((a->dgamma == nullptr) == (a->dbeta == nullptr) && (a->dgamma == nullptr) == (a->partial == nullptr), PD_ERR_ARG,
           "pd_layernorm_bwd: dgamma, dbeta and partial go together");
  PD_CHECK(a->dxsum == nullptr || a->partial != nullptr, PD_ERR_ARG, "pd_layernorm_bwd: dxsum needs dgamma, dbeta and a partial workspace of 3 C floats per block");
  const int grid = ln_bwd_grid(a->rows, a->C);
  hipStream_t st = (hipStream_t)stream;
  const int np = (a->C / 8 + 63) / 64;
  const bool dxs = a->dxsum != nullptr;
  PD_CHECK(!dxs || np <= 3, PD_ERR_UNSUPPORTED, "pd_layernorm_bwd: dxsum needs C <= 1536 (C=%d)", a->C);
  if (a->dtype != PD_F32 && a->dtype != PD_BF16 && a->dtype != PD_F16) { set_error("pd_layernorm_bwd: bad dtype"); return PD_ERR_ARG; }
#define PD_LN_BWD(NP_)                                                                                             \
  do {                                                                                                             \
    if (dxs) {                                                                                                     \
      if (a->dtype == PD_F32) hipLaunchKernelGGL((layernorm_bwd_kernel<float, NP_, (NP_ <= 3)>), dim3(grid), dim3(256), 0, st, *a); \
      else if (a->dtype == PD_F16) hipLaunchKernelGGL((layernorm_bwd_kernel<half_t, NP_, (NP_ <= 3)>), dim3(grid), dim3(256), 0, st, *a); \
      else hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, NP_, (NP_ <= 3)>), dim3(grid), dim3(256), 0, st, *a);           \
    } else if (a->dtype == PD_F32) hipLaunchKernelGGL((layernorm_bwd_kernel<float, NP_, false>), dim3(grid), dim3(256), 0, st, *a); \
    else if (a->dtype == PD_F16) hipLaunchKernelGGL((layernorm_bwd_kernel<half_t, NP_, false>), dim3(grid), dim3(256), 0, st, *a); \
    else hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, NP_, false>), dim3(grid), dim3(256), 0, st, *a);           \
  } while (0)
  if (np == 1) PD_LN_BWD(1); else if (np == 2) PD_LN_BWD(2); else if (np == 3) PD_LN_BWD(3); else PD_LN_BWD(4);
#undef PD_LN_BWD
  PD_LAUNCH_CHECK();
  if (a->partial) {
    const int ns = dxs ? 3 : 2;
    hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((ns * a->C + 15) / 16), dim3(256), 0, st, (const float*)a->partial, grid, a->C,
                       a->dgamma, a->dbeta, ns, a->dxsum);
    PD_LAUNCH_CHECK();
  }
  return PD_OK;
}

extern "C" int pd_token_embedding_grad(const pd_token_embedding_grad_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->rows > 0 && a->dim > 0 && a->num_classes > 0 && a->d && a->dtable && a->row_stride >= a->dim, PD_ERR_ARG,
           "pd_token_embedding_grad: bad args");
  if (!a->labels) return PD_OK;                        // unconditional step (all-zero context): no class-embedding gradient
  const unsigned grid = (unsigned)((a->dim + 255) / 256);
  if (a->dtype == PD_F32) hipLaunchKernelGGL(token_embedding_grad_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(token_embedding_grad_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(token_embedding_grad_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else { set_error("pd_token_embedding_grad: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_geglu_bwd(const pd_geglu_bwd_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->rows > 0 && a->inner > 0 && a->inner % 8 == 0 && a->x && a->dy && a->dx, PD_ERR_ARG, "pd_geglu_bwd: bad args");
  if (a->sums) {
    PD_CHECK(a->B > 0 && a->sum_splits > 0 && a->rows % a->B == 0 && a->inner % 256 == 0, PD_ERR_ARG,
             "pd_geglu_bwd: column sums need inner %% 256 == 0 and rows %% B == 0 (rows=%lld B=%d inner=%d splits=%d)", a->rows, a->B, a->inner, a->sum_splits);
    const long long blocks = (long long)a->B * a->sum_splits * (a->inner / 256);
    PD_CHECK(blocks < (1ll << 31), PD_ERR_SHAPE, "pd_geglu_bwd: %lld workgroups", blocks);
    if (a->dtype == PD_F32) hipLaunchKernelGGL(geglu_bwd_sums_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    else if (a->dtype == PD_BF16) hipLaunchKernelGGL(geglu_bwd_sums_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    else if (a->dtype == PD_F16) hipLaunchKernelGGL(geglu_bwd_sums_kernel<half_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    else { set_error("pd_geglu_bwd: bad dtype"); return PD_ERR_ARG; }
    PD_LAUNCH_CHECK();
    return PD_OK;
  }
  const size_t total = (size_t)a->rows * (a->inner / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (a->dtype == PD_F32) hipLaunchKernelGGL(geglu_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_BF16) hipLaunchKernelGGL(geglu_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->dtype == PD_F16) hipLaunchKernelGGL(geglu_bwd_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else { set_error("pd_geglu_bwd: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}
