// pd_attn_d8: softmax(q k^T / sqrt(8)) v, head_dim = 8, streaming softmax (scores never leave registers).
//
// Per wave: 32 queries of one (batch, head).  Keys are visited in tiles of 32:
//   S^T[key][query] = K[32 x 8] . Q^T[8 x 32]        one 32x32 MFMA (bf16: K=16 slot half used; fp32: 4 x 32x32x2)
//     -> D layout puts the QUERY on the lane and 16 keys in the lane's registers, so the row max / exp of the
//        softmax is lane-local (one cross-half exchange per tile), and the exponentiated tile is ALREADY the
//        B operand of the next product (rows of D == k index of B), no LDS round trip:
//   O^T[row][query] += A[row][key] . P^T[key][query]  A rows 0..7 = V^T (d), rows 8..15 = 1.0  (=> row sum l for free)
// Workgroup = 4 waves = 128 queries; K and V^T tiles of 256 keys are staged through double-buffered LDS (V transposed on
// the way, next tile's global loads in flight during the current tile's math, one barrier per tile).
#include <stdlib.h>
#include <type_traits>
#include "pd_common.h"

namespace pd {

constexpr int KT = 256;   // keys per LDS tile (double-buffered: one barrier per tile)
static_assert(KT == 256, "knmax is read as one f32x4 (KT / 64 slices)");

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// max over each row of 16 lanes with DPP-modified VALU ops (no LDS crossbar traffic): xor 1, xor 2, mirror within 8, mirror within 16
__device__ __forceinline__ float row16_max(float x) {
#define PD_DPP_MAX(ctrl) x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), ctrl, 0xf, 0xf, false)))
  PD_DPP_MAX(0xB1);    // quad_perm:[1,0,3,2]
  PD_DPP_MAX(0x4E);    // quad_perm:[2,3,0,1]
  PD_DPP_MAX(0x141);   // row_half_mirror
  PD_DPP_MAX(0x140);   // row_mirror
#undef PD_DPP_MAX
  return x;
}

__device__ __forceinline__ unsigned short bits16(bf16_t v) { return v; }
__device__ __forceinline__ unsigned short bits16(half_t v) { return v.bits; }

// primary template: the 16-bit element types (bf16, fp16); exact fp32 is the specialisation below
template <typename T> struct AttnOps {
  // deferred-rescale threshold (log2 domain): p = 2^(s - m) <= 2^THR must stay finite in the storage type (fp16: < 65504)
  static constexpr float RESCALE_THR = sizeof(T) == 2 && !std::is_same<T, bf16_t>::value ? 14.0f : 16.0f;
  static constexpr int VT_PITCH = (KT + 8) * 2;   // bytes per V^T row (pad -> rows on distinct banks)
  struct QF { s16x8 v; };
  static __device__ __forceinline__ QF load_q(const T* q, int h, float scale) {
    QF f; f.v = (s16x8)(0);
    if (h == 0) {
      s16x8 raw = *(const s16x8*)q;
      {
        float x[8];
        typename Elem<T>::Frag fr; fr.v = raw;
        Elem<T>::unpack(fr, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] *= scale;
        f.v = Elem<T>::pack(x).v;
      }
    }
    return f;
  }
  // The reference maximum m rides in the k-slots 8..10 of the QK^T MFMA that head_dim 8 leaves unused (B rows = -m split in
  // three T-precision terms, A columns = the constants M_C0, 1, 1 of the shared LDS slot): s' = q.k - m_eff leaves the matrix
  // pipe ready for exp2 with C = 0, i.e. without a 16-register block holding -m.  Returns m_eff, the value actually
  // subtracted (|m_eff - m| <= 2^-20 |m|; any reference works as long as every consumer uses the same one).
  static constexpr float M_C0 = std::is_same<T, bf16_t>::value ? 1.0f : 4096.0f;   // fp16: |m| up to ~1e8
  static constexpr bool M_IN_C = false;
  static __device__ __forceinline__ float set_m(QF& q, f32x16&, int h, float m) {
    const T hi = Elem<T>::from_f(-m * (1.0f / M_C0));
    const float r1 = fmaf(-M_C0, Elem<T>::to_f(hi), -m);
    const T mid = Elem<T>::from_f(r1);
    const float r2 = r1 - Elem<T>::to_f(mid);
    const T lo = Elem<T>::from_f(r2);
    if (h) { q.v[0] = (short)bits16(hi); q.v[1] = (short)bits16(mid); q.v[2] = (short)bits16(lo); }
    return -(fmaf(M_C0, Elem<T>::to_f(hi), Elem<T>::to_f(mid)) + Elem<T>::to_f(lo));
  }
  // x as three T-precision terms: x ~= M_C0 * hi + mid + lo (24 / 33 mantissa bits for bf16 / fp16)
  static __device__ __forceinline__ void split3(float x, T& hi, T& mid, T& lo) {
    hi = Elem<T>::from_f(x * (1.0f / M_C0));
    const float r1 = fmaf(-M_C0, Elem<T>::to_f(hi), x);
    mid = Elem<T>::from_f(r1);
    lo = Elem<T>::from_f(r1 - Elem<T>::to_f(mid));
  }
  // backward kernels: a per-row constant c rides in the same three k-slots (the other operand holds M_C0, 1, 1): product + c
  static __device__ __forceinline__ void set_terms(QF& q, int h, float c) {
    T hi, mid, lo;
    split3(c, hi, mid, lo);
    if (h) { q.v[0] = (short)bits16(hi); q.v[1] = (short)bits16(mid); q.v[2] = (short)bits16(lo); }
  }
  static __device__ __forceinline__ void set_ones(QF& q, int h) {
    if (h) { q.v[0] = (short)bits16(Elem<T>::from_f(M_C0)); q.v[1] = (short)bits16(Elem<T>::from_f(1.0f)); q.v[2] = q.v[1]; }
  }
  static __device__ __forceinline__ void store_terms(unsigned char* row16, float c) {   // a 16-byte LDS row: [hi mid lo 0 0 0 0 0]
    T t[8];
    split3(c, t[0], t[1], t[2]);
#pragma unroll
    for (int i = 3; i < 8; ++i) t[i] = Elem<T>::from_f(0.0f);
    *(s16x8*)row16 = *(const s16x8*)t;
  }
  static __device__ __forceinline__ void init_const_slot(unsigned char* slot) {   // k-slots 8..15 of every K row
    T c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = Elem<T>::from_f(i == 0 ? M_C0 : i < 3 ? 1.0f : 0.0f);
    *(s16x8*)slot = *(const s16x8*)c;
  }
  static __device__ __forceinline__ float q_norm2(const QF& q) {
    float x[8], t = 0.f;
    typename Elem<T>::Frag fr; fr.v = q.v;
    Elem<T>::unpack(fr, x);
#pragma unroll
    for (int j = 0; j < 8; ++j) t += x[j] * x[j];
    return t;
  }
  static __device__ __forceinline__ float dot(const QF& x, const QF& y) {   // this lane's share of sum_d x_d y_d
    float a[8], b[8], t = 0.f;
    typename Elem<T>::Frag fa, fb; fa.v = x.v; fb.v = y.v;
    Elem<T>::unpack(fa, a); Elem<T>::unpack(fb, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) t += a[j] * b[j];
    return t;
  }
  // |k|^2 of a staged key row: four v_dot2c_f32_bf16 on the packed pairs (no unpacking)
  static __device__ __forceinline__ float k_norm2(const typename Elem<T>::Frag& k) {
    const u32x4 w = __builtin_bit_cast(u32x4, k.v);
    float n2 = 0.f;
    if constexpr (std::is_same<T, bf16_t>::value) {
      typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // copy the element to a scalar first: hipcc 7.2 folds `__builtin_bit_cast(X, vec[j])` of an ext-vector ELEMENT to element 0
        // for every j (round-3 finding: this norm was 4 x the first pair's -- the same defect that made round 2 blame v_dot2_f32_f16)
        const uint32_t wj = w[j];
        const bf2 pr = __builtin_bit_cast(bf2, wj);
        n2 = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, n2, false);
      }
    } else {
      // fp16: plain fp32 arithmetic on the unpacked values (v_dot2_f32_f16 through __builtin_amdgcn_fdot2 returned norms that
      // under-estimated |k|^2 on gfx950, which made the score bound below invalid: p = 2^(s - m) overflowed fp16)
      float x[8];
      Elem<T>::unpack(k, x);
#pragma unroll
      for (int j = 0; j < 8; ++j) n2 += x[j] * x[j];
    }
    return n2;
  }
  // V row -> column vp of the 8 V^T rows: the bf16 halves are stored as they are (ds_write_b16 / _d16_hi), no conversion
  static __device__ __forceinline__ void store_vt(unsigned char* vl, int vp, const typename Elem<T>::Frag& v) {
#pragma unroll
    for (int d = 0; d < 8; ++d) *(unsigned short*)(vl + d * VT_PITCH + vp * 2) = (unsigned short)v.v[d];
  }
  // S^T = K . Q^T ; a = K tile fragment (lane: key r; h==0 holds d 0..7, h==1 zeros)
  static constexpr int KROW = 16;                 // bytes per K row in LDS
  static constexpr int KSTEP = 32 * KROW;         // bytes between consecutive 32-key sub-tiles
  // lane's K-fragment address for sub-tile 0: h == 0 -> row r; h == 1 -> the shared constant slot after the tile
  // (k-slots 8..15 of the 32x32x16 MFMA: see set_m); kstep = 0 keeps h == 1 lanes on that slot
  static __device__ __forceinline__ int kaddr(int r, int h) { return h ? KT * KROW : r * KROW; }
  static __device__ __forceinline__ int kstep(int h) { return h ? 0 : KSTEP; }
  struct KF { s16x8 v; };
  struct VF { u32x4 v[2]; };
  static __device__ __forceinline__ KF load_k(const unsigned char* ka) { KF f; f.v = *(const s16x8*)ka; return f; }
  static __device__ __forceinline__ f32x16 qk(const KF& a, const QF& q, const f32x16& c) {
    return Elem<T>::mma16(a.v, q.v, c);      // forward: c stays the zero it was initialised to (set_m never writes it)
  }
  static __device__ __forceinline__ VF load_v(const unsigned char* vrow, int key0) {
    VF f;
    f.v[0] = *(const u32x4*)(vrow + key0 * 2);
    f.v[1] = *(const u32x4*)(vrow + (key0 + 16) * 2);
    return f;
  }
  // V^T row position of key t: within each 16-key block the middle two groups of 4 are swapped, so the 8 keys one lane
  // feeds to a PV k-step (16s + 4h + {0..3}, 16s + 8 + 4h + {0..3}) are 16 contiguous bytes -> one ds_read_b128
  static __device__ __forceinline__ int vpos(int t) { return t ^ ((((t >> 2) ^ (t >> 3)) & 1) * 12); }
  static __device__ __forceinline__ int vlane_off(int h) { return h * 16; }
  // Two probabilities -> one dword of the PV MFMA's B operand.  bf16 (round 3): the upper halves of the two fp32 words by ONE
  // v_perm_b32 (full-rate, ~3 cycles at this occupancy) instead of v_cvt_pk_bf16_f32 (4.8 cycles measured,
  // scripts/micro/exp_variants.hip): truncation instead of round-to-nearest.  The row sum l is accumulated by the same MFMA from
  // the same truncated values (the ones row of A), so the -2^-9 mean bias divides out of O = (P V) / l; the random part has the
  // variance of round-to-nearest (one ulp wide either way).  8 of these per 32 x 32 tile: ~14 of its ~217 cycles.
  static __device__ __forceinline__ uint32_t pack_p(float lo, float hi) {
    if constexpr (std::is_same<T, bf16_t>::value) return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
    else return Pack16<T>::pack(lo, hi);
  }
  // O^T += A . P^T for one 32-key sub-tile; p = exponentiated tile (fp32 accumulator layout)
  static __device__ __forceinline__ f32x16 pv(const VF& a, const f32x16& p, f32x16 o) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // B fragment: element j <-> key 16s + 8(j>>2) + 4h + (j&3) == accumulator register 8s + j
      uint32_t bw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bw[j] = pack_p(p[8 * s + 2 * j], p[8 * s + 2 * j + 1]);
      u32x4 bv = {bw[0], bw[1], bw[2], bw[3]};
      o = Elem<T>::mma16(__builtin_bit_cast(s16x8, a.v[s]), __builtin_bit_cast(s16x8, bv), o);
    }
    return o;
  }
};

template <> struct AttnOps<float> {
  static constexpr float RESCALE_THR = 16.0f;
  static constexpr int VT_PITCH = (KT + 4) * 4;
  struct QF { f32x4 v; };
  static __device__ __forceinline__ QF load_q(const float* q, int h, float scale) {
    QF f; f.v = *(const f32x4*)(q + 4 * h);
    f.v *= scale;
    return f;
  }
  static constexpr bool M_IN_C = true;   // exact fp32: all 8 k-slots are d; -m rides in the C operand
  static __device__ __forceinline__ float set_m(QF&, f32x16& negm, int, float m) { negm = (f32x16)(-m); return m; }
  static __device__ __forceinline__ void init_const_slot(unsigned char*) {}
  static __device__ __forceinline__ float q_norm2(const QF& q) {
    return q.v[0] * q.v[0] + q.v[1] * q.v[1] + q.v[2] * q.v[2] + q.v[3] * q.v[3];
  }
  static __device__ __forceinline__ float dot(const QF& x, const QF& y) {
    return x.v[0] * y.v[0] + x.v[1] * y.v[1] + x.v[2] * y.v[2] + x.v[3] * y.v[3];
  }
  static __device__ __forceinline__ float k_norm2(const Elem<float>::Frag& k) {
    float n2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) n2 += k.lo[j] * k.lo[j] + k.hi[j] * k.hi[j];
    return n2;
  }
  static __device__ __forceinline__ void store_vt(unsigned char* vl, int vp, const Elem<float>::Frag& v) {
#pragma unroll
    for (int d = 0; d < 4; ++d) { *(float*)(vl + d * VT_PITCH + vp * 4) = v.lo[d]; *(float*)(vl + (4 + d) * VT_PITCH + vp * 4) = v.hi[d]; }
  }
  static constexpr int KROW = 32;
  static constexpr int KSTEP = 32 * KROW;
  static __device__ __forceinline__ int kaddr(int r, int h) { return r * KROW + h * 16; }   // d = 4h + i
  static __device__ __forceinline__ int kstep(int) { return KSTEP; }
  static __device__ __forceinline__ int vpos(int t) { return t; }
  static __device__ __forceinline__ int vlane_off(int h) { return h * 16; }
  struct KF { f32x4 v; };
  struct VF { f32x4 v[4]; };
  static __device__ __forceinline__ KF load_k(const unsigned char* ka) { KF f; f.v = *(const f32x4*)ka; return f; }
  static __device__ __forceinline__ f32x16 qk(const KF& a, const QF& q, f32x16 c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[i], q.v[i], c, 0, 0, 0);
    return c;
  }
  static __device__ __forceinline__ VF load_v(const unsigned char* vrow, int key0) {
    VF f;
#pragma unroll
    for (int g = 0; g < 4; ++g) f.v[g] = *(const f32x4*)(vrow + (key0 + 8 * g) * 4);   // keys 8g + 4h + (0..3)
    return f;
  }
  static __device__ __forceinline__ f32x16 pv(const VF& a, const f32x16& p, f32x16 o) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[g][i], p[4 * g + i], o, 0, 0, 0);
    return o;
  }
};

// WPB waves per workgroup share every staged K / V^T tile: with 8 waves (512 threads, 256 queries) each thread stages a key
// every other tile, i.e. half the staging VALU / LDS-write work per query of the 4-wave form at the same 4 waves per SIMD
template <typename T, int QB, int WPB>
__global__ __launch_bounds__(WPB * 64) void attn_kernel(const pd_attn_args a) {
  constexpr int QPW = 32 * QB;                       // queries per wave
  constexpr int QPB = WPB * QPW;                     // queries per workgroup
  using E = Elem<T>;
  using Ops = AttnOps<T>;
#ifdef PD_ABL_NOPIPE
  constexpr bool PIPE = false;
#else
  constexpr bool PIPE = sizeof(T) == 2 && QB == 1;   // 16 more registers: the other forms would lose a wave per SIMD
#endif
  constexpr int KROW = Ops::KROW;
  constexpr int VBYTES = 10 * Ops::VT_PITCH;         // rows 0..7 = V^T, row 8 = 1.0, row 9 = 0
  __shared__ __attribute__((aligned(16))) unsigned char klds[2][KT * KROW + 16];   // + one 16-B slot: k-slots 8..15 of every row
  __shared__ __attribute__((aligned(16))) unsigned char vlds[2][VBYTES];
  __shared__ __attribute__((aligned(16))) float knmax[2][KT / 64];   // max |k| over each 64-key slice (one staging wave): Cauchy-Schwarz score bound

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // 1-D grid with an XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch), so giving each XCD a
  // contiguous range of work items keeps all query blocks of one (batch, head) -- which stream the same K / V -- on
  // one L2 (speed only; any placement is correct)
  const int nqb = (a.N + QPB - 1) / QPB;
  const int total = nqb * a.heads * a.B;
  int item = blockIdx.x;
  if ((total & 7) == 0) item = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  const int qb = item % nqb;
  const int head = (item / nqb) % a.heads, b = item / (nqb * a.heads);
  const int N = a.N;
  const size_t bh = ((size_t)b * a.heads + head) * N;
  const T* qp = (const T*)a.q + bh * 8;
  const T* kp = (const T*)a.k + bh * 8;
  const T* vp = (const T*)a.v + bh * 8;

  // fold softmax scale 8^-1/2 and log2(e) into q: p = exp2(s' - m')
  const float qscale = 0.35355339059327373f * 1.4426950408889634f;
  // Deferred-rescale online softmax (per query block).  `m` is the reference maximum (log2 domain) shared by both lane
  // halves of a query; -m rides in the QK^T MFMA itself (spare k-slots for the 16-bit types, the C operand for fp32: Ops::set_m),
  // so s' = S - m leaves the matrix pipe ready for exp2.
  // m is only raised when some s' exceeds RESCALE_THR (p <= 2^THR otherwise).  A tile needs the exact row max only if
  // some score COULD exceed m + THR: s = q.k <= |q| * max|k| (Cauchy-Schwarz), checked once per 256-key tile.
#ifdef PD_ABL_THR
  constexpr float RESCALE_THR = 1e30f;   // ablation only: never take the exact-max path after the first tile
#else
  constexpr float RESCALE_THR = Ops::RESCALE_THR;
#endif
  int query[QB];
  typename Ops::QF qf[QB];
  f32x16 o[QB], negm[QB];
  float qn[QB], m[QB];
  bool first[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    query[j] = qb * QPB + (wave * QB + j) * 32 + r;
    qf[j] = Ops::load_q(qp + (size_t)min(query[j], N - 1) * 8, h, qscale);
    float t = Ops::q_norm2(qf[j]);
    t += __shfl_xor(t, 32);
    qn[j] = sqrtf(t) * 1.00001f + 1e-6f;
    o[j] = (f32x16)(0.f); negm[j] = (f32x16)(0.f); m[j] = 0.f; first[j] = true;
  }

  // constant LDS content, written once: the shared K slot of k-slots 8..15, all-ones row 8 of V^T (A rows 8..15 -> l)
#pragma unroll
  for (int b2 = 0; b2 < 2; ++b2) {
    if (tid == 0) { *(f32x4*)(klds[b2] + KT * KROW) = (f32x4)(0.f); Ops::init_const_slot(klds[b2] + KT * KROW); }
    if (tid < KT) { *(T*)(vlds[b2] + 8 * Ops::VT_PITCH + tid * E::BYTES) = E::from_f(1.0f); *(T*)(vlds[b2] + 9 * Ops::VT_PITCH + tid * E::BYTES) = E::from_f(0.0f); }
  }
  // A rows of O^T += A . P^T: 0..7 = V^T (d), 8 and 12 = ones (the two rows read back as l, lane halves h = 0 / 1); every other
  // row reads the zero row -- the matrix pipe multiplies zeros instead of 22 rows of live data (the kernel runs power-limited:
  // -3 % time on the DMA-staged kernel)
  const int vrow_off = (r < 8 ? r : (r == 8 || r == 12) ? 8 : 9) * Ops::VT_PITCH + Ops::vlane_off(h);
  const int ka0 = Ops::kaddr(r, h), kst = Ops::kstep(h);

  // staging: thread (t & 255) owns K row t and V row t of a 256-key tile; with 8 waves the two halves of the workgroup take
  // the even / odd tiles in turn (wave-uniform), so the staging work is spread over all waves
  const int st = tid & (KT - 1);
  const int my_parity = (WPB * 64 > KT) ? (tid / KT) : 0;
  auto mine = [&](int tile) { return (WPB * 64 > KT) ? ((tile & 1) == my_parity) : true; };
  typename E::Frag stk, stv;
  auto issue = [&](int k0) {
    if (!mine(k0 / KT)) return;
    const int key = k0 + st;
    if (key < N) { stk = E::load(kp + (size_t)key * 8); stv = E::load(vp + (size_t)key * 8); }
    else { stk = E::zero(); stv = E::zero(); }
  };
  auto commit = [&](int b2, int k0) {
    if (!mine(k0 / KT)) return;
    E::store(klds[b2] + st * KROW, stk);
#ifndef PD_ABL_NONORM
    float n2 = row16_max(Ops::k_norm2(stk));
    n2 = fmaxf(n2, __shfl_xor(n2, 16));
    n2 = fmaxf(n2, __shfl_xor(n2, 32));
    if ((st & 63) == 0) knmax[b2][st >> 6] = sqrtf(n2) * 1.00002f;
#endif
#ifndef PD_ABL_NOVT
    Ops::store_vt(vlds[b2], Ops::vpos(st), stv);
#endif
  };

  issue(0);
  commit(0, 0);
  if (KT < N) issue(KT);
  __syncthreads();
  for (int k0 = 0, cur = 0; k0 < N; k0 += KT, cur ^= 1) {
    const unsigned char* kl = klds[cur];
    const unsigned char* vrow = vlds[cur] + vrow_off;
    // one bound for the whole 256-key tile: if even |q| * max|k| cannot push a score above m + THR, the 8 sub-tiles
    // run the check-free body (QK^T MFMA -> 16 x v_exp -> 8 x cvt_pk -> 2 PV MFMAs, nothing else)
    const f32x4 kn4 = *(const f32x4*)knmax[cur];             // KT / 64 = 4 slices: one 16-byte LDS read
    const float kn8 = fmaxf(fmaxf(kn4[0], kn4[1]), fmaxf(kn4[2], kn4[3]));
    const bool full_tile = k0 + KT <= N;
    bool need = false;
#pragma unroll
    for (int j = 0; j < QB; ++j) need = need || first[j] || (qn[j] * kn8 - m[j] > RESCALE_THR);
#ifdef PD_ABL_NOCHECK   // ablation only
    if (full_tile && (k0 > 0 || !__builtin_amdgcn_ballot_w64(need))) {
#else
    if (full_tile && !__builtin_amdgcn_ballot_w64(need)) {
#endif
      // check-free body: per sub-tile one K and one V^T fragment read feed all QB query blocks
      if constexpr (PIPE) {
        // software-pipelined: the QK^T MFMA of sub-tile i+1 (and the LDS reads of its successors) are issued before the
        // exponentials of sub-tile i, so their latency hides under the wave's own v_exp stream instead of in a stall that
        // other waves have to fill
        typename Ops::KF kfn = Ops::load_k(kl + ka0);
        typename Ops::VF vfn = Ops::load_v(vrow, 0);
        f32x16 sn = Ops::qk(kfn, qf[0], negm[0]);
        kfn = Ops::load_k(kl + ka0 + kst);
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
          const typename Ops::VF vf = vfn;
          f32x16 s = sn;
          if (sub + 1 < KT / 32) {
            sn = Ops::qk(kfn, qf[0], negm[0]);
            if (sub + 2 < KT / 32) kfn = Ops::load_k(kl + ka0 + (sub + 2) * kst);
            vfn = Ops::load_v(vrow, (sub + 1) * 32);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
          o[0] = Ops::pv(vf, s, o[0]);
        }
      } else {
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
          const typename Ops::KF kf = Ops::load_k(kl + ka0 + sub * kst);
          const typename Ops::VF vf = Ops::load_v(vrow, sub * 32);
          f32x16 s[QB];
#pragma unroll
          for (int j = 0; j < QB; ++j) s[j] = Ops::qk(kf, qf[j], negm[j]);
#pragma unroll
          for (int j = 0; j < QB; ++j) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[j][i] = __builtin_amdgcn_exp2f(s[j][i]);
            o[j] = Ops::pv(vf, s[j], o[j]);
          }
        }
      }
    } else {
#pragma unroll 1
      for (int sub = 0; sub < KT / 32; ++sub) {
        const int kb = sub * 32;
        if (k0 + kb >= N) break;
        const typename Ops::KF kf = Ops::load_k(kl + ka0 + sub * kst);
        const typename Ops::VF vf = Ops::load_v(vrow, kb);
#pragma unroll
        for (int j = 0; j < QB; ++j) {
          f32x16 s = Ops::qk(kf, qf[j], negm[j]);
          // mask keys beyond N (accumulator register i <-> key (i&3) + 8(i>>2) + 4h)
          if (k0 + kb + 32 > N) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int key = k0 + kb + (i & 3) + 8 * (i >> 2) + 4 * h;
              if (key >= N) s[i] = -INFINITY;
            }
          }
          const float bound = qn[j] * knmax[cur][sub >> 1] - m[j];   // upper bound of every s' of this lane in this sub-tile
          if (__builtin_amdgcn_ballot_w64(first[j] || bound > RESCALE_THR)) {   // wave-uniform
            float tmax = s[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, s[i]);
            if (__builtin_amdgcn_ballot_w64(first[j] || tmax > RESCALE_THR)) {
              const float t2 = fmaxf(tmax, __shfl_xor(tmax, 32));           // finite: the first tile holds key 0
              const float m_new = Ops::set_m(qf[j], negm[j], h, m[j] + (first[j] ? t2 : fmaxf(t2, 0.f)));
              const float delta = m_new - m[j];             // what the scores of this and of later tiles lose
              const float sc = first[j] ? 1.f : __builtin_amdgcn_exp2f(-delta);
              // only rows 0..15 of O^T are meaningful (d 0..7 and the all-ones rows): registers 0..7
#pragma unroll
              for (int i = 0; i < 8; ++i) o[j][i] *= sc;
#pragma unroll
              for (int i = 0; i < 16; ++i) s[i] -= delta;
              m[j] = m_new;
              first[j] = false;
            }
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
          o[j] = Ops::pv(vf, s, o[j]);
        }
      }
    }
#ifndef PD_ABL_NOSTAGE   // ablation only: the first two tiles' LDS content is reused, no staging, no barrier
    if (k0 + KT < N) {
      commit(cur ^ 1, k0 + KT);              // tile k0+KT: loaded during the previous tile's math
      if (k0 + 2 * KT < N) issue(k0 + 2 * KT);
    }
#ifndef PD_ABL_NOBARRIER
    __syncthreads();
#endif
#endif
  }

#pragma unroll
  for (int j = 0; j < QB; ++j) {
    if (query[j] < N) {
      // lane (query, h): registers 0..3 = O^T rows 4h..4h+3 (d), register 4 = row 8 + 4h = l
      const float inv = 1.0f / o[j][4];
      // log2-domain log-sum-exp of the scaled scores, kept for the backward (p = exp2(s' - lse))
      if (a.lse && h == 0) a.lse[bh + query[j]] = m[j] + __builtin_amdgcn_logf(o[j][4]);
      T* dst = (T*)a.out + ((size_t)b * N + query[j]) * (a.heads * 8) + head * 8 + 4 * h;
      store4(dst, o[j][0] * inv, o[j][1] * inv, o[j][2] * inv, o[j][3] * inv);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same attention with the staging taken off the vector pipe (16-bit types, 8 waves = 256 queries per workgroup, needs
// pd_attn_args.kmax2).  In attn_kernel every staged key costs its thread two global loads, |k|^2, a 6-step cross-lane max,
// a 16-byte and eight 2-byte LDS writes (V transposed on the way): 7 % of the kernel's time at N = 4096, on the VALU port
// that bounds it.  Here
//   * K and V tiles keep their global layout [key][8] in LDS, so a tile is eight 1-KiB global_load_lds pieces (one per wave,
//     no registers, no VALU); the DMA of tile t+1 is issued before tile t's math and retired (vmcnt(0)) before the barrier
//     that ends it,
//   * the V^T fragments of O^T += [V^T; 1] . P^T come from transposed LDS reads (ds_read_b64_tr_b16): lane 4q+p of a
//     16-lane group supplies key row q, d columns 4p..4p+3 (p < 2) or a block of ones (p >= 2: A rows 8..15 -> l),
//   * the score bound uses the producer's max |k|^2 per (batch, head) instead of per-tile norms: once
//     |q| * max|k| - m <= THR every later tile runs the check-free body with no LDS read in front of it.
#ifndef PD_ATTN_DMA_KT
#define PD_ATTN_DMA_KT 256
#endif
template <typename T>
__global__ __launch_bounds__(512) void attn_glds_kernel(const pd_attn_args a) {
  static_assert(sizeof(T) == 2, "16-bit element types");
  using Ops = AttnOps<T>;
  constexpr int KT = PD_ATTN_DMA_KT;                                                  // keys per LDS tile (one barrier per tile)
  constexpr int KROW = 16, TILE = KT * KROW;
  // one LDS array, addressed by byte offsets (plain integers keep every access a ds_* instruction):
  //   K tiles [2][TILE] | V tiles [2][TILE] | constants [ONES] ([1 0 0 0 | 0 0 0 0] per 16 bytes, see voff below) | k-slots 8..15 of every K row (16 B)
  constexpr int ONES = TILE + 256;                     // the ones lanes read up to 128 + 8 bytes past a tile-sized block
  constexpr int K_OFF = 0, V_OFF = 2 * TILE, ONES_OFF = 4 * TILE, KCONST_OFF = 4 * TILE + ONES;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * TILE + ONES + 16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nqb = (a.N + 255) / 256;
  const int total = nqb * a.heads * a.B;
  int item = blockIdx.x;
  if ((total & 7) == 0) item = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);   // XCD-aware remap, as attn_kernel
  const int qb = item % nqb;
  const int head = (item / nqb) % a.heads, b = item / (nqb * a.heads);
  const int N = a.N;
  const size_t bh = ((size_t)b * a.heads + head) * N;
  const T* qp = (const T*)a.q + bh * 8;

  const float qscale = 0.35355339059327373f * 1.4426950408889634f;
  constexpr float RESCALE_THR = Ops::RESCALE_THR;
  const int query = qb * 256 + wave * 32 + r;
  typename Ops::QF qf = Ops::load_q(qp + (size_t)min(query, N - 1) * 8, h, qscale);
  float qn = Ops::q_norm2(qf);
  qn += __shfl_xor(qn, 32);
  qn = sqrtf(qn) * 1.00001f + 1e-6f;
  const float kn = sqrtf(a.kmax2[b * a.heads + head]) * 1.00002f;
  float qk_bound = qn * kn;                       // every scaled score of this query is <= qk_bound (Cauchy-Schwarz)
  f32x16 o = (f32x16)(0.f), zero = (f32x16)(0.f);
  float m = 0.f;

  // constant LDS content
  if (tid == 0) Ops::init_const_slot(lds + KCONST_OFF);
  for (int i16 = tid; i16 < ONES / 16; i16 += 512) {     // every 16 bytes: [1 0 0 0 | 0 0 0 0]
    T one8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) one8[i] = Elem<T>::from_f(i == 0 ? 1.0f : 0.0f);
    *(s16x8*)(lds + ONES_OFF + i16 * 16) = *(const s16x8*)one8;
  }
  // K fragment offset in buffer 0: h == 0 -> row r of the sub-tile; h == 1 -> the constant slot (every sub-tile, both buffers)
  const int koff = h ? KCONST_OFF : K_OFF + r * KROW;
  const int kst = h ? 0 : 32 * KROW, kbuf = h ? 0 : TILE;
  // V^T fragment offsets (transposed read; EXEC is all ones wherever they are used).  Key-row lanes sit on banks
  // 16h + 32j .. +15, the ones lanes on 16h + 16 + 32j ..: no conflict inside a 32-lane half; lanes 16..31 of each half repeat
  // lanes 0..15 (A rows 16..31 are never read back)
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  // Only A rows 0..7 (d), 8 and 12 (the two rows whose sums are read back as l) matter: columns 8 and 12 are 1, every other
  // column of 8..15 and all of rows 16..31 (lanes 16..31 of each half) are 0 -- the matrix pipe multiplies zeros instead of
  // 22 rows of live data (power, i.e. clock, on a kernel that runs DVFS-limited)
  const bool vrow = (lane & 16) == 0 && tp < 2;
  const int voff = vrow ? V_OFF + (4 * h + tq) * KROW + 8 * tp : ONES_OFF + (16 * h + 16) * 4 + ((lane & 16) ? 8 : 0);
  const int vbuf = vrow ? TILE : 0;
  auto load_v = [&](int base, int sub) {      // A operands of the two PV MFMAs of 32-key sub-tile `sub`
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s* lp;
    typename Ops::VF f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      // read j: k-slots 4j..4j+3 of this lane half = keys 32 sub + 16 s2 + 8 j + 4 h + (0..3)
      const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + base + (32 * sub + 16 * s2) * KROW));
      const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + base + (32 * sub + 16 * s2 + 8) * KROW));
      const s16x8 fr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      f.v[s2] = __builtin_bit_cast(u32x4, fr);
    }
    return f;
  };

  // staging: waves 0..3 copy the four 1-KiB pieces of the K tile, waves 4..7 those of the V tile (keys past N re-read the
  // last row: their scores are masked below, their p = 0 multiplies finite values).  The DMA is inline asm: the compiler
  // would order every later ds_read behind a global_load_lds it can see (vmcnt(0) in the middle of the tile).
  const unsigned char* gsrc = (const unsigned char*)((const T*)(wave < 4 ? a.k : a.v) + bh * 8);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned sdst = __builtin_amdgcn_readfirstlane(lds_base + (wave < 4 ? K_OFF : V_OFF) + (wave & 3) * 1024);
  auto stage = [&](int b2, int k0) {
    #pragma unroll
    for (int pc = 0; pc < KT / 256; ++pc) {       // KT / 64 pieces per operand tile, 4 waves per operand
      const int key = min(k0 + ((wave & 3) + 4 * pc) * 64 + lane, N - 1);
      const unsigned char* src = gsrc + (size_t)key * 16;
      const unsigned dst = sdst + b2 * TILE + pc * 4096;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  };

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // the reference maximum starts at the exact maximum over the first 32 keys (one QK^T MFMA, nothing accumulated): from the
  // first tile on, a tile whose bound allows it takes the check-free body -- the first tile used to pay the exact-max path for
  // all of its 256 keys (1/16 of the work at N = 4096, a quarter at N = 1024).  N >= 1024 here: no key of this block is padding.
  {
    const f32x16 s0 = Ops::qk(Ops::load_k(lds + koff), qf, zero);
    float tmax = s0[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, s0[i]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    m = Ops::set_m(qf, zero, h, tmax);
    // kmax2 is a HARD precondition (>= max |k|^2 of this (sample, head): pd_linear.kmax2_out after a pd_zero).  Safety net for a
    // slot that is zero / stale / belongs to another K: a first-sub-tile score above the "bound" proves it wrong ->
    // this wave keeps the exact running-maximum path for the whole launch (no unbounded 2^(s - m); costs speed, not results)
    if (tmax > qk_bound) qk_bound = INFINITY;
  }
  for (int k0 = 0, cur = 0; k0 < N; k0 += KT, cur ^= 1) {
    if (k0 + KT < N) stage(cur ^ 1, k0 + KT);      // buffer cur^1 was last read before the barrier that ended the previous tile
    const int kl = koff + cur * kbuf;
    const int vb = voff + cur * vbuf;
    const bool full_tile = k0 + KT <= N;
    const bool need = qk_bound - m > RESCALE_THR;
    if (full_tile && !__builtin_amdgcn_ballot_w64(need)) {
      // check-free body, software-pipelined as in attn_kernel
      typename Ops::KF kfn = Ops::load_k(lds + kl);
      typename Ops::VF vfn = load_v(vb, 0);
      f32x16 sn = Ops::qk(kfn, qf, zero);
      kfn = Ops::load_k(lds + kl + kst);
#pragma unroll
      for (int sub = 0; sub < KT / 32; ++sub) {
        const typename Ops::VF vf = vfn;
        f32x16 s = sn;
        if (sub + 1 < KT / 32) {
          sn = Ops::qk(kfn, qf, zero);
          if (sub + 2 < KT / 32) kfn = Ops::load_k(lds + kl + (sub + 2) * kst);
          vfn = load_v(vb, sub + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
        o = Ops::pv(vf, s, o);
      }
    } else {
#pragma unroll 1
      for (int sub = 0; sub < KT / 32; ++sub) {
        const int kb = sub * 32;
        if (k0 + kb >= N) break;
        const typename Ops::KF kf = Ops::load_k(lds + kl + sub * kst);
        const typename Ops::VF vf = load_v(vb, sub);
        f32x16 s = Ops::qk(kf, qf, zero);
        if (k0 + kb + 32 > N) {      // mask keys beyond N (accumulator register i <-> key (i&3) + 8(i>>2) + 4h)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = k0 + kb + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (key >= N) s[i] = -INFINITY;
          }
        }
        if (__builtin_amdgcn_ballot_w64(qk_bound - m > RESCALE_THR)) {   // wave-uniform
          float tmax = s[0];
#pragma unroll
          for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, s[i]);
          if (__builtin_amdgcn_ballot_w64(tmax > RESCALE_THR)) {
            const float t2 = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = Ops::set_m(qf, zero, h, m + fmaxf(t2, 0.f));
            const float delta = m_new - m;
            const float sc = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] *= sc;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] -= delta;
            m = m_new;
          }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
        o = Ops::pv(vf, s, o);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA piece of the next tile has landed ...
    __syncthreads();                                   // ... and so has everybody's; nobody still reads buffer `cur`
  }

  if (query < N) {
    const float inv = 1.0f / o[4];
    if (a.lse && h == 0) a.lse[bh + query] = m + __builtin_amdgcn_logf(o[4]);
    T* dst = (T*)a.out + ((size_t)b * N + query) * (a.heads * 8) + head * 8 + 4 * h;
    store4(dst, o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward (autograd of F.scaled_dot_product_attention, AttnProcessor2_0).  P is recomputed from the forward's
// log-sum-exp; with head_dim 8 the exp, not the MFMA, prices a tile, so the pass is split in two kernels that each keep
// their reduction lane-local instead of one kernel with atomics:
//   dQ  kernel: query on the lane (forward's layout).  S^T = K.Q^T - lse, dP^T = V.dO^T - delta (both constants ride in the
//               MFMA C operand), dS^T = exp2(S^T) * dP^T stays in registers as the B operand of dQ^T += K^T . dS^T.
//   dKV kernel: key on the lane.  S = Q.K^T - lse[query], dP = dO.V^T - delta[query] (C operand read per query row from
//               LDS), P and dS are the B operands of dV^T += dO^T . P and dK^T += Q^T . dS.
// delta = rowsum(dO * O) is produced by the dQ kernel and reused by the dKV kernel (same stream).
// Gradients land in one NHWC tensor [B][N][3*heads*8] = [dq | dk | dv], the layout of the fused q/k/v projection's output
// gradient, so the projection's pd_conv (input gradient) and pd_conv_wgrad consume it directly.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const pd_attn_bwd_args a) {
  using E = Elem<T>;
  using Ops = AttnOps<T>;
  constexpr int KROW = Ops::KROW;
  __shared__ __attribute__((aligned(16))) unsigned char klds[2][KT * KROW + 16];
  __shared__ __attribute__((aligned(16))) unsigned char vlds[2][KT * KROW + 16];
  __shared__ __attribute__((aligned(16))) unsigned char ktlds[2][9 * Ops::VT_PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nqb = (a.N + 127) / 128;
  const int total = nqb * a.heads * a.B;
  int item = blockIdx.x;
  if ((total & 7) == 0) item = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  const int qb = item % nqb;
  const int head = (item / nqb) % a.heads, b = item / (nqb * a.heads);
  const int N = a.N, C = a.heads * 8;
  const size_t bh = ((size_t)b * a.heads + head) * N;
  const T* kp = (const T*)a.k + bh * 8;
  const T* vp = (const T*)a.v + bh * 8;
  const float qscale = 0.35355339059327373f * 1.4426950408889634f;

  const int query = qb * 128 + wave * 32 + r, qc = min(query, N - 1);
  const typename Ops::QF qf0 = Ops::load_q((const T*)a.q + (bh + qc) * 8, h, qscale);
  const typename Ops::QF dof0 = Ops::load_q((const T*)a.dout + ((size_t)b * N + qc) * C + head * 8, h, 1.0f);
  const typename Ops::QF of = Ops::load_q((const T*)a.o + ((size_t)b * N + qc) * C + head * 8, h, 1.0f);
  float delta = Ops::dot(dof0, of);
  delta += __shfl_xor(delta, 32);
  if (h == 0 && query < N) a.delta[bh + query] = delta;
  // -lse and -delta: in the C operand (fp32), or in the three spare k-slots of the two QK-shaped MFMAs (16-bit types: no
  // 2 x 16-register constant blocks -> one more wave per SIMD)
  constexpr bool SLOTS = !Ops::M_IN_C;
  f32x16 negl = (f32x16)(0.f), negd = (f32x16)(0.f);
  typename Ops::QF qf = qf0, dof = dof0;
  if constexpr (SLOTS) { Ops::set_terms(qf, h, -a.lse[bh + qc]); Ops::set_terms(dof, h, -delta); }
  else { negl = (f32x16)(-a.lse[bh + qc]); negd = (f32x16)(-delta); }
  f32x16 dq = (f32x16)(0.f);

#pragma unroll
  for (int b2 = 0; b2 < 2; ++b2) {
    if (tid == 0) {
      *(f32x4*)(klds[b2] + KT * KROW) = (f32x4)(0.f); *(f32x4*)(vlds[b2] + KT * KROW) = (f32x4)(0.f);
      if constexpr (SLOTS) { Ops::init_const_slot(klds[b2] + KT * KROW); Ops::init_const_slot(vlds[b2] + KT * KROW); }
    }
    *(T*)(ktlds[b2] + 8 * Ops::VT_PITCH + tid * E::BYTES) = E::from_f(0.0f);
  }
  const int vrow_off = (r < 8 ? r : 8) * Ops::VT_PITCH + Ops::vlane_off(h);   // A rows 8..31: the zero row
  const int ka0 = Ops::kaddr(r, h), kst = Ops::kstep(h);

  typename E::Frag stk, stv;
  auto issue = [&](int k0) {
    const int key = k0 + tid;
    if (key < N) { stk = E::load(kp + (size_t)key * 8); stv = E::load(vp + (size_t)key * 8); }
    else { stk = E::zero(); stv = E::zero(); }
  };
  auto commit = [&](int b2) {
    E::store(klds[b2] + tid * KROW, stk);
    E::store(vlds[b2] + tid * KROW, stv);
    float kv[8];
    E::unpack(stk, kv);
    const int vp_ = Ops::vpos(tid);
#pragma unroll
    for (int d = 0; d < 8; ++d) *(T*)(ktlds[b2] + d * Ops::VT_PITCH + vp_ * E::BYTES) = E::from_f(kv[d]);
  };
  auto body = [&](const unsigned char* kl, const unsigned char* vl, const unsigned char* ktrow, int sub) {
    const typename Ops::KF kf = Ops::load_k(kl + ka0 + sub * kst);
    const typename Ops::KF vf = Ops::load_k(vl + ka0 + sub * kst);
    const typename Ops::VF ktf = Ops::load_v(ktrow, sub * 32);
    f32x16 s = Ops::qk(kf, qf, negl);
    const f32x16 dp = Ops::qk(vf, dof, negd);
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]) * dp[i];
    dq = Ops::pv(ktf, s, dq);
  };

  issue(0);
  commit(0);
  if (KT < N) issue(KT);
  __syncthreads();
  for (int k0 = 0, cur = 0; k0 < N; k0 += KT, cur ^= 1) {
    const unsigned char* kl = klds[cur];
    const unsigned char* vl = vlds[cur];
    const unsigned char* ktrow = ktlds[cur] + vrow_off;
    if (k0 + KT <= N) {
#pragma unroll
      for (int sub = 0; sub < KT / 32; ++sub) body(kl, vl, ktrow, sub);
    } else {
#pragma unroll 1
      for (int sub = 0; sub < KT / 32 && k0 + sub * 32 < N; ++sub) body(kl, vl, ktrow, sub);   // padded keys: K^T columns are zero
    }
    if (k0 + KT < N) {
      commit(cur ^ 1);
      if (k0 + 2 * KT < N) issue(k0 + 2 * KT);
    }
    __syncthreads();
  }
  if (query < N) {
    const float sc = 0.35355339059327373f;     // d(scale * q.k)/dq
    T* dst = (T*)a.dqkv + ((size_t)b * N + query) * (3 * C) + head * 8 + 4 * h;
    store4(dst, dq[0] * sc, dq[1] * sc, dq[2] * sc, dq[3] * sc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const pd_attn_bwd_args a) {
  using E = Elem<T>;
  using Ops = AttnOps<T>;
  constexpr int KROW = Ops::KROW;
  constexpr int QT = sizeof(T) == 4 ? 128 : 256;       // queries per LDS tile (fp32 validation mode: LDS budget)
  // 16-bit types: every Q / dO row is followed (QT rows later) by a 16-byte extension row [hi mid lo 0 ...] = -lse / -delta of
  // that query in three T-precision terms: k-slots 8..10 of the A operand, against (M_C0, 1, 1) in K / V -- the per-row constants
  // ride in the MFMA instead of a 16-register C block assembled from eight LDS reads per sub-tile (161 -> fewer registers:
  // one more wave per SIMD).  fp32: the C-operand form (nl / nd).
  constexpr bool SLOTS = !Ops::M_IN_C;
  constexpr int ROWS = SLOTS ? 2 * QT * KROW : QT * KROW + 16;
  __shared__ __attribute__((aligned(16))) unsigned char qlds[2][ROWS];
  __shared__ __attribute__((aligned(16))) unsigned char dolds[2][ROWS];
  __shared__ __attribute__((aligned(16))) unsigned char qtlds[2][9 * Ops::VT_PITCH];
  __shared__ __attribute__((aligned(16))) unsigned char dotlds[2][9 * Ops::VT_PITCH];
  __shared__ __attribute__((aligned(16))) float nl[2][SLOTS ? 4 : QT], nd[2][SLOTS ? 4 : QT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nkb = (a.N + 127) / 128;
  const int total = nkb * a.heads * a.B;
  int item = blockIdx.x;
  if ((total & 7) == 0) item = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  const int kb = item % nkb;
  const int head = (item / nkb) % a.heads, b = item / (nkb * a.heads);
  const int N = a.N, C = a.heads * 8;
  const size_t bh = ((size_t)b * a.heads + head) * N;
  const T* qp = (const T*)a.q + bh * 8;
  const T* dop = (const T*)a.dout + (size_t)b * N * C + head * 8;
  const float qscale = 0.35355339059327373f * 1.4426950408889634f;

  const int key = kb * 128 + wave * 32 + r, kc = min(key, N - 1);
  typename Ops::QF kfr = Ops::load_q((const T*)a.k + (bh + kc) * 8, h, 1.0f);
  typename Ops::QF vfr = Ops::load_q((const T*)a.v + (bh + kc) * 8, h, 1.0f);
  if constexpr (SLOTS) { Ops::set_ones(kfr, h); Ops::set_ones(vfr, h); }
  f32x16 dk = (f32x16)(0.f), dv = (f32x16)(0.f);

#pragma unroll
  for (int b2 = 0; b2 < 2; ++b2) {
    if constexpr (!SLOTS) {
      if (tid == 0) { *(f32x4*)(qlds[b2] + QT * KROW) = (f32x4)(0.f); *(f32x4*)(dolds[b2] + QT * KROW) = (f32x4)(0.f); }
    }
    if (tid < QT) {
      *(T*)(qtlds[b2] + 8 * Ops::VT_PITCH + tid * E::BYTES) = E::from_f(0.0f);
      *(T*)(dotlds[b2] + 8 * Ops::VT_PITCH + tid * E::BYTES) = E::from_f(0.0f);
    }
  }
  const int vrow_off = (r < 8 ? r : 8) * Ops::VT_PITCH + Ops::vlane_off(h);   // A rows 8..31: the zero row
  // K-row style addressing of the Q / dO row images; the bf16 zero slot sits after QT rows here
  const int ka0 = SLOTS ? (h ? QT * KROW + r * KROW : r * KROW) : Ops::kaddr(r, h);      // h == 1: the extension rows
  const int kst = SLOTS ? 32 * KROW : Ops::kstep(h);

  typename E::Frag stq, stdo;
  float stl = 0.f, std_ = 0.f;
  auto issue = [&](int q0) {
    const int query = q0 + tid;
    if (tid < QT && query < N) {
      stq = E::load(qp + (size_t)query * 8); stdo = E::load(dop + (size_t)query * C);
      stl = -a.lse[bh + query]; std_ = -a.delta[bh + query];
    } else { stq = E::zero(); stdo = E::zero(); stl = 0.f; std_ = 0.f; }
  };
  auto commit = [&](int b2) {
    if (tid < QT) {
      float qv[8], dv_[8];
      E::unpack(stq, qv);
#pragma unroll
      for (int d = 0; d < 8; ++d) qv[d] *= qscale;
      const typename E::Frag qs = E::pack(qv);
      E::store(qlds[b2] + tid * KROW, qs);
      E::store(dolds[b2] + tid * KROW, stdo);
      E::unpack(qs, qv);
      E::unpack(stdo, dv_);
      const int vp_ = Ops::vpos(tid);
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        *(T*)(qtlds[b2] + d * Ops::VT_PITCH + vp_ * E::BYTES) = E::from_f(qv[d]);
        *(T*)(dotlds[b2] + d * Ops::VT_PITCH + vp_ * E::BYTES) = E::from_f(dv_[d]);
      }
      if constexpr (SLOTS) {
        Ops::store_terms(qlds[b2] + (QT + tid) * KROW, stl);
        Ops::store_terms(dolds[b2] + (QT + tid) * KROW, std_);
      } else { nl[b2][tid] = stl; nd[b2][tid] = std_; }
    }
  };
  auto body = [&](int cur, int sub) {
    const typename Ops::KF qa = Ops::load_k(qlds[cur] + ka0 + sub * kst);
    const typename Ops::KF doa = Ops::load_k(dolds[cur] + ka0 + sub * kst);
    f32x16 cl = (f32x16)(0.f), cd = (f32x16)(0.f);       // accumulator register i <-> query (i&3) + 8(i>>2) + 4h of the sub-tile
    if constexpr (!SLOTS) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *(const f32x4*)&nl[cur][sub * 32 + 8 * g + 4 * h];
        const f32x4 d4 = *(const f32x4*)&nd[cur][sub * 32 + 8 * g + 4 * h];
#pragma unroll
        for (int i = 0; i < 4; ++i) { cl[4 * g + i] = l4[i]; cd[4 * g + i] = d4[i]; }
      }
    }
    f32x16 p = Ops::qk(qa, kfr, cl);
    f32x16 ds = Ops::qk(doa, vfr, cd);
#pragma unroll
    for (int i = 0; i < 16; ++i) { p[i] = __builtin_amdgcn_exp2f(p[i]); ds[i] *= p[i]; }
    dv = Ops::pv(Ops::load_v(dotlds[cur] + vrow_off, sub * 32), p, dv);
    dk = Ops::pv(Ops::load_v(qtlds[cur] + vrow_off, sub * 32), ds, dk);
  };

  issue(0);
  commit(0);
  if (QT < N) issue(QT);
  __syncthreads();
  for (int q0 = 0, cur = 0; q0 < N; q0 += QT, cur ^= 1) {
    if (q0 + QT <= N) {
#pragma unroll
      for (int sub = 0; sub < QT / 32; ++sub) body(cur, sub);
    } else {
#pragma unroll 1
      for (int sub = 0; sub < QT / 32 && q0 + sub * 32 < N; ++sub) body(cur, sub);   // padded queries: zero rows, p = 1, ds = 0
    }
    if (q0 + QT < N) {
      commit(cur ^ 1);
      if (q0 + 2 * QT < N) issue(q0 + 2 * QT);
    }
    __syncthreads();
  }
  if (key < N) {
    const float ln2 = 0.6931471805599453f;      // Q^T carried scale * log2(e); dK = scale * dS^T q
    T* dst = (T*)a.dqkv + ((size_t)b * N + key) * (3 * C) + head * 8 + 4 * h;
    store4(dst + C, dk[0] * ln2, dk[1] * ln2, dk[2] * ln2, dk[3] * ln2);
    store4(dst + 2 * C, dv[0], dv[1], dv[2], dv[3]);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Backward in ONE pass (round 5; 16-bit engines, N >= 512, needs the pd_attn_bwd_args.slab workspace).  The two kernels above
// each recompute P = exp2(S - lse): 32 exponentials and 10 MFMAs per 32 x 32 tile, on a vector issue port that prices the tile.
// Here the key stays on the lane (dK, dV lane-local, as in the dK/dV kernel) and dS additionally goes through a wave-private LDS
// image to come back TRANSPOSED (ds_read_b64_tr_b16: query on the lane, 8 keys per register group) as the B operand of
// dQ^T += K^T . dS^T: 16 exponentials and 8 MFMAs per tile.
//   workgroup = 8 waves x NSUB 32-key sub-tiles = 256 NSUB keys of one (sample, head); K, V, K^T fragments and the dK / dV
//               accumulators live in registers for the whole kernel;
//   query tiles of 256 (Q / dO row images with their -lse / -delta extension rows, Q^T / dO^T images: the dK/dV kernel's staging;
//               delta = rowsum(dO o O) is formed while staging, no pre-pass).  Every wave walks the tile's 8 query sub-tiles with NO
//               barrier in between; per sub-tile its NSUB key sub-tiles run as one instruction stream (`step`: shared fragment reads,
//               all score MFMAs up front, one key sub-tile's transposition round trip hidden behind the other's vector work).
//               (First version: a Latin-square walk with one barrier per sub-tile into a shared LDS accumulator, 2 waves per SIMD
//               stalled on the transposition: 1.30 ms per layer = the SUM of the two kernels it replaces; second: software-pipelined
//               bodies, no step barriers: 1.25 ms.)
//   at the end of a tile the 8 waves' partial dQ go through LDS (one private block each), are added in wave order and leave for
//   slab[key block][...] (fp32) or, with one key block, straight to dqkv; attn_dq_reduce_kernel adds the key blocks in order.
//   Run-to-run bit-identical.
constexpr int FUSED_TR_PITCH = 72;        // bytes per key row of a transposition image (64 + 8: 2-way bank conflicts at most)
constexpr int FUSED_TR_IMG = 32 * FUSED_TR_PITCH;
template <typename T, int NSUB>
__global__ __launch_bounds__(512) void attn_bwd_fused_kernel(const pd_attn_bwd_args a, float* __restrict__ slab, int nkb) {
  static_assert(sizeof(T) == 2, "16-bit element types");
  using E = Elem<T>;
  using Ops = AttnOps<T>;
  constexpr int KROW = 16, QT = 256, KPB = 256 * NSUB;
  constexpr int ROWS = 2 * QT * KROW;                   // row image + extension rows
  constexpr int VTB = 9 * Ops::VT_PITCH;
  // dynamic LDS: qlds[2] | dolds[2] | qtlds[2] | dotlds[2] | transposition images [8 waves][2] | partial dQ [8 waves][QT][8] fp32
  constexpr int Q_OFF = 0, DO_OFF = 2 * ROWS, QT_OFF = 4 * ROWS, DOT_OFF = 4 * ROWS + 2 * VTB, TR_OFF = 4 * ROWS + 4 * VTB,
                DQ_OFF = TR_OFF + 8 * 2 * FUSED_TR_IMG;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int total = nkb * a.heads * a.B;
  int item = blockIdx.x;
  if ((total & 7) == 0) item = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  const int kb = item % nkb;
  const int head = (item / nkb) % a.heads, b = item / (nkb * a.heads);
  const int N = a.N, C = a.heads * 8;
  const size_t bh = ((size_t)b * a.heads + head) * N;
  const T* qp = (const T*)a.q + bh * 8;
  const T* dop = (const T*)a.dout + (size_t)b * N * C + head * 8;
  const T* op = (const T*)a.o + (size_t)b * N * C + head * 8;
  const float qscale = 0.35355339059327373f * 1.4426950408889634f;

  typename Ops::QF kfr[NSUB], vfr[NSUB];
  s16x8 kt[NSUB][2];                                    // K^T as the A operand of dQ^T += K^T . dS^T: lane (row d = r < 8, h), k = key 16 s + 8 h + j
  f32x16 dk[NSUB], dv[NSUB];
#pragma unroll
  for (int ks = 0; ks < NSUB; ++ks) {
    const int key0 = kb * KPB + (wave * NSUB + ks) * 32;
    const int key = key0 + r, kc = min(key, N - 1);
    kfr[ks] = Ops::load_q((const T*)a.k + (bh + kc) * 8, h, 1.0f);
    vfr[ks] = Ops::load_q((const T*)a.v + (bh + kc) * 8, h, 1.0f);
    Ops::set_ones(kfr[ks], h); Ops::set_ones(vfr[ks], h);
    dk[ks] = (f32x16)(0.f); dv[ks] = (f32x16)(0.f);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      s16x8 f = (s16x8)(0);
      if (r < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kk = key0 + 16 * s2 + 8 * h + j;
          if (kk < N) f[j] = (short)bits16(((const T*)a.k)[(bh + kk) * 8 + r]);      // a key past N adds nothing to dQ
        }
      }
      kt[ks][s2] = f;
    }
  }

#pragma unroll
  for (int b2 = 0; b2 < 2; ++b2) {
    if (tid < QT) {
      *(T*)(lds + QT_OFF + b2 * VTB + 8 * Ops::VT_PITCH + tid * E::BYTES) = E::from_f(0.0f);
      *(T*)(lds + DOT_OFF + b2 * VTB + 8 * Ops::VT_PITCH + tid * E::BYTES) = E::from_f(0.0f);
    }
  }
  const int vrow_off = (r < 8 ? r : 8) * Ops::VT_PITCH + Ops::vlane_off(h);   // A rows 8..31: the zero row
  const int ka0 = h ? QT * KROW + r * KROW : r * KROW;                        // h == 1: the extension rows (-lse / -delta terms)
  constexpr int kst = 32 * KROW;
  unsigned char* const trw = lds + TR_OFF + wave * (2 * FUSED_TR_IMG);        // this wave's two transposition images [key 32][query 32] T
  // transposed read (T10): lane 4q + p of a 16-lane group supplies key row q, query columns 4p .. 4p+3 of the group's 16 queries;
  // lane i receives query i, element q' = key row q'.  This lane: query r, keys 16 s + 8 h + (0..3 | 4..7)
  const int tr_rd = ((lane & 15) >> 2) * FUSED_TR_PITCH + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2 + 8 * h * FUSED_TR_PITCH;
  const int tr_wr = r * FUSED_TR_PITCH + 8 * h;                               // + 16 g: registers 4g .. 4g+3 = queries 8g + 4h + (0..3)
  float* const dq_mine = (float*)(lds + DQ_OFF) + wave * (QT * 8);

  typename E::Frag stq, stdo, sto;
  float stl = 0.f;
  auto issue = [&](int q0) {
    const int query = q0 + tid;
    if (tid < QT && query < N) {
      stq = E::load(qp + (size_t)query * 8); stdo = E::load(dop + (size_t)query * C); sto = E::load(op + (size_t)query * C);
      stl = -a.lse[bh + query];
    } else { stq = E::zero(); stdo = E::zero(); sto = E::zero(); stl = 0.f; }
  };
  auto commit = [&](int b2, int q0) {
    if (tid < QT) {
      float qv[8], dv_[8], ov[8];
      E::unpack(stq, qv);
#pragma unroll
      for (int d = 0; d < 8; ++d) qv[d] *= qscale;
      const typename E::Frag qs = E::pack(qv);
      E::store(lds + Q_OFF + b2 * ROWS + tid * KROW, qs);
      E::store(lds + DO_OFF + b2 * ROWS + tid * KROW, stdo);
      E::unpack(qs, qv);
      E::unpack(stdo, dv_);
      E::unpack(sto, ov);
      float delta = 0.f;
#pragma unroll
      for (int d = 0; d < 8; ++d) delta += dv_[d] * ov[d];
      const int vp_ = Ops::vpos(tid);
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        *(T*)(lds + QT_OFF + b2 * VTB + d * Ops::VT_PITCH + vp_ * E::BYTES) = E::from_f(qv[d]);
        *(T*)(lds + DOT_OFF + b2 * VTB + d * Ops::VT_PITCH + vp_ * E::BYTES) = E::from_f(dv_[d]);
      }
      Ops::store_terms(lds + Q_OFF + b2 * ROWS + (QT + tid) * KROW, stl);
      Ops::store_terms(lds + DO_OFF + b2 * ROWS + (QT + tid) * KROW, -delta);
      if (kb == 0 && q0 + tid < N) a.delta[bh + q0 + tid] = delta;            // (kept for callers that read it back)
    }
  };
  typedef short v4s __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) v4s* lp;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  // One query sub-tile against this wave's NSUB key sub-tiles, as ONE instruction stream: the Q / dO row fragments and the Q^T / dO^T
  // fragments are read once for all of them, the score MFMAs of every key sub-tile are issued first (the matrix pipe runs them while the
  // vector pipe starts on the first one's exponentials), and a key sub-tile's dS sits in its transposition image for the length of the
  // next one's vector work before it is read back -- the LDS round trip is hidden without a second pair of images.
  auto step = [&](int cur, int sub, f32x16& dq) {
    // LLVM's MFMA / exp interleaving strategy over the step (same box: none 1.228 ms per configs[1] layer incl. the reduce, (1) 1.300, (2) 1.215,
    // (3) 1.210; -DPD_ATTN_BWD_IGLP=n builds another)
#ifndef PD_ATTN_BWD_IGLP
#define PD_ATTN_BWD_IGLP 3
#endif
    __builtin_amdgcn_iglp_opt(PD_ATTN_BWD_IGLP);
    const typename Ops::KF qa = Ops::load_k(lds + Q_OFF + cur * ROWS + ka0 + sub * kst);
    const typename Ops::KF doa = Ops::load_k(lds + DO_OFF + cur * ROWS + ka0 + sub * kst);
    const typename Ops::VF dotf = Ops::load_v(lds + DOT_OFF + cur * VTB + vrow_off, sub * 32);
    const typename Ops::VF qtf = Ops::load_v(lds + QT_OFF + cur * VTB + vrow_off, sub * 32);
    const f32x16 zero = (f32x16)(0.f);
    f32x16 p[NSUB], ds[NSUB];
#pragma unroll
    for (int ks = 0; ks < NSUB; ++ks) {               // accumulator register i <-> query (i&3) + 8(i>>2) + 4h of the sub-tile, lane <-> key
      p[ks] = Ops::qk(qa, kfr[ks], zero);
      ds[ks] = Ops::qk(doa, vfr[ks], zero);
    }
    uint32_t bw[NSUB][2][4];
#pragma unroll
    for (int ks = 0; ks < NSUB; ++ks) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { p[ks][i] = __builtin_amdgcn_exp2f(p[ks][i]); ds[ks][i] *= p[ks][i]; }
      // dS packed ONCE: the B operand of dK^T += Q^T . dS and -- the same dwords -- the rows of the transposition image
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 4; ++j) bw[ks][s2][j] = Ops::pack_p(ds[ks][8 * s2 + 2 * j], ds[ks][8 * s2 + 2 * j + 1]);
#pragma unroll
      for (int g = 0; g < 4; ++g) {                   // registers 4g .. 4g+3 -> 8 bytes at [key r][query 8g + 4h]
        const u32x2 w2 = {bw[ks][g >> 1][(g & 1) * 2], bw[ks][g >> 1][(g & 1) * 2 + 1]};
        *(u32x2*)(trw + ks * FUSED_TR_IMG + tr_wr + 16 * g) = w2;
      }
    }
#pragma unroll
    for (int ks = 0; ks < NSUB; ++ks) {
      dv[ks] = Ops::pv(dotf, p[ks], dv[ks]);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const u32x4 bv = {bw[ks][s2][0], bw[ks][s2][1], bw[ks][s2][2], bw[ks][s2][3]};
        dk[ks] = E::mma16(__builtin_bit_cast(s16x8, qtf.v[s2]), __builtin_bit_cast(s16x8, bv), dk[ks]);
      }
      // the image read back transposed: dQ^T (rows 0..7 = d) += K^T . dS^T
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(trw + ks * FUSED_TR_IMG + tr_rd + (16 * s2) * FUSED_TR_PITCH));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(trw + ks * FUSED_TR_IMG + tr_rd + (16 * s2 + 4) * FUSED_TR_PITCH));
        const s16x8 bt = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        dq = E::mma16(kt[ks][s2], bt, dq);
      }
    }
  };

  issue(0);
  commit(0, 0);
  if (QT < N) issue(QT);
  __syncthreads();
  for (int q0 = 0, cur = 0; q0 < N; q0 += QT, cur ^= 1) {
    const int nsub = min(8, (N - q0 + 31) / 32);      // (uniform) query sub-tiles of this tile that hold queries
#pragma unroll 1
    for (int sub = 0; sub < nsub; ++sub) {
      f32x16 dq = (f32x16)(0.f);                      // dQ^T of the sub-tile (rows 0..7 = d in registers 0..3, h), this wave's keys
      step(cur, sub, dq);
      *(f32x4*)(dq_mine + (sub * 32 + r) * 8 + 4 * h) = (f32x4){dq[0], dq[1], dq[2], dq[3]};      // -> this wave's private LDS block
    }
    __syncthreads();                                  // every wave's block is complete (and nobody reads staging buffer `cur` any more)
    {   // the tile's dQ summed over the 8 waves in order: 512 threads x 16 bytes
      const int query = q0 + (tid >> 1), half = tid & 1;
      if (query < N) {
        f32x4 v = (f32x4)(0.f);
#pragma unroll
        for (int w = 0; w < 8; ++w) v += *(const f32x4*)((const float*)(lds + DQ_OFF) + w * (QT * 8) + tid * 4);
        if (nkb == 1) {
          const float sc = 0.35355339059327373f;      // d(scale * q.k)/dq
          store4((T*)a.dqkv + ((size_t)b * N + query) * (3 * C) + head * 8 + 4 * half, v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc);
        } else {
          *(f32x4*)(slab + (((size_t)kb * a.B * a.heads + (size_t)b * a.heads + head) * N + query) * 8 + 4 * half) = v;
        }
      }
    }
    if (q0 + QT < N) {
      commit(cur ^ 1, q0 + QT);
      if (q0 + 2 * QT < N) issue(q0 + 2 * QT);
    }
    __syncthreads();
  }
  const float ln2 = 0.6931471805599453f;              // Q^T carried scale * log2(e); dK = scale * dS^T q
#pragma unroll
  for (int ks = 0; ks < NSUB; ++ks) {
    const int key = kb * KPB + (wave * NSUB + ks) * 32 + r;
    if (key < N) {
      T* dst = (T*)a.dqkv + ((size_t)b * N + key) * (3 * C) + head * 8 + 4 * h;
      store4(dst + C, dk[ks][0] * ln2, dk[ks][1] * ln2, dk[ks][2] * ln2, dk[ks][3] * ln2);
      store4(dst + 2 * C, dv[ks][0], dv[ks][1], dv[ks][2], dv[ks][3]);
    }
  }
}

// dq = scale * sum over the key blocks (in order) of the one-pass kernel's partial dQ; one thread per (sample, token, head)
template <typename T>
__global__ __launch_bounds__(256) void attn_dq_reduce_kernel(const float* __restrict__ slab, int nkb, int B, int heads, int N, T* __restrict__ dqkv) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // (b * heads + head) * N + n
  if (i >= (size_t)B * heads * N) return;
  const int n = (int)(i % N), head = (int)((i / N) % heads), b = (int)(i / ((size_t)N * heads));
  const size_t per = (size_t)B * heads * N * 8;
  f32x4 lo = (f32x4)(0.f), hi = (f32x4)(0.f);
  for (int k = 0; k < nkb; ++k) {
    lo += *(const f32x4*)(slab + k * per + i * 8);
    hi += *(const f32x4*)(slab + k * per + i * 8 + 4);
  }
  const float sc = 0.35355339059327373f;
  T* dst = dqkv + ((size_t)b * N + n) * (3 * heads * 8) + head * 8;
  store4(dst, lo[0] * sc, lo[1] * sc, lo[2] * sc, lo[3] * sc);
  store4(dst + 4, hi[0] * sc, hi[1] * sc, hi[2] * sc, hi[3] * sc);
}

}  // namespace pd

extern "C" int pd_attn_d8(const pd_attn_args* a, void* stream) {
  using namespace pd;
  PD_CHECK(a != nullptr, PD_ERR_ARG, "pd_attn_d8: null args");
  PD_CHECK(a->B > 0 && a->heads > 0 && a->N > 0, PD_ERR_SHAPE, "pd_attn_d8: bad shape");
  PD_CHECK(a->q && a->k && a->v && a->out, PD_ERR_ARG, "pd_attn_d8: null pointer");
  // Query blocks per wave.  QB = 2 (K / V^T fragments shared by two independent chains, 2 waves/SIMD) measured 3 % slower
  // than QB = 1 (4 waves/SIMD) at N = 4096: the tile time is pinned by the exp + MFMA issue mix (scripts/micro/), not by
  // LDS traffic or occupancy.  Kept selectable for future shapes.
#ifdef PD_ATTN_QB2
  const int qbw = (a->N >= 2048) ? 2 : 1;
#else
  const int qbw = 1;
#endif
  // 8-wave workgroups (256 queries share each staged tile) once there are enough query blocks to fill the chip with them
  const bool wpb4_only = diag_env("PD_ATTN_WPB4", 0) != 0;      // diagnostic: same-box A/B
  const bool wide = !wpb4_only && qbw == 1 && a->N >= 1024 && (long long)(a->N / 256) * a->heads * a->B >= 1024;
  const int qpb = wide ? 256 : 128 * qbw;
  PD_CHECK((long long)((a->N + qpb - 1) / qpb) * a->heads * a->B < (1ll << 31), PD_ERR_SHAPE, "pd_attn_d8: grid too large");
  dim3 grid(((a->N + qpb - 1) / qpb) * a->heads * a->B);
  hipStream_t st = (hipStream_t)stream;
  // diagnostic (scripts/bench_concurrent.py): extra dynamic LDS per workgroup caps the kernel's occupancy, so that another
  // stream's MFMA-bound convolutions can co-reside on every CU beside this VALU-bound kernel
  static const int lds_pad = getenv("PD_ATTN_LDS_PAD") ? atoi(getenv("PD_ATTN_LDS_PAD")) : 0;
  if (lds_pad > 0 && a->dtype == PD_BF16 && wide) {
    static LdsAttr attr_a, attr_b;
    (void)ensure_lds(attr_a, attn_kernel<bf16_t, 1, 8>, lds_pad);
    (void)ensure_lds(attr_b, attn_glds_kernel<bf16_t>, lds_pad);
    if (a->kmax2) hipLaunchKernelGGL((attn_glds_kernel<bf16_t>), grid, dim3(512), lds_pad, st, *a);     // the shipped kernel, occupancy-capped
    else hipLaunchKernelGGL((attn_kernel<bf16_t, 1, 8>), grid, dim3(512), lds_pad, st, *a);
    PD_LAUNCH_CHECK();
    return PD_OK;
  }
  const bool no_glds = diag_env("PD_ATTN_NO_GLDS", 0) != 0;      // diagnostic: same-box A/B
  if (a->kmax2 && wide && !no_glds && a->dtype != PD_F32) {
    if (a->dtype == PD_BF16) hipLaunchKernelGGL((attn_glds_kernel<bf16_t>), grid, dim3(512), 0, st, *a);
    else if (a->dtype == PD_F16) {
      hipLaunchKernelGGL((attn_glds_kernel<half_t>), grid, dim3(512), 0, st, *a);
    } else { set_error("pd_attn_d8: bad dtype"); return PD_ERR_ARG; }
    PD_LAUNCH_CHECK();
    return PD_OK;
  }
  if (a->dtype == PD_F32) {
    hipLaunchKernelGGL((attn_kernel<float, 1, 4>), dim3(((a->N + 127) / 128) * a->heads * a->B), dim3(256), 0, st, *a);
  } else if (a->dtype == PD_BF16) {
    if (qbw == 2) hipLaunchKernelGGL((attn_kernel<bf16_t, 2, 4>), grid, dim3(256), 0, st, *a);
    else if (wide) hipLaunchKernelGGL((attn_kernel<bf16_t, 1, 8>), grid, dim3(512), 0, st, *a);
    else hipLaunchKernelGGL((attn_kernel<bf16_t, 1, 4>), grid, dim3(256), 0, st, *a);
  } else if (a->dtype == PD_F16) {
    if (qbw == 2) hipLaunchKernelGGL((attn_kernel<half_t, 2, 4>), grid, dim3(256), 0, st, *a);
    else if (wide) hipLaunchKernelGGL((attn_kernel<half_t, 1, 8>), grid, dim3(512), 0, st, *a);
    else hipLaunchKernelGGL((attn_kernel<half_t, 1, 4>), grid, dim3(256), 0, st, *a);
  } else { set_error("pd_attn_d8: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}


// one-pass form: 8 waves x 2 sub-tiles of 32 keys per workgroup
static constexpr int FUSED_KPB = 512;
extern "C" size_t pd_attn_d8_bwd_workspace(const pd_attn_bwd_args* a) {
  if (!a || a->dtype == PD_F32 || a->B < 1 || a->heads < 1 || a->N < FUSED_KPB) return 0;
  const size_t nkb = ((size_t)a->N + FUSED_KPB - 1) / FUSED_KPB;
  return nkb * (size_t)a->B * a->heads * a->N * 8 * sizeof(float);      // (one key block writes dq directly, but a uniform answer keeps callers simple)
}

template <typename T>
static int launch_attn_bwd_fused(const pd_attn_bwd_args* a, hipStream_t st) {
  using namespace pd;
  constexpr int LDS = 4 * (2 * 256 * 16) + 4 * (9 * AttnOps<T>::VT_PITCH) + 8 * 2 * FUSED_TR_IMG + 8 * 256 * 8 * 4;
  auto kern = attn_bwd_fused_kernel<T, 2>;
  static LdsAttr attr;
  if (!ensure_lds(attr, kern, LDS)) {
    set_error("pd_attn_d8_bwd: cannot reserve %d bytes of LDS", LDS);
    return PD_ERR_LAUNCH;             // (pd_attn_d8_bwd falls back to the two-kernel path)
  }
  const int nkb = (a->N + FUSED_KPB - 1) / FUSED_KPB;
  PD_CHECK((long long)nkb * a->heads * a->B < (1ll << 31), PD_ERR_SHAPE, "pd_attn_d8_bwd: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)(nkb * a->heads * a->B)), dim3(512), LDS, st, *a, a->slab, nkb);
  PD_LAUNCH_CHECK();
  if (nkb > 1) {
    const size_t rows = (size_t)a->B * a->heads * a->N;
    hipLaunchKernelGGL(attn_dq_reduce_kernel<T>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, (const float*)a->slab, nkb, a->B, a->heads, a->N,
                       (T*)a->dqkv);
    PD_LAUNCH_CHECK();
  }
  return PD_OK;
}

extern "C" int pd_attn_d8_bwd(const pd_attn_bwd_args* a, void* stream) {
  using namespace pd;
  PD_CHECK(a != nullptr, PD_ERR_ARG, "pd_attn_d8_bwd: null args");
  PD_CHECK(a->B > 0 && a->heads > 0 && a->N > 0, PD_ERR_SHAPE, "pd_attn_d8_bwd: bad shape");
  PD_CHECK(a->q && a->k && a->v && a->o && a->dout && a->lse && a->delta && a->dqkv, PD_ERR_ARG, "pd_attn_d8_bwd: null pointer");
  // one pass (round 5) when the caller brought the workspace.  PD_ATTN_BWD_FUSED=0: diagnostic override (same-box A/B)
  if (a->slab != nullptr && diag_env("PD_ATTN_BWD_FUSED", 1) != 0) {
    const size_t need = pd_attn_d8_bwd_workspace(a);
    if (need > 0 && a->slab_bytes >= need && (a->dtype == PD_BF16 || a->dtype == PD_F16)) {
      const int rc = a->dtype == PD_BF16 ? launch_attn_bwd_fused<bf16_t>(a, (hipStream_t)stream) : launch_attn_bwd_fused<half_t>(a, (hipStream_t)stream);
      if (rc != PD_ERR_LAUNCH) return rc;     // the 154 KB of dynamic LDS could not be reserved on this device: the two kernels below
    }
  }
  PD_CHECK((long long)((a->N + 127) / 128) * a->heads * a->B < (1ll << 31), PD_ERR_SHAPE, "pd_attn_d8_bwd: grid too large");
  const dim3 grid(((a->N + 127) / 128) * a->heads * a->B);
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == PD_F32) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<float>, grid, dim3(256), 0, st, *a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<float>, grid, dim3(256), 0, st, *a);
  } else if (a->dtype == PD_BF16) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<bf16_t>, grid, dim3(256), 0, st, *a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<bf16_t>, grid, dim3(256), 0, st, *a);
  } else if (a->dtype == PD_F16) {      // fp16 training (round 5): dS = p (dP - delta) carries the loss scale; p <= 1, no overflow risk of its own
    hipLaunchKernelGGL(attn_bwd_dq_kernel<half_t>, grid, dim3(256), 0, st, *a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<half_t>, grid, dim3(256), 0, st, *a);
  } else { set_error("pd_attn_d8_bwd: bad dtype"); return PD_ERR_ARG; }
  PD_LAUNCH_CHECK();
  return PD_OK;
}
