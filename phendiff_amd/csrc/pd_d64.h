// LDS layouts and MFMA operand helpers shared by the head_dim-64 attention kernels (forward: sd_kernels.hip, backward:
// sd_bwd_kernels.hip).  A [rows][64] tile sits row-major in LDS; `load_vt` reads the A operand X^T (32 d x 16 rows) of an
// MFMA whose B operand is a score tile packed by `pack_p` (row order = the accumulator register order of a 32x32 tile).
#pragma once
#include "pd_common.h"
#include "pd_stage.h"

namespace pd {

template <typename T> struct D64 {          // primary: 16-bit element types (bf16, fp16); fp32 below
  static constexpr int KP = 128 + 16, VP = 128 + 64;      // row pitches: conflict-free ds_read_b128 rows / 4-row transposed blocks
  typedef typename Elem<T>::Frag Frag;
  static __device__ __forceinline__ int vt_lane_off(int lane) {      // block row q <-> key 4h + q, columns 16*cg + 4*pp of a 32-d row tile
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    return (4 * (g >> 1) + q) * VP + (16 * (g & 1) + 4 * pp) * 2;
  }
  // A fragment of V^T for k-step s (16 keys) of a 32-key sub-tile: element j <-> key 16s + 8(j>>2) + 4h + (j&3) (P's register order)
  static __device__ __forceinline__ Frag load_vt(const unsigned char* base) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s* lp;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + 8 * VP));
    Frag f; f.v = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
  }
  // (the same from two explicit addresses: the swizzled backward tiles below)
  static __device__ __forceinline__ Frag load_vt2(const unsigned char* lo_, const unsigned char* hi_) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s* lp;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lo_));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(hi_));
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
// Tiles of the BACKWARD kernels (round 6).  There every [64 rows][64 d] tile is read both ways -- as row fragments (ds_read_b128, 16 rows per
// pass: needs a pitch of 4 x odd dwords) and as transposed 4-row blocks (ds_read_b64_tr_b16, 4 rows x 64 B per half-wave: needs a pitch of
// 16 x odd dwords) -- and no single pitch serves both: at VP = 192 B the row reads were 4-way bank-conflicted (PMC: 252 M conflict cycles per
// launch in BOTH backward kernels = 46-60 % of their LDS-active cycles, the LDS pipe 70 % busy; the forward, with separate K / V pitches: 0).
// 16-bit engines: 128-byte rows, the 16-byte slot index XOR-ed with a key of the row -- key(row) = (t0 << 2) | (t2 << 1) | t1 of t = (row >> 1) & 7:
// eight same-parity rows of any aligned 16 get eight different slots (row reads), rows R and R + 2 of an aligned 4 get different slot HALVES
// (transposed reads).  Brute-forced over both access patterns: conflict-free.  fp32: the padded pitch as before.
template <typename T> struct D64B {
  static constexpr bool SW = sizeof(T) == 2;
  static constexpr int ES = Elem<T>::BYTES;
  static constexpr int P = SW ? 128 : D64<T>::VP;
  typedef typename Elem<T>::Frag Frag;
  static __device__ __forceinline__ int key(int row) { const int t = (row >> 1) & 7; return ((t & 1) << 2) | ((t >> 2) << 1) | ((t >> 1) & 1); }
  // staging store of piece (row, 8-element sub-block `sub`)
  static __device__ __forceinline__ int store_off(int row, int sub) { return SW ? row * P + ((sub ^ key(row)) << 4) : row * P + sub * 8 * ES; }
  // row fragment of lane (r, h): d = 16 ks + 8 h + (0..7) of row r (+ a multiple of 16 rows): off(ks) = SW ? base ^ (32 ks) : base + 16 ks ES
  static __device__ __forceinline__ int row_base(int r, int h) { return SW ? r * P + ((h ^ key(r)) << 4) : r * P + 8 * h * ES; }
  static __device__ __forceinline__ int row_off(int base, int ks) { return SW ? (base ^ (ks << 5)) : base + ks * 16 * ES; }
  // transposed fragment (see D64::load_vt): per lane FOUR offsets [8-row half][32-d half] (added to tile + 16-row-aligned base row * P)
  struct VtOff { int o[2][2]; };
  static __device__ __forceinline__ VtOff vt_off(int lane) {
    VtOff v;
    if constexpr (SW) {
      const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
      for (int hi = 0; hi < 2; ++hi)
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
          const int row = 4 * (g >> 1) + q + 8 * hi, slot = 2 * (g & 1) + (pp >> 1) + 4 * dh;
          v.o[hi][dh] = row * P + ((slot ^ key(row)) << 4) + 8 * (pp & 1);
        }
    } else {
      const int t = D64<T>::vt_lane_off(lane);
#pragma unroll
      for (int hi = 0; hi < 2; ++hi)
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) v.o[hi][dh] = t + dh * 32 * ES;       // (fp32: load_vt adds the row steps itself)
    }
    return v;
  }
  static __device__ __forceinline__ Frag load_vt(const unsigned char* tile_rows, const VtOff& v, int dh) {
    if constexpr (SW) return D64<T>::load_vt2(tile_rows + v.o[0][dh], tile_rows + v.o[1][dh]);
    else return D64<T>::load_vt(tile_rows + v.o[0][dh]);
  }
};

template <> struct D64<float> {
  static constexpr int KP = 256 + 16, VP = 256 + 16;
  typedef Elem<float>::Frag Frag;
  static __device__ __forceinline__ int vt_lane_off(int lane) { return (4 * (lane >> 5)) * VP + (lane & 31) * 4; }   // key 4h, d = r
  static __device__ __forceinline__ Frag load_vt(const unsigned char* base) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.lo[j] = *(const float*)(base + j * VP); f.hi[j] = *(const float*)(base + (8 + j) * VP); }
    return f;
  }
  static __device__ __forceinline__ Frag pack_p(const f32x16& p, int s) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.lo[j] = p[8 * s + j]; f.hi[j] = p[8 * s + 4 + j]; }
    return f;
  }
};

}  // namespace pd
