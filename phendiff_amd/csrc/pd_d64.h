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
  static __device__ __forceinline__ Frag pack_p(const f32x16& p, int s) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = Pack16<T>::pack(p[8 * s + 2 * j], p[8 * s + 2 * j + 1]);
    Frag f; f.v = __builtin_bit_cast(s16x8, (u32x4){w[0], w[1], w[2], w[3]});
    return f;
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
