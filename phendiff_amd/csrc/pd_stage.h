// Global -> register -> LDS staging pieces shared by the forward implicit-GEMM conv (conv_igemm.hip) and the
// weight-gradient kernel (wgrad.hip): 8-channel pieces loaded with buffer loads (out-of-range offset -> zeros) and written
// to LDS through the GroupNorm affine (+ SiLU) the consuming convolution sees.
#pragma once
#include <type_traits>
#include "pd_common.h"

namespace pd {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB_OFF = 0xC0000000u;   // > any tensor we accept (< 2 GiB): buffer loads return 0

// primary template: the 16-bit element types (bf16, fp16); fp32 is the specialisation below
template <typename T> struct Stage {
  struct R { u32x4 v; };
  static __device__ __forceinline__ uint32_t word(const R& r, int j) { return r.v[j]; }      // channel pair j of the piece
  static __device__ __forceinline__ u32x4 raw(const R& r) { return r.v; }
  static __device__ __forceinline__ R load(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    R r; r.v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); return r;
  }
  // silu(x * sc + sh) of one 8-channel piece, unconditionally: no branch, no store (conv_kernel's compile-time prologue form calls it directly).
  // fp16 engine: the GroupNorm affine in fp32 straight from the packed halves, y kept as a packed fp16 pair (v_fma_mixlo / mixhi_f16:
  // unpack and re-pack are free); the sigmoid's exponent argument, the exponential and the reciprocal in fp32 (v_fma_mix_f32 reads
  // the fp16 halves directly); y * sigmoid back to packed fp16 by v_fma_mixlo / mixhi_f16.  A first version ran the whole SiLU on
  // the packed halves (v_pk_mul_f16, v_exp_f16, v_rcp_f16): same instruction time (the 16-bit transcendentals issue at the fp32
  // rate and need a v_pack_b32_f16 per pair) but the fp16 exponent argument raised the UNet's error from 1.5e-3 to 2.0e-3.
  static __device__ __forceinline__ u32x4 xform_gs(const R& in, const float (&sc)[8], const float (&sh)[8]) {
    u32x4 o;
    if constexpr (std::is_same<T, half_t>::value) {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t wj = in.v[j];          // (a bit_cast straight from the vector ELEMENT reads element 0 for every j: hipcc 7.2)
        const h2 x = __builtin_bit_cast(h2, wj);
        h2 y;
        y.x = (_Float16)__builtin_fmaf((float)x.x, sc[2 * j], sh[2 * j]);
        y.y = (_Float16)__builtin_fmaf((float)x.y, sc[2 * j + 1], sh[2 * j + 1]);
        const float e0 = __builtin_amdgcn_exp2f((float)y.x * -1.4426950408889634f), e1 = __builtin_amdgcn_exp2f((float)y.y * -1.4426950408889634f);
        const float r0 = __builtin_amdgcn_rcpf(1.0f + e0), r1 = __builtin_amdgcn_rcpf(1.0f + e1);
        h2 o2;
        o2.x = (_Float16)((float)y.x * r0);
        o2.y = (_Float16)((float)y.y * r1);
        o[j] = __builtin_bit_cast(uint32_t, o2);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 x;                                     // packed fp32 math: one VALU slot per channel pair
        float xl, xh;
        const uint32_t wj = in.v[j];
        Pack16<T>::unpack(wj, xl, xh);
        x.x = xl; x.y = xh;
        x = silu_fast2(x * (f32x2){sc[2 * j], sc[2 * j + 1]} + (f32x2){sh[2 * j], sh[2 * j + 1]});
        o[j] = Pack16<T>::pack(x.x, x.y);
      }
    }
    return o;
  }
  // y = silu?(x*sc + sh) on 8 packed 16-bit values, zeroed when !valid.  `affine` / `silu` are kernel-uniform: the two common
  // forms (GroupNorm + SiLU; plain copy) are whole separate paths -- written as per-element `if`s the compiler turned both flags
  // into two v_cndmask per element (16 of a piece's ~70 VALU instructions, round-3 ISA reading)
  template <bool ZERO = true>
  static __device__ __forceinline__ void xform_store(unsigned char* dst, const R& in, const float (&sc)[8], const float (&sh)[8],
                                                     bool affine, bool silu, bool valid) {
    u32x4 o = in.v;
    if (affine && silu) {
      o = xform_gs(in, sc, sh);
      if (ZERO && !valid) o = (u32x4)(0u);
    } else if (affine || silu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 x;
        float xl, xh;
        Pack16<T>::unpack(in.v[j], xl, xh);
        x.x = xl; x.y = xh;
        if (affine) x = x * (f32x2){sc[2 * j], sc[2 * j + 1]} + (f32x2){sh[2 * j], sh[2 * j + 1]};
        if (silu) x = silu_fast2(x);
        o[j] = Pack16<T>::pack(x.x, x.y);
      }
      if (ZERO && !valid) o = (u32x4)(0u);
    }
    *(u32x4*)dst = o;
  }
};
template <> struct Stage<float> {
  struct R { u32x4 a, b; };
  static __device__ __forceinline__ uint32_t word(const R& r, int j) { return r.a[j]; }      // (16-bit engines only; never called)
  static __device__ __forceinline__ u32x4 raw(const R& r) { return r.a; }
  static __device__ __forceinline__ R load(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    R r; r.a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); r.b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
    return r;
  }
  template <bool ZERO = true>
  static __device__ __forceinline__ void xform_store(unsigned char* dst, const R& in, const float (&sc)[8], const float (&sh)[8],
                                                     bool affine, bool silu, bool valid) {
    u32x4 oa = in.a, ob = in.b;
    if (affine || silu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = __uint_as_float(in.a[j]), y = __uint_as_float(in.b[j]);
        if (affine) { x = x * sc[j] + sh[j]; y = y * sc[4 + j] + sh[4 + j]; }
        if (silu) { x = silu_f(x); y = silu_f(y); }   // parity mode: accurate division
        oa[j] = __float_as_uint(x); ob[j] = __float_as_uint(y);
      }
      if (ZERO && !valid) { oa = (u32x4)(0u); ob = (u32x4)(0u); }
    }
    *(u32x4*)dst = oa; *((u32x4*)dst + 1) = ob;
  }
};


// ---- MFMA operand fragments whose 8 elements are 8 consecutive ROWS of an LDS image at one column (a transposed read):
// bf16 via ds_read_b64_tr_b16 (lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3; lane i receives column i),
// fp32 (validation) via eight ds_read_b32.  Used by the weight-gradient GEMM (K = pixels) and the d = 64 attention (V^T).
template <typename T> struct FragLd {       // primary: 16-bit element types
  // lane base address: pixel 8h+q, channels 16*cg + 4*pp (see file header); second read 4 pixels further
  static __device__ __forceinline__ unsigned lane_off(int lane, int pitch) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    return (unsigned)((8 * (g >> 1) + q) * pitch + (16 * (g & 1) + 4 * pp) * 2);
  }
  template <int PITCH>
  static __device__ __forceinline__ typename Elem<T>::Frag load(const unsigned char* base) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s* lp;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + 4 * PITCH));
    typename Elem<T>::Frag f;
    f.v = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
  }
};
template <> struct FragLd<float> {
  static __device__ __forceinline__ unsigned lane_off(int lane, int pitch) {
    return (unsigned)(8 * (lane >> 5) * pitch + (lane & 31) * 4);
  }
  template <int PITCH>
  static __device__ __forceinline__ Elem<float>::Frag load(const unsigned char* base) {
    Elem<float>::Frag f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.lo[j] = *(const float*)(base + j * PITCH); f.hi[j] = *(const float*)(base + (4 + j) * PITCH); }
    return f;
  }
};


}  // namespace pd
