// Shared device helpers for libphendiff_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/phendiff_hip.h"

namespace pd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;  // raw bf16 bits
struct half_t { unsigned short bits; };   // raw IEEE fp16 bits: a distinct type, so templates can tell the two 16-bit formats apart

void set_error(const char* fmt, ...);

// Kernel-selecting diagnostic switch, read from the environment at EVERY dispatch (a linear scan of environ: ~0.1 us): one process can
// capture the same trajectory under several switch settings and replay them in alternation (scripts/ab_cells.py: the same-box,
// same-thermal-state A/B the whole-workload decisions are made by).  The driver's run carries none of them (bench.py diagnostic_env).
static inline int diag_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

#define PD_CHECK(cond, code, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      pd::set_error(__VA_ARGS__);        \
      return (code);                     \
    }                                    \
  } while (0)

// A dynamic LDS size above 64 KiB needs hipFuncAttributeMaxDynamicSharedMemorySize, and that attribute belongs to the function ON ONE
// DEVICE: one flag per (call site, device), so a process that drives several GPUs sets it on each of them (a function-local
// `static bool` covered the first device only).  Benign race between host threads: they write the same value.
struct LdsAttr { bool done[16] = {}; };
template <typename K> static inline bool ensure_lds(LdsAttr& f, K kern, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
  if (dev >= 0 && f.done[dev]) return true;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  if (dev >= 0) f.done[dev] = true;
  return true;
}

#define PD_LAUNCH_CHECK()                                              \
  do {                                                                 \
    hipError_t e_ = hipGetLastError();                                 \
    if (e_ != hipSuccess) {                                            \
      pd::set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return PD_ERR_LAUNCH;                                            \
    }                                                                  \
  } while (0)

// ---- bf16 <-> f32 ------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN.
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

// ---- two 16-bit storage values in one dword <-> two fp32 (round-to-nearest-even), for either 16-bit element type
template <typename T> struct Pack16;
template <> struct Pack16<bf16_t> {
  static __device__ __forceinline__ void unpack(uint32_t w, float& lo, float& hi) {
    lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);
  }
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) { return pack2bf(lo, hi); }
};
template <> struct Pack16<half_t> {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ void unpack(uint32_t w, float& lo, float& hi) {
    const h2 v = __builtin_bit_cast(h2, w);
    lo = (float)v.x; hi = (float)v.y;
  }
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
    const h2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
  }
};
template <> struct Pack16<float> {     // (lets code shared with the fp32 engine compile; never called)
  static __device__ __forceinline__ void unpack(uint32_t w, float& lo, float& hi) { lo = hi = __uint_as_float(w); }
  static __device__ __forceinline__ uint32_t pack(float lo, float) { return __float_as_uint(lo); }
};
__device__ __forceinline__ float h2f(half_t v) { return (float)__builtin_bit_cast(_Float16, v.bits); }
__device__ __forceinline__ half_t f2h(float f) { half_t h; h.bits = __builtin_bit_cast(unsigned short, (_Float16)f); return h; }

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
// throughput form: v_exp_f32 + v_rcp_f32 (1 ulp each), 5 VALU ops instead of ~15 for the IEEE division
__device__ __forceinline__ float silu_fast(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}

// two lanes of fp32 per instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 run at full rate on CDNA: half the VALU issue
// slots of the scalar form); the transcendentals stay scalar
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 silu_fast2(f32x2 v) {
  const f32x2 t = v * (f32x2)(-1.4426950408889634f);
  f32x2 e; e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  const f32x2 d = e + (f32x2)(1.0f);
  f32x2 r; r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
  return v * r;
}

// ---- element traits: an "8-element fragment" is what one lane feeds to one MFMA k16-step ----------
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int BYTES = 4;
  struct Frag { f32x4 lo, hi; };   // 8 fp32
  static __device__ __forceinline__ void unpack(const Frag& f, float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = f.lo[i]; o[4 + i] = f.hi[i]; }
  }
  static __device__ __forceinline__ Frag pack(const float (&o)[8]) {
    Frag f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { f.lo[i] = o[i]; f.hi[i] = o[4 + i]; }
    return f;
  }
  static __device__ __forceinline__ Frag zero() { Frag f; f.lo = (f32x4)(0.f); f.hi = (f32x4)(0.f); return f; }
  static __device__ __forceinline__ Frag load(const void* p) {
    Frag f; f.lo = *(const f32x4*)p; f.hi = *((const f32x4*)p + 1); return f;
  }
  // fragment through a buffer resource: per-lane byte offset in a VGPR, the wave-uniform part in an SGPR
  static __device__ __forceinline__ Frag load_buf(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0), b = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16, soff, 0);
    Frag f; f.lo = __builtin_bit_cast(f32x4, a); f.hi = __builtin_bit_cast(f32x4, b); return f;
  }
  static __device__ __forceinline__ void store(void* p, const Frag& f) { *(f32x4*)p = f.lo; *((f32x4*)p + 1) = f.hi; }
  // D[32x32] += A[32 x 8k] * B[8k x 32]; exact fp32 (v_mfma_f32_32x32x2_f32 x4, k order consistent in A and B)
  static __device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b, f32x16 c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[i], b.lo[i], c, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[i], b.hi[i], c, 0, 0, 0);
    return c;
  }
  static __device__ __forceinline__ f32x4 mma_16x16x32(const Frag&, const Frag&, f32x4 c) { return c; }   // (16-bit engines only; never called)
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int BYTES = 2;
  struct Frag { s16x8 v; };        // 8 bf16
  static __device__ __forceinline__ void unpack(const Frag& f, float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bf2f((bf16_t)f.v[i]);
  }
  static __device__ __forceinline__ Frag pack(const float (&o)[8]) {
    Frag f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f.v[i] = (short)f2bf(o[i]);
    return f;
  }
  static __device__ __forceinline__ Frag zero() { Frag f; f.v = (s16x8)(0); return f; }
  static __device__ __forceinline__ Frag load(const void* p) { Frag f; f.v = *(const s16x8*)p; return f; }
  static __device__ __forceinline__ Frag load_buf(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    Frag f; f.v = __builtin_bit_cast(s16x8, (u4)__builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0)); return f;
  }
  static __device__ __forceinline__ void store(void* p, const Frag& f) { *(s16x8*)p = f.v; }
  static __device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b, f32x16 c) { return mma16(a.v, b.v, c); }
  static __device__ __forceinline__ f32x16 mma16(s16x8 a, s16x8 b, f32x16 c) {      // D += A[32 x 16] . B[16 x 32] on raw 16-bit lanes
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  // D[16 x 16] += A[16 x 32] . B[32 x 16]: lane (i = lane % 16, g = lane / 16) holds row / column i, k = 8 g .. 8 g + 7; D: column i, rows 4 g .. 4 g + 3
  static __device__ __forceinline__ f32x4 mma_16x16x32(const Frag& a, const Frag& b, f32x4 c) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
  static __device__ __forceinline__ bf16_t from_f(float v) { return f2bf(v); }
};

template <> struct Elem<half_t> {
  static constexpr int BYTES = 2;
  struct Frag { s16x8 v; };        // 8 fp16
  static __device__ __forceinline__ void unpack(const Frag& f, float (&o)[8]) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const h8 h = __builtin_bit_cast(h8, f.v);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)h[i];
  }
  static __device__ __forceinline__ Frag pack(const float (&o)[8]) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (_Float16)o[i];
    Frag f; f.v = __builtin_bit_cast(s16x8, h);
    return f;
  }
  static __device__ __forceinline__ Frag zero() { Frag f; f.v = (s16x8)(0); return f; }
  static __device__ __forceinline__ Frag load(const void* p) { Frag f; f.v = *(const s16x8*)p; return f; }
  static __device__ __forceinline__ Frag load_buf(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    Frag f; f.v = __builtin_bit_cast(s16x8, (u4)__builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0)); return f;
  }
  static __device__ __forceinline__ void store(void* p, const Frag& f) { *(s16x8*)p = f.v; }
  static __device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b, f32x16 c) { return mma16(a.v, b.v, c); }
  static __device__ __forceinline__ f32x16 mma16(s16x8 a, s16x8 b, f32x16 c) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mma_16x16x32(const Frag& a, const Frag& b, f32x4 c) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a.v), __builtin_bit_cast(h8, b.v), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f(half_t v) { return h2f(v); }
  static __device__ __forceinline__ half_t from_f(float v) { return f2h(v); }
};

// XCD-aware work order (speed only: MI355X_MICROARCH "Workgroup dispatch": blocks b, b + 8, b + 16, ... share an XCD and its 4-MiB L2).  Returns the
// position in a work list of `total` items such that XCD label b % 8 takes a CONTIGUOUS run of the list -- bijective for any total -- in
// dispatch order: neighbours in the list (which share operand panels) then run on one L2 instead of on eight.
__device__ __forceinline__ int xcd_chunk_index(int b, int total) {
  const int x = b & 7, j = b >> 3, q = total >> 3, r = total & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// store 4 consecutive output elements
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  *(f32x4*)p = (f32x4){a, b, c, d};
}
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
  uint2 v; v.x = pack2bf(a, b); v.y = pack2bf(c, d);
  *(uint2*)p = v;
}
__device__ __forceinline__ void store4(half_t* p, float a, float b, float c, float d) {
  uint2 v; v.x = Pack16<half_t>::pack(a, b); v.y = Pack16<half_t>::pack(c, d);
  *(uint2*)p = v;
}
__device__ __forceinline__ void load4(const half_t* p, float (&o)[4]) {
  const uint2 v = *(const uint2*)p;
  Pack16<half_t>::unpack(v.x, o[0], o[1]); Pack16<half_t>::unpack(v.y, o[2], o[3]);
}
__device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
  f32x4 v = *(const f32x4*)p; o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&o)[4]) {
  uint2 v = *(const uint2*)p;
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}

}  // namespace pd
