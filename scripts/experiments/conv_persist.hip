// pd_conv, persistent form: 3x3 stride-1 NHWC convolutions on outputs >= 32 pixels wide (the bulk of every UNet here).
//
// Same implicit GEMM as conv_igemm.hip (D[co][pixel] = sum_k W[co][k] X[k][pixel], 8 x 32-pixel x 64-co tiles, 32-channel K chunks
// whose halo tile is staged once through LDS and reused by the 9 taps), but organised around what bounded that kernel:
//   * a workgroup there lived for one tile -- load, wait, transform, barrier, 2..16 chunks of MFMA, epilogue -- so the first
//     activation load of every tile was exposed, only ~24 KB per workgroup were ever in flight, and the 64-channel 256^2 layers
//     ran at 2.2-3.6 TB/s of algorithmic traffic with the matrix pipe 1/3 busy;
//   * activation loads and weight-fragment loads shared one in-order vmcnt queue: every weight wait behind a freshly issued
//     activation load also waited for that load.
// Here 2 x CUs PERSISTENT workgroups of 8 waves walk the (pixel tile, co tile) items, and the waves have two ROLES:
//   waves 4-7  LOADERS  global -> registers -> GroupNorm affine + SiLU + zero padding -> LDS, always two chunks ahead in
//              registers (two sets, ~48 KB per workgroup in flight) and one chunk ahead in LDS, across item boundaries: the next
//              tile's first chunk is transformed while the current tile's last chunk multiplies and its epilogue stores;
//   waves 0-3  MFMA     weight fragments straight from L2 (their own vmcnt queue: no activation load in it), activation
//              fragments from LDS, 4 MFMAs per k-step; epilogue per wave through a private 2.5 KB LDS patch (accumulators ->
//              [pixel][32 co] -> 16-byte coalesced residual loads / stores + GroupNorm statistics), no workgroup barrier in it.
// One s_barrier per chunk joins the roles (LDS slot s&1 written during chunk s-1, read during chunk s).
// GroupNorm statistics: one (sum, sumsq) row per wave half-tile (128 pixels), i.e. 2 statistic tiles per 8 x 32 tile.
#include <type_traits>
#include <stdlib.h>
#include "pd_common.h"
#include "pd_stage.h"
#include "pd_conv.h"

namespace pd {

namespace {
constexpr int P_TH = 8, P_TW = 32;
constexpr int P_IN_TW = P_TW + 2, P_NPIX = (P_TH + 2) * P_IN_TW;        // 10 x 34 halo tile
constexpr int P_KSTEPS = 18;                                            // 9 taps x 2 k16-steps per 32-channel chunk
constexpr int P_NLOAD = 128;                                            // loader threads (waves 4-5)
constexpr int P_NIT = (P_NPIX * 4 + P_NLOAD - 1) / P_NLOAD;             // 8-channel pieces per loader thread per chunk (11)
}  // namespace

template <typename T> struct PersistCfg {
  static constexpr int BYTES = Elem<T>::BYTES;
  static constexpr int CHB = 32 * BYTES;                                // one pixel's 32-channel chunk
  static constexpr int PITCH = 2 * CHB + 16;                            // [slot 0 | slot 1 | pad]: odd number of 16-B slots
  static constexpr int TILE_BYTES = ((P_NPIX * PITCH + 15) / 16) * 16;
  static constexpr int EPF_PITCH = 32 * 4 + 16;                         // epilogue patch: [32 pixels][32 co] fp32 per MFMA wave
  static constexpr int EP_WAVE = 33 * EPF_PITCH;                        // + one row: bias + temb of the 32 channels
  static constexpr int LDS_BYTES = TILE_BYTES + 4 * EP_WAVE;
};

// `c ? a : b` with two lvalues is an lvalue: clang selects the ADDRESS and loads through it, which keeps whatever the operands
// live in (a lambda closure, the kernarg struct) in scratch memory.  rv() makes an operand a value.
template <typename V> static __device__ __forceinline__ V rv(V v) { return v; }

// item i -> (sample, tile row, tile column, co tile): the co tiles of a pixel tile are 8 items apart, i.e. on one XCD / L2 at
// about the same time (round-robin dispatch; speed only)
static __device__ __forceinline__ void persist_decode(int i, int n_pt, int nco, int tpi, int tiles_x, int& n, int& ty, int& tx, int& co_t) {
  int pt;
  if ((n_pt & 7) == 0) { const int q = i >> 3; co_t = q % nco; pt = (q / nco) * 8 + (i & 7); }
  else { co_t = i % nco; pt = i / nco; }
  n = pt / tpi;
  const int rem = pt - n * tpi;
  ty = rem / tiles_x; tx = rem - ty * tiles_x;
}

// ---- loader-wave pieces as free functions with every launch constant passed BY VALUE (no lambda closure: a closure holding
// references to the four buffer descriptors made LLVM select between closure FIELD ADDRESSES, which kept the closure, the kernarg
// struct and the register sets in scratch)
// one chunk in flight: 6 x 16-byte pieces + the chunk's GroupNorm scale / shift, ONE value per lane (lane l < 32: scale of channel l,
// lane 32 + l: shift) -- the writer fetches its 8 + 8 values with ds_bpermute instead of holding 16 registers per set
template <typename T> struct PSet { typename Stage<T>::R r[P_NIT]; float scsh; unsigned vmask; bool plain; };

// The loader's loads are INLINE ASM, outside hipcc's s_waitcnt bookkeeping, and waited for by hand (cdna_hip_programming.md 5.7
// form (ii)): the compiler-counted form could not express "wait for the older register set, leave the younger one in flight" at
// the head of the two-set loop (it emitted vmcnt(0) there: zero prefetch).  A set is ALWAYS 6 pieces + 1 scale/shift word, so
// "all but the younger set's loads" is the constant vmcnt(LOADS); no other vector-memory instruction exists in the loader role.
template <typename T> struct PAsm;
template <> struct PAsm<bf16_t> {
  static constexpr int LOADS = P_NIT + 1;
  static __device__ __forceinline__ void load(Stage<bf16_t>::R& r, __amdgpu_buffer_rsrc_t rs, unsigned off) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r.v) : "v"(off), "s"(rs) : "memory");
  }
  static __device__ __forceinline__ void wait_older(PSet<bf16_t>& S) {
    static_assert(LOADS == 12, "vmcnt literal below");
    asm volatile("s_waitcnt vmcnt(12)" : "+v"(S.r[0].v), "+v"(S.r[1].v), "+v"(S.r[2].v), "+v"(S.r[3].v), "+v"(S.r[4].v), "+v"(S.r[5].v),
                 "+v"(S.r[6].v), "+v"(S.r[7].v), "+v"(S.r[8].v), "+v"(S.r[9].v), "+v"(S.r[10].v), "+v"(S.scsh) :: "memory");
  }
};
template <> struct PAsm<float> {
  static constexpr int LOADS = 2 * P_NIT + 1;
  static __device__ __forceinline__ void load(Stage<float>::R& r, __amdgpu_buffer_rsrc_t rs, unsigned off) {
    asm volatile("buffer_load_dwordx4 %0, %2, %3, 0 offen\n\tbuffer_load_dwordx4 %1, %2, %3, 0 offen offset:16"
                 : "=&v"(r.a), "=&v"(r.b) : "v"(off), "s"(rs) : "memory");
  }
  static __device__ __forceinline__ void wait_older(PSet<float>& S) {
    static_assert(LOADS == 23, "vmcnt literal below");
    asm volatile("s_waitcnt vmcnt(23)" : "+v"(S.r[0].a), "+v"(S.r[0].b), "+v"(S.r[1].a), "+v"(S.r[1].b), "+v"(S.r[2].a), "+v"(S.r[2].b),
                 "+v"(S.r[3].a), "+v"(S.r[3].b), "+v"(S.r[4].a), "+v"(S.r[4].b), "+v"(S.r[5].a), "+v"(S.r[5].b), "+v"(S.r[6].a),
                 "+v"(S.r[6].b), "+v"(S.r[7].a), "+v"(S.r[7].b), "+v"(S.r[8].a), "+v"(S.r[8].b), "+v"(S.r[9].a), "+v"(S.r[9].b),
                 "+v"(S.r[10].a), "+v"(S.r[10].b), "+v"(S.scsh) :: "memory");
  }
};

// weight-fragment ring of the MFMA waves: inline-asm loads + hand-counted waits, like the loader's sets.  RULE for every asm load
// in this file: it is waited for (by an asm wait naming its destination) before its register can die -- an asm load left in
// flight lands in a register the compiler has meanwhile given to something else (an address: memory fault).
template <typename T> struct PRing;
#define COMMA ,
#define PD_WAITCNT_CASE(n, ops) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" : ops :: "memory")
template <> struct PRing<bf16_t> {
  using Frag = Elem<bf16_t>::Frag;
  static __device__ __forceinline__ void load(Frag& f, const bf16_t* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f.v) : "v"(ptr) : "memory");
  }
  template <int YOUNGER> static __device__ __forceinline__ void wait(Frag& f) {      // YOUNGER fragments stay in flight
    constexpr int N = YOUNGER;                                                         // one load per fragment
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(f.v) :: "memory");
    PD_WAITCNT_CASE(1, "+v"(f.v)); PD_WAITCNT_CASE(2, "+v"(f.v)); PD_WAITCNT_CASE(3, "+v"(f.v)); PD_WAITCNT_CASE(4, "+v"(f.v));
    PD_WAITCNT_CASE(5, "+v"(f.v));
    else static_assert(N < 0, "add a vmcnt literal");
  }
};
template <> struct PRing<float> {
  using Frag = Elem<float>::Frag;
  static __device__ __forceinline__ void load(Frag& f, const float* ptr) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(f.lo), "=&v"(f.hi) : "v"(ptr) : "memory");
  }
  template <int YOUNGER> static __device__ __forceinline__ void wait(Frag& f) {
    constexpr int N = 2 * YOUNGER;                                                     // two loads per fragment
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(f.lo), "+v"(f.hi) :: "memory");
    PD_WAITCNT_CASE(2, "+v"(f.lo) COMMA "+v"(f.hi)); PD_WAITCNT_CASE(4, "+v"(f.lo) COMMA "+v"(f.hi));
    PD_WAITCNT_CASE(6, "+v"(f.lo) COMMA "+v"(f.hi)); PD_WAITCNT_CASE(8, "+v"(f.lo) COMMA "+v"(f.hi));
    PD_WAITCNT_CASE(10, "+v"(f.lo) COMMA "+v"(f.hi));
    else static_assert(N < 0, "add a vmcnt literal");
  }
};

struct PLoadGeom {      // plain scalars, by value
  int G, n_pt, nco, tpi, tiles_x, pad, upsample, Hin, Win, Hc, Wc, n_main, nchunks, cin;
  unsigned C0, C1, Ct0, Ct1;
};

template <typename T, bool TAIL>
static __device__ __forceinline__ void persist_load(PSet<T>& S, int (&spix)[P_NIT], int& ld_n, int& ld_k, int& ld_chunk, int ltid,
                                                    const PLoadGeom g, __amdgpu_buffer_rsrc_t rs0, __amdgpu_buffer_rsrc_t rs1,
                                                    __amdgpu_buffer_rsrc_t rt0, __amdgpu_buffer_rsrc_t rt1,
                                                    const float* scale, const float* shift, const void* dummy, bool live) {
  using E = Elem<T>;
  const int sub = ltid & 3;
  // Every call issues EXACTLY the same vector-memory instructions (6 piece loads + 4 scale / shift loads), live or not: s_waitcnt
  // vmcnt is an in-order count, and the compiler must assume the smaller count wherever a load is conditional -- which made each
  // wait for chunk s+1 also wait for the just-issued chunks s+2 and s+3 (no prefetch left).  A dead call (past the last chunk, or
  // no GroupNorm) loads from out-of-range buffer offsets: zeros, no memory traffic.
  // item geometry / source pixels once per ITEM (uniform branch; the loads below are inline asm with hand-counted waits, so a
  // compiler that duplicates them into both arms changes nothing: every path executes each of them once)
  if (live && ld_chunk == 0) {
    asm volatile("" : "+v"(ltid));      // opaque: the per-piece tile coordinates are recomputed per item, not kept in registers
    int ty, tx, co_t;
    persist_decode(blockIdx.x + ld_k * g.G, g.n_pt, g.nco, g.tpi, g.tiles_x, ld_n, ty, tx, co_t);
    const int y0 = ty * P_TH, x0 = tx * P_TW;
#pragma unroll
    for (int i = 0; i < P_NIT; ++i) {
      const int pix = (ltid + P_NLOAD * i) >> 2;
      const int u = pix / P_IN_TW, vv = pix - u * P_IN_TW;
      const int iy = y0 - g.pad + u, ix = x0 - g.pad + vv;
      // upsample 1: nearest x2 (Upsample2D); upsample 2: zero-stuffed x2 (input gradient of a stride-2 conv)
      const bool ok = pix < P_NPIX && iy >= 0 && iy < g.Hc && ix >= 0 && ix < g.Wc && (g.upsample != 2 || (((iy | ix) & 1) == 0));
      const int sy = g.upsample ? (iy >> 1) : iy, sx = g.upsample ? (ix >> 1) : ix;
      spix[i] = ok ? (ld_n * g.Hin + sy) * g.Win + sx : -1;
    }
  }
  const int chunk = ld_chunk;
  // source of this chunk: main [x0 | x1] or (fused 1x1 shortcut) tail [t0 | t1]; one descriptor per chunk, selected as a VALUE
  // (selects of by-value parameters: scalar s_cselect, no branches -- a branch here duplicates the loads into both arms and the
  // vmcnt bookkeeping of the merged paths turns conservative again)
  const bool tail = TAIL && chunk >= g.n_main;
  const int cc0 = tail ? (chunk - g.n_main) * 32 : chunk * 32;
  const int cfirst = tail ? (int)g.Ct0 : (int)g.C0;
  const bool second = cc0 >= cfirst;
  const int cch = second ? cc0 - cfirst : cc0;
  const unsigned cs = tail ? (second ? g.Ct1 : g.Ct0) : (second ? g.C1 : g.C0);
  const __amdgpu_buffer_rsrc_t rs = tail ? (second ? rt1 : rt0) : (second ? rs1 : rs0);
  const bool plain = tail;
  S.plain = plain;
  const unsigned cbytes = (unsigned)(cch + sub * 8) * E::BYTES;
  unsigned vm = 0;
#pragma unroll
  for (int i = 0; i < P_NIT; ++i) {
    // out-of-range as ARITHMETIC (or-ing 0xC0000000 into the offset), not a select the compiler may turn into a branch
    const unsigned dead = (unsigned)(spix[i] >> 31) & OOB_OFF;                 // padding pixel (spix = -1): out of range
    const unsigned off = ((unsigned)spix[i] * cs * E::BYTES + cbytes) | dead | (live ? 0u : OOB_OFF);
    vm |= (spix[i] >= 0 ? 1u : 0u) << i;
    PAsm<T>::load(S.r[i], rs, off);
  }
  S.vmask = vm;
  {
    // GroupNorm scale / shift of this chunk's 32 channels: ONE load per lane from a per-lane address (lane < 32: scale, else
    // shift) -- no select behind the load, which the compiler would schedule (and wait for) right after issuing it.  Without
    // GroupNorm, or past the last chunk, a valid dummy word is read and ignored.
    const int lane = ltid & 63;
    const bool use = live && !plain && scale != nullptr;
    const float* base = use ? (lane < 32 ? scale : shift) : (const float*)dummy;
    const int idx = use ? ld_n * g.cin + chunk * 32 + (lane & 31) : 0;
    const float* src = base + idx;
    asm volatile("global_load_dword %0, %1, off" : "=v"(S.scsh) : "v"(src) : "memory");
  }
  {
    const int nc = ld_chunk + (live ? 1 : 0);
    const bool wrap = nc == g.nchunks;
    ld_chunk = wrap ? 0 : nc;
    ld_k += wrap ? 1 : 0;
  }
}

template <typename T>
static __device__ __forceinline__ void persist_write(PSet<T>& S, unsigned char* buf, int ltid, bool affine, bool do_silu) {
  using E = Elem<T>;
  asm volatile("" : "+v"(ltid));     // as in persist_load: per-piece LDS addresses are recomputed, not hoisted
  const int sub = ltid & 3;
  PAsm<T>::wait_older(S);            // this set has landed; the younger set's loads stay in flight
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = __shfl(S.scsh, sub * 8 + j); sh[j] = __shfl(S.scsh, 32 + sub * 8 + j); }
#pragma unroll
  for (int i = 0; i < P_NIT; ++i) {
    const int pix = (ltid + P_NLOAD * i) >> 2;
    if (pix < P_NPIX)
      Stage<T>::xform_store(buf + pix * PersistCfg<T>::PITCH + sub * 8 * E::BYTES, S.r[i], sc, sh, affine && !S.plain,
                            do_silu && !S.plain, ((S.vmask >> i) & 1u) != 0);
  }
}

template <typename T, bool TAIL>
__global__ __launch_bounds__(384, sizeof(T) == 2 ? 3 : 2) void conv_persist_kernel(const ConvP p) {
  using E = Elem<T>;
  using Frag = typename E::Frag;
  using SR = typename Stage<T>::R;
  using Cfg = PersistCfg<T>;
  constexpr int CHB = Cfg::CHB, PITCH = Cfg::PITCH;
  constexpr int NIT = P_NIT, KSTEPS = P_KSTEPS, IN_TW = P_IN_TW, NPIX = P_NPIX;
  constexpr int NF = 4;                                                  // 32-pixel fragments (tile rows) per MFMA wave

  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- work items: (pixel tile, co tile), the co tiles of a pixel tile 8 items apart = on one XCD / L2 at about the same time
  const int tpi = p.tiles_x * p.tiles_y;
  const int n_pt = p.B * tpi, nco = p.n_co_tiles;
  const int total = n_pt * nco;
  const int G = gridDim.x;
  const int n_my = (int)blockIdx.x < total ? (total - (int)blockIdx.x + G - 1) / G : 0;
  const int total_chunks = n_my * p.nchunks;
#define PD_DECODE(ordinal, n, ty, tx, co_t) persist_decode(blockIdx.x + (ordinal) * G, n_pt, nco, tpi, p.tiles_x, n, ty, tx, co_t)

#ifdef PD_DIAG_NO_LOADER
  if (wave >= 4) return;
#endif
#ifdef PD_DIAG_NO_MFMA
  if (wave < 4) return;
#endif
  if (wave >= 4) {
    // ======================================================= LOADER waves ========================================================
    const int ltid = tid - 256;
    const void* const x0p = p.x0; const void* const x1p = p.x1; const void* const t0p = p.t0; const void* const t1p = p.t1;
    const unsigned b0v = p.bytes0, b1v = p.bytes1, tb0v = p.tbytes0, tb1v = p.tbytes1;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)x0p, 0, b0v, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(x1p ? rv(x1p) : rv(x0p)), 0, b1v, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt0 = __builtin_amdgcn_make_buffer_rsrc((void*)(TAIL ? rv(t0p) : rv(x0p)), 0, TAIL ? rv(tb0v) : rv(b0v), 0x00020000);
    const __amdgpu_buffer_rsrc_t rt1 = __builtin_amdgcn_make_buffer_rsrc((void*)((TAIL && t1p) ? rv(t1p) : rv(x0p)), 0, (TAIL && t1p) ? rv(tb1v) : rv(b0v), 0x00020000);
    const bool affine = p.scale != nullptr;
    const bool do_silu = p.silu != 0;
    PLoadGeom g;
    g.G = G; g.n_pt = n_pt; g.nco = nco; g.tpi = tpi; g.tiles_x = p.tiles_x; g.pad = p.pad; g.upsample = p.upsample;
    g.Hin = p.Hin; g.Win = p.Win; g.Hc = p.upsample ? p.Hin * 2 : p.Hin; g.Wc = p.upsample ? p.Win * 2 : p.Win;
    g.n_main = p.n_main; g.nchunks = p.nchunks; g.cin = p.C0 + p.C1; g.C0 = p.C0; g.C1 = p.C1; g.Ct0 = p.Ct0; g.Ct1 = p.Ct1;
    const float* const scale = p.scale; const float* const shift = p.shift;

    PSet<T> R0, R1;
    // load cursor: (item ordinal, chunk) of the next chunk to fetch, with that item's source pixels
    int ld_k = 0, ld_chunk = 0, ld_n = 0;
    int spix[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) spix[i] = -1;
#define load(S, live) persist_load<T, TAIL>(S, spix, ld_n, ld_k, ld_chunk, ltid, g, rs0, rs1, rt0, rt1, scale, shift, x0p, live)
#define write(S, slot) persist_write<T>(S, lds + (slot) * CHB, ltid, affine, do_silu)

#ifdef PD_DIAG_LOADER_IDLE      // timing experiment only (garbage results): the loader just keeps the barrier protocol
    __syncthreads();
    for (int s = 0; s < total_chunks; ++s) __syncthreads();
    return;
#endif
    load(R0, total_chunks > 0);
    load(R1, total_chunks > 1);
    write(R0, 0);
    load(R0, total_chunks > 2);
    __syncthreads();
    for (int s = 0; s < total_chunks; s += 2) {
      // chunk s multiplies out of slot 0: slot 1 takes chunk s+1 (register set R1), which then refills with chunk s+3.  The
      // writes past the last chunk land in a slot nobody reads (zeros from dead loads): unconditional like the loads.
      write(R1, 1);
      load(R1, s + 3 < total_chunks);
      __syncthreads();
      if (s + 1 >= total_chunks) break;
      write(R0, 0);
      load(R0, s + 4 < total_chunks);
      __syncthreads();
    }
#undef load
#undef write
    // the two dead sets issued past the last chunk are still in flight: land them before the registers go away
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ========================================================= MFMA waves ==========================================================
  const int wp = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int rbase[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) rbase[f] = ((wp * NF + f) * IN_TW + r) * PITCH + h * 8 * E::BYTES;

  const int main_ksteps = (TAIL ? p.n_main : p.nchunks) * KSTEPS;
  const int all_ksteps = main_ksteps + (TAIL ? p.n_tail * 2 : 0);
  const int last_kstep = all_ksteps - 1;
  // weight-fragment ring, AD k-steps ahead (KSTEPS % AR == 0: static ring indices).  Two MFMA waves per SIMD x 5 fragments in
  // flight: with 2 ahead a k-step took ~L2 latency / 2 = 550 cycles instead of its 128-256 MFMA cycles (loader idle: 1.0 PF)
  constexpr int AR = 6, AD = AR - 1;
  Frag aring[AR];
  f32x16 acc[NF];

  // per-item state of this wave
  int it_n = 0, it_ty = 0, it_tx = 0, it_ct32 = 0;
  bool active = false;
  const T* wbase = (const T*)p.w + lane * 8;
  auto begin_item = [&](int k) __attribute__((always_inline)) {   // decode item k and start its weight ring (AD fragments)
    int co_t;
    PD_DECODE(k, it_n, it_ty, it_tx, co_t);
    it_ct32 = co_t * 2 + wc;
    active = (it_ct32 * 32) < p.Cout_pad;
    // an inactive wave (co tile past Cout_pad) streams tile 0's weights and multiplies nothing: same instruction stream
    wbase = (const T*)p.w + (size_t)(active ? it_ct32 : 0) * all_ksteps * 512 + lane * 8;
    // compiler-counted loads here (NOT the inline-asm form): these registers stay live across the epilogue, where the allocator
    // spills -- an asm load's destination counts as written at once, so a spill / reuse of it before the data lands would let
    // the late data overwrite whatever the register holds by then (an address: memory fault)
#pragma unroll
    for (int i = 0; i < AD; ++i) aring[i] = E::load(wbase + (size_t)min(i, last_kstep) * 512);
  };

  // One chunk = 18 k-steps of 4 MFMAs.  Weight fragments: inline-asm loads two k-steps ahead, waited for by hand with the
  // constant vmcnt(AD loads) -- the ring is the only load stream of this role inside a chunk, so the count is exact; behind an
  // epilogue it over-waits (for that item's output stores), never under-waits.  The activation fragment of accumulator f is
  // re-read for the NEXT k-step right behind the MFMA that consumed it (same registers: an MFMA reads its operands in its first
  // passes, the LDS data lands ~100 cycles later), so every LDS read has three MFMAs (96+ cycles) to land.
  auto mma_chunk = [&](int chunk, const unsigned char* buf, auto last_c) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;    // last main chunk of the item: no prefetch past it (nothing left in flight)
    const int g0 = chunk * KSTEPS;
    Frag bc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) bc[f] = E::load(buf + rbase[f]);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const bool prefetch = !LAST || ks + AD < KSTEPS;
#ifdef PD_ABL_W0      // timing experiment: every weight fragment load hits fragment 0 (L1-resident): prices the weight stream
      if (prefetch) PRing<T>::load(aring[(ks + AD) % AR], wbase);
#else
      if (prefetch) PRing<T>::load(aring[(ks + AD) % AR], wbase + (size_t)(g0 + ks + AD) * 512);
#endif
      const int ns = ks + 1, tap = ns >> 1;
      const int toff = ((tap / 3) * IN_TW + (tap % 3)) * PITCH + (ns & 1) * 16 * E::BYTES;
      // fragments younger than the one needed: AD in steady state, fewer over the last AD k-steps of an item
      constexpr int dummy_ = 0; (void)dummy_;
      if (!LAST || ks + AD < KSTEPS) PRing<T>::template wait<AD>(aring[ks % AR]);
      else if (KSTEPS - 1 - ks == 4) PRing<T>::template wait<4>(aring[ks % AR]);
      else if (KSTEPS - 1 - ks == 3) PRing<T>::template wait<3>(aring[ks % AR]);
      else if (KSTEPS - 1 - ks == 2) PRing<T>::template wait<2>(aring[ks % AR]);
      else if (KSTEPS - 1 - ks == 1) PRing<T>::template wait<1>(aring[ks % AR]);
      else PRing<T>::template wait<0>(aring[ks % AR]);
#ifndef PD_ABL_NOPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        acc[f] = E::mma(aring[ks % AR], bc[f], acc[f]);
#ifndef PD_ABL_NOLDS     // timing experiment: the activation fragments are read once per chunk only
        if (ns < KSTEPS) bc[f] = E::load(buf + rbase[f] + toff);
#endif
      }
#ifndef PD_ABL_NOPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
  };
  auto mma_tail = [&](int chunk, const unsigned char* buf) __attribute__((always_inline)) {      // fused 1x1 shortcut: centre tap only, 2 k-steps
    constexpr int CENTER = (IN_TW + 1) * PITCH;
    const T* wt = wbase + (size_t)(main_ksteps + (chunk - p.n_main) * 2) * 512;
    const Frag a0 = E::load(wt), a1 = E::load(wt + 512);     // compiler-counted loads (two per chunk: nothing to pipeline)
    Frag bc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) bc[f] = E::load(buf + rbase[f] + CENTER);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = E::mma(a0, bc[f], acc[f]);
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int f = 0; f < NF; ++f) bc[f] = E::load(buf + rbase[f] + CENTER + 16 * E::BYTES);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = E::mma(a1, bc[f], acc[f]);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- epilogue of one item, wave-local: fragment -> private LDS patch [32 px][32 co] -> 16-byte pieces.  Every global access
  // is an UNCONDITIONAL buffer load / store (out-of-range offset = dropped): a fixed instruction stream, exact wait counts.
  constexpr int EPC = 16 / E::BYTES;                 // channels per 16-byte piece
  constexpr int PPP = 32 / EPC;                      // pieces per pixel (32 co per wave)
  constexpr int PXP = 64 / PPP;                      // pixels per pass
  constexpr int NPASS = 32 / PXP;
  unsigned char* patch = lds + Cfg::TILE_BYTES + wave * Cfg::EP_WAVE;
  const int piece = lane % PPP, prow = lane / PPP;
  const unsigned ybytes = (unsigned)((size_t)p.B * p.Hout * p.Wout * p.Cout * E::BYTES);      // < 2 GiB (checked on the host)
  const void* const yp = p.y; const void* const resp = p.residual; const void* const statp = p.stats;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)yp, 0, ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(resp ? rv(resp) : rv(yp)), 0, resp ? rv(ybytes) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rstat = __builtin_amdgcn_make_buffer_rsrc((void*)(statp ? rv(statp) : rv(yp)), 0,
                                                                          statp ? (unsigned)((size_t)p.B * 2 * tpi * p.Cout * 8) : 0u, 0x00020000);
  const bool want_stats = statp != nullptr;
  auto epilogue = [&](int n, int ty, int tx, int ct32) __attribute__((always_inline)) {
    const int co = ct32 * 32 + piece * EPC;
    const bool co_ok = co < p.Cout;                  // Cout % 8 == 0 (checked on the host): a piece is all real or all padding
    // bias + time embedding of the wave's 32 channels (fp32): fetched once per item into a row of the patch (lane l < 32 holds
    // channel l), read back per pass -- eight registers less than keeping them across the four fragments
    {
      const int cl = ct32 * 32 + (lane & 31);
      float b = 0.f;
      if (lane < 32 && cl < p.Cout) {
        b = p.bias[cl];
        if (p.temb) b += p.temb[(size_t)n * p.temb_stride + cl];
      }
      if (lane < 32) *(float*)(patch + 32 * Cfg::EPF_PITCH + lane * 4) = b;
    }
    float ssum[EPC], ssq[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      // the patch holds the fp32 accumulators; bias / temb / residual are added in fp32 below: ONE rounding, at the store
      const int oy = ty * P_TH + wp * NF + f;
      u32x4 rr[NPASS];
      unsigned yoff[NPASS];
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {            // residual loads first: they fly while the patch is written
        const int ox = tx * P_TW + ps * PXP + prow;
        const bool ok = co_ok && oy < p.Hout && ox < p.Wout;
        yoff[ps] = (unsigned)((((n * p.Hout + oy) * p.Wout + ox) * p.Cout + co) * E::BYTES) | (ok ? 0u : OOB_OFF);
        rr[ps] = __builtin_amdgcn_raw_buffer_load_b128(rres, yoff[ps], 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(f32x4*)(patch + r * Cfg::EPF_PITCH + (8 * g + 4 * h) * 4) = (f32x4){acc[f][4 * g], acc[f][4 * g + 1], acc[f][4 * g + 2], acc[f][4 * g + 3]};
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int px = ps * PXP + prow;
        float x[EPC];
#pragma unroll
        for (int q = 0; q < EPC / 4; ++q) {
          const f32x4 t4 = *(const f32x4*)(patch + px * Cfg::EPF_PITCH + (piece * EPC + 4 * q) * 4);
          const f32x4 b4 = *(const f32x4*)(patch + 32 * Cfg::EPF_PITCH + (piece * EPC + 4 * q) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) x[4 * q + j] = t4[j] + b4[j];
        }
        u32x4 v;
        if (E::BYTES == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lo = x[(2 * j) % EPC] + __uint_as_float(rr[ps][j] << 16);
            const float hi = x[(2 * j + 1) % EPC] + __uint_as_float(rr[ps][j] & 0xffff0000u);
            v[j] = pack2bf(lo, hi);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = __float_as_uint(x[j % EPC] + __uint_as_float(rr[ps][j]));
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, ry, yoff[ps], 0, 0);
        // GroupNorm statistics of what was STORED (the consumer normalises the rounded tensor); dropped pixels count as nothing
        const float live = (yoff[ps] & 0x80000000u) ? 0.f : 1.f;
        if (E::BYTES == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(v[j] << 16) * live, hi = __uint_as_float(v[j] & 0xffff0000u) * live;
            ssum[(2 * j) % EPC] += lo; ssq[(2 * j) % EPC] += lo * lo;
            ssum[(2 * j + 1) % EPC] += hi; ssq[(2 * j + 1) % EPC] += hi * hi;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float xx = __uint_as_float(v[j]) * live; ssum[j % EPC] += xx; ssq[j % EPC] += xx * xx; }
        }
      }
      __builtin_amdgcn_wave_barrier();               // the patch is rewritten by the next fragment
    }
    {
      // lanes with equal `piece` hold the same channels: fixed-order butterfly over the prow bits (deterministic)
#pragma unroll
      for (int m = PPP; m < 64; m <<= 1) {
#pragma unroll
        for (int j = 0; j < EPC; ++j) { ssum[j] += __shfl_xor(ssum[j], m); ssq[j] += __shfl_xor(ssq[j], m); }
      }
      const int T2 = 2 * tpi;
      const int t2 = (ty * p.tiles_x + tx) * 2 + wp;
      const unsigned so = (unsigned)((((n * T2 + t2) * p.Cout + co) * 2) * 4) | ((want_stats && prow == 0 && co_ok) ? 0u : OOB_OFF);
#pragma unroll
      for (int q = 0; q < 2 * EPC / 4; ++q) {
        u32x4 v4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int e = q * 4 + j; v4[j] = __float_as_uint((e & 1) ? ssq[e >> 1] : ssum[e >> 1]); }
        __builtin_amdgcn_raw_buffer_store_b128(v4, rstat, so + 16 * q, 0, 0);
      }
    }
  };

#ifdef PD_DIAG_MFMA_IDLE        // timing experiment only (garbage results): the MFMA waves just keep the barrier protocol
  __syncthreads();
  for (int s2 = 0; s2 < total_chunks; ++s2) __syncthreads();
  return;
#endif
  if (n_my > 0) begin_item(0);
  __syncthreads();                                    // slot 0 holds chunk 0
  int s = 0;
  for (int k = 0; k < n_my; ++k) {
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = (f32x16)(0.f);
    const int n = it_n, ty = it_ty, tx = it_tx, ct32 = it_ct32;
    const bool act = active;
    // main chunks, then (fused shortcut) tail chunks: two loops, so the accumulators flow through straight-line call sites (an
    // if / else per chunk made the register allocator keep two accumulator sets)
    const int n_main_chunks = TAIL ? p.n_main : p.nchunks;
    int chunk = 0;
    for (; chunk < n_main_chunks - 1; ++chunk, ++s) {
      mma_chunk(chunk, lds + (s & 1) * CHB, std::false_type{});
      __syncthreads();
    }
    mma_chunk(chunk, lds + (s & 1) * CHB, std::true_type{});
    if (chunk == p.nchunks - 1 && k + 1 < n_my) begin_item(k + 1);     // next item's first weights fly during the epilogue
    __syncthreads();
    ++chunk; ++s;
    if constexpr (TAIL) {
      for (; chunk < p.nchunks; ++chunk, ++s) {
        mma_tail(chunk, lds + (s & 1) * CHB);
        if (chunk == p.nchunks - 1 && k + 1 < n_my) begin_item(k + 1);
        __syncthreads();
      }
    }
    if (act) epilogue(n, ty, tx, ct32);
  }
}

static int persist_grid_cache = 0;
static int persist_cus() {
  if (persist_grid_cache == 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
    persist_grid_cache = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }
  return persist_grid_cache;
}

template <typename T, bool TAIL>
static int launch_persist(const ConvP& p, hipStream_t st) {
  using Cfg = PersistCfg<T>;
  static_assert(Cfg::LDS_BYTES <= 160 * 1024, "tile too large");
  auto kern = conv_persist_kernel<T, TAIL>;
  static bool attr_set = false;   // per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PD_ERR_LAUNCH; }
    attr_set = true;
  }
  ConvP q = p;
  q.tiles_x = (p.Wout + P_TW - 1) / P_TW;
  q.tiles_y = (p.Hout + P_TH - 1) / P_TH;
  q.tiles_x_shift = -1;
  q.n_co_tiles = (p.Cout_pad + 63) / 64;
  const long long total = (long long)p.B * q.tiles_x * q.tiles_y * q.n_co_tiles;
  const int per_cu = (160 * 1024) / Cfg::LDS_BYTES >= 2 ? 2 : 1;         // bf16: two 59 KB workgroups per CU; fp32: one
  long long grid = (long long)persist_cus() * per_cu;
  if (grid > total) grid = total;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(384), Cfg::LDS_BYTES, st, q);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

template <typename T> int launch_conv_persist(const ConvP& p, hipStream_t st) {
  return p.n_tail > 0 ? launch_persist<T, true>(p, st) : launch_persist<T, false>(p, st);
}
template int launch_conv_persist<float>(const ConvP&, hipStream_t);
template int launch_conv_persist<bf16_t>(const ConvP&, hipStream_t);

int conv_persist_stat_tiles(int Hout, int Wout) {
  return 2 * ((Hout + P_TH - 1) / P_TH) * ((Wout + P_TW - 1) / P_TW);
}

bool conv_persist_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("PD_NO_PERSIST_CONV"); v = (e && e[0] == '1') ? 0 : 1; }   // diagnostic A/B switch
  return v == 1;
}

}  // namespace pd
