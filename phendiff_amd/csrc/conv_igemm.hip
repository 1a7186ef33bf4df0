// pd_conv: implicit-GEMM convolution on gfx950 MFMA, NHWC, LDS-staged halo tiles.
//
// GEMM view:  D[co][pixel] = sum_k W[co][k] * X[k][pixel],  k = (tap, ci).
//   A operand = packed weights   (lane: row = co,    8 consecutive ci)   -> straight from global/L2, 1 KiB coalesced per wave,
//                                                                           prefetched two k-steps ahead in registers
//   B operand = activation tile  (lane: col = pixel, 8 consecutive ci)   -> ds_read_b128 from the LDS halo tile, one k-step ahead
//   D         : lane owns one pixel and 16 output channels (4 runs of 4 consecutive co) -> 8/16-byte NHWC stores
// Workgroup = 256 threads = 4 waves = 2 (pixel halves) x 2 (co halves of 32); tile = TH*TW pixels x 64 co.
// K loop: chunks of 32 input channels; per chunk the halo tile is staged global -> regs -> (GroupNorm affine,
// SiLU, zero padding) -> LDS once and reused by all KS*KS taps (9x LDS reuse instead of 9x global re-reads).
// Software pipeline (DB = double-buffered LDS): while the MFMAs of chunk c run out of LDS[c&1], the same waves
// transform the registers of chunk c+1 and write LDS[(c+1)&1] BETWEEN the MFMAs (matrix and vector pipes overlap),
// then issue the global loads of chunk c+2; one barrier per chunk.
// LDS pixel pitch = 32 ch + 16 B pad: an odd number of 16-B slots, so a wave's 32 consecutive pixels hit 16
// distinct slots per ds_read_b128 lane group (conflict-free for TW = 32).
#include <type_traits>
#include <utility>
#include <stdio.h>
#include <stdlib.h>
#include "pd_common.h"
#include "pd_stage.h"
#include "pd_conv.h"

#ifndef PD_S2_SINGLE
#define PD_S2_SINGLE 1
#endif
// Diagnostic (scripts/experiments/overlap_pair.py, -DPD_CONV_PRIO_BASE=2): wave priority of the conv kernel outside / inside its MFMA
// clusters -- the controlled attention || convolution co-residency experiment of round 3.  The shipped build keeps 0 / 1.
#ifndef PD_CONV_PRIO_BASE
#define PD_CONV_PRIO_BASE 0
#endif
// 16-bit 3x3 stride-1 double-buffered variants multiply with v_mfma_f32_16x16x32 instead of 32x32x16 (see M16 in conv_kernel).
// -DPD_CONV_M16=0 builds the 32x32x16 form everywhere (same-box A/B).
// Round 6: depth of the weight-fragment register ring of the 8-wide 3x3 tiles (<= 128 pixels, 16-bit engines; 0: the general 3); must divide 18.
// A 64-pixel tile has ONE 32-cycle MFMA per k-step and wave, so the general ring fetched a weight fragment 64 matrix cycles before its use -- far less
// than an L2 round trip, let alone the HBM trip of the trajectory where 1.7 GB of weights pass between two uses of a layer.  Five k-steps ahead
// (24 registers of 94): 1 280 -> 1 280 @8x8 with cold weights 117 -> 94 us, 2 560 -> 1 280 227 -> 170 us; same box, whole workloads: SD img2img
// +1.9 %, SD fine-tuning +0.6 % (profiles/r6_ab_conv8_ring_*.log).  9 / 18 deep: no further gain.
#ifndef PD_CONV_AR8
#define PD_CONV_AR8 6
#endif
#ifndef PD_CONV_AR8_64   // ... of the 64-pixel tile (8 x 8 images)
#define PD_CONV_AR8_64 PD_CONV_AR8
#endif
#ifndef PD_CONV_PIXMAP16   // 1: staging pieces dealt to the lanes pixel-fastest (conv_kernel PIXMAP16) -- diagnostic builds: removes the LDS store conflicts and is
                           // 1.5 % SLOWER on the headline (15.25 vs 15.48 images/s, same box): a 16-lane group of the global load then touches 16 lines instead of 4
#define PD_CONV_PIXMAP16 0
#endif
#ifndef PD_CONV_TAILPLAIN_WPS   // workgroups per CU the prologue-free convolutions with a fused 1x1 tail are compiled for (0: as the tail-free form, 3 -- where they spill 20 B / lane)
#define PD_CONV_TAILPLAIN_WPS 0
#endif
#ifndef PD_CONV_RMAP     // 1: bank-conflict-free row order of the 8 x 8 tile's fragments (conv_kernel RMAP) -- diagnostic builds: parity-green, neutral as an op
                         // (94.2 vs 94.4 us, 176 vs 177 us) and on SD img2img (9.943 vs 9.948): the LDS pipe was never what this tile waited for
#define PD_CONV_RMAP 0
#endif
#ifndef PD_CONV_M16
#define PD_CONV_M16 1
#endif
// Experiment switches (round 4): the 16x16x32 form behind a GroupNorm prologue too; the wave-priority raise of the MFMA clusters.
#ifndef PD_CONV_M16_GN
#define PD_CONV_M16_GN 0
#endif
#ifndef PD_CONV_MFMA_PRIO
#define PD_CONV_MFMA_PRIO 1
#endif
// PRO (compile-time GroupNorm + SiLU prologue, branch-free staging, MFMA / vector interleave by sched_group_barrier): which launches take it
// (0 = none, 1 = the two-tile form, 2 = the one-tile form too), its waves per SIMD (register budget), the MFMA slot at which the staging
// work of a chunk starts, and whether the activation fragments of the next k-step are read ahead.
#ifndef PD_CONV_PRO
#define PD_CONV_PRO 1
#endif
#ifndef PD_CONV_PRO_WPS
#define PD_CONV_PRO_WPS 3
#endif
#ifndef PD_CONV_PRO_START
#define PD_CONV_PRO_START 8
#endif
#ifndef PD_CONV_PRO_DSR
#define PD_CONV_PRO_DSR 1
#endif
#ifndef PD_CONV_PRO_M16
#define PD_CONV_PRO_M16 0     /* 16x16x32 MFMAs in the PRO launches: 0 = none, 1 = the two-tile form, 2 = the one-tile form too */
#endif
#ifndef PD_CONV_PRO_PV
#define PD_CONV_PRO_PV 24     /* non-transcendental vector instructions of one staged piece (counted in the ISA) */
#endif
namespace pd {



// Ablation switches for diagnostic builds (scripts/ablate_conv.sh); never defined in the shipped library.
#ifdef PD_ABL_W0
#define PD_WIDX(i) 0            /* every weight fragment load hits fragment 0 (L1-resident): prices the weight stream */
#else
#define PD_WIDX(i) (i)
#endif
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int... I, typename F> __device__ __forceinline__ void pd_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void pd_static_for(F&& f) { pd_static_for_impl(std::make_integer_sequence<int, N>{}, f); }
// PRO schedule (conv_kernel): a staged piece is PRO_OPS steps (2 halves x 9 stages x 2 channel pairs, then its LDS store); their issue
// costs in tenths of a cycle (profiles/r3_exp_variants.log), and the number of steps -- over the NIT pieces of a chunk -- that are
// emitted by the end of MFMA slot S of NS when the staging work starts at slot PS0 and is dealt out by cost
constexpr int PRO_OPS = 37;
constexpr int pro_op_cost(int k) {
  if (k >= 36) return 100;
  const int stg = (k % 18) >> 1;
  return (stg == 3 || stg == 4 || stg == 6 || stg == 7) ? 87 : (stg == 0 ? 60 : (stg == 8 ? 106 : 58));
}
constexpr int pro_ops_done(int S, int NS, int PS0, int NIT) {
  if (S < PS0) return 0;
  if (S >= NS - 1) return NIT * PRO_OPS;
  long long tot = 0;
  for (int k = 0; k < PRO_OPS; ++k) tot += pro_op_cost(k);
  tot *= NIT;
  const long long target = (long long)(S - PS0 + 1) * tot / (NS - PS0);
  int done = 0;
  long long dc = 0;
  while (done < NIT * PRO_OPS && dc + pro_op_cost(done % PRO_OPS) <= target) { dc += pro_op_cost(done % PRO_OPS); ++done; }
  return done;
}
__device__ __forceinline__ unsigned pd_lin_block() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }
#ifdef PD_STAMPS   // diagnostic build only (scripts/stamp_conv.py): phase timestamps of the first workgroups
__device__ unsigned long long pd_conv_stamps[4096 * 16];
#define PD_STAMP(k)                                                                                   \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    unsigned long long t_;                                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if (threadIdx.x == 0 && pd_lin_block() < 4096) pd_conv_stamps[pd_lin_block() * 16 + (k)] = t_;            \
  } while (0)
#else
#define PD_STAMP(k) do {} while (0)
#endif
// NCO = 2 (3x3 stride 1, 256-pixel tiles, Cout % 128 == 0): a workgroup computes TWO 64-channel output tiles from one staged
// halo tile -- every staged (GroupNorm + SiLU-transformed) activation and every LDS fragment read feeds twice the MFMAs, and the
// grid of the 128 / 256-channel layers (4096 / 2048 workgroups of the NCO = 1 form on 768 slots = 5.33 / 2.67 rounds) becomes
// 2048 / 1024 workgroups on 512 slots = whole rounds.  128 accumulator registers: two workgroups per CU instead of three.
// PLAIN (compile time): no GroupNorm / SiLU prologue -- the input gradients of every convolution, the latent-diffusion UNet's convolutions
// (its GroupNorms are applied by pd_gn_apply), the upsamplers.  The 16 scale / shift registers and the transform are gone, which is what lets
// the 16x16x32 MFMA form (M16 below) fit the register budgets.
template <typename T, int KS, int STRIDE, int TH, int TW, bool DB, bool TAIL, int NCO = 1, bool PLAIN = false, int PRO = 0, bool STACK = false>
__global__ __launch_bounds__(256, NCO == 2 ? 2 : (PRO ? PD_CONV_PRO_WPS : ((TAIL && PLAIN && PD_CONV_TAILPLAIN_WPS) ? PD_CONV_TAILPLAIN_WPS : (((KS == 1 || (KS == 3 && STRIDE == 1 && TH * TW == 256)) && sizeof(T) == 2) ? 3 : 2)))) void conv_kernel(const ConvP p) {
  static_assert(PRO == 0 || (!PLAIN && DB && KS == 3 && STRIDE == 1 && sizeof(T) == 2), "compile-time prologue: 16-bit 3x3 stride-1 double-buffered launches");   // 1x1: fits 168 registers without spilling -> 3 workgroups per CU
  static_assert(!TAIL || (DB && KS == 3 && STRIDE == 1), "fused shortcut tail: 3x3 stride-1 double-buffered variant only");
  static_assert(NCO == 1 || (NCO == 2 && DB && KS == 3 && STRIDE == 1), "two output tiles per workgroup: 3x3 stride-1 double-buffered variant only");
  // STACK (round 6): the tile is TH / 8 whole 8 x 8 IMAGES (samples n, n + 1, ...) stacked vertically, each with its own zero halo rows in LDS --
  // an 8 x 8 level has 64 pixels per sample, and a 64-pixel tile streams a 32 co x 16 k weight fragment per MFMA; two images per tile halve
  // that and the barriers per MFMA.  NHWC output of 8 x 8 images is contiguous over samples, so tile row R = 8 j + y of sample n + j lies at
  // row R of sample n.  Per-sample operands: temb (a wave's fragments belong to ONE sample: n + wp), statistics (one reduction per sample);
  // no GroupNorm prologue (dispatch).
  static_assert(!STACK || (KS == 3 && STRIDE == 1 && TW == 8 && TH == 16 && NCO == 1 && !PLAIN && PRO == 0), "stacked 8 x 8 images: the 16 x 8 tile");
  constexpr int SG = STACK ? TH / 8 : 1;   // images per tile
  using E = Elem<T>;
  using Frag = typename E::Frag;
  using SR = typename Stage<T>::R;
  constexpr int TP = TH * TW;              // pixels per workgroup tile
  constexpr int NF = TP / 64;              // 32-pixel fragments per wave
  constexpr int RPF = 32 / TW;             // tile rows per fragment (TW == 32 -> 1)
  constexpr int IN_TH = STACK ? SG * 10 : (TH - 1) * STRIDE + KS;
  constexpr int IN_TW = (TW - 1) * STRIDE + KS;
  constexpr int NPIX = IN_TH * IN_TW;
  // double-buffered variants interleave the two buffers per pixel -- [chunk c: 32 ch | chunk c+1: 32 ch | 16 B pad] -- so both
  // share ONE pad: 144-byte pixels (bf16; an odd number of 16-B slots, conflict-free like the 80-byte single pitch) make the
  // 10 x 34 halo tile 48 960 B instead of 2 x 27 200 B, which (with the 168-register budget) admits a third workgroup per CU
  // M16 (PLAIN, 16-bit, 3x3, stride 1, double-buffered, TW >= 16, one output tile per workgroup): the MFMAs are v_mfma_f32_16x16x32 -- per (32 co x 32 pixel) block and tap
  // 4 MFMAs over K = 32 channels instead of 2 x 32x32x16 over K = 16: the same FLOPs, cycles, operand bytes and registers, but the
  // chip holds a higher clock on this shape under the convolution's load (it runs at 1.9-2.0 GHz, profiles/r3_conv_clock.json;
  // MI355X_MICROARCH.md "DVFS give-back" item 7; a timing-only build of this kernel with the shape swapped: -4.1 % on the sum of 3x3).
  // The form needs 4 (NCO = 1) / 8 (NCO = 2) registers more for its A operands (a tap's two operands are prefetched a whole tap ahead) and
  // spilled behind a GroupNorm prologue (16 scale / shift registers): shipped for the PLAIN instantiations only, where it fits (162 VGPRs,
  // no scratch) -- same box: the SD UNet's 3x3 convolutions -4.3 % (16 x 16 levels -12 %), SD img2img +1.4...1.6 %, SD fine-tuning +1.1 %.
  //   A: the PACKED weights are unchanged (32 co x 16 k fragments, lane (r, h) = 8 channels 16 half + 8 h of row r): lane (i, g) of the
  //      16 co x 32 k operand ti reads the 16 bytes of old lane (16 ti + i, g & 1) of half g >> 1 -- four 256-byte runs per wave;
  //   B: lane (i, g) reads pixel i of its 16-pixel run, channels 8 g .. 8 g + 7 (one ds_read_b128); pixel pitch 160 B (10 slots):
  //      the 144-byte pitch is 2-way conflicted for this access, 160 is conflict-free (brute force over the ds_read_b128 lane groups);
  //   D: acq[c][f][2 ti + tj]: lane (i, g) owns pixel 16 tj + i of fragment f and channels 16 ti + 4 g .. + 3 of tile c.
  constexpr bool M16 = PD_CONV_M16 && (PLAIN || (PD_CONV_M16_GN && (NCO == 1 || PD_CONV_M16_GN >= 2)) || (PRO && PD_CONV_PRO_M16 >= (NCO == 2 ? 1 : 2))) && sizeof(T) == 2 && (KS == 3 || KS == 2) && STRIDE == 1 && DB && TW >= 16;
  constexpr int CHB = 32 * E::BYTES;                 // bytes of one 32-channel chunk of a pixel
  constexpr int PITCH = DB ? 2 * CHB + (M16 ? 32 : 16) : CHB + 16;
  constexpr int NIT = (NPIX * 4 + 255) / 256;
  constexpr int TAPS = KS * KS;
  constexpr int KSTEPS = TAPS * 2;
  constexpr int LDS_TILE = DB ? CHB : 0;             // offset of the second buffer
  static_assert(RPF >= 1 && TW * RPF == 32, "TW must divide 32");

  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  if (PD_CONV_PRIO_BASE) __builtin_amdgcn_s_setprio(PD_CONV_PRIO_BASE);

  // ---- block -> (pixel tile, co tile, sample) = (x, y, z): the co tiles of one pixel tile are tiles-per-image blocks
  // apart in dispatch order, i.e. on the same XCD / L2 whenever tiles-per-image % 8 == 0 (speed only)
  // co_major (round 6; layers whose weights outweigh their input activations): the work list runs channel tile -> sample -> pixel tile and
  // every XCD takes a contiguous run of it (xcd_chunk_index), so a channel tile's weights are streamed through ONE L2, not eight
  int co_t = blockIdx.y, n = blockIdx.z * SG, bx = blockIdx.x;
  if (!STACK && p.co_major) {
    const int gx = gridDim.x, gxz = gridDim.x * gridDim.z;
    const int idx = xcd_chunk_index((int)pd_lin_block(), gxz * (int)gridDim.y);
    co_t = idx / gxz;
    const int rem = idx - co_t * gxz;
    n = rem / gx; bx = rem - n * gx;
  }
  int tx, ty;
  if (p.tiles_x_shift >= 0) { tx = bx & (p.tiles_x - 1); ty = bx >> p.tiles_x_shift; }
  else { ty = bx / p.tiles_x; tx = bx - ty * p.tiles_x; }
  PD_STAMP(0);
  const int y0 = ty * TH, x0 = tx * TW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int ct32 = co_t * (2 * NCO) + wc;              // this wave's (first) 32-co tile; NCO = 2: the second one is ct32 + 2
  const bool wave_active = (ct32 * 32) < p.Cout_pad;   // wave-uniform (NCO = 2 is only launched with Cout_pad % 128 == 0: always)

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x0, 0, p.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x1 ? p.x1 : p.x0), 0, p.bytes1, 0x00020000);

  const __amdgpu_buffer_rsrc_t rt0 = __builtin_amdgcn_make_buffer_rsrc((void*)(TAIL ? p.t0 : p.x0), 0, TAIL ? p.tbytes0 : p.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rt1 = __builtin_amdgcn_make_buffer_rsrc((void*)((TAIL && p.t1) ? p.t1 : p.x0), 0, (TAIL && p.t1) ? p.tbytes1 : p.bytes0, 0x00020000);

  // ---- staging bookkeeping: this thread's pieces (pixel, 8-channel sub-block)
  // Round 6: which (pixel, 8-channel sub-block) piece a thread stages.  Per 64 lanes the pieces are the same 16 pixels x 4 sub-blocks either way (same global
  // lines), but with lane = 4 pixel + sub a 16-lane pass of the LDS store covers 4 pixels x 64 B, and at the 144-byte (and 80-byte) pixel pitch pixels p and
  // p + 2 overlap in 8 banks: every store pass ran twice (PMC: 19-28 % of the convolutions' LDS-active cycles were conflict cycles).  lane = 16 sub + pixel puts
  // 16 different pixels' same sub-block into a pass: 36 p mod 64 = 16 different multiples of 4, conflict-free.  (The 160-byte pitch of the 16x16x32 form repeats
  // every 8 pixels: it keeps the old map.)
  constexpr bool PIXMAP16 = PD_CONV_PIXMAP16 && !M16;
  const int sub = PIXMAP16 ? (tid >> 4) & 3 : tid & 3;
  auto stage_pix = [&](int i) { return PIXMAP16 ? ((tid >> 6) + 4 * i) * 16 + (tid & 15) : (tid + 256 * i) >> 2; };
  const int Hc = p.upsample ? p.Hin * 2 : p.Hin;
  const int Wc = p.upsample ? p.Win * 2 : p.Win;
  int spix[NIT];   // linear source pixel index (n, sy, sx) or -1 (zero padding / no such piece)
  constexpr int TAIL_CENTRE = 1 << 30;
  // branch-free (round 4: the nested `if`s were 16 divergent branches = 2.2k cycles of every workgroup's prologue, s_memtime stamps)
  // upsample 1: nearest x2 (Upsample2D);  upsample 2: zero-stuffed x2 (the stride-2 conv's input gradient is a stride-1 conv over dY
  // with zeros between its samples: samples at the EVEN positions for a pad-1 forward);  upsample 3: at the ODD positions
  // (Downsample2D(padding=0)'s (0,1,0,1)-padded forward)
  const int ups = p.upsample ? 1 : 0;                     // source coordinate = conv coordinate >> ups
  const int ph_mask = p.upsample >= 2 ? 1 : 0, ph_want = p.upsample == 3 ? 1 : 0;   // zero-stuffing: both coordinates' low bit must equal ph_want
  // (input-side phase, pd_conv_args.phase_in: the source is in_step times as large and the launch reads its pixels
  // (in_step iy + in_oy, in_step ix + in_ox); in_step = 1, offsets 0 otherwise)
  const int istep = p.in_step, Ws = p.Win * istep;
  const int iy_base = y0 * STRIDE - p.pad, ix_base = x0 * STRIDE - p.pad_x, n_base = n * p.Hin * istep + p.in_oy;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int pix = stage_pix(i);
    const int u = pix / IN_TW, vv = pix - u * IN_TW;
    const int sj = STACK ? u / 10 : 0, su = u - sj * 10;         // STACK: image within the tile, row within its 10-row halo block
    const int iy = STACK ? su - 1 : iy_base + u, ix = ix_base + vv;
    const bool ok = (pix < NPIX) & ((unsigned)iy < (unsigned)Hc) & ((unsigned)ix < (unsigned)Wc) & (!STACK || n + sj < p.B)
                    & (((iy & ph_mask) == (ph_want & ph_mask)) & ((ix & ph_mask) == (ph_want & ph_mask)));
    int v = STACK ? ((n + sj) * 8 + iy) * 8 + ix : (n_base + (iy >> ups) * istep) * Ws + (ix >> ups) * istep + p.in_ox;
    // fused 1x1 tail: its chunks multiply the CENTRE tap only, so they need the tile's own pixels, not the halo ring around them
    // (a third more pixels: 10 x 34 against 8 x 32) -- bit 30 marks the pixels a tail chunk loads (an index is < 2^26: the tensors
    // are < 2 GiB at >= 64 bytes per pixel)
    if (TAIL) v |= (((unsigned)((STACK ? su : u) - KS / 2) < (unsigned)(STACK ? 8 : TH)) & ((unsigned)(vv - KS / 2) < (unsigned)TW)) ? TAIL_CENTRE : 0;
    spix[i] = ok ? v : -1;
  }
  const bool affine = PRO || (!PLAIN && p.scale != nullptr);
  const bool do_silu = PRO || (!PLAIN && p.silu != 0);
  const int cin = p.C0 + p.C1;

  SR stage[NIT];
  float sc[8], sh[8];
  bool stage_plain = false;   // the staged chunk is a tail chunk: no affine / SiLU
  auto issue_loads = [&](int chunk) {
    // source of this chunk: 0/1 = main [x0 | x1], 2/3 = tail [t0 | t1]
    int src, cch;
    if (!TAIL || chunk < p.n_main) { cch = chunk * 32; src = cch < p.C0 ? 0 : 1; if (src == 1) cch -= p.C0; }
    else { cch = (chunk - p.n_main) * 32; src = cch < p.Ct0 ? 2 : 3; if (src == 3) cch -= p.Ct0; }
    stage_plain = TAIL && src >= 2;
    // byte offset of (pixel, sub-block) in the source, recomputed per chunk (6 multiplies: keeping them across the chunk loop
    // costs 6 registers that decide between 2 and 3 resident workgroups per CU)
    const unsigned cs = src == 0 ? p.C0 : (src == 1 ? p.C1 : (src == 2 ? p.Ct0 : p.Ct1));
    const unsigned cbytes = (unsigned)(cch + sub * 8) * E::BYTES;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
#ifdef PD_ABL_X0
      const unsigned off = OOB_OFF;   // ablation: no activation traffic
#else
      const bool want = TAIL ? (stage_plain ? spix[i] >= TAIL_CENTRE : spix[i] >= 0) : spix[i] >= 0;
      const unsigned off = want ? (unsigned)(TAIL ? spix[i] & (TAIL_CENTRE - 1) : spix[i]) * cs * E::BYTES + cbytes : OOB_OFF;
#endif
      stage[i] = src == 0 ? Stage<T>::load(rs0, off) : (src == 1 ? Stage<T>::load(rs1, off)
                 : (src == 2 ? Stage<T>::load(rt0, off) : Stage<T>::load(rt1, off)));
    }
    if ((PRO && !stage_plain) || (!PRO && affine && !stage_plain)) {
      const int cg = chunk * 32 + sub * 8;          // channel index in the concatenated main input
      const float* ps = p.scale + (size_t)n * cin + cg;
      const float* pb = p.shift + (size_t)n * cin + cg;
      const f32x4 a0 = *(const f32x4*)ps, a1 = *((const f32x4*)ps + 1);
      const f32x4 b0 = *(const f32x4*)pb, b1 = *((const f32x4*)pb + 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) { sc[j] = a0[j]; sc[4 + j] = a1[j]; sh[j] = b0[j]; sh[4 + j] = b1[j]; }
    }
  };
  // Padding pixels (spix < 0) hold zeros in BOTH LDS buffers from the start of the kernel (zero_padding below) and are never
  // written again: the per-piece "if (!valid) zero the 8 transformed values" (5 vector instructions per piece, ~6 % of the
  // kernel's vector work by the round-3 instruction counters) becomes an exec-masked store.
  auto write_piece = [&](int i, unsigned char* buf, bool tail_piece = false) {
    const int pix = stage_pix(i);
    if constexpr (PRO) {
      if (TAIL && tail_piece) {          // (compile-time at every call site) a fused-tail chunk is staged as it is
        if (pix < NPIX && spix[i] >= 0) *(u32x4*)(buf + pix * PITCH + sub * 8 * E::BYTES) = Stage<T>::raw(stage[i]);
        return;
      }
      // no branch: a lane without a pixel to write (padding, past the tile) stores to the dump slot behind the tile, so that a piece is
      // straight-line code in the same basic block as the wave's MFMAs (sched_group_barrier below places it between them)
      const u32x4 o = Stage<T>::xform_gs(stage[i], sc, sh);
      const bool ok = pix < NPIX && spix[i] >= 0;
      *(u32x4*)(ok ? buf + pix * PITCH + sub * 8 * E::BYTES : lds + NPIX * PITCH) = o;
    } else
    if (pix < NPIX && spix[i] >= 0)
      Stage<T>::template xform_store<false>(buf + pix * PITCH + sub * 8 * E::BYTES, stage[i], sc, sh, affine && !stage_plain,
                                            do_silu && !stage_plain, true);
  };
  auto zero_padding = [&]() {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int pix = stage_pix(i);
      if (pix < NPIX && spix[i] < 0) {
        unsigned char* d = lds + pix * PITCH + sub * 8 * E::BYTES;
#pragma unroll
        for (int q = 0; q < (DB ? 2 : 1); ++q)
#pragma unroll
          for (int b16 = 0; b16 < 8 * E::BYTES; b16 += 16) *(u32x4*)(d + q * LDS_TILE + b16) = (u32x4)(0u);
      }
    }
  };

  // conv_in (cond_unet_2d.py:127-129): the 3x3 conv over <= 3 fp32 NCHW channels is run as a 1x1 conv over 32 virtual
  // channels k = ci*9 + ky*3 + kx (27 real, rest zero) gathered here straight into the LDS tile; thread = pixel.
  auto stage_im2col = [&](unsigned char* buf) {
    if constexpr (KS == 1 && STRIDE == 1) {
      const int py = tid / TW, px = tid % TW;
      const int oy = y0 + py, ox = x0 + px;
      const float* src = (const float*)p.x0;
      float v[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) v[k] = 0.f;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            // UNCONDITIONAL loads from clamped (always valid) addresses, masked afterwards: guarded loads make the compiler wait
            // for each one before issuing the next (27 serial L2 round trips = most of this workgroup's lifetime)
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            const bool ok = ci < p.C0r && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
            const int cic = min(ci, p.C0r - 1), iyc = min(max(iy, 0), p.Hin - 1), ixc = min(max(ix, 0), p.Win - 1);
            const float xv = src[(((size_t)n * p.C0r + cic) * p.Hin + iyc) * p.Win + ixc];
            v[ci * 9 + ky * 3 + kx] = ok ? xv : 0.f;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = v[q * 8 + j];
        E::store(buf + tid * PITCH + q * 8 * E::BYTES, E::pack(o));
      }
    }
  };

  // ---- per-lane LDS read bases for the B (activation) fragments
  // RMAP (round 6, the 8 x 8 tile = 8 x 8 images): the four 8-pixel rows of fragment fi are tile rows {0, 4, 1, 5} + 2 fi instead of 4 fi + {0..3}.  A 16-lane pass
  // of the ds_read_b128 then covers halo-tile pixels 10 y + (0..7) and 10 (y + 4) + (0..7) = 16 different residues mod 16 (the 144-byte pixel pitch repeats
  // its banks every 16 pixels); rows y and y + 1 put pixels p and p + 16 into one pass: every pass ran twice (PMC: conflict cycles 0.64 of the LDS-active
  // cycles of this instantiation, by far the most conflicted kernel of the SD workloads).  Only the lane -> pixel map changes (here and in the epilogue).
  constexpr bool RMAP = PD_CONV_RMAP && TH == 8 && TW == 8 && KS == 3 && STRIDE == 1 && !STACK;
  auto frag_row = [&](int fi, int q) { return RMAP ? (q >> 1) + 4 * (q & 1) + 2 * fi : fi * RPF + q; };     // tile row of 8-pixel run q of fragment fi
  const int li = lane & 15, lg = lane >> 4;          // M16: row / column within a 16 x 16 operand, k group
  constexpr int TJ_STEP = (TW >= 32 ? 16 : IN_TW) * PITCH;   // M16: bytes from the first to the second 16-pixel run of a fragment
  int rbase[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int fi = wp * NF + f;
    if constexpr (M16) {
      // one VGPR: fragment f is a compile-time distance from fragment 0 (folded into the ds_read offset field)
      rbase[f] = f == 0 ? ((wp * NF * RPF) * IN_TW + li) * PITCH + lg * 8 * E::BYTES : 0;
    } else {
      const int py = frag_row(fi, r / TW), px = r % TW;
      const int ly = STACK ? py + 2 * (py >> 3) : py * STRIDE;      // STACK: two halo rows between images
      rbase[f] = (ly * IN_TW + px * STRIDE) * PITCH + h * 8 * E::BYTES;
    }
  }

  f32x16 acc[NCO][NF];                    // 32x32x16 form
  f32x4 acq[NCO][NF][4];                  // M16 form (the unused one of the two is dead code)
  const int main_ksteps = (TAIL ? p.n_main : p.nchunks) * KSTEPS;
  const int all_ksteps = main_ksteps + (TAIL ? p.n_tail * 2 : 0);
  // Weight fragments through a buffer resource: address = base + per-lane VGPR offset (fixed) + SGPR offset (k-step, tile):
  // the per-k-step address is scalar arithmetic (round 3: the 64-bit VGPR address add per load was 5 % of the kernel's
  // vector instructions), and a prefetch past the last k-step is out of range = zeros instead of a clamped index.
  const unsigned wfrag = 512u * E::BYTES;                                       // bytes of one 32 co x 16 k fragment block
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)(p.Cout_pad / 32) * (unsigned)all_ksteps * wfrag, 0x00020000);
  const unsigned wlane = (unsigned)lane * 8u * E::BYTES;
  const unsigned wtile = (unsigned)ct32 * (unsigned)all_ksteps * wfrag;         // this wave's (first) 32-co tile
  const unsigned wstep = 2u * (unsigned)all_ksteps * wfrag;                     // NCO = 2: bytes between this wave's two 32-co tiles
  auto load_w = [&](int c, int kstep) { return E::load_buf(rw, wlane, wtile + (unsigned)c * wstep + (unsigned)PD_WIDX(kstep) * wfrag); };

  // Weight (A) fragments live in a register ring of AR entries, prefetched AD k-steps ahead and CONTINUOUSLY across
  // chunk boundaries (a chunk's fragments are contiguous with the next chunk's), so L2 latency (~600-800 cycles under
  // load) is covered by AD x 4 MFMAs.  The ring index is static because AR divides KSTEPS.  The MFMA loop is kept
  // free of branches (prefetch index clamped, not guarded; have_next / wave_active are compile-time) so that it stays
  // ONE scheduling region with counted waits.
  constexpr int AR = (KS == 3 && TW == 8 && STRIDE == 1 && sizeof(T) == 2 && TP <= 128 && PD_CONV_AR8 > 0) ? (TP == 64 ? PD_CONV_AR8_64 : PD_CONV_AR8) : ((KSTEPS % 3 == 0) ? 3 : 2);
  static_assert(KSTEPS % AR == 0, "the ring index is static");
  constexpr int AD = AR - 1;
  Frag aring[AR][NCO];
  // M16: the two 16 co x 32 k operands of a tap, ring of two taps (tap t in entry t & 1, tap t + 1 prefetched at the start of tap t;
  // a chunk has 9 taps, so the entry prefetched last moves to entry 0 at the end of a chunk: 8 NCO register moves)
  static_assert(!M16 || NF % 2 == 0, "M16 walks a fragment pair per half-step");
  Frag aq[2][NCO][2];
  const unsigned wlane16 = (unsigned)(((lg >> 1) * 64 + (lg & 1) * 32 + li) * 16);
  auto load_w16 = [&](int c, int tap_index, int ti) {
    return E::load_buf(rw, wlane16 + (unsigned)ti * 256u, wtile + (unsigned)c * wstep + (unsigned)PD_WIDX(2 * tap_index) * wfrag);
  };

  // one chunk of MFMAs out of `buf`; when DB, pieces of the NEXT chunk are transformed + written to `nbuf` in between
  auto mma_chunk = [&](int chunk, const unsigned char* buf, unsigned char* nbuf, auto have_next_c, auto active_c) {
    constexpr int NEXT = (int)decltype(have_next_c)::value;       // 0: last chunk; 1: stage the next chunk; 2 (PRO): the next chunk is a fused-tail chunk (staged as it is)
    constexpr bool HAVE_NEXT = NEXT != 0, NEXT_PLAIN = NEXT == 2;
    constexpr bool ACTIVE = decltype(active_c)::value;
    const int g0 = chunk * KSTEPS;
    Frag bc[NF];                           // activation fragments of the current k-step (3 waves per SIMD cover the LDS latency)
    int piece = 0;
#ifndef PD_NO_IGLP
    if constexpr (!PRO) __builtin_amdgcn_iglp_opt(0);          // interleave the region's LDS reads / staging VALU work with the MFMAs (same-box A/B: -1 % per forward)
#endif
    if constexpr (PRO && ACTIVE) {
      // PRO: the chunk is ONE basic block whose instruction order is written out slot by slot (a slot = one MFMA = 32 matrix-pipe
      // cycles during which the vector issue port is free for ~24) and pinned by sched_barrier(0): after every MFMA the LDS read that
      // refills the fragment it just consumed with the NEXT k-step's (four slots ahead of its use), then this slot's share of the
      // staging work of the next chunk.  A piece (8 channels of one pixel) is 9 steps per channel pair -- unpack, affine, scale,
      // exp, exp, +1, rcp, rcp, multiply + pack -- walked two pairs at a time so that consecutive instructions are independent;
      // steps are dealt to the slots from PD_CONV_PRO_START on by their issue cost (transcendentals 8.7 cycles, packed fp32 5.8,
      // the rest ~3-5: profiles/r3_exp_variants.log).  Without this order every piece was a block of ~45 vector instructions BEHIND
      // its three k-steps of MFMAs: an in-order wave ran them one after the other (matrix and vector pipes co-executed in 18 % of the
      // matrix-busy cycles, r3_conv64_inst_mix.txt).
      constexpr int NM = NCO * NF;                    // MFMAs per k-step
      constexpr int NS = KSTEPS * NM;                 // MFMA slots per chunk
      constexpr int PS0 = PD_CONV_PRO_START < NS - 8 ? PD_CONV_PRO_START : 0;
      constexpr int NS16 = 2 * NS, PS16 = 2 * PS0;    // the 16x16x32 form: twice the slots of half the length
      f32x2 py[4], pw[4];
      u32x4 po;
      auto piece_op = [&](auto ic, auto kc) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, k = decltype(kc)::value;
        if constexpr (k < 36) {
          constexpr int half = k / 18, kk = k % 18, stg = kk >> 1, j = half * 2 + (kk & 1);
          if constexpr (stg == 0) { const uint32_t wj = Stage<T>::word(stage[i], j); float lo, hi; Pack16<T>::unpack(wj, lo, hi); pw[j].x = lo; pw[j].y = hi; }
          // (fp16: y stays fp32 here, the non-PRO path -- Stage<half_t>::xform_gs -- rounds y to fp16 before the sigmoid: the two forms of an
          //  fp16 layer agree to fp16 rounding, NOT bit for bit, and dispatch picks the form from B, W and PD_CONV_PRO; bf16 / fp32 are unaffected)
          else if constexpr (stg == 1) py[j] = pw[j] * (f32x2){sc[2 * j], sc[2 * j + 1]} + (f32x2){sh[2 * j], sh[2 * j + 1]};
          else if constexpr (stg == 2) pw[j] = py[j] * (f32x2)(-1.4426950408889634f);
          else if constexpr (stg == 3) pw[j].x = __builtin_amdgcn_exp2f(pw[j].x);
          else if constexpr (stg == 4) pw[j].y = __builtin_amdgcn_exp2f(pw[j].y);
          else if constexpr (stg == 5) pw[j] = pw[j] + (f32x2)(1.0f);
          else if constexpr (stg == 6) pw[j].x = __builtin_amdgcn_rcpf(pw[j].x);
          else if constexpr (stg == 7) pw[j].y = __builtin_amdgcn_rcpf(pw[j].y);
          else { const f32x2 o = py[j] * pw[j]; po[j] = Pack16<T>::pack(o.x, o.y); }
          // pin the step to its slot: the values are pure arithmetic, which instruction selection otherwise sinks to its user (the piece's LDS
          // store) -- an empty volatile asm that "modifies" the step's result keeps it above the slot's sched_barrier
          if constexpr (stg == 1) asm volatile("" : "+v"(py[j]));
          else if constexpr (stg == 8) { uint32_t w = po[j]; asm volatile("" : "+v"(w)); po[j] = w; }
          else asm volatile("" : "+v"(pw[j]));
        } else {
          const int pix = stage_pix(i);
          const bool ok = pix < NPIX && spix[i] >= 0;
          *(u32x4*)(ok ? nbuf + pix * PITCH + sub * 8 * E::BYTES : lds + NPIX * PITCH) = NEXT_PLAIN ? Stage<T>::raw(stage[i]) : po;
        }
      };
      // steps [lo, hi) of the chunk's staging work that slot S of NSL runs (a fused-tail chunk: only the NIT stores, spread evenly)
      auto run_ops = [&](auto s_c, auto nsl_c, auto ps_c) __attribute__((always_inline)) {
        constexpr int S = decltype(s_c)::value, NSL = decltype(nsl_c)::value, PS = decltype(ps_c)::value;
        if constexpr (NEXT_PLAIN) {
          constexpr int lo = (S * NIT) / NSL, hi = ((S + 1) * NIT) / NSL;
          pd_static_for<hi - lo>([&](auto gc) __attribute__((always_inline)) {
            piece_op(std::integral_constant<int, lo + decltype(gc)::value>{}, std::integral_constant<int, 36>{});
          });
        } else if constexpr (HAVE_NEXT) {
          constexpr int lo = S == 0 ? 0 : pro_ops_done(S - 1, NSL, PS, NIT), hi = pro_ops_done(S, NSL, PS, NIT);
          pd_static_for<hi - lo>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = lo + decltype(gc)::value;
            piece_op(std::integral_constant<int, g / PRO_OPS>{}, std::integral_constant<int, g % PRO_OPS>{});
          });
        }
      };
      if constexpr (!M16) {
#pragma unroll
        for (int f = 0; f < NF; ++f) bc[f] = E::load(buf + rbase[f]);                    // k-step 0: tap 0, first half
        __builtin_amdgcn_sched_barrier(0);
        pd_static_for<NS>([&](auto sc_) __attribute__((always_inline)) {
          constexpr int S = decltype(sc_)::value, ks = S / NM, m = S % NM, f = m / NCO, c = m % NCO;
          constexpr int tapn = (ks + 1) >> 1, sn = (ks + 1) & 1;
          constexpr int toffn = ((tapn / KS) * IN_TW + (tapn % KS)) * PITCH + sn * 16 * E::BYTES;
          if constexpr (m == 0) {
  #pragma unroll
            for (int cc = 0; cc < NCO; ++cc) aring[(ks + AD) % AR][cc] = load_w(cc, g0 + ks + AD);
          }
          acc[c][f] = E::mma(aring[ks % AR][c], bc[f], acc[c][f]);
          if constexpr (c == NCO - 1 && ks + 1 < KSTEPS) bc[f] = E::load(buf + rbase[f] + toffn);
          run_ops(sc_, std::integral_constant<int, NS>{}, std::integral_constant<int, PS0>{});
          __builtin_amdgcn_sched_barrier(0);
        });
  } else {
        // 16x16x32 form: a slot is one 16-cycle MFMA; half-step hs = (tap, pixel half s) = 16 NCO / 2 MFMAs over the tap's A operands
        // aq[tap & 1][c][ti] (the next tap's are loaded at s == 0) and the B operands bq[f][tj] of fragments 2 s, 2 s + 1 -- each
        // bq is consumed by NCO x 2 consecutive MFMAs and then refilled with the NEXT half-step's fragment (>= 6 slots ahead of its use)
        constexpr int FH = NF / 2;
        constexpr int MH = NCO * FH * 4;                // MFMAs per half-step
        static_assert(NS16 == KSTEPS * MH, "slot count");
        Frag bq[FH][2];
#pragma unroll
        for (int f = 0; f < FH; ++f)
#pragma unroll
          for (int tj = 0; tj < 2; ++tj) bq[f][tj] = E::load(buf + rbase[0] + f * (RPF * IN_TW * PITCH) + tj * TJ_STEP);
        __builtin_amdgcn_sched_barrier(0);
        pd_static_for<NS16>([&](auto sc_) __attribute__((always_inline)) {
          constexpr int S = decltype(sc_)::value, hs = S / MH, m = S % MH, tap = hs >> 1, s2 = hs & 1;
          // slot order inside a half-step: (f, tj) outer -- one B operand --, (c, ti) inner
          constexpr int f = m / (2 * NCO * 2), tj = (m / (NCO * 2)) % 2, c = (m / 2) % NCO, ti = m % 2;
          constexpr int tapn = (hs + 1) >> 1, sn = (hs + 1) & 1;
          constexpr int toffn = ((tapn / KS) * IN_TW + (tapn % KS)) * PITCH;
          if constexpr (m == 0 && s2 == 0) {
#pragma unroll
            for (int cc = 0; cc < NCO; ++cc)
#pragma unroll
              for (int t2 = 0; t2 < 2; ++t2) aq[(tap + 1) & 1][cc][t2] = load_w16(cc, chunk * TAPS + tap + 1, t2);
          }
          acq[c][s2 * FH + f][2 * ti + tj] = E::mma_16x16x32(aq[tap & 1][c][ti], bq[f][tj], acq[c][s2 * FH + f][2 * ti + tj]);
          if constexpr (c == NCO - 1 && ti == 1 && hs + 1 < KSTEPS)
            bq[f][tj] = E::load(buf + rbase[0] + (sn * FH + f) * (RPF * IN_TW * PITCH) + tj * TJ_STEP + toffn);
          run_ops(sc_, std::integral_constant<int, NS16>{}, std::integral_constant<int, PS16>{});
          __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int c = 0; c < NCO; ++c) { aq[0][c][0] = aq[1][c][0]; aq[0][c][1] = aq[1][c][1]; }      // (TAPS is odd: see the general form below)
      }
    } else
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if constexpr (ACTIVE && M16) {
        // half-step ks = (tap, pixel half s): the tap's two A operands x the four 16-pixel runs of fragments 2 s', 2 s' + 1
        const int tap = ks >> 1, s = ks & 1;
        if (s == 0) {
#pragma unroll
          for (int c = 0; c < NCO; ++c)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) aq[(tap + 1) & 1][c][ti] = load_w16(c, chunk * TAPS + tap + 1, ti);
        }
        constexpr int FH = NF >= 2 ? NF / 2 : 1;   // (NF = 1 tiles are never M16)
        const int toff = ((tap / KS) * IN_TW + (tap % KS)) * PITCH;
        Frag bq[FH][2];
#pragma unroll
        for (int f = 0; f < FH; ++f)
#pragma unroll
          for (int tj = 0; tj < 2; ++tj) bq[f][tj] = E::load(buf + rbase[0] + (s * FH + f) * (RPF * IN_TW * PITCH) + tj * TJ_STEP + toff);
        __builtin_amdgcn_s_setprio(PD_CONV_PRIO_BASE + PD_CONV_MFMA_PRIO);
#pragma unroll
        for (int c = 0; c < NCO; ++c)
#pragma unroll
          for (int f = 0; f < FH; ++f)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
              for (int tj = 0; tj < 2; ++tj)
                acq[c][s * FH + f][2 * ti + tj] = E::mma_16x16x32(aq[tap & 1][c][ti], bq[f][tj], acq[c][s * FH + f][2 * ti + tj]);
        __builtin_amdgcn_s_setprio(PD_CONV_PRIO_BASE);
      } else if constexpr (ACTIVE) {
#pragma unroll
        for (int c = 0; c < NCO; ++c) aring[(ks + AD) % AR][c] = load_w(c, g0 + ks + AD);
        {
          const int tap = ks >> 1, s = ks & 1;
          const int toff = ((tap / KS) * IN_TW + (tap % KS)) * PITCH + s * 16 * E::BYTES;
#pragma unroll
          for (int f = 0; f < NF; ++f) bc[f] = E::load(buf + rbase[f] + toff);
        }
        __builtin_amdgcn_s_setprio(PD_CONV_PRIO_BASE + PD_CONV_MFMA_PRIO);   // keeps the 4-MFMA cluster together and ahead of the other wave's VALU work (+5 %)
#pragma unroll
        for (int c = 0; c < NCO; ++c)
#pragma unroll
          for (int f = 0; f < NF; ++f) acc[c][f] = E::mma(aring[ks % AR][c], bc[f], acc[c][f]);
        __builtin_amdgcn_s_setprio(PD_CONV_PRIO_BASE);
      }
      if constexpr (DB && HAVE_NEXT) {
        // spread the NIT pieces of the next chunk evenly over the k-steps
        const int due = ((ks + 1) * NIT) / KSTEPS;
#pragma unroll
        for (int i = 0; i < NIT; ++i)
          if (i >= piece && i < due) write_piece(i, nbuf);
        piece = due;
      }
    }
    if constexpr (ACTIVE && M16 && TAPS % 2 == 1) {       // TAPS odd: the operands prefetched for the next chunk's tap 0 sit in entry 1
#pragma unroll
      for (int c = 0; c < NCO; ++c) { aq[0][c][0] = aq[1][c][0]; aq[0][c][1] = aq[1][c][1]; }
    }
  };
  // tail chunk (fused 1x1 shortcut): 2 k-steps at the centre tap; its two weight fragments are loaded up front
  auto mma_tail = [&](int chunk, const unsigned char* buf, unsigned char* nbuf, auto have_next_c, auto active_c) {
    constexpr bool HAVE_NEXT = decltype(have_next_c)::value;
    constexpr bool ACTIVE = decltype(active_c)::value;
    constexpr int CENTER = ((KS / 2) * IN_TW + (KS / 2)) * PITCH;
    if constexpr (ACTIVE && M16) {
      const int tt = (main_ksteps >> 1) + (chunk - p.n_main);          // tap-pair index of this tail chunk in the packed k order
      Frag ta[NCO][2];
#pragma unroll
      for (int c = 0; c < NCO; ++c) { ta[c][0] = load_w16(c, tt, 0); ta[c][1] = load_w16(c, tt, 1); }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        constexpr int FH = NF >= 2 ? NF / 2 : 1;   // (NF = 1 tiles are never M16)
        Frag bq[FH][2];
#pragma unroll
        for (int f = 0; f < FH; ++f)
#pragma unroll
          for (int tj = 0; tj < 2; ++tj) bq[f][tj] = E::load(buf + rbase[0] + (s * FH + f) * (RPF * IN_TW * PITCH) + tj * TJ_STEP + CENTER);
        __builtin_amdgcn_s_setprio(PD_CONV_PRIO_BASE + PD_CONV_MFMA_PRIO);
#pragma unroll
        for (int c = 0; c < NCO; ++c)
#pragma unroll
          for (int f = 0; f < FH; ++f)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
              for (int tj = 0; tj < 2; ++tj)
                acq[c][s * FH + f][2 * ti + tj] = E::mma_16x16x32(ta[c][ti], bq[f][tj], acq[c][s * FH + f][2 * ti + tj]);
        __builtin_amdgcn_s_setprio(PD_CONV_PRIO_BASE);
      }
    } else if constexpr (ACTIVE) {
      const int kt = main_ksteps + (chunk - p.n_main) * 2;
      Frag a0[NCO], a1[NCO];
#pragma unroll
      for (int c = 0; c < NCO; ++c) { a0[c] = load_w(c, kt); a1[c] = load_w(c, kt + 1); }
      Frag b0[NF], b1[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) { b0[f] = E::load(buf + rbase[f] + CENTER); b1[f] = E::load(buf + rbase[f] + CENTER + 16 * E::BYTES); }
      __builtin_amdgcn_s_setprio(PD_CONV_PRIO_BASE + PD_CONV_MFMA_PRIO);
#pragma unroll
      for (int c = 0; c < NCO; ++c) {
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[c][f] = E::mma(a0[c], b0[f], acc[c][f]);
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[c][f] = E::mma(a1[c], b1[f], acc[c][f]);
      }
      __builtin_amdgcn_s_setprio(PD_CONV_PRIO_BASE);
    }
    if constexpr (HAVE_NEXT) {
#pragma unroll
      for (int i = 0; i < NIT; ++i) write_piece(i, nbuf, true);
    }
  };
  // chunk driver: the last chunk is peeled (HAVE_NEXT = false) so the accumulators flow through two call sites
  // instead of an if/else diamond (which made the register allocator keep two accumulator sets)
  using std::true_type; using std::false_type;

  PD_STAMP(7);
  // bias / temb first: they are older than the HBM loads below in the in-order vmcnt queue, so initialising the
  // accumulators does not wait for the activation tile
  // Round 3: all of them issued back to back, ONE wait -- the per-group `if (temb) { load; add }` form made the compiler wait for
  // each (bias, temb) pair before issuing the next: 4 serial L2 round trips = 4.6k of a 64-channel workgroup's 37k cycles
  // (s_memtime stamps).  The time-embedding rows come through a buffer resource (no tensor -> zero records -> zeros; the padded
  // channels of a partial last tile lie beyond the resource and read zeros).
  f32x4 bt[NCO][4];
  u32x4 tv[NCO][4];
  if constexpr (M16) {
    // lane (i, g): registers r of quad (ti, tj) <-> co = 16 ti + 4 g + r: two (bias, temb) pairs per tile instead of four
    if (wave_active) {
      const __amdgpu_buffer_rsrc_t rtemb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.temb ? p.temb : p.bias), 0,
          p.temb ? ((unsigned)n * (unsigned)p.temb_stride + (unsigned)p.Cout) * 4u : 0u, 0x00020000);
#pragma unroll
      for (int c = 0; c < NCO; ++c)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          const int co = (ct32 + 2 * c) * 32 + 16 * ti + 4 * lg;
          bt[c][ti] = *(const f32x4*)(p.bias + co);
          tv[c][ti] = __builtin_amdgcn_raw_buffer_load_b128(rtemb, (unsigned)(n * p.temb_stride + co) * 4u, 0, 0);
        }
    }
  } else
  if (wave_active) {
    const int nt = STACK ? min(n + wp, p.B - 1) : n;     // STACK: this wave's fragments are the two halves of image n + wp (static_assert: two images)
    const __amdgpu_buffer_rsrc_t rtemb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.temb ? p.temb : p.bias), 0,
        p.temb ? ((unsigned)nt * (unsigned)p.temb_stride + (unsigned)p.Cout) * 4u : 0u, 0x00020000);   // up to the end of row n's slice
#pragma unroll
    for (int c = 0; c < NCO; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = (ct32 + 2 * c) * 32 + 8 * g + 4 * h;
        bt[c][g] = *(const f32x4*)(p.bias + co);
        tv[c][g] = __builtin_amdgcn_raw_buffer_load_b128(rtemb, (unsigned)(nt * p.temb_stride + co) * 4u, 0, 0);
      }
  }
  if (!p.im2col3) issue_loads(0);      // everything below overlaps the HBM latency of chunk 0
  if (wave_active) {
#pragma unroll
    for (int c = 0; c < NCO; ++c)
#pragma unroll
      for (int g = 0; g < (M16 ? 2 : 4); ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) bt[c][g][i] += __uint_as_float(tv[c][g][i]);
  }
  PD_STAMP(8);
  // accumulators start at bias[co] + temb[n][co] (lane (pixel, h), register i <-> co = 8(i>>2) + 4h + (i&3)):
  // the epilogue then has no per-channel loads at all
  if constexpr (M16) {
#pragma unroll
    for (int c = 0; c < NCO; ++c)
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) acq[c][f][q] = wave_active ? bt[c][q >> 1] : (f32x4)(0.f);
    if (wave_active) {
#pragma unroll
      for (int c = 0; c < NCO; ++c) { aq[0][c][0] = load_w16(c, 0, 0); aq[0][c][1] = load_w16(c, 0, 1); }
    }
  } else {
#pragma unroll
  for (int c = 0; c < NCO; ++c) {
    f32x16 init = (f32x16)(0.f);
    if (wave_active) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) init[4 * g + i] = bt[c][g][i];
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[c][f] = init;
  }
  if (wave_active) {
#pragma unroll
    for (int i = 0; i < AD; ++i)
#pragma unroll
      for (int c = 0; c < NCO; ++c) aring[i][c] = load_w(c, i);
  }
  }
  PD_STAMP(9);
  if (!p.im2col3) zero_padding();
  if (DB) {
    if (p.im2col3) {
      stage_im2col(lds);
    } else {
      write_piece(0, lds);
      PD_STAMP(10);
#pragma unroll
      for (int i = 1; i < NIT; ++i) write_piece(i, lds);
    }
    if (p.nchunks > 1) issue_loads(1);
    PD_STAMP(1);
    __syncthreads();
    PD_STAMP(2);
    int chunk = 0;
    // with a tail every main chunk has a successor; PRO: the last main chunk (which stages the first tail chunk as it is) is peeled below
    const int main_loop_end = TAIL ? (PRO ? p.n_main - 1 : p.n_main) : p.nchunks - 1;
    for (; chunk < main_loop_end; ++chunk) {
      unsigned char* buf = lds + (chunk & 1) * LDS_TILE;
      unsigned char* nbuf = lds + ((chunk + 1) & 1) * LDS_TILE;
      if constexpr (PRO) mma_chunk(chunk, buf, nbuf, true_type{}, true_type{});      // (PRO launches have no idle wave: Cout_pad % 64 == 0)
      else if (wave_active) mma_chunk(chunk, buf, nbuf, true_type{}, true_type{});
      else mma_chunk(chunk, buf, nbuf, true_type{}, false_type{});
      if (chunk + 2 < p.nchunks) issue_loads(chunk + 2);
      if (chunk == 0) PD_STAMP(3);
      __syncthreads();
      if (chunk == 0) PD_STAMP(4);
    }
    if constexpr (TAIL && PRO) {
      mma_chunk(chunk, lds + (chunk & 1) * LDS_TILE, lds + ((chunk + 1) & 1) * LDS_TILE, std::integral_constant<int, 2>{}, true_type{});
      if (chunk + 2 < p.nchunks) issue_loads(chunk + 2);
      __syncthreads();
      ++chunk;
    }
    if constexpr (TAIL) {
      for (; chunk + 1 < p.nchunks; ++chunk) {
        unsigned char* buf = lds + (chunk & 1) * LDS_TILE;
        unsigned char* nbuf = lds + ((chunk + 1) & 1) * LDS_TILE;
        if (wave_active) mma_tail(chunk, buf, nbuf, true_type{}, true_type{});
        else mma_tail(chunk, buf, nbuf, true_type{}, false_type{});
        if (chunk + 2 < p.nchunks) issue_loads(chunk + 2);
        __syncthreads();
      }
      if (wave_active) mma_tail(chunk, lds + (chunk & 1) * LDS_TILE, lds, false_type{}, true_type{});
    } else {
      if (wave_active) mma_chunk(chunk, lds + (chunk & 1) * LDS_TILE, lds, false_type{}, true_type{});
    }
    __syncthreads();
    PD_STAMP(5);
  } else {
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
      if (chunk > 0) __syncthreads();
#pragma unroll
      for (int i = 0; i < NIT; ++i) write_piece(i, lds);
      __syncthreads();
      if (chunk + 1 < p.nchunks) issue_loads(chunk + 1);
      if (wave_active) mma_chunk(chunk, lds, lds, false_type{}, true_type{});
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------------
  const int co_w = ct32 * 32;
  if (p.out_mode == PD_OUT_NCHW_F32) {
    // conv_out: <= 4 real channels, fp32 planes; lanes (pixels) are contiguous along x
    if (!wave_active) return;
    if constexpr (M16) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          const int fi = wp * NF + f, pp = tj * 16 + li;
          const int oy = y0 + fi * RPF + pp / TW, ox = x0 + pp % TW;
          if (oy >= p.Hout || ox >= p.Wout) continue;
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) {
            const int co = co_w + 16 * ti + 4 * lg;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (co + i < p.Cout) ((float*)p.y)[(((size_t)n * p.Cout + co + i) * p.Hout + oy) * p.Wout + ox] = acq[0][f][2 * ti + tj][i];
          }
        }
      return;
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int fi = wp * NF + f;
      const int oy = y0 + frag_row(fi, r / TW), ox = x0 + r % TW;
      if (oy >= p.Hout || ox >= p.Wout) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co_w + 8 * g + 4 * h;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (co + i < p.Cout) ((float*)p.y)[(((size_t)n * p.Cout + co + i) * p.Hout + oy) * p.Wout + ox] = acc[0][f][4 * g + i];
      }
    }
    return;
  }
  // NHWC / head-major: stage the 64-channel tile through LDS as [pixel][co] so that the residual add and the
  // stores are fully coalesced 16-byte accesses (a pixel's 64 channels = one 128-B line in bf16).
  constexpr int EP_PITCH = 64 * E::BYTES + 16;
  constexpr int EPC = 16 / E::BYTES;           // channels per 16-byte piece
  constexpr int PPP = 64 / EPC;                // pieces per pixel
  constexpr int PXI = 256 / PPP;               // pixels per iteration
  constexpr int NEP = TP / PXI;                // iterations (pieces per thread)
  static_assert(PXI % TW == 0 || TW % PXI == 0, "epilogue piece map");
  const int piece = tid % PPP, prow = tid / PPP;
  // Piece `it` of this thread is pixel (py0 + DPY(it), px0 + DPX(it)) of the tile with compile-time DPY / DPX (no carry between
  // them: PXI divides TW or the other way round), so its byte offset in the NHWC output is one 32-bit base + a scalar step --
  // round 3: the 64-bit multiplies per piece of the old address computation were a third of the epilogue's VALU time, and
  // the guarded residual load was waited for right after its issue (one exposed L2 / HBM round trip per piece: +10k cycles
  // of a 49k-cycle workgroup on the 64-channel 256^2 layers).  Residual pieces are now loaded up front, unconditionally,
  // through a buffer resource (invalid pixel / no residual tensor -> out-of-range offset -> zeros, no traffic); stores go
  // through a buffer resource too (out-of-range -> dropped), so the loop has no divergent branch.
  const int py0 = prow / TW, px0 = prow % TW;
  // (sub-pixel phase of an upsampling convolution: this launch writes every out_step-th pixel of a tensor out_step^2 times as large)
  const int os = p.out_step, HF = p.Hout * os, WF = p.Wout * os;
  const unsigned out_bytes = (unsigned)p.B * HF * WF * p.Cout * E::BYTES;     // < 2 GiB (checked by pd_conv)
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.out_mode == PD_OUT_NHWC ? out_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual ? p.residual : p.y), 0, p.residual ? out_bytes : 0u, 0x00020000);
  const unsigned px_bytes = (unsigned)p.Cout * E::BYTES * os, row_bytes = (unsigned)WF * p.Cout * E::BYTES * os;   // steps of one tile pixel / row
  auto piece_off = [&](int it, int co) -> unsigned {       // byte offset of piece `it` (or OOB_OFF) for the 64-channel tile at `co`
    const int dpy = (it * PXI) / TW, dpx = (it * PXI) % TW;
    const int oy = y0 + py0 + dpy, ox = x0 + px0 + dpx;
    const unsigned off = (unsigned)((n * HF + (y0 + py0) * os + p.out_oy) * WF + (x0 + px0) * os + p.out_ox) * (unsigned)(p.Cout * E::BYTES)
                         + (unsigned)co * E::BYTES + (unsigned)dpy * row_bytes + (unsigned)dpx * px_bytes;
    // STACK: tile row 8 j + y is row y of sample n + j, and the 8 x 8 NHWC images are contiguous over samples: the same offset formula
    if constexpr (STACK) return (n + ((py0 + dpy) >> 3) < p.B && co < p.Cout) ? off : OOB_OFF;
    return (oy < p.Hout && ox < p.Wout && co < p.Cout) ? off : OOB_OFF;
  };
  if (!DB) __syncthreads();                 // DB: the chunk loop already ended on a barrier
#pragma unroll
  for (int cth = 0; cth < NCO; ++cth) {     // NCO = 2: the two 64-channel tiles go through the same LDS staging area in turn
  const int co_tile = co_t * NCO + cth;
  const int co = co_tile * 64 + piece * EPC;
  u32x4 rr[NEP];
  if (p.out_mode == PD_OUT_NHWC) {
#pragma unroll
    for (int it = 0; it < NEP; ++it) rr[it] = __builtin_amdgcn_raw_buffer_load_b128(rres, piece_off(it, co), 0, 0);
  }
  if (cth > 0) __syncthreads();
  if (wave_active) {
    if constexpr (M16) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int plin = (wp * NF + f) * 32 + (q & 1) * 16 + li;
          const f32x4 v = acq[cth][f][q];
          store4((T*)(lds + plin * EP_PITCH) + wc * 32 + (q >> 1) * 16 + 4 * lg, v[0], v[1], v[2], v[3]);
        }
    } else {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int plin = RMAP ? frag_row(wp * NF + f, r / TW) * TW + r % TW : (wp * NF + f) * 32 + r;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        store4((T*)(lds + plin * EP_PITCH) + wc * 32 + 8 * g + 4 * h, acc[cth][f][4 * g], acc[cth][f][4 * g + 1], acc[cth][f][4 * g + 2], acc[cth][f][4 * g + 3]);
    }
    }
  }
  __syncthreads();
  float ssum[SG][EPC], ssq[SG][EPC];           // GroupNorm statistics of what is stored (consumer's norm input); STACK: per image of the tile
#pragma unroll
  for (int g = 0; g < SG; ++g)
#pragma unroll
    for (int j = 0; j < EPC; ++j) { ssum[g][j] = 0.f; ssq[g][j] = 0.f; }
  static_assert(!STACK || (NEP % SG == 0 && (NEP / SG) * PXI == 64), "a thread's pieces of one image are consecutive iterations");
  if (p.out_mode == PD_OUT_NHWC) {
#pragma unroll
    for (int it = 0; it < NEP; ++it) {
      const int plin = it * PXI + prow;
      const unsigned off = piece_off(it, co);
      u32x4 v = *(const u32x4*)(lds + plin * EP_PITCH + piece * 16);
      if (p.residual) {                        // kernel-uniform
        if constexpr (E::BYTES == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float lo, hi, rl, rh;
            Pack16<T>::unpack(v[j], lo, hi); Pack16<T>::unpack(rr[it][j], rl, rh);
            v[j] = Pack16<T>::pack(lo + rl, hi + rh);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = __float_as_uint(__uint_as_float(v[j]) + __uint_as_float(rr[it][j]));
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(v, ry, off, 0, 0);
      if (p.stats) {                           // kernel-uniform
        const int sg = STACK ? it / (NEP / SG) : 0;     // (compile-time: the loop is unrolled)
        if (off == OOB_OFF) v = (u32x4)(0u);   // pixels beyond the image / padded channels do not count
        if constexpr (E::BYTES == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float lo, hi;
            Pack16<T>::unpack(v[j], lo, hi);
            ssum[sg][2 * j] += lo; ssq[sg][2 * j] += lo * lo; ssum[sg][2 * j + 1] += hi; ssq[sg][2 * j + 1] += hi * hi;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float x = __uint_as_float(v[j]); ssum[sg][j] += x; ssq[sg][j] += x * x; }
        }
      }
    }
  } else if (co < p.Cout) {  // PD_OUT_QKV_HEADS: [which][B][heads][N][8]
#pragma unroll
    for (int it = 0; it < NEP; ++it) {
      const int plin = it * PXI + prow;
      const int oy = y0 + plin / TW, ox = x0 + plin % TW;
      if (oy >= p.Hout || ox >= p.Wout) continue;
      const u32x4 v = *(const u32x4*)(lds + plin * EP_PITCH + piece * 16);
      const int Cq = p.heads * 8;
      const int which = co / Cq, cc = co - which * Cq;
      const size_t N = (size_t)p.Hout * p.Wout;
      const size_t tok = (size_t)oy * p.Wout + ox;
      *(u32x4*)((T*)p.y + ((((size_t)which * p.B + n) * p.heads + (cc >> 3)) * N + tok) * 8 + (cc & 7)) = v;
    }
  }
  if (p.stats) {   // kernel-uniform
    // per-thread partials -> LDS [thread][2*EPC]; then (channel, sum|sumsq) threads add the PXI pixel-rows in a fixed
    // order (deterministic, conflict-free: for one prow the 128 readers cover 512 contiguous bytes)
    float* red = (float*)(lds + TP * EP_PITCH);
#pragma unroll
    for (int sg = 0; sg < SG; ++sg) {            // STACK: one reduction per image of the tile (sample n + sg)
      if (sg > 0) __syncthreads();
#pragma unroll
      for (int q = 0; q < 2 * EPC / 4; ++q) {
        f32x4 v4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int e = q * 4 + j; v4[j] = e < EPC ? ssum[sg][e] : ssq[sg][e - EPC]; }
        *(f32x4*)(red + tid * (2 * EPC) + q * 4) = v4;
      }
      __syncthreads();
      if (tid < 128) {
        const int c = tid >> 1, which = tid & 1;
        const int pc = c / EPC, j = c % EPC;
        const int cog = co_tile * 64 + c;
        float tot = 0.f;
#pragma unroll 8
        for (int pr = 0; pr < PXI; ++pr) tot += red[(pr * PPP + pc) * (2 * EPC) + which * EPC + j];
        if (cog < p.Cout && n + sg < p.B) {
          const int tile = p.stat_tile_base + ty * p.tiles_x + tx;
          p.stats[(((size_t)(n + sg) * p.stat_tiles + tile) * p.Cout + cog) * 2 + which] = tot;
        }
      }
    }
  }
  }   // cth
  PD_STAMP(6);
}

template <typename T, int KS, int STRIDE, int TH, int TW, bool TAIL = false, int NCO = 1, bool PLAIN = false, int PRO = 0, bool STACK = false>
static int launch_conv(const ConvP& p, hipStream_t st) {
  constexpr int IN_TH = STACK ? (TH / 8) * 10 : (TH - 1) * STRIDE + KS, IN_TW = (TW - 1) * STRIDE + KS;
  constexpr int PITCH = 32 * Elem<T>::BYTES + 16;
  constexpr int LDS_TILE = ((IN_TH * IN_TW * PITCH + 15) / 16) * 16;
  // double-buffer when two tiles fit comfortably -- and leave room for a second workgroup: the stride-2 halo tile (9 x 65 pixels)
  // double-buffered is 84 KB = ONE workgroup (4 waves) per CU; single-buffered 47 KB admits two, which overlap each other
  constexpr bool DB = 2 * LDS_TILE <= 100 * 1024 && !(STRIDE == 2 && PD_S2_SINGLE);
  constexpr bool M16 = PD_CONV_M16 && (PLAIN || (PD_CONV_M16_GN && (NCO == 1 || PD_CONV_M16_GN >= 2)) || (PRO && PD_CONV_PRO_M16 >= (NCO == 2 ? 1 : 2))) && sizeof(T) == 2 && (KS == 3 || KS == 2) && STRIDE == 1 && TW >= 16;        // (conv_kernel: 160-byte pixels)
  constexpr int LDS_DB = ((IN_TH * IN_TW * (2 * 32 * Elem<T>::BYTES + (M16 ? 32 : 16)) + 15) / 16) * 16;   // interleaved buffers, one shared pad
  constexpr int EPI_BYTES = TH * TW * (64 * Elem<T>::BYTES + 16) + 256 * 64;   // output tile + stats scratch [256][2*EPC] fp32
  constexpr int LDS_MAIN = (DB ? LDS_DB : LDS_TILE) + (PRO ? 16 : 0);          // PRO: + the dump slot behind the tile
  constexpr int LDS_BYTES = LDS_MAIN > EPI_BYTES ? LDS_MAIN : EPI_BYTES;
  static_assert(LDS_BYTES <= 160 * 1024, "tile too large");
  static_assert(NCO == 1 || DB, "NCO = 2 is a double-buffered variant");
  auto kern = conv_kernel<T, KS, STRIDE, TH, TW, DB, TAIL, NCO, PLAIN, PRO, STACK>;
  if (LDS_BYTES > 64 * 1024) {
    static LdsAttr attr;   // per instantiation, per device
    if (!ensure_lds(attr, kern, LDS_BYTES)) { set_error("pd_conv: cannot reserve %d bytes of LDS", LDS_BYTES); return PD_ERR_LAUNCH; }
  }
  ConvP q = p;
  q.tiles_x = (p.Wout + TW - 1) / TW;
  q.tiles_y = STACK ? 1 : (p.Hout + TH - 1) / TH;
  q.tiles_x_shift = -1;
  for (int sft = 0; sft < 16; ++sft) if ((1 << sft) == q.tiles_x) q.tiles_x_shift = sft;
  q.n_co_tiles = (p.Cout_pad + 63) / 64;
  if (q.stat_tiles == 0) q.stat_tiles = q.tiles_x * q.tiles_y;                 // ordinary launch: the statistic tiles are this launch's tiles
  else { q.stat_tiles *= q.tiles_x * q.tiles_y; q.stat_tile_base *= q.tiles_x * q.tiles_y; }      // phase launch: (phases, phase index) so far
  {
    // channel-tile-major XCD order: OPT-IN (PD_CONV_XCD=1: whenever the grid has >= 64 workgroups; 2: where the weights outweigh the input
    // activations).  Measured (docs/LAB_r6.md section 9, same box): with rule 2 the latent-diffusion trajectory is 1.1 % SLOWER (9.686 vs
    // 9.789 images/s) and the fine-tuning step 0.8 % slower -- the weights the XCDs re-read come out of the Infinity Cache cheaply, the
    // activations every XCD then has to read do not pay for it.  Default: the sample-major order of rounds 1-5.
    const long long es = (long long)sizeof(T), cin = (long long)p.C0 + p.C1;
    const long long wbytes = (long long)p.Cout_pad * cin * KS * KS * es, abytes = (long long)p.B * p.Hin * p.Win * cin * es;
    const long long blocks = (long long)q.tiles_x * q.tiles_y * (q.n_co_tiles / NCO) * p.B;
    const int env = diag_env("PD_CONV_XCD", 0);
    q.co_major = !STACK && blocks >= 64 && blocks < (1ll << 30) && !p.im2col3 && (env == 1 || (env == 2 && wbytes > abytes));
  }
  hipLaunchKernelGGL(kern, dim3(q.tiles_x * q.tiles_y, q.n_co_tiles / NCO, STACK ? (p.B + TH / 8 - 1) / (TH / 8) : p.B), dim3(256), LDS_BYTES, st, q);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

// tile shape by output width: 32-wide rows when possible (bank-conflict-free), else squarer tiles
static void tile_shape(int ksize, int stride, int hout, int wout, int* th, int* tw) {
  const int tp = (ksize == 3 && stride == 1 && wout <= 8 && hout <= 8) ? 64
                 : ((ksize == 3 && stride == 2) || (ksize == 3 && stride == 1 && wout < 16 && hout <= 16)) ? 128 : 256;
  *tw = wout >= 32 ? 32 : (wout >= 16 ? 16 : 8);
  *th = tp / *tw;
}

template <typename T>
static int dispatch_conv(const ConvP& p, int ksize, int stride, hipStream_t st) {
  // tile shape by output width: 32-wide rows when possible (bank-conflict-free), else squarer tiles
  const int w = p.Wout;
  if (ksize == 3 && stride == 1) {
    // 8-wide images (the SD UNet's innermost level): a 16 x 8 tile wastes half of an 8 x 8 image's tile instead of three quarters
    const bool tiny = p.Hout <= 16;
    // Two 64-channel output tiles per workgroup (NCO = 2; 16-bit engines, NHWC / head-major output, Cout a multiple of 128) when
    // the estimate below says so: workgroups run in rounds of (resident workgroups per CU) x CUs -- three per CU for NCO = 1,
    // two for NCO = 2 -- and the last round is as long as a full one; per MFMA the NCO = 2 form is ~5 % faster behind a
    // GroupNorm prologue (the transform and the LDS fragment reads are shared by two tiles) and ~16 % slower without one
    // (same-box: 256 -> 256 @64^2 0.219 -> 0.185 ms, 128 -> 128 @128^2 0.234 -> 0.197, 512 -> 256 0.403 -> 0.343; no prologue
    // 0.184 -> 0.194; 64 -> 128 @128^2 0.110 -> 0.116: K too small).  PD_CONV_NCO=1 / 2: diagnostic override (same-box A/B).
    if constexpr (sizeof(T) == 2) {
      const int nco_env = diag_env("PD_CONV_NCO", 0);
      static int cus_of[16] = {0};      // per device (a process that drives several GPUs: each its own count; benign race: same value)
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
      if (cus_of[dev] == 0) {
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cus_of[dev] = c;
      }
      const int cus = cus_of[dev];
      bool nco2 = false;
      if (w >= 32 && p.Cout_pad % 128 == 0 && p.Cout == p.Cout_pad && p.out_mode != PD_OUT_NCHW_F32) {
        const long long wgs1 = (long long)((p.Wout + 31) / 32) * ((p.Hout + 7) / 8) * (p.Cout_pad / 64) * p.B, wgs2 = wgs1 / 2;
        auto eff = [](long long wgs, long long slots) { return (double)wgs / (double)(((wgs + slots - 1) / slots) * slots); };
        const double speed2 = (p.scale != nullptr && p.C0 + p.C1 >= 128) ? 1.05 : 0.84;
        nco2 = nco_env ? nco_env == 2 : eff(wgs2, 2ll * cus) * speed2 > eff(wgs1, 3ll * cus);
      }
      // no GroupNorm / SiLU prologue (input gradients, the SD UNet's convolutions, upsamplers): the PLAIN instantiations = 16x16x32 MFMAs
      const bool plain_off = diag_env("PD_CONV_PLAIN", 1) == 0;      // diagnostic: same-box A/B
      const bool plain = !plain_off && p.scale == nullptr && !p.silu && !p.im2col3 && w >= 16;
      // (the two-tile form keeps the 32x32x16 MFMAs: its 16x16x32 form needs 32 registers of A operands and spills at 256)
      // GroupNorm + SiLU prologue known at compile time (PRO): branch-free staging interleaved with the MFMAs.  PD_CONV_PRO=0 / 1 / 2: diagnostic override (none / the two-tile form / the one-tile form too)
      const int pro_lvl = diag_env("PD_CONV_PRO", PD_CONV_PRO);
      const bool gs = p.scale != nullptr && p.silu != 0 && !p.im2col3 && w >= 32 && p.Cout_pad % 64 == 0 && p.out_mode != PD_OUT_NCHW_F32;
      if (nco2 && gs && pro_lvl >= 1)
        return p.n_tail > 0 ? launch_conv<T, 3, 1, 8, 32, true, 2, false, 1>(p, st) : launch_conv<T, 3, 1, 8, 32, false, 2, false, 1>(p, st);
#ifdef PD_CONV_PRO_NCO1     // diagnostic builds only (-DPD_CONV_PRO_NCO1, then PD_CONV_PRO=2): the one-tile form under PRO is +-0.5 % and spills 28-64 B -- not in the shipped library
      if (!nco2 && gs && pro_lvl >= 2)
        return p.n_tail > 0 ? launch_conv<T, 3, 1, 8, 32, true, 1, false, 1>(p, st) : launch_conv<T, 3, 1, 8, 32, false, 1, false, 1>(p, st);
#endif
      if (nco2) return p.n_tail > 0 ? launch_conv<T, 3, 1, 8, 32, true, 2>(p, st) : launch_conv<T, 3, 1, 8, 32, false, 2>(p, st);
      if (plain) {
        if (p.n_tail > 0) return w >= 32 ? launch_conv<T, 3, 1, 8, 32, true, 1, true>(p, st) : launch_conv<T, 3, 1, 16, 16, true, 1, true>(p, st);
        return w >= 32 ? launch_conv<T, 3, 1, 8, 32, false, 1, true>(p, st) : launch_conv<T, 3, 1, 16, 16, false, 1, true>(p, st);
      }
    }
    // images of at most 8 x 8 pixels (the SD UNet's innermost level at 512 x 512): an 8 x 8 tile = two 32-pixel fragments, one per
    // wave pair -- the 16 x 8 tile spent half of its MFMAs on rows below the image (round 3: 1 280 -> 1 280 @8x8 417 TF/s)
    const bool one8 = p.Hout <= 8 && w <= 8;
    // 8 x 8 IMAGES (the SD UNet's innermost level at 512 x 512): two samples per 128-pixel tile (STACK), see conv_kernel.  OPT-IN (PD_CONV_STACK=1): as an op
    // 5-10 % faster than the 64-pixel tile with cold weights (94 -> 84-89 us, 170 -> 161 us), on the whole workloads neutral (same box: SD img2img
    // 10.243 vs 10.235 images/s, fine-tuning 243.75 vs 243.73 samples/s; profiles/r6_ab_conv8_stack_*.log) -- default: the 64-pixel tile
    const bool stack = p.Hout == 8 && w == 8 && p.Hin == 8 && p.Win == 8 && !p.upsample && p.in_step == 1 && p.out_step == 1 && p.pad == 1 && p.pad_x == 1 &&
                       p.scale == nullptr && !p.silu && !p.im2col3 && p.out_mode == PD_OUT_NHWC && p.B >= 2 && diag_env("PD_CONV_STACK", 0) != 0;
    if (stack) return p.n_tail > 0 ? launch_conv<T, 3, 1, 16, 8, true, 1, false, 0, true>(p, st) : launch_conv<T, 3, 1, 16, 8, false, 1, false, 0, true>(p, st);
    if (p.n_tail > 0) {
      if (w >= 32) return launch_conv<T, 3, 1, 8, 32, true>(p, st);
      if (w >= 16) return launch_conv<T, 3, 1, 16, 16, true>(p, st);
      if (one8) return launch_conv<T, 3, 1, 8, 8, true>(p, st);
      if (tiny) return launch_conv<T, 3, 1, 16, 8, true>(p, st);
      return launch_conv<T, 3, 1, 32, 8, true>(p, st);
    }
    if (w >= 32) return launch_conv<T, 3, 1, 8, 32>(p, st);
    if (w >= 16) return launch_conv<T, 3, 1, 16, 16>(p, st);
    if (one8) return launch_conv<T, 3, 1, 8, 8>(p, st);
    if (tiny) return launch_conv<T, 3, 1, 16, 8>(p, st);
    return launch_conv<T, 3, 1, 32, 8>(p, st);
  }
  if (ksize == 3 && stride == 2) {
    if (w >= 32) return launch_conv<T, 3, 2, 4, 32>(p, st);
    if (w >= 16) return launch_conv<T, 3, 2, 8, 16>(p, st);
    return launch_conv<T, 3, 2, 16, 8>(p, st);
  }
  if (ksize == 2 && stride == 1) {     // a phase of the sub-pixel upsampling convolution: always without a prologue (PLAIN: 16x16x32 MFMAs in the 16-bit engines)
    if (w >= 32) return launch_conv<T, 2, 1, 8, 32, false, 1, true>(p, st);
    if (w >= 16) return launch_conv<T, 2, 1, 16, 16, false, 1, true>(p, st);      // (the SD UNet's 16 -> 32 upsampler: 1 280 channels)
    set_error("pd_conv: the 2x2 phase form needs Wout >= 16");
    return PD_ERR_UNSUPPORTED;
  }
  if (ksize == 1 && stride == 1) {
    if (w >= 32) return launch_conv<T, 1, 1, 8, 32>(p, st);
    if (w >= 16) return launch_conv<T, 1, 1, 16, 16>(p, st);
    return launch_conv<T, 1, 1, 32, 8>(p, st);
  }
  set_error("pd_conv: unsupported ksize=%d stride=%d", ksize, stride);
  return PD_ERR_UNSUPPORTED;
}

}  // namespace pd

extern "C" int pd_conv(const pd_conv_args* a, void* stream) {
  using namespace pd;
  PD_CHECK(a != nullptr, PD_ERR_ARG, "pd_conv: null args");
  PD_CHECK(a->dtype == PD_F32 || a->dtype == PD_BF16 || a->dtype == PD_F16, PD_ERR_ARG, "pd_conv: bad dtype %d", a->dtype);
  PD_CHECK(a->B > 0 && a->Hin > 0 && a->Win > 0 && a->Hout > 0 && a->Wout > 0, PD_ERR_SHAPE, "pd_conv: bad spatial shape");
  PD_CHECK(a->C0 > 0 && a->C0 % 32 == 0 && a->C1 >= 0 && a->C1 % 32 == 0, PD_ERR_SHAPE,
           "pd_conv: C0=%d C1=%d must be multiples of 32", a->C0, a->C1);
  PD_CHECK(a->Cout > 0 && a->Cout_pad >= a->Cout && a->Cout_pad % 32 == 0, PD_ERR_SHAPE, "pd_conv: bad Cout/Cout_pad");
  PD_CHECK(a->Cout % 8 == 0 || a->out_mode == PD_OUT_NCHW_F32, PD_ERR_SHAPE, "pd_conv: NHWC / head-major output needs Cout %% 8 == 0");
  PD_CHECK(a->x0 && a->w_packed && a->bias && a->y, PD_ERR_ARG, "pd_conv: null pointer");
  PD_CHECK((a->C1 == 0) == (a->x1 == nullptr), PD_ERR_ARG, "pd_conv: x1/C1 mismatch");
  PD_CHECK((a->scale == nullptr) == (a->shift == nullptr), PD_ERR_ARG, "pd_conv: scale/shift mismatch");
  PD_CHECK(a->phase >= 0 && a->phase <= 4, PD_ERR_ARG, "pd_conv: phase %d", a->phase);
  PD_CHECK(a->phase_in == 0 || (a->phase_in == 1 && a->phase), PD_ERR_ARG, "pd_conv: phase_in %d without a phase", a->phase_in);
  if (a->phase) {
    PD_CHECK(a->ksize == 2 && a->stride == 1 && !a->upsample && !a->scale && !a->silu && !a->tail_x0 && !a->im2col3 && (!a->residual || a->phase_in) &&
             a->out_mode == PD_OUT_NHWC && a->Hout == a->Hin && a->Wout == a->Win && a->C1 == 0 && !(a->phase_in && a->stats_out), PD_ERR_UNSUPPORTED,
             "pd_conv: a sub-pixel phase is a plain 2x2 convolution over one source with Hout = Hin, Wout = Win and NHWC output (residual only with phase_in, statistics only without)");
    PD_CHECK((size_t)a->B * a->Hout * a->Wout * 4 * (a->phase_in ? a->C0 : a->Cout) * (a->dtype == PD_F32 ? 4 : 2) < 0x80000000ull, PD_ERR_SHAPE,
             "pd_conv: the upsampled NHWC tensor exceeds 2 GiB (32-bit buffer offsets); split the batch");
  }
  PD_CHECK(a->ksize == 1 || a->ksize == 3 || (a->ksize == 2 && a->phase), PD_ERR_UNSUPPORTED, "pd_conv: ksize %d", a->ksize);
  PD_CHECK(a->stride == 1 || (a->stride == 2 && a->ksize == 3), PD_ERR_UNSUPPORTED, "pd_conv: stride %d", a->stride);
  PD_CHECK(!(a->upsample && a->stride != 1) && a->upsample >= 0 && a->upsample <= 3, PD_ERR_UNSUPPORTED, "pd_conv: upsample %d with stride %d", a->upsample, a->stride);
  if (!a->phase) {
    const int hc = a->upsample ? 2 * a->Hin : a->Hin, wc = a->upsample ? 2 * a->Win : a->Win;
    const int extra = (a->ksize == 3 && a->pad == 0) ? 1 : 0;   // asymmetric (0,1,0,1) zero pad of Downsample2D(padding=0)
    const int ho = (hc + 2 * a->pad + extra - a->ksize) / a->stride + 1;
    const int wo = (wc + 2 * a->pad + extra - a->ksize) / a->stride + 1;
    PD_CHECK(ho == a->Hout && wo == a->Wout, PD_ERR_SHAPE, "pd_conv: Hout/Wout (%d,%d) != expected (%d,%d)", a->Hout, a->Wout, ho, wo);
  }
  if (a->out_mode == PD_OUT_QKV_HEADS)
    PD_CHECK(a->heads > 0 && a->Cout == 3 * a->heads * 8, PD_ERR_SHAPE, "pd_conv: QKV mode needs Cout == 3*heads*8");
  PD_CHECK(a->out_mode == PD_OUT_NHWC || a->residual == nullptr, PD_ERR_UNSUPPORTED, "pd_conv: residual needs NHWC output");
  if (a->im2col3) {
    PD_CHECK(a->im2col3 <= 3 && a->ksize == 1 && a->C0 == 32 && a->C1 == 0 && !a->scale && !a->upsample && a->pad == 0,
             PD_ERR_UNSUPPORTED, "pd_conv: im2col3 mode needs ksize=1, C0=32 (27 real), <= 3 source channels, no GroupNorm");
  }
  PD_CHECK(!a->stats_out || a->out_mode == PD_OUT_NHWC, PD_ERR_UNSUPPORTED, "pd_conv: stats_out needs NHWC output");
  if (a->tail_x0) {
    PD_CHECK(a->ksize == 3 && a->stride == 1 && !a->upsample && !a->im2col3, PD_ERR_UNSUPPORTED,
             "pd_conv: fused 1x1 tail needs a 3x3 stride-1 conv");
    PD_CHECK(a->tail_C0 > 0 && a->tail_C0 % 32 == 0 && a->tail_C1 >= 0 && a->tail_C1 % 32 == 0 && ((a->tail_C1 == 0) == (a->tail_x1 == nullptr)),
             PD_ERR_SHAPE, "pd_conv: tail channels must be multiples of 32");
    PD_CHECK((size_t)a->B * a->Hin * a->Win * (a->tail_C0 > a->tail_C1 ? a->tail_C0 : a->tail_C1) * (a->dtype == PD_F32 ? 4 : 2) < 0x80000000ull,
             PD_ERR_SHAPE, "pd_conv: tail tensor exceeds 2 GiB");
  } else {
    PD_CHECK(a->tail_C0 == 0 && a->tail_C1 == 0 && a->tail_x1 == nullptr, PD_ERR_ARG, "pd_conv: tail_C without tail_x0");
  }
  const size_t esz = a->dtype == PD_F32 ? 4 : 2;
  const size_t bytes0 = (size_t)a->B * a->Hin * a->Win * a->C0 * esz * (a->phase_in ? 4 : 1), bytes1 = (size_t)a->B * a->Hin * a->Win * a->C1 * esz;
  PD_CHECK(bytes0 < 0x80000000ull && bytes1 < 0x80000000ull, PD_ERR_SHAPE,
           "pd_conv: source tensor exceeds 2 GiB (32-bit buffer offsets); split the batch");
  PD_CHECK(a->out_mode != PD_OUT_NHWC || (size_t)a->B * a->Hout * a->Wout * a->Cout * esz < 0x80000000ull, PD_ERR_SHAPE,
           "pd_conv: NHWC output exceeds 2 GiB (32-bit buffer offsets); split the batch");
  ConvP p{};
  p.B = a->B; p.Hin = a->Hin; p.Win = a->Win; p.Hout = a->Hout; p.Wout = a->Wout;
  p.C0 = a->C0; p.C1 = a->C1; p.Cout = a->Cout; p.Cout_pad = a->Cout_pad;
  p.pad = a->pad; p.upsample = a->upsample; p.silu = a->silu; p.out_mode = a->out_mode; p.heads = a->heads;
  p.n_main = (a->C0 + a->C1) / 32;
  p.n_tail = (a->tail_C0 + a->tail_C1) / 32;
  p.nchunks = p.n_main + p.n_tail;
  p.Ct0 = a->tail_C0; p.Ct1 = a->tail_C1; p.t0 = a->tail_x0; p.t1 = a->tail_x1;
  p.tbytes0 = (unsigned)((size_t)a->B * a->Hin * a->Win * a->tail_C0 * esz);
  p.tbytes1 = (unsigned)((size_t)a->B * a->Hin * a->Win * a->tail_C1 * esz);
  p.bytes0 = (unsigned)bytes0; p.bytes1 = (unsigned)(a->x1 ? bytes1 : bytes0);
  p.x0 = a->x0; p.x1 = a->x1; p.scale = a->scale; p.shift = a->shift; p.w = a->w_packed; p.bias = a->bias;
  p.temb = a->temb; p.temb_stride = a->temb_stride; p.residual = a->residual; p.y = a->y;
  p.stats = a->stats_out; p.im2col3 = a->im2col3 ? 1 : 0; p.C0r = a->im2col3;
  p.pad_x = a->pad; p.out_step = 1; p.in_step = 1;
  if (a->phase) {          // phase 1 + 2 a + b: rows start at oy - (1 - a), columns at ox - (1 - b)
    const int pa = (a->phase - 1) >> 1, pb = (a->phase - 1) & 1;
    if (a->phase_in) {     // input gradient of that phase: dense output, rows start at oy - a in the phase's sub-image of the source
      p.pad = pa; p.pad_x = pb; p.in_step = 2; p.in_oy = pa; p.in_ox = pb;
    } else {
      p.pad = 1 - pa; p.pad_x = 1 - pb; p.out_step = 2; p.out_oy = pa; p.out_ox = pb;
      p.stat_tiles = 4; p.stat_tile_base = a->phase - 1;        // (launch_conv scales both by the tiles per phase)
    }
  }
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == PD_F32) return dispatch_conv<float>(p, a->ksize, a->stride, st);
  if (a->dtype == PD_F16) return dispatch_conv<half_t>(p, a->ksize, a->stride, st);
  return dispatch_conv<bf16_t>(p, a->ksize, a->stride, st);
}

#ifdef PD_STAMPS
extern "C" int pd_debug_conv_occupancy(int lds_bytes) {
  int nb = -1;
  const void* kerns[2] = {(const void*)pd::conv_kernel<pd::bf16_t, 3, 1, 8, 32, true, false>,
                          (const void*)pd::conv_kernel<pd::bf16_t, 3, 1, 8, 32, true, false, 1, true>};      // GroupNorm-prologue form; PLAIN (16x16x32, 160-byte pixels: 54 400 B)
  for (int k = 0; k < 2; ++k) {
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kerns[k], 256, (size_t)lds_bytes);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, kerns[k]);
    printf("occupancy API (%s): %d blocks/CU at %d B dyn LDS (err %d); numRegs %d sharedStatic %zu maxDyn %d\n", k ? "PLAIN" : "GN", nb, lds_bytes, (int)e,
           fa.numRegs, fa.sharedSizeBytes, fa.maxDynamicSharedSizeBytes);
  }
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("device: sharedMemPerBlock %zu sharedMemPerMultiprocessor %zu regsPerBlock %d CUs %d maxThreadsPerMP %d\n", pr.sharedMemPerBlock,
         pr.sharedMemPerMultiprocessor, pr.regsPerBlock, pr.multiProcessorCount, pr.maxThreadsPerMultiProcessor);
  return nb;
}
extern "C" int pd_debug_read_conv_stamps(unsigned long long* host, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pd::pd_conv_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int pd_conv_stat_tiles(int Hout, int Wout, int ksize, int stride) {
  int th, tw;
  pd::tile_shape(ksize, stride, Hout, Wout, &th, &tw);
  return ((Hout + th - 1) / th) * ((Wout + tw - 1) / tw);
}
