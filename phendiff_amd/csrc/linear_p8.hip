// linear_p8_kernel (round 6): Y[m][n] = sum_k X[m][k] W[n][k] + bias[n] (+ R[m][n] | GEGLU) for the big nn.Linear layers of the
// latent-diffusion UNet (BasicTransformerBlock's FeedForward / attention projections and their input gradients:
// custom_pipeline_stable_diffusion_img2img.py:680-686, utils_training.py:459-496) on a 256-token x 256-channel workgroup tile with
// the K loop cut into four phases per 64-channel K tile (eight per pair of LDS buffers), the structure of the programming guide's
// 256 x 256 template restated for this library's operand formats:
//   * 8 waves = 2 groups (g = wave / 4) x 4 channel columns (wc = wave % 4); a wave owns 128 tokens x 64 channels as four quadrants
//     (token half xh) x (channel half wh): tokens 128 xh + 64 g + (0..63), channels of packed 32-channel tile 2 wc + wh.  128
//     accumulator registers (32 16x16 tiles), v_mfma_f32_16x16x32 (the shape the chip clocks higher on, MI355X_MICROARCH "DVFS give-back" 7).
//   * every K tile is staged as four 16-KiB half tiles [W lo | X lo | W hi | X hi] by LDS-DMA (global_load_lds_dwordx4, 2 per wave and
//     half tile, no registers, no ds_write): W lo / W hi = the EVEN / ODD packed weight tiles of the 256 channels in pd_conv's
//     fragment order (1-KiB lane-linear fragments of 32 channels x 16 k: a 16x16x32 A fragment reads four 256-byte runs of it,
//     conflict-free; even / odd = the (value, gate) pairing of the fused GEGLU), X lo / X hi = 128 tokens x 64 k, 128-byte rows with
//     the 16-byte slots XOR-swizzled through the SOURCE address (slot q of row r holds source slot q ^ (r & 7)): conflict-free
//     ds_read_b128 for the 16-token B fragments.
//   * phase p of a K tile: P1 reads W lo (4 ds_read_b128) + X lo (8), multiplies quadrant (lo, lo); P2 reads W hi (4): (X lo, W hi);
//     P3 reads X hi (8, over the X lo registers): (hi, hi); P4 reads nothing: (X hi, W lo kept from P1).  16 MFMAs per phase and wave.
//   * each phase also issues ONE half tile of DMA, three half tiles ahead: P1 -> X hi of tile t + 1 (other buffer), P2 / P3 / P4 ->
//     W lo / X lo / W hi of tile t + 2 (THIS buffer: those halves were last read in P1 / P1 / P2).  One counted `s_waitcnt vmcnt(6)` per
//     K tile (in P4: everything up to P1's piece has landed = tile t + 1 complete, the three younger half tiles stay in flight
//     across the barriers), never vmcnt(0) inside the loop.
//   * the two groups run one barrier apart (group 1 takes one extra s_barrier up front, group 0 one at the end): while one group's
//     waves multiply, their SIMD partners of the other group read LDS and issue DMA.
// Ordering rules the schedule obeys (cdna_hip_programming.md "Read a staged buffer one phase AFTER the wait that retires it"):
//   RAW: a half tile is read no earlier than the phase after the vmcnt wait that covers it (P4's wait -> reads from the next P1 on).
//   WAR: a half tile is re-filled two phases after its last read -- or one phase after (W lo, P1 -> P2) because P1 retires its four
//        W reads with `s_waitcnt lgkmcnt(8)` BEFORE its first barrier (the W reads are issued first, order pinned by sched_barrier).
// No other vector-memory instruction may sit between the prologue and the end of the loop (the waits count DMA pieces): the bias is
// added in the epilogue.
#include "pd_common.h"
#include "pd_stage.h"
#include "pd_linear.h"

namespace pd {

template <typename T, bool GLU>
__global__ __launch_bounds__(512, 2) void linear_p8_kernel(const LinP p) {
  static_assert(sizeof(T) == 2, "16-bit element types");
  using E = Elem<T>;
  using Frag = typename E::Frag;
  constexpr int ES = 2, TM = 256, TN = 256;
  constexpr int HALF = 16384, BUF = 4 * HALF;             // [W lo | X lo | W hi | X hi]
  constexpr int H_WLO = 0, H_XLO = 1, H_WHI = 2, H_XHI = 3;
  constexpr int TNO = GLU ? TN / 2 : TN;
  constexpr int EP_PITCH = TN * ES + 16;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // [2][BUF]; epilogue [TM][EP_PITCH]

  // ---- block -> (token tile, channel tile).  Blocks b, b + 8, b + 16, ... share an XCD (and its L2); XCD label x = b % 8 takes a
  // contiguous run of the tile list (bijective for any tile count), and the list walks panels of 4 token tiles x 8 channel tiles, so
  // the 32 workgroups an XCD runs together re-use 4 X tiles and 8 W tiles out of its L2 instead of 1 + 32.
  int tt, ct;
  {
    const int T_ = p.t_tiles * p.c_tiles;
    const int b = blockIdx.x, x = b & 7, j = b >> 3;
    const int q = T_ >> 3, r = T_ & 7;
    const int idx = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    const int per_tp = 4 * p.c_tiles;
    const int tp = idx / per_tp;
    int rem = idx - tp * per_tp;
    const int th = min(4, p.t_tiles - 4 * tp);
    const int cp = rem / (th * 8);
    rem -= cp * th * 8;
    const int cw = min(8, p.c_tiles - 8 * cp);
    const int dt = rem / cw;
    tt = 4 * tp + dt; ct = 8 * cp + (rem - dt * cw);
  }
  const long long m0 = (long long)tt * TM;
  const int n0 = ct * TN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, kq = lane >> 4;
  const int ksteps = p.K / 16, nk = p.K / 64;
  const int last_ct32 = p.N_pad / 32 - 1;
  const long long Mrows = p.M;

  // ---- DMA sources: per half tile this wave moves pieces 2 wave and 2 wave + 1 (1 KiB each)
  //   W half hh: piece f -> packed tile (n0 / 32 + 2 (f / 4) + hh), k-step16 f % 4 of the K tile; lane-linear
  //   X half hh: piece pc -> rows 8 pc .. 8 pc + 7 of the half's 128 tokens; lane -> row 8 pc + lane / 8, LDS slot lane % 8 <- source slot (lane % 8) ^ (row % 8)
  const unsigned char* wsrc[2][2];
  const unsigned char* xsrc[2][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int c32 = min(n0 / 32 + 2 * (wave >> 1) + hh, last_ct32);      // tiles beyond N_pad do not exist: clamp (never stored)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      wsrc[hh][j] = (const unsigned char*)p.w + (((size_t)c32 * ksteps + 2 * (wave & 1) + j) * 512 + lane * 8) * ES;
      const int row = 16 * wave + 8 * j + (lane >> 3);
      const long long m = m0 + 128 * hh + row;
      const long long mc = m < Mrows ? m : Mrows - 1;                      // rows past M re-read the last row (never stored)
      xsrc[hh][j] = (const unsigned char*)p.x + ((size_t)mc * p.x_stride + (((lane & 7) ^ (row & 7)) * 8)) * ES;
    }
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto dma = [&](const unsigned char* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  // half tile `h4` of K tile `kt` into buffer `b2` (two pieces per wave; skipped past the end of K)
  auto stage = [&](int kt, int b2, int h4) {
    if (kt < nk) {
      const unsigned base = __builtin_amdgcn_readfirstlane(lds_base + b2 * BUF + h4 * HALF + wave * 2048);
      const int hh = h4 >> 1;
      if ((h4 & 1) == 0) {
        dma(wsrc[hh][0] + (size_t)kt * 4096, base);
        dma(wsrc[hh][1] + (size_t)kt * 4096, base + 1024);
      } else {
        dma(xsrc[hh][0] + (size_t)kt * 128, base);
        dma(xsrc[hh][1] + (size_t)kt * 128, base + 1024);
      }
    }
  };

  // ---- fragment read offsets inside a buffer
  //   X fragment (token block b of 16, k-step32 s) of half hh: row 64 g + 16 b + l15, source slot 4 s + kq at physical slot (4 s + kq) ^ (l15 & 7)
  //   W fragment (channel block a of 16, k-step32 s) of half hh: piece (wc * 4 + 2 s + (kq >> 1)), lane slot (kq & 1) * 32 + 16 a + l15
  const int xrow = (64 * g + l15) * 128;
  const int xs0 = ((0 + kq) ^ (l15 & 7)) * 16, xs1 = ((4 + kq) ^ (l15 & 7)) * 16;
  const int wlane = (wc * 4 + (kq >> 1)) * 1024 + ((kq & 1) * 32 + l15) * 16;

  f32x4 acc[2][2][2][4];                                  // [xh][wh][a][b]
#pragma unroll
  for (int xh = 0; xh < 2; ++xh)
#pragma unroll
    for (int wh = 0; wh < 2; ++wh)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[xh][wh][a][b] = (f32x4)(0.f);

  Frag xf[4][2], w0[2][2], w1[2][2];                      // [b][s], [a][s]
  auto read_w = [&](Frag (&w)[2][2], const unsigned char* half) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int s = 0; s < 2; ++s) w[a][s] = E::load(half + wlane + s * 2048 + a * 256);
  };
  auto read_x = [&](const unsigned char* half) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      xf[b][0] = E::load(half + xrow + b * 2048 + xs0);
      xf[b][1] = E::load(half + xrow + b * 2048 + xs1);
    }
  };
  auto mma_quad = [&](f32x4 (&c)[2][4], const Frag (&w)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) c[a][b] = E::mma_16x16x32(w[a][s], xf[b][s], c[a][b]);
    __builtin_amdgcn_s_setprio(0);
  };
#define P8_BAR()                          \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

  // ---- prologue: K tile 0 (four half tiles) + the first three half tiles of K tile 1; tile 0 has landed when at most those three
  // (6 pieces per wave) are in flight
  stage(0, 0, H_WLO); stage(0, 0, H_XLO); stage(0, 0, H_WHI); stage(0, 0, H_XHI);
  stage(1, 1, H_WLO); stage(1, 1, H_XLO); stage(1, 1, H_WHI);
  if (nk >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  P8_BAR();
  if (g == 1) P8_BAR();                                   // group 1 runs one barrier behind group 0 from here on

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const unsigned char* buf = lds + cur * BUF;
    // P1: (X lo, W lo)
    read_w(w0, buf + H_WLO * HALF);
    __builtin_amdgcn_sched_barrier(0);
    read_x(buf + H_XLO * HALF);
    __builtin_amdgcn_sched_barrier(0);
    stage(kt + 1, cur ^ 1, H_XHI);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");    // the four W lo reads (issued first) are back: P2 may re-fill W lo
    P8_BAR();
    mma_quad(acc[0][0], w0);
    P8_BAR();
    // P2: (X lo, W hi)
    read_w(w1, buf + H_WHI * HALF);
    __builtin_amdgcn_sched_barrier(0);
    stage(kt + 2, cur, H_WLO);
    P8_BAR();
    mma_quad(acc[0][1], w1);
    P8_BAR();
    // P3: (X hi, W hi)
    read_x(buf + H_XHI * HALF);
    __builtin_amdgcn_sched_barrier(0);
    stage(kt + 2, cur, H_XLO);
    P8_BAR();
    mma_quad(acc[1][1], w1);
    P8_BAR();
    // P4: (X hi, W lo); K tile kt + 1 complete when only P2..P4's pieces (of tile kt + 2) are still in flight
    asm volatile("" ::: "memory");
    stage(kt + 2, cur, H_WHI);
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P8_BAR();
    mma_quad(acc[1][0], w0);
    P8_BAR();
  }
  if (g == 0) P8_BAR();                                   // pairs with group 1's last barrier
  __syncthreads();                                        // every wave is done with the buffers: the epilogue reuses the LDS
#undef P8_BAR

  // ---- epilogue: + bias (module order), [token][channels] image through LDS, then coalesced 16-byte residual loads / stores.
  // D layout of v_mfma_f32_16x16x32: lane (l15, kq) holds token l15 of the block, channels 4 kq + (0..3) of the 16-channel block
  f32x4 bv[2][2];                                          // [wh][a]
#pragma unroll
  for (int wh = 0; wh < 2; ++wh) {
    const int c32 = min(n0 / 32 + 2 * wc + wh, last_ct32);
    const int cob = GLU ? ((c32 & 1) ? p.N / 2 : 0) + 32 * (c32 >> 1) : c32 * 32;
#pragma unroll
    for (int a = 0; a < 2; ++a) bv[wh][a] = *(const f32x4*)(p.bias + cob + 16 * a + 4 * kq);
  }
#pragma unroll
  for (int xh = 0; xh < 2; ++xh)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int tok = 128 * xh + 64 * g + 16 * b + l15;
      T* row = (T*)(lds + tok * EP_PITCH);
      if constexpr (GLU) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          f32x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (acc[xh][0][a][b][i] + bv[0][a][i]) * gelu_f<ES>(acc[xh][1][a][b][i] + bv[1][a][i]);
          store4(row + 32 * wc + 16 * a + 4 * kq, o[0], o[1], o[2], o[3]);
        }
      } else {
#pragma unroll
        for (int wh = 0; wh < 2; ++wh)
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const f32x4 o = acc[xh][wh][a][b] + bv[wh][a];
            store4(row + 32 * (2 * wc + wh) + 16 * a + 4 * kq, o[0], o[1], o[2], o[3]);
          }
      }
    }
  __syncthreads();
  constexpr int EPC = 8;                                 // channels per 16-byte piece
  constexpr int PPT = TNO / EPC;                         // pieces per token
  constexpr int TPI = 512 / PPT;                         // tokens per iteration
  constexpr int NIT = TM / TPI;
  const int piece = tid % PPT, trow = tid / PPT;
  const int NO = GLU ? p.N / 2 : p.N;
  const int co = ct * TNO + piece * EPC;
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

template <typename T, bool GLU>
static int launch_p8(const LinP& p, hipStream_t st) {
  constexpr int MAIN = 2 * 4 * 16384, EPI = 256 * (256 * 2 + 16);
  constexpr int LDS = MAIN > EPI ? MAIN : EPI;
  auto kern = linear_p8_kernel<T, GLU>;
  static LdsAttr attr;
  if (!ensure_lds(attr, kern, LDS)) {
    set_error("pd_linear: cannot reserve %d bytes of LDS", LDS);
    return PD_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.t_tiles * p.c_tiles)), dim3(512), LDS, st, p);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

int launch_linear_p8(const LinP& p, int dtype, bool glu, hipStream_t st) {
  if (dtype == PD_F16) return glu ? launch_p8<half_t, true>(p, st) : launch_p8<half_t, false>(p, st);
  return glu ? launch_p8<bf16_t, true>(p, st) : launch_p8<bf16_t, false>(p, st);
}

}  // namespace pd
