// GroupNorm finalize of ONE sample by one workgroup: the tail of a producer kernel (round 5).
//
// pd_gn_finalize is 41 launches of ~8 us per UNet forward, each on the critical path between a producer (the convolution whose
// epilogue wrote the per-tile channel sums) and its consumer (the next convolution's GroupNorm prologue): kernel + two launch
// boundaries ~ 12 us x 41 = 0.5 ms of a 19-ms forward.  With pd_conv_args.fin the producer finishes the job itself: every workgroup
// publishes its statistic rows (release fence), bumps a per-sample counter, and the workgroup that arrives LAST for a sample (it
// sees every other workgroup's rows after an acquire fence) folds them into scale / shift.  The samples of a launch finish in
// grid order, so all but the last sample's fold overlap the launch's remaining workgroups.  The fold is a fixed-order fp64 tree
// over (tile slice, channel): the result does not depend on which workgroup runs it.
#pragma once
#include "pd_common.h"

namespace pd {

// scratch: >= 2 * 256 doubles + 2 * 64 floats of LDS (8-byte aligned), free for this workgroup's use; 256 threads, all of them call.
__device__ __forceinline__ void gn_finalize_sample(const pd_gn_finalize_args& a, int n, unsigned char* scratch, int tid) {
  double* ps = (double*)scratch;
  double* pq = ps + 256;
  float* mean_s = (float*)(pq + 256);
  float* rstd_s = mean_s + 64;
  const int C = a.C0 + a.C1;
  const int gs = C / a.groups;
  // channel blocks of whole groups, at most 64 channels: 64 channel lanes x 4 tile slices (gs > 64: one group per block, lanes loop)
  const int CB = gs <= 64 ? (64 / gs) * gs : gs;
  const int cl = tid & 63, sl = tid >> 6;
  for (int c_lo = 0; c_lo < C; c_lo += CB) {
    const int c_hi = min(C, c_lo + CB);
    const int ng = (c_hi - c_lo) / gs;
    if (gs <= 64) {
      double s = 0.0, q = 0.0;
      const int c = c_lo + cl;
      if (c < c_hi) {
        const bool first = c < a.C0;
        const float* st = first ? a.stats0 : a.stats1;
        const int Cs = first ? a.C0 : a.C1, T = first ? a.T0 : a.T1, cc = first ? c : c - a.C0;
        const float* base = st + ((size_t)n * T * Cs + cc) * 2;
#pragma unroll 4
        for (int t = sl; t < T; t += 4) {
          const float2 v = *(const float2*)(base + (size_t)t * Cs * 2);
          s += (double)v.x; q += (double)v.y;
        }
      }
      ps[tid] = s; pq[tid] = q;
      __syncthreads();
      if (tid < ng) {
        double ds = 0.0, dq = 0.0;
        for (int k = tid * gs; k < (tid + 1) * gs; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) { ds += ps[j * 64 + k]; dq += pq[j * 64 + k]; }
        const double cnt = (double)gs * (double)a.HW;
        const double mean = ds / cnt;
        double var = dq / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[tid] = (float)mean;
        rstd_s[tid] = (float)(1.0 / sqrt(var + (double)a.eps));
        if (a.mean) { a.mean[n * a.groups + c_lo / gs + tid] = mean_s[tid]; a.rstd[n * a.groups + c_lo / gs + tid] = rstd_s[tid]; }
      }
      __syncthreads();
    } else {
      // wide groups (the latent-diffusion UNet's 2560-channel concatenations: 80 channels per group): every thread sums a strided share of
      // the group's (channel, tile) pairs, fixed-order tree over the 256 partials
      double s = 0.0, q = 0.0;
      for (int c = c_lo + tid; c < c_hi; c += 256) {
        const bool first = c < a.C0;
        const float* st = first ? a.stats0 : a.stats1;
        const int Cs = first ? a.C0 : a.C1, T = first ? a.T0 : a.T1, cc = first ? c : c - a.C0;
        const float* base = st + ((size_t)n * T * Cs + cc) * 2;
        for (int t = 0; t < T; ++t) {
          const float2 v = *(const float2*)(base + (size_t)t * Cs * 2);
          s += (double)v.x; q += (double)v.y;
        }
      }
      ps[tid] = s; pq[tid] = q;
      __syncthreads();
      for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) { ps[tid] += ps[tid + off]; pq[tid] += pq[tid + off]; }
        __syncthreads();
      }
      if (tid == 0) {
        const double cnt = (double)gs * (double)a.HW;
        const double mean = ps[0] / cnt;
        double var = pq[0] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[0] = (float)mean;
        rstd_s[0] = (float)(1.0 / sqrt(var + (double)a.eps));
        if (a.mean) { a.mean[n * a.groups + c_lo / gs] = mean_s[0]; a.rstd[n * a.groups + c_lo / gs] = rstd_s[0]; }
      }
      __syncthreads();
    }
    for (int c = c_lo + tid; c < c_hi; c += 256) {
      const int g = (c - c_lo) / gs;
      float sc = rstd_s[g] * a.gamma[c];
      float sh = a.beta[c] - mean_s[g] * sc;
      if (a.temb) {      // ResnetBlock2D(time_embedding_norm = "scale_shift"): (x*sc + sh)*(1 + s) + t
        const float* row = a.temb + (size_t)n * a.temb_stride;
        const float s1 = 1.0f + row[c];
        sc *= s1;
        sh = sh * s1 + row[C + c];
      }
      a.scale[(size_t)n * C + c] = sc;
      a.shift[(size_t)n * C + c] = sh;
    }
    __syncthreads();
  }
}

// The tail of a producer workgroup (256 threads, all of them call; `lds` = at least 4.6 KB of LDS no longer in use): publish, count, and --
// for the last arrival of sample n -- fold.  expected = workgroups per sample over all the launches that share the counter.
__device__ __forceinline__ void gn_fused_finalize_tail(const pd_gn_finalize_args* fin, unsigned* counter, unsigned expected, int n,
                                                       unsigned char* lds, int tid) {
  __threadfence();                 // release: this workgroup's statistic rows are visible device-wide before its count is
  __syncthreads();
  unsigned* flag = (unsigned*)lds;
  if (tid == 0) {
    const unsigned old = atomicAdd(counter + n, 1u);
    const unsigned last = (old + 1u == expected) ? 1u : 0u;
    if (last) counter[n] = 0u;     // ready for the next forward (the next kernel boundary orders this store)
    *flag = last;
  }
  __syncthreads();
  const unsigned last = *flag;
  __syncthreads();                 // (the flag word is part of the scratch area below)
  if (last) {
    __threadfence();               // acquire: every other workgroup's rows
    gn_finalize_sample(*fin, n, lds + 16, tid);
  }
}

}  // namespace pd
