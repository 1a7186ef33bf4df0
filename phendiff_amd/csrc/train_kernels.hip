// Training-step building blocks: bandwidth-bound fused passes over flat fp32 buffers.
// The UNet backward itself is NOT built yet; these are the pieces around it.
//   pd_diffusion_loss : loss + dL/d(model_output) for epsilon / sample (SNR-weighted) / v_prediction  (utils_training.py:415-433)
//   pd_sumsq          : deterministic sum of squares of a flat buffer (global grad norm, utils_training.py:438-440)
//   pd_adamw_ema      : grad clip scaling + torch AdamW update + diffusers EMAModel.step in ONE pass  (:452-454, :553-556)
#include "pd_common.h"

namespace pd {

constexpr int RED_BLOCKS = 1024;

__device__ __forceinline__ double block_sum(double v, double* sm) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i];
  return t;   // valid in thread 0
}

__global__ __launch_bounds__(256) void loss_kernel(const pd_loss_args a) {
#pragma clang fp contract(off)
  __shared__ double sm[4];
  double acc = 0.0;
  const float inv_n = 1.0f / (float)a.numel;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.numel; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / a.per_sample;
    float target, w = 1.0f;
    if (a.pred_type == PD_PRED_EPSILON) target = a.noise[i];
    else if (a.pred_type == PD_PRED_SAMPLE) { target = a.clean[i]; w = a.weight[n]; }
    else target = a.sa[n] * a.noise[i] - a.sb[n] * a.clean[i];      // get_velocity
    const float d = a.model_out[i] - target;
    acc += (double)(w * (d * d));
    if (a.grad_out) a.grad_out[i] = (2.0f * w * d) * inv_n * a.grad_scale;
  }
  const double t = block_sum(acc, sm);
  if (threadIdx.x == 0) a.partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(const double* partial, int nblocks, double scale, float* out) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
  const double t = block_sum(acc, sm);
  if (threadIdx.x == 0) out[0] = (float)(t * scale);
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* x, int64_t n, double* partial) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const double v = x[i]; acc += v * v; }
  const double t = block_sum(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// norm = sqrt(sum partial); clip coefficient as torch.nn.utils.clip_grad_norm_: min(1, max_norm / (norm + 1e-6))
__global__ __launch_bounds__(256) void norm_finalize_kernel(const double* partial, int nblocks, float max_norm, float* norm_out, float* coef_out) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
  const double t = block_sum(acc, sm);
  if (threadIdx.x == 0) {
    const float nrm = (float)sqrt(t);
    norm_out[0] = nrm;
    const float c = max_norm / (nrm + 1e-6f);
    coef_out[0] = c < 1.0f ? c : 1.0f;
  }
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(const pd_adamw_ema_args a) {
#pragma clang fp contract(off)
  const float coef = a.clip_coef ? a.clip_coef[0] : 1.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.numel; i += (int64_t)gridDim.x * 256) {
    const float g = a.grad[i] * coef;
    float p = a.param[i];
    p = p * (1.0f - a.lr * a.weight_decay);                          // param.mul_(1 - lr * wd)
    const float m = a.exp_avg[i] + (g - a.exp_avg[i]) * (1.0f - a.beta1);   // exp_avg.lerp_(grad, 1 - beta1)
    const float v = a.exp_avg_sq[i] * a.beta2 + (g * g) * (1.0f - a.beta2);
    const float denom = sqrtf(v) / a.bias_correction2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);                               // addcdiv_(exp_avg, denom, value=-step_size)
    a.param[i] = p; a.exp_avg[i] = m; a.exp_avg_sq[i] = v;
    if (a.ema) { const float s = a.ema[i]; a.ema[i] = s - a.one_minus_decay * (s - p); }   // EMAModel.step
    if (a.zero_grad) a.grad[i] = 0.0f;
  }
}

}  // namespace pd

using namespace pd;

extern "C" int pd_diffusion_loss(const pd_loss_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->numel > 0 && a->per_sample > 0 && a->numel % a->per_sample == 0, PD_ERR_ARG, "pd_diffusion_loss: bad sizes");
  PD_CHECK(a->model_out && a->partial && a->loss_out, PD_ERR_ARG, "pd_diffusion_loss: null pointer");
  PD_CHECK(a->pred_type >= 0 && a->pred_type <= 2, PD_ERR_ARG, "pd_diffusion_loss: bad prediction type");
  if (a->pred_type == PD_PRED_EPSILON) PD_CHECK(a->noise, PD_ERR_ARG, "pd_diffusion_loss: epsilon needs noise");
  if (a->pred_type == PD_PRED_SAMPLE) PD_CHECK(a->clean && a->weight, PD_ERR_ARG, "pd_diffusion_loss: sample needs clean + SNR weights");
  if (a->pred_type == PD_PRED_V) PD_CHECK(a->clean && a->noise && a->sa && a->sb, PD_ERR_ARG, "pd_diffusion_loss: v_prediction needs clean, noise, sa, sb");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_kernel, dim3(RED_BLOCKS), dim3(256), 0, st, *a);
  PD_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)a->partial, RED_BLOCKS, 1.0 / (double)a->numel, a->loss_out);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_grad_norm(const float* grad, int64_t numel, double* partial, float max_norm, float* norm_out, float* clip_coef_out, void* stream) {
  PD_CHECK(grad && numel > 0 && partial && norm_out && clip_coef_out, PD_ERR_ARG, "pd_grad_norm: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_kernel, dim3(RED_BLOCKS), dim3(256), 0, st, grad, numel, partial);
  PD_LAUNCH_CHECK();
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)partial, RED_BLOCKS, max_norm, norm_out, clip_coef_out);
  PD_LAUNCH_CHECK();
  return PD_OK;
}

extern "C" int pd_adamw_ema(const pd_adamw_ema_args* a, void* stream) {
  PD_CHECK(a != nullptr && a->numel > 0 && a->param && a->grad && a->exp_avg && a->exp_avg_sq, PD_ERR_ARG, "pd_adamw_ema: bad args");
  PD_CHECK(a->bias_correction2_sqrt > 0.f, PD_ERR_ARG, "pd_adamw_ema: bias_correction2_sqrt must be > 0");
  const int64_t blocks = (a->numel + 255) / 256;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, *a);
  PD_LAUNCH_CHECK();
  return PD_OK;
}
