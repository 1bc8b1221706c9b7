// npp_loss_adam.hip -- K4 adaptive robust pixel loss (forward + closed-form backward)
// and K8 fused Adam.  Both are tiny, HBM/latency-bound elementwise kernels.
//
// K4 restates models/mse_calculator.py:13-27 -> robust_loss_pytorch/adaptive.py:183-204
// -> distribution.py:171-210 (nll = rho + log c + logZ(alpha)) -> general.py:85-118
// ('otherwise' branch; alpha is confined to (0.001,1.999) by adaptive.py:146-164) ->
// distribution.py:90-114,143-169 + cubic_spline.py:65-97 (log-partition spline).
// Backward (derivation in oracle/npp_oracle.py: robust_nll_grads):
//   beta = 2-alpha, u = (x/c)^2/beta + 1, e = alpha/2
//   d rho/dx = x/c^2 u^(e-1) ; d rho/dc = -x^2/c^3 u^(e-1) ;
//   d rho/da = -(2/a^2)(u^e - 1) + (beta/a) u^e (ln(u)/2 + e (x/c)^2/(beta^2 u))
#include "npp_common.h"

namespace npp {

// (body: pixel_loss_body in npp_common.h -- shared with the fused patch-in launch of npp_conv.hip)
__global__ __launch_bounds__(256) void pixel_loss_kernel(PixelLossArgs a) { pixel_loss_body(a, (int)blockIdx.x, (int)gridDim.x); }
// nbatch independent losses in one launch (blockIdx.y): predictions / gradients (N,3) back to back, 6 latents and one loss word
// per problem; the targets are shared when gt_stride == 0.
__global__ __launch_bounds__(256) void pixel_loss_batched_kernel(PixelLossArgs a, int64_t gt_stride) {
  const int64_t b = blockIdx.y;
  a.pred += b * a.N * 3; a.dpred += b * a.N * 3; a.gt += b * gt_stride;
  if (a.latents) { a.latents += b * 6; a.dlatent += b * 6; }
  a.loss_out += b;
  pixel_loss_body(a, (int)blockIdx.x, (int)gridDim.x);
}

// torch.optim.Adam single-tensor maths (helpers.py:164): the gradient is the sum of the
// split-K slabs written by npp_mlp_wgrad, so this kernel is also the wgrad reduction.
// hp != nullptr: step_size and 1/sqrt(1 - b2^t) are read from device memory ([0], [1]) so that a
// captured HIP graph can be replayed with the values of the current step.
// tail (optional, handled by one extra block): a second small parameter group -- the adaptive-loss
// latents, whose gradient accumulator is consumed and cleared -- and an accumulator to clear for the
// next iteration, so that one launch replaces optimizer.step() over both groups plus zero_grad().
// (AdamTail, adam_tail_block: npp_common.h -- shared with the fused Adam + re-pack launch of npp_api.hip)
// VEC = 4: one float4 per thread and array (n, slab_stride multiples of 4, 16-byte aligned pointers), four slab loads in
// flight before the (order-preserving, hence bit-identical) summation: the kernel is a pure HBM stream of
// (n_slabs + 5) * 4 B per parameter.
template <int VEC>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ g,
                                                   int64_t n, int n_slabs, int64_t slab_stride,
                                                   float step_size, float b1, float b2, float inv_sqrt_bc2,
                                                   float eps, const float* __restrict__ hp, AdamTail tail) {
  if (hp) { step_size = hp[0]; inv_sqrt_bc2 = hp[1]; }
  if ((int64_t)blockIdx.x * blockDim.x * VEC >= n) {      // the extra block
    adam_tail_block(tail, step_size, b1, b2, inv_sqrt_bc2, eps);
    return;
  }
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (i >= n) return;
  if (VEC > 1 && i + VEC > n) {                            // the last n % VEC elements (the parameter count is odd): scalar
    for (int64_t e = i; e < n; ++e) {
      float ge = 0.0f;
      for (int sl = 0; sl < n_slabs; ++sl) ge += g[(int64_t)sl * slab_stride + e];
      const float me = b1 * m[e] + (1.0f - b1) * ge;
      const float ve = b2 * v[e] + (1.0f - b2) * ge * ge;
      m[e] = me;
      v[e] = ve;
      p[e] = p[e] - step_size * (me / (sqrtf(ve) * inv_sqrt_bc2 + eps));
    }
    return;
  }
  auto ld = [&](int sl) { return *(const vec_t*)(g + (int64_t)sl * slab_stride + i); };
  vec_t gi = (vec_t)(0.0f);
  int sl = 0;
  for (; sl + 4 <= n_slabs; sl += 4) {
    const vec_t a0 = ld(sl), a1 = ld(sl + 1), a2 = ld(sl + 2), a3 = ld(sl + 3);
    gi += a0; gi += a1; gi += a2; gi += a3;
  }
  for (; sl < n_slabs; ++sl) gi += ld(sl);
  vec_t mi = *(const vec_t*)(m + i), vi = *(const vec_t*)(v + i), pi = *(const vec_t*)(p + i);
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const float ge = gi[e];
    const float me = b1 * mi[e] + (1.0f - b1) * ge;
    const float ve = b2 * vi[e] + (1.0f - b2) * ge * ge;
    mi[e] = me;
    vi[e] = ve;
    const float denom = sqrtf(ve) * inv_sqrt_bc2 + eps;
    pi[e] = pi[e] - step_size * (me / denom);
  }
  *(vec_t*)(m + i) = mi;
  *(vec_t*)(v + i) = vi;
  *(vec_t*)(p + i) = pi;
}

static void adam_launch(float* p, float* m, float* v, const float* g, int64_t n, int n_slabs, int64_t slab_stride,
                        float step_size, float b1, float b2, float inv_sqrt_bc2, float eps, const float* hp,
                        const AdamTail& tail, bool with_tail, hipStream_t s) {
  const bool vec = slab_stride % 4 == 0 && (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) & 15) == 0;
  const int64_t threads = vec ? (n + 3) / 4 : n;
  const dim3 grid((unsigned)((threads + 255) / 256 + (with_tail ? 1 : 0)));
  if (vec)
    hipLaunchKernelGGL(adam_kernel<4>, grid, dim3(256), 0, s, p, m, v, g, n, n_slabs, slab_stride, step_size, b1, b2,
                       inv_sqrt_bc2, eps, hp, tail);
  else
    hipLaunchKernelGGL(adam_kernel<1>, grid, dim3(256), 0, s, p, m, v, g, n, n_slabs, slab_stride, step_size, b1, b2,
                       inv_sqrt_bc2, eps, hp, tail);
}

// Gradient of the parameter blob as one tensor (sum of the split-K slabs): what loss.backward() leaves in
// .grad for an optimiser that is not npp_adam_step (compatibility form, torch.optim.Adam).
__global__ __launch_bounds__(256) void grad_reduce_kernel(const float* __restrict__ g, int n_slabs, int64_t slab_stride,
                                                          int64_t n, float* __restrict__ out, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? out[i] : 0.0f;
  for (int k = 0; k < n_slabs; ++k) s += g[(int64_t)k * slab_stride + i];
  out[i] = s;
}

// a2 stand-alone: Embedder.embed (models/embedder.py:11-56) on an arbitrary (N, d) fp32 input:
// out = [x | sin(f0 x) | cos(f0 x) | ... ] in blocks of d columns (include_input drops the first block).
struct FourierArgs {
  float freq[NPP_N_FREQ];
  int32_t n_freq, d, include_input;
};
__global__ __launch_bounds__(256) void fourier_kernel(const float* __restrict__ x, int64_t N, FourierArgs a,
                                                      float* __restrict__ out) {
  const int od = a.d * (2 * a.n_freq + (a.include_input ? 1 : 0));
  const int64_t total = N * od;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / od;
    const int c = (int)(idx - r * od);
    int blk = c / a.d;
    const int i = c - blk * a.d;
    const float v = x[r * a.d + i];
    float o;
    if (a.include_input) {
      if (blk == 0) { out[idx] = v; continue; }
      blk -= 1;
    }
    const float arg = v * a.freq[blk >> 1];       // p_fn(x * freq), fp32 like the reference
    o = (blk & 1) ? cosf(arg) : sinf(arg);
    out[idx] = o;
  }
}

}  // namespace npp

using namespace npp;

extern "C" int npp_pixel_loss(const float* d_pred, const float* d_gt, const float* d_mask, int64_t N,
                              const float* d_latents, const float* d_spline, int n_knots, float x_scale,
                              float weight, float* d_loss, float* d_dpred, float* d_dlatent, void* stream) {
  if (N <= 0 || !d_pred || !d_gt || !d_latents || !d_spline || !d_loss || !d_dpred || !d_dlatent || n_knots < 2) {
    set_error("npp_pixel_loss: bad arguments (N=%lld, n_knots=%d)", (long long)N, n_knots);
    return NPP_ERR_ARG;
  }
  const PixelLossArgs a{d_pred, d_gt, d_mask, N, d_latents, d_spline, n_knots, x_scale, weight, d_loss, d_dpred, d_dlatent};
  hipLaunchKernelGGL(pixel_loss_kernel, dim3((unsigned)pixel_loss_blocks(N)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("npp_pixel_loss");
}

extern "C" int npp_pixel_loss_batched(const float* d_pred, const float* d_gt, int64_t gt_stride, int64_t N, int nbatch,
                                      const float* d_latents, const float* d_spline, int n_knots, float x_scale, float weight,
                                      float* d_loss, float* d_dpred, float* d_dlatent, void* stream) {
  if (N <= 0 || nbatch < 1 || nbatch > 65535 || !d_pred || !d_gt || !d_latents || !d_spline || !d_loss || !d_dpred || !d_dlatent ||
      n_knots < 2 || gt_stride < 0) {
    set_error("npp_pixel_loss_batched: bad arguments (N=%lld, nbatch=%d, n_knots=%d)", (long long)N, nbatch, n_knots);
    return NPP_ERR_ARG;
  }
  const PixelLossArgs a{d_pred, d_gt, nullptr, N, d_latents, d_spline, n_knots, x_scale, weight, d_loss, d_dpred, d_dlatent};
  hipLaunchKernelGGL(pixel_loss_batched_kernel, dim3((unsigned)pixel_loss_blocks(N), (unsigned)nbatch), dim3(256), 0, (hipStream_t)stream, a,
                     gt_stride);
  return check_launch("npp_pixel_loss_batched");
}

// The non-adaptive switches of img2mse (models/mse_calculator.py:19-23) -- 'l2': coef = 1; 'robust_loss' = lossfun(diff, alpha = 2,
// scale = 0.1) = 0.5 (diff / 0.1)^2: coef = 50 -- for nbatch problems: loss[b] += weight * coef * mean(x^2), dpred = its gradient;
// x = diff * mask + (1 - mask) * diff * 0.3 with the (N) mask shared by the problems (nullable).
extern "C" int npp_pixel_loss_quad(const float* d_pred, const float* d_gt, int64_t gt_stride, const float* d_mask, int64_t N, int nbatch,
                                   float coef, float weight, float* d_loss, float* d_dpred, void* stream) {
  if (N <= 0 || nbatch < 1 || nbatch > 65535 || !d_pred || !d_gt || !d_loss || !d_dpred || gt_stride < 0 || !(coef > 0.0f)) {
    set_error("npp_pixel_loss_quad: bad arguments (N=%lld, nbatch=%d, coef=%g)", (long long)N, nbatch, (double)coef);
    return NPP_ERR_ARG;
  }
  PixelLossArgs a{d_pred, d_gt, d_mask, N, nullptr, nullptr, 0, 0.0f, weight, d_loss, d_dpred, nullptr, nullptr, coef};
  hipLaunchKernelGGL(pixel_loss_batched_kernel, dim3((unsigned)pixel_loss_blocks(N), (unsigned)nbatch), dim3(256), 0, (hipStream_t)stream, a,
                     gt_stride);
  return check_launch("npp_pixel_loss_quad");
}

extern "C" int npp_adam_step(float* d_p, float* d_m, float* d_v, const float* d_gslabs, int64_t n, int n_slabs,
                             int64_t slab_stride, float lr, float beta1, float beta2, float eps, int step,
                             void* stream) {
  if (n <= 0 || !d_p || !d_m || !d_v || !d_gslabs || n_slabs < 1 || step < 1) {
    set_error("npp_adam_step: bad arguments");
    return NPP_ERR_ARG;
  }
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  adam_launch(d_p, d_m, d_v, d_gslabs, n, n_slabs, slab_stride, step_size, beta1, beta2, inv_sqrt_bc2, eps, nullptr,
              AdamTail{}, false, (hipStream_t)stream);
  return check_launch("npp_adam_step");
}

extern "C" int npp_adam_step_net(float* d_p, float* d_m, float* d_v, const float* d_gslabs, int64_t n, int n_slabs,
                                 int64_t slab_stride, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat,
                                 int n_lat, float* d_zero, int n_zero, float lr, float beta1, float beta2, float eps,
                                 int step, void* stream) {
  if (n <= 0 || !d_p || !d_m || !d_v || !d_gslabs || n_slabs < 1 || step < 1 || n_lat < 0 || n_zero < 0 ||
      (n_lat > 0 && (!d_lat || !d_lat_m || !d_lat_v || !d_dlat)) || (n_zero > 0 && !d_zero)) {
    set_error("npp_adam_step_net: bad arguments");
    return NPP_ERR_ARG;
  }
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const AdamTail tail{d_lat, d_lat_m, d_lat_v, d_dlat, n_lat, d_zero, n_zero};
  adam_launch(d_p, d_m, d_v, d_gslabs, n, n_slabs, slab_stride, step_size, beta1, beta2, inv_sqrt_bc2, eps, nullptr, tail,
              true, (hipStream_t)stream);
  return check_launch("npp_adam_step_net");
}

extern "C" int npp_adam_step_dev(float* d_p, float* d_m, float* d_v, const float* d_gslabs, int64_t n, int n_slabs,
                                 int64_t slab_stride, float beta1, float beta2, float eps, const float* d_hp,
                                 void* stream) {
  if (n <= 0 || !d_p || !d_m || !d_v || !d_gslabs || n_slabs < 1 || !d_hp) {
    set_error("npp_adam_step_dev: bad arguments");
    return NPP_ERR_ARG;
  }
  adam_launch(d_p, d_m, d_v, d_gslabs, n, n_slabs, slab_stride, 0.0f, beta1, beta2, 0.0f, eps, d_hp, AdamTail{}, false,
              (hipStream_t)stream);
  return check_launch("npp_adam_step_dev");
}

extern "C" int npp_grad_reduce(const float* d_gslabs, int n_slabs, int64_t slab_stride, int64_t n, float* d_grad,
                               int accumulate, void* stream) {
  if (!d_gslabs || !d_grad || n <= 0 || n_slabs < 1 || slab_stride < n) {
    set_error("npp_grad_reduce: bad arguments");
    return NPP_ERR_ARG;
  }
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_gslabs,
                     n_slabs, slab_stride, n, d_grad, accumulate);
  return check_launch("npp_grad_reduce");
}

extern "C" int npp_fourier_fwd(const float* d_x, int64_t N, int d, const float* freqs, int n_freq, int include_input,
                               float* d_out, void* stream) {
  if (N < 0 || d < 1 || n_freq < 0 || n_freq > NPP_N_FREQ || !freqs || (N > 0 && (!d_x || !d_out))) {
    set_error("npp_fourier_fwd: bad arguments (N=%lld d=%d n_freq=%d)", (long long)N, d, n_freq);
    return NPP_ERR_ARG;
  }
  if (N == 0) return NPP_OK;
  FourierArgs a{};
  for (int j = 0; j < n_freq; ++j) a.freq[j] = freqs[j];
  a.n_freq = n_freq; a.d = d; a.include_input = include_input ? 1 : 0;
  const int64_t total = N * d * (2 * n_freq + a.include_input);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(fourier_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_x, N, a, d_out);
  return check_launch("npp_fourier_fwd");
}
