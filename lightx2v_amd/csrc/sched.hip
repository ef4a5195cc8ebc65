// One denoise step's sampler update as ONE elementwise kernel: the fp32 CFG combine of WanModel.infer
// (models/networks/wan/model.py:218) + WanScheduler.step_post (models/schedulers/wan/scheduler.py:322-360: flow-matching x0
// prediction, UniPC-bh2 corrector :224-320 and predictor :130-222, solver order <= 2), or the 4-step-distilled scheduler's
// step_post (schedulers/wan/step_distill/scheduler.py:40-56).
//
// Bound: HBM.  Algorithmic bytes per latent element (Wan, order 2, CFG): reads cond 4 + uncond 4 + latents 2 + last_sample 4 + m0 4
// + m1 4, writes x0 4 + sample 4 + latents 4 (+ noise_pred 4) = 34-38 B; 4.8 M elements at 720p x 81 frames = 0.18 GB, ~30 us at the
// achievable HBM rate, in place of ~25 torch launches with their temporaries.
//
// Numerics: the reference runs this update as a sequence of separate fp32 torch ops, so every product, sum and quotient below is
// rounded on its own — `#pragma clang fp contract(off)` keeps the compiler from fusing a*b+c, the division is the correctly rounded
// IEEE one (hipcc default) exactly where the reference divides by r_k, and the literal `0.0f + x` / `c * 0.0f` terms reproduce what
// the reference computes when a lower-order step passes Python's integer 0 for the missing difference term (sign of zero
// included).  All scalar coefficients are computed by the host exactly as the reference computes them (fp32 0-dim tensors) and
// passed by value.  tests/test_gpu_sched.py requires bit-equality with oracle.WanSchedulerOracle over whole trajectories.
#include <string.h>

#include "x2v_common.h"

namespace x2v {

struct UniPcCoef {
  float guide;     // CFG scale (model.py:218)
  float sigma_i;   // sigmas[step_index]                         x0 = sample - sigma_i * model_output
  // corrector (_coeffs(step_index, step_index - 1, order_c, ...)):
  float c_a;       // sigma_t / sigma_s0
  float c_b;       // alpha_t * h_phi_1
  float c_c;       // alpha_t * B_h
  float c_rk;      // r_1 (order 2)
  float c_rho0;    // rhos_c[0] (order 2)
  float c_rhol;    // rhos_c[-1]
  // predictor (_coeffs(step_index + 1, step_index, order_p, ...)):
  float p_a, p_b, p_c, p_rk;
};

template <bool LAT_BF16>
__global__ __launch_bounds__(256) void unipc_step_kernel(const float* __restrict__ cond, const float* __restrict__ uncond, const void* __restrict__ lat,
                                                         const float* __restrict__ last_sample, const float* __restrict__ m0p, const float* __restrict__ m1p,
                                                         float* __restrict__ noise_pred, float* __restrict__ x0_out, float* __restrict__ sample_out,
                                                         float* __restrict__ lat_out, UniPcCoef c, int order_c, int order_p, int64_t n) {
#pragma clang fp contract(off)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float mo = cond[i];
    if (uncond != nullptr) {  // noise_pred = uncond + guide * (cond - uncond)
      const float u = uncond[i];
      const float d = mo - u;
      const float g = c.guide * d;
      mo = u + g;
    }
    if (noise_pred != nullptr) noise_pred[i] = mo;
    float sample = LAT_BF16 ? bf2f(reinterpret_cast<const unsigned short*>(lat)[i]) : reinterpret_cast<const float*>(lat)[i];
    const float sm = c.sigma_i * mo;
    const float x0 = sample - sm;
    if (order_c > 0) {  // multistep_uni_c_bh_update(this_model_output = x0, last_sample, order_c)
      const float m0 = m0p[i];
      const float ta = c.c_a * last_sample[i];
      const float tb = c.c_b * m0;
      const float xt = ta - tb;
      float corr;
      if (order_c >= 2) {
        const float dm = m1p[i] - m0;
        const float d1 = dm / c.c_rk;
        corr = c.c_rho0 * d1;
      } else {
        corr = 0.0f;
      }
      const float dx = x0 - m0;
      const float rl = c.c_rhol * dx;
      float t;
      if (order_c >= 2) t = corr + rl;
      else t = 0.0f + rl;  // Python int 0 + tensor
      const float ct = c.c_c * t;
      sample = xt - ct;
    }
    x0_out[i] = x0;
    sample_out[i] = sample;
    {  // multistep_uni_p_bh_update(sample, order_p) with m0 = x0
      const float ta = c.p_a * sample;
      const float tb = c.p_b * x0;
      const float xt = ta - tb;
      float pt;
      if (order_p >= 2) {
        const float dm = m0p[i] - x0;  // previous x0 is now model_outputs[-2]
        const float d1 = dm / c.p_rk;
        const float pred = 0.5f * d1;  // rhos_p = [0.5]
        pt = c.p_c * pred;
      } else {
        pt = c.p_c * 0.0f;  // (alpha_t * B_h) * 0
      }
      lat_out[i] = xt - pt;
    }
  }
}

// step_distill/scheduler.py:40-56: x0 = latents - sigma * flow_pred; if not last: x0 = (1 - nxt) * x0 + nxt * noise; latents = x0.to(latents.dtype)
template <bool LAT_BF16>
__global__ __launch_bounds__(256) void distill_step_kernel(const float* __restrict__ cond, const float* __restrict__ uncond, const void* __restrict__ lat,
                                                           const float* __restrict__ noise, float* __restrict__ noise_pred, void* __restrict__ lat_out, float guide,
                                                           float sigma, float one_minus_next, float next, int64_t n) {
#pragma clang fp contract(off)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float mo = cond[i];
    if (uncond != nullptr) {
      const float u = uncond[i];
      const float d = mo - u;
      const float g = guide * d;
      mo = u + g;
    }
    if (noise_pred != nullptr) noise_pred[i] = mo;
    const float sample = LAT_BF16 ? bf2f(reinterpret_cast<const unsigned short*>(lat)[i]) : reinterpret_cast<const float*>(lat)[i];
    const float sm = sigma * mo;
    float x0 = sample - sm;
    if (noise != nullptr) {
      const float a = one_minus_next * x0;
      const float b = next * noise[i];
      x0 = a + b;
    }
    if (LAT_BF16) reinterpret_cast<unsigned short*>(lat_out)[i] = f2bf(x0);
    else reinterpret_cast<float*>(lat_out)[i] = x0;
  }
}

static unsigned grid_for(int64_t n) {
  const int64_t blocks = (n + 255) / 256;
  return (unsigned)(blocks < 2048 ? blocks : 2048);
}

}  // namespace x2v
using namespace x2v;

extern "C" __attribute__((visibility("default"))) int x2v_unipc_step_f32(const float* cond, const float* uncond, const void* latents, int latents_bf16,
                                                                         const float* last_sample, const float* m0, const float* m1, float* noise_pred,
                                                                         float* x0_out, float* sample_out, float* latents_out, const float* coef, int order_c,
                                                                         int order_p, int64_t n, void* stream) {
  X2V_REQUIRE(cond && latents && x0_out && sample_out && latents_out && coef, X2V_E_ARG, "unipc_step: null pointer");
  X2V_REQUIRE(n >= 0 && order_c >= 0 && order_c <= 2 && order_p >= 1 && order_p <= 2, X2V_E_SHAPE, "unipc_step: n=%lld order_c=%d order_p=%d", (long long)n, order_c,
              order_p);
  X2V_REQUIRE(order_c == 0 || (last_sample && m0), X2V_E_ARG, "unipc_step: the corrector needs last_sample and m0");
  X2V_REQUIRE((order_c < 2 || m1) && (order_p < 2 || m0), X2V_E_ARG, "unipc_step: an order-2 update needs the older model outputs");
  if (n == 0) return X2V_OK;
  UniPcCoef c;
  static_assert(sizeof(UniPcCoef) == 12 * sizeof(float), "coef is 12 floats (x2v.h)");
  ::memcpy(&c, coef, sizeof c);  // host array, read at call time
  if (latents_bf16)
    hipLaunchKernelGGL(unipc_step_kernel<true>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, cond, uncond, latents, last_sample, m0, m1, noise_pred, x0_out,
                       sample_out, latents_out, c, order_c, order_p, n);
  else
    hipLaunchKernelGGL(unipc_step_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, cond, uncond, latents, last_sample, m0, m1, noise_pred, x0_out,
                       sample_out, latents_out, c, order_c, order_p, n);
  X2V_LAUNCH_CHECK("unipc_step launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_distill_step_f32(const float* cond, const float* uncond, float guide, const void* latents, int latents_bf16,
                                                                           const float* noise, float sigma, float one_minus_next, float sigma_next,
                                                                           float* noise_pred, void* latents_out, int64_t n, void* stream) {
  X2V_REQUIRE(cond && latents && latents_out, X2V_E_ARG, "distill_step: null pointer");
  X2V_REQUIRE(n >= 0, X2V_E_SHAPE, "distill_step: n=%lld", (long long)n);
  if (n == 0) return X2V_OK;
  if (latents_bf16)
    hipLaunchKernelGGL(distill_step_kernel<true>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, cond, uncond, latents, noise, noise_pred, latents_out, guide,
                       sigma, one_minus_next, sigma_next, n);
  else
    hipLaunchKernelGGL(distill_step_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, cond, uncond, latents, noise, noise_pred, latents_out, guide,
                       sigma, one_minus_next, sigma_next, n);
  X2V_LAUNCH_CHECK("distill_step launch");
  return X2V_OK;
}
