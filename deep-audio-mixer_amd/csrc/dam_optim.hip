// dam_optim.hip -- Adam with L2 weight decay (torch.optim.Adam semantics, not AdamW) over one flat buffer.
//
// Replaces optimizer.step() at model_trainer.py:37 for the optimizer the reference constructs in
// training.ipynb cell 11: torch.optim.Adam(model.parameters(), weight_decay=1e-5)  (lr 1e-3, betas (0.9, 0.999),
// eps 1e-8, amsgrad False).  All parameters live in ONE contiguous buffer (the parameters are views of it),
// so the step is a single HBM-bound launch and the data-parallel gradient exchange a single all-reduce.
// The step counter lives on the device so the launch sequence is hipGraph-capturable.
#include "dam_common.h"

namespace dam {
namespace {

// hyper (optional, device): {lr, beta1, beta2, eps, weight_decay, grad_scale, 1-beta1, 1-beta2} -- read at run time so
// that a captured hipGraph follows param_groups edits / LR schedulers; nullptr = the scalar arguments (baked into the
// launch).  1-beta is formed in double on the host, as torch does (1.0f - 0.999f is off by 5e-5 relative).
struct AdamHyper { float lr, beta1, beta2, eps, wd, gscale, omb1, omb2; };

__device__ __forceinline__ AdamHyper load_hyper(const float* __restrict__ h, AdamHyper k) {
    if (h) { k.lr = h[0]; k.beta1 = h[1]; k.beta2 = h[2]; k.eps = h[3]; k.wd = h[4]; k.gscale = h[5]; k.omb1 = h[6]; k.omb2 = h[7]; }
    return k;
}

__global__ void adam_tick_kernel(long long* __restrict__ step, float* __restrict__ derived, const float* __restrict__ hyper,
                                 AdamHyper k) {
    k = load_hyper(hyper, k);
    const long long t = *step + 1;
    *step = t;
    const double bc1 = 1.0 - pow((double)k.beta1, (double)t);
    const double bc2 = 1.0 - pow((double)k.beta2, (double)t);
    derived[0] = (float)((double)k.lr / bc1);    // step_size
    derived[1] = (float)sqrt(bc2);               // bias_correction2_sqrt
}

__device__ __forceinline__ void adam_one(float& pv, float gv, float& mv, float& vv, const AdamHyper& k, float step_size, float bc2s) {
    gv = fmaf(k.wd, pv, gv * k.gscale);
    mv = fmaf(k.beta1, mv, k.omb1 * gv);         // lerp form: m + (g - m)*(1-b1) differs by rounding only
    vv = fmaf(k.beta2, vv, k.omb2 * gv * gv);
    const float denom = sqrtf(vv) / bc2s + k.eps;
    pv = pv - step_size * (mv / denom);
}

__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, const float* __restrict__ derived,
                                                        const float* __restrict__ hyper, AdamHyper k) {
    k = load_hyper(hyper, k);
    const float step_size = derived[0], bc2s = derived[1];
    const int64_t n4 = n / 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) { float a = pv[j], b = mv[j], c = vv[j]; adam_one(a, gv[j], b, c, k, step_size, bc2s); pv[j] = a; mv[j] = b; vv[j] = c; }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) {
        const int64_t i = 4 * n4 + threadIdx.x;
        float a = p[i], b = m[i], c = v[i];
        adam_one(a, g[i], b, c, k, step_size, bc2s);
        p[i] = a; m[i] = b; v[i] = c;
    }
}

}  // namespace
}  // namespace dam

extern "C" int dam_adam_l2_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                    int64_t* step, float* derived2, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, float grad_scale, const float* hyper_dev, void* stream) {
    using namespace dam;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !step || !derived2 || n <= 0) return DAM_ERR_BAD_ARG;
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return DAM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const AdamHyper k{lr, beta1, beta2, eps, weight_decay, grad_scale, (float)(1.0 - (double)beta1), (float)(1.0 - (double)beta2)};
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, (long long*)step, derived2, hyper_dev, k);
    DAM_CHECK_LAUNCH();
    int64_t blocks = cdiv(n / 4 + 1, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n, derived2,
                       hyper_dev, k);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}
