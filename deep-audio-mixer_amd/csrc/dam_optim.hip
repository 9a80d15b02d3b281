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

// state[0] = step (as float64 bits in two floats is awkward) -> keep a separate int64 + two derived floats
__global__ void adam_tick_kernel(long long* __restrict__ step, float* __restrict__ derived, float lr, float beta1, float beta2) {
    const long long t = *step + 1;
    *step = t;
    const double bc1 = 1.0 - pow((double)beta1, (double)t);
    const double bc2 = 1.0 - pow((double)beta2, (double)t);
    derived[0] = (float)((double)lr / bc1);      // step_size
    derived[1] = (float)sqrt(bc2);               // bias_correction2_sqrt
}

__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, int64_t n, const float* __restrict__ derived, float beta1,
                                 float beta2, float eps, float weight_decay, float grad_scale) {
    const float step_size = derived[0], bc2s = derived[1];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pv = p[i];
        float gv = g[i] * grad_scale;
        gv = fmaf(weight_decay, pv, gv);
        const float mv = fmaf(beta1, m[i], (1.0f - beta1) * gv);         // lerp form: m + (g - m)*(1-b1) differs by rounding only
        const float vv = fmaf(beta2, v[i], (1.0f - beta2) * gv * gv);
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2s + eps;
        p[i] = pv - step_size * (mv / denom);
    }
}

}  // namespace
}  // namespace dam

extern "C" int dam_adam_l2_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                    int64_t* step, float* derived2, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, float grad_scale, void* stream) {
    using namespace dam;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !step || !derived2 || n <= 0) return DAM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, (long long*)step, derived2, lr, beta1, beta2);
    DAM_CHECK_LAUNCH();
    int64_t blocks = cdiv(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n, derived2,
                       beta1, beta2, eps, weight_decay, grad_scale);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}
