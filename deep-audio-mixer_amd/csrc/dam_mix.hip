// dam_mix.hip -- applying smoothed per-chunk gains to the original audio (full-song inference tail).
//
// Replaces inference_utils.py:12-41 (interpolate_mask: piecewise-constant stretch of the gain sequence to sample
// resolution) fused with inference_utils.py:143 (mixed_tracks[track] = loaded_tracks[track] * mask), so the
// 8-byte-per-sample mask is never materialised: sample n of every channel is scaled by
// gains[min(n / seg, n_gains-1)], seg = int(n_samples / n_gains).  HBM-bound, one pass.
#include "dam_common.h"

namespace dam {
namespace {

template <typename T>
__global__ void gain_ramp_apply_kernel(const T* __restrict__ audio, const T* __restrict__ gains, int64_t n_samples,
                                       int n_gains, int64_t seg, int64_t total, T* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i % n_samples;                   // audio is [rows][n_samples]
        int64_t k = n_gains > 1 ? n / seg : 0;
        if (k > n_gains - 1) k = n_gains - 1;
        out[i] = audio[i] * gains[k];
    }
}

}  // namespace
}  // namespace dam

extern "C" int dam_gain_ramp_apply(const void* audio, const void* gains, int is_f64, int64_t rows, int64_t n_samples,
                                   int n_gains, void* out, void* stream) {
    using namespace dam;
    if (!audio || !gains || !out || rows <= 0 || n_samples <= 0 || n_gains <= 0 || n_gains > n_samples) return DAM_ERR_BAD_ARG;
    const int64_t total = rows * n_samples, seg = n_samples / n_gains;
    int64_t blocks = cdiv(total, 256);
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (is_f64)
        hipLaunchKernelGGL(gain_ramp_apply_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, (const double*)audio,
                           (const double*)gains, n_samples, n_gains, seg, total, (double*)out);
    else
        hipLaunchKernelGGL(gain_ramp_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)audio,
                           (const float*)gains, n_samples, n_gains, seg, total, (float*)out);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}
