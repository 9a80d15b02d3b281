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


// mix[r][n] = sum_s audio[s][r][n] * gains[s][min(n / seg, n_gains-1)], plus per-(row, block) max-abs partials.
template <typename T>
__global__ __launch_bounds__(256) void mixdown_kernel(const T* __restrict__ audio, const T* __restrict__ gains, int S,
                                                      int64_t n_samples, int n_gains, int64_t seg, T* __restrict__ mix,
                                                      T* __restrict__ peak_partial) {
    __shared__ T red[256];
    const int row = blockIdx.y, rows = gridDim.y;
    T m = 0;
    for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < n_samples; n += (int64_t)gridDim.x * blockDim.x) {
        int64_t k = n_gains > 1 ? n / seg : 0;
        if (k > n_gains - 1) k = n_gains - 1;
        T a = 0;
        for (int s = 0; s < S; ++s) a += audio[((int64_t)s * rows + row) * n_samples + n] * gains[(int64_t)s * n_gains + k];
        mix[(int64_t)row * n_samples + n] = a;
        const T b = a < 0 ? -a : a;
        m = b > m ? b : m;
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] = red[threadIdx.x + st] > red[threadIdx.x] ? red[threadIdx.x + st] : red[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) peak_partial[(int64_t)row * gridDim.x + blockIdx.x] = red[0];
}

// mix[r][:] /= max_n |mix[r][n]|   (librosa.util.normalize(x, axis=1): rows whose peak is below `tiny` are left alone)
template <typename T>
__global__ __launch_bounds__(256) void peak_normalize_kernel(T* __restrict__ mix, int64_t n_samples, const T* __restrict__ peak_partial,
                                                             int nblk, T tiny) {
    __shared__ T red[256];
    const int row = blockIdx.y;
    T m = 0;
    for (int i = threadIdx.x; i < nblk; i += 256) { const T v = peak_partial[(int64_t)row * nblk + i]; m = v > m ? v : m; }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] = red[threadIdx.x + st] > red[threadIdx.x] ? red[threadIdx.x + st] : red[threadIdx.x];
        __syncthreads();
    }
    const T peak = red[0];
    if (peak < tiny) return;
    for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < n_samples; n += (int64_t)gridDim.x * blockDim.x)
        mix[(int64_t)row * n_samples + n] /= peak;
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

extern "C" int64_t dam_mixdown_workspace_elems(int64_t rows) { return rows > 0 ? rows * 1024 : 0; }

extern "C" int dam_mixdown_peak_normalize(const void* audio, const void* gains, int is_f64, int n_stems, int64_t rows,
                                          int64_t n_samples, int n_gains, int normalize, void* mix, void* workspace,
                                          void* stream) {
    using namespace dam;
    if (!audio || !gains || !mix || !workspace || n_stems <= 0 || rows <= 0 || n_samples <= 0 || n_gains <= 0 || n_gains > n_samples)
        return DAM_ERR_BAD_ARG;
    if (rows > 65535) return DAM_ERR_UNSUPPORTED;
    const int64_t seg = n_samples / n_gains;
    int64_t bx = cdiv(n_samples, 256 * 8);
    if (bx > 1024) bx = 1024;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)bx, (unsigned)rows);
    if (is_f64) {
        hipLaunchKernelGGL(mixdown_kernel<double>, grid, dim3(256), 0, st, (const double*)audio, (const double*)gains, n_stems,
                           n_samples, n_gains, seg, (double*)mix, (double*)workspace);
        DAM_CHECK_LAUNCH();
        if (normalize) {
            hipLaunchKernelGGL(peak_normalize_kernel<double>, grid, dim3(256), 0, st, (double*)mix, n_samples, (const double*)workspace,
                               (int)bx, 2.2250738585072014e-308);
            DAM_CHECK_LAUNCH();
        }
    } else {
        hipLaunchKernelGGL(mixdown_kernel<float>, grid, dim3(256), 0, st, (const float*)audio, (const float*)gains, n_stems,
                           n_samples, n_gains, seg, (float*)mix, (float*)workspace);
        DAM_CHECK_LAUNCH();
        if (normalize) {
            hipLaunchKernelGGL(peak_normalize_kernel<float>, grid, dim3(256), 0, st, (float*)mix, n_samples, (const float*)workspace,
                               (int)bx, 1.17549435e-38f);
            DAM_CHECK_LAUNCH();
        }
    }
    return DAM_OK;
}
