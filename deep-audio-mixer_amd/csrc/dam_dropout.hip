// dam_dropout.hip -- inverted dropout for the scalar models' ConvBlock2d (models/model_scalar_1s.py:177,187-188,
// models/model_scalar_2s.py:35,45-46: nn.Dropout(p) applied only while self.training).
//
// torch's Philox stream cannot be reproduced bit for bit outside torch, so parity here is distributional: every element
// is kept with probability 1-p and scaled by 1/(1-p).  The mask is a pure function of (seed, call offset, element
// index) -- a counter-based generator, nothing is stored: the backward pass regenerates it from the saved offset.
// The call offset lives on the DEVICE (dam_dropout_tick advances it), so captured hipGraphs draw fresh masks on replay.
#include "dam_common.h"

namespace dam {
namespace {

__device__ __forceinline__ unsigned mix64(unsigned long long z) {       // splitmix64 finaliser, high 32 bits
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 32);
}

__global__ void dropout_tick_kernel(long long* counter, long long n, long long* snapshot) {
    const long long c = *counter;
    *snapshot = c;
    *counter = c + n;
}

__global__ void dropout_apply_kernel(const float* __restrict__ x, int64_t n4, float p, float scale, unsigned long long seed,
                                     const long long* __restrict__ snapshot, float* __restrict__ y) {
    const unsigned long long base = seed * 0xD1342543DE82EF95ull + (unsigned long long)(*snapshot);
    const unsigned thr = (unsigned)fminf(p * 4294967296.0f, 4294967295.0f);      // keep iff r >= thr
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const unsigned long long c = base + 4ull * (unsigned long long)i;
        float4 o;
        o.x = mix64(c) >= thr ? v.x * scale : 0.f;
        o.y = mix64(c + 1) >= thr ? v.y * scale : 0.f;
        o.z = mix64(c + 2) >= thr ? v.z * scale : 0.f;
        o.w = mix64(c + 3) >= thr ? v.w * scale : 0.f;
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

// The augmentation draw of data/dataset.py:164-168 (`np.random.uniform(0.6, 1.4) * audio`, one draw per track of an
// item, the mix included :198-199) as a counter-based function of (seed, global item index, track): reproducible whatever
// the batch composition, the worker count or the rank sharding, and drawn where the STFT kernel consumes it.
//   u = top 24 bits of splitmix64(seed * K + item * 4096 + track) / 2^24;  gain = lo + (hi - lo) * u
__global__ void augment_gains_kernel(unsigned long long seed, const long long* __restrict__ items, long long first_item,
                                     int n_items, int n_tracks, float lo, float hi, float* __restrict__ gains) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items * n_tracks) return;
    const int it = i / n_tracks, k = i - it * n_tracks;
    const unsigned long long item = (unsigned long long)(items ? items[it] : first_item + it);
    const unsigned r = mix64(seed * 0xD1342543DE82EF95ull + item * 4096ull + (unsigned long long)k);
    gains[i] = lo + (hi - lo) * ((float)(r >> 8) * (1.0f / 16777216.0f));
}

}  // namespace
}  // namespace dam

extern "C" int dam_augment_gains_f32(uint64_t seed, const int64_t* items, int64_t first_item, int n_items, int n_tracks,
                                     float lo, float hi, float* gains, void* stream) {
    if (!gains || n_items <= 0 || n_tracks <= 0 || n_tracks > 4096 || !(hi >= lo)) return DAM_ERR_BAD_ARG;
    const int total = n_items * n_tracks;
    hipLaunchKernelGGL(dam::augment_gains_kernel, dim3((unsigned)dam::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long)seed, (const long long*)items, (long long)first_item, n_items, n_tracks, lo, hi,
                       gains);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_dropout_tick(int64_t* counter, int64_t n, int64_t* snapshot, void* stream) {
    if (!counter || !snapshot || n <= 0) return DAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(dam::dropout_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)counter, (long long)n,
                       (long long*)snapshot);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_dropout_apply_f32(const float* x, int64_t n, float p, uint64_t seed, const int64_t* snapshot, float* y,
                                     void* stream) {
    using namespace dam;
    if (!x || !y || !snapshot || n <= 0 || !(p >= 0.f && p < 1.f)) return DAM_ERR_BAD_ARG;
    if (n % 4) return DAM_ERR_UNSUPPORTED;
    int64_t blocks = cdiv(n / 4, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n / 4, p,
                       1.0f / (1.0f - p), (unsigned long long)seed, (const long long*)snapshot, y);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}
