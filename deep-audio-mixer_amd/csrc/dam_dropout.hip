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

}  // namespace
}  // namespace dam

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
