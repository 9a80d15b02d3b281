// dam_layout.hip -- NCHW stem input -> NHWC with 16 zero-padded channels.
//
// The ResNet stem (models/model_resnet.py:64,97: conv1 over the S <= 16 stacked stem spectrograms [B,S,F,T]) then takes the
// same strip / row-streaming kernels as layer1 instead of the gather path of the tile kernels.  One read of the S planes,
// one 64-byte write per pixel; HBM-bound.
#include "dam_common.h"

namespace dam {
namespace {

// A workgroup takes 256 consecutive pixels per round: plane reads coalesce across the threads (one pixel each), the 64-byte pixel
// records go through LDS so that consecutive lanes STORE consecutive 16-byte pieces (a thread storing its own pixel's four quads
// wrote 16 bytes at a 64-byte pitch per instruction: 25.7 us for 34 MB in + 68 MB out).
__global__ __launch_bounds__(256) void nchw_to_nhwc16_kernel(const float* __restrict__ x, int C, int64_t HW, float* __restrict__ y) {
    __shared__ float4 t[256 * 4 + 4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const float* xb = x + (int64_t)b * C * HW;
    float4* yb = reinterpret_cast<float4*>(y + (int64_t)b * HW * 16);
    for (int64_t p0 = blockIdx.x * (int64_t)256; p0 < HW; p0 += (int64_t)gridDim.x * 256) {
        const int64_t p = p0 + tid;
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = (c < C && p < HW) ? xb[(int64_t)c * HW + p] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) t[tid * 4 + q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        __syncthreads();
        const int64_t n4 = (HW - p0 < 256 ? HW - p0 : 256) * 4;            // quads of this round
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = k * 256 + tid;
            if (i < n4) yb[p0 * 4 + i] = t[i];
        }
        __syncthreads();
    }
}

}  // namespace
}  // namespace dam

extern "C" int dam_nchw_to_nhwc16_f32(const float* x, int B, int C, int64_t HW, float* y, void* stream) {
    using namespace dam;
    if (!x || !y || B <= 0 || B > 65535 || C <= 0 || C > 16 || HW <= 0) return DAM_ERR_BAD_ARG;
    int64_t blocks = cdiv(HW, 256);
    if (blocks > 2048) blocks = 2048;       // (grid-stride over rounds of 256 pixels)
    hipLaunchKernelGGL(nchw_to_nhwc16_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, C, HW, y);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}
