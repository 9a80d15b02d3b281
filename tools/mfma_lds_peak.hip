// tools/mfma_lds_peak.hip -- what v_mfma_f32_16x16x4_f32 sustains when its operands come from LDS the way the row-streaming
// weight gradient and the strip convolution feed it: per step one ds_read_b32 for A and NINE for B (256 contiguous bytes per
// instruction, conflict-free), then 9 independent MFMAs; operands of step t+1 are requested before the MFMAs of step t.
// One wave per SIMD (W = 1, as those kernels' compute waves) or two; G workgroups of 256 * W threads, one per CU.
// MODE 0: register operands only (the same loop without the reads); 1: LDS operands; 2: LDS operands + a co-resident wave per
// SIMD that streams 16-byte global loads into LDS writes (the loader waves' traffic: 4 KB per CU per 16 compute steps).
// Prints time per MFMA per SIMD (hipEvents), shader clocks per MFMA (s_memtime / clock64) and the sustained shader clock
// (clock64 against wall_clock64, 100 MHz), so that a chip-wide limit (time per MFMA rises with G at constant clocks per MFMA:
// the clock fell) can be told from a per-CU one.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_lds_peak.hip -o tools/mfma_lds_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void loop(float* out, long long* clocks, const float4* __restrict__ src, int iters, int compute_waves) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid * 16; e < 64 * 1024; e += blockDim.x * 16) *reinterpret_cast<float4*>(smem + e) = make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f);
    __syncthreads();
    if (wave >= compute_waves) {
        // loader role (MODE 2 only): 16 B per lane per step from global memory into LDS, two requests in flight
        if (MODE != 2) return;
        const size_t stride = (size_t)gridDim.x * 4 * 64;
        size_t idx = ((size_t)blockIdx.x * 4 + (wave - compute_waves)) * 64 + lane;
        float4 v0 = src[idx], v1 = src[idx + stride];
        for (int it = 0; it < iters; it += 16) {       // 4 KB per 16 compute steps and CU (the weight gradient moves 0.23 KB per step)
            *reinterpret_cast<float4*>(smem + 65536 + (wave - compute_waves) * 1024 + lane * 16) = v0;
            v0 = v1;
            idx += stride;
            v1 = src[(idx + stride) & ((1u << 24) - 1)];
        }
        if (v0.x == 123.f) out[1] = v0.x;
        return;
    }
    v4f acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int base = (wave & 3) * 2048 + (lane >> 4) * 64 + (lane & 15) * 4;
    float av[2], bv[2][9];
    auto load = [&](int t, int buf) {
        if (MODE == 0) {
            av[buf] = 1.0f + t;
#pragma unroll
            for (int k = 0; k < 9; ++k) bv[buf][k] = 2.0f + k;
        } else {
            const int o = base + (t & 15) * 256;
            av[buf] = *reinterpret_cast<const float*>(smem + 32768 + o);
#pragma unroll
            for (int k = 0; k < 9; ++k) bv[buf][k] = *reinterpret_cast<const float*>(smem + o + (k / 3) * 2048 + (k % 3) * 64);
        }
    };
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it += 16) {
        load(0, 0);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t + 1 < 16) load(t + 1, (t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t & 1], bv[t & 1][k], acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 123.456f) out[0] = s;
    if (tid == 0 && blockIdx.x == gridDim.x / 2) { clocks[0] = t1 - t0; clocks[1] = w1 - w0; }
}

template <int MODE>
void run(int grid, int compute_waves, int iters, const float4* src) {
    float* out; long long* clk;
    hipMalloc(&out, 8); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 64 * compute_waves + (MODE == 2 ? 256 : 0);
    const size_t lds = 65536 + 4096;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop<MODE>, dim3(grid), dim3(threads), lds, 0, out, clk, src, iters, compute_waves);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const int wps = compute_waves / 4;
    const double per_simd = (double)iters * 9 * wps * ((grid + 255) / 256);
    const double flops = (double)grid * compute_waves * iters * 9 * 2048.0;
    printf("mode %d  grid %4d  compute waves/SIMD %d : %7.1f us  %6.2f ns/MFMA/SIMD  %5.1f clk/MFMA/SIMD  shader clock %4.0f MHz  %6.1f TFLOP/s\n", MODE,
           grid, wps, ms * 1e3, ms * 1e6 / per_simd, (double)c[0] / (iters * 9.0 * wps), c[1] > 0 ? (double)c[0] / c[1] * 100.0 : 0.0,
           flops / (ms * 1e-3) * 1e-12);
    hipFree(out); hipFree(clk);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 8000;      // steps per wave (multiple of 16)
    float4* src; hipMalloc(&src, (size_t)(1u << 24) * 16 + (1u << 22)); hipMemset(src, 0, (size_t)(1u << 24) * 16);
    for (int grid : {32, 128, 232, 256}) {
        run<0>(grid, 4, iters, src);
        run<1>(grid, 4, iters, src);
        run<2>(grid, 4, iters, src);
        run<1>(grid, 8, iters, src);
    }
    return 0;
}
