// tools/mfma_peak.hip -- what v_mfma_f32_16x16x4_f32 sustains on this chip: a register-only loop (no memory), NACC independent
// accumulators per wave, W waves per SIMD, G workgroups.  Prints time per MFMA per SIMD in ns (hipEvents) and in shader clocks
// (s_memtime), so the sustained clock under matrix load and the issue cost of dependent chains can be read off separately.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(1024) void mfma_loop(float* out, long long* clocks, int iters, float a0, float b0) {
    v4f acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clocks[0] = t1 - t0;
}

template <int NACC>
void run(int grid, int waves_per_simd, int iters) {
    float* out; long long* clk;
    hipMalloc(&out, 4); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(threads), 0, 0, out, clk, iters, 1.0f, 2.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double per_simd = (double)iters * 4 * NACC * waves_per_simd * ((grid + 255) / 256);   // MFMAs through one SIMD's pipe
    const double flops = (double)grid * threads / 64 * iters * 4 * NACC * 2048.0;
    printf("nacc %d  grid %4d  waves/SIMD %d : %7.1f us  %6.2f ns/MFMA/SIMD  %6.1f clk/MFMA/SIMD (s_memtime)  %6.1f TFLOP/s\n", NACC, grid,
           waves_per_simd, ms * 1e3, ms * 1e6 / per_simd, (double)c / (iters * 4.0 * NACC * waves_per_simd), flops / (ms * 1e-3) * 1e-12);
    hipFree(out); hipFree(clk);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    for (int grid : {1, 64, 160, 256}) {
        for (int w : {1, 2, 4}) {
            run<1>(grid, w, iters);
            run<4>(grid, w, iters);
            run<8>(grid, w, iters);
        }
    }
    return 0;
}
