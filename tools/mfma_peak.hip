// Diagnostic (not part of the library): sustained v_mfma_f32_16x16x4_f32 rate, with and without ds_read_b128 feeding.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) sm[i] = (float)(i % 7) * 0.001f;
    __syncthreads();
    v4f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f){0, 0, 0, 0};
    float4 a = make_float4(1.f, 2.f, 3.f, 4.f), b = make_float4(0.5f, 0.25f, 0.125f, 1.f);
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            b = *reinterpret_cast<const float4*>(&sm[((it * 64 + lane) * 4) & 16383]);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, bool LDS>
void run(const char* name, int blocks, int iters) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flop = (double)blocks * 4 * iters * NACC * 4 * 2048.0;
    printf("%-28s blocks %5d: %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
    hipFree(out);
}
int main() {
    run<4, false>("4 acc, regs only", 256, 20000);
    run<4, false>("4 acc, regs only", 512, 20000);
    run<4, false>("4 acc, regs only", 1024, 20000);
    run<4, true>("4 acc + ds_read_b128", 512, 20000);
    run<1, true>("1 acc + ds_read_b128", 512, 40000);
    run<16, true>("16 acc + ds_read_b128", 512, 5000);
    return 0;
}
