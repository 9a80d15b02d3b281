// Diagnostic: what a co-resident wave's VALU / SALU / LDS instructions cost next to a wave streaming MFMAs on the same SIMD.
// 512 threads = 2 waves per SIMD: waves 0-3 issue a fixed number of v_mfma_f32_16x16x4_f32, waves 4-7 run `other` work.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE, int PRIO, int SWAP, int YIELD>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int mfma_iters, int other_iters) {
    __shared__ __attribute__((aligned(16))) float sm[4096];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 512) sm[i] = 1.f;
    __syncthreads();
    float res = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if ((wave < 4) != (SWAP != 0)) {
        v4f acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (v4f){0, 0, 0, 0};
        float a = 1.f + lane, b = 0.5f;
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                    if (YIELD == 1) asm volatile("s_nop 0");
                }
            if (YIELD == 2) asm volatile("s_sleep 1");
            if (YIELD == 3) asm volatile("s_nop 7");
            if (YIELD == 4) asm volatile("s_nop 15\n\ts_nop 15");
            if (YIELD == 5) asm volatile("s_setprio 0\n\ts_nop 3\n\ts_setprio 1");
        }
        for (int i = 0; i < 4; ++i) res += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    } else if (MODE != 0) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        if (MODE == 1) {            // independent VALU: 16 v_fma per iteration on 8 chains
            float x[8];
            for (int i = 0; i < 8; ++i) x[i] = lane + i;
            for (int it = 0; it < other_iters; ++it) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(0.5f));
            }
            for (int i = 0; i < 8; ++i) res += x[i];
        } else if (MODE == 2) {     // SALU: 16 s_add per iteration
            int s = other_iters;
            for (int it = 0; it < other_iters; ++it) {
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
            }
            res = (float)s;
        } else if (MODE == 3) {     // LDS: 4 ds_read_b128 per iteration
            float accx = 0.f;
            for (int it = 0; it < other_iters; ++it) {
                float4 v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("ds_read_b128 %0, %1" : "=v"(v[r]) : "v"((((it + r) * 256 + lane * 4) & 4095) * 4));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int r = 0; r < 4; ++r) accx += v[r].x;
            }
            res = accx;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + (SWAP ? wave ^ 4 : wave)] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = res;
}
template <int MODE, int PRIO, int SWAP = 0, int YIELD = 0>
void run(const char* name, int mfma_iters, int other_iters, double other_ops_per_iter) {
    const int blocks = 256;
    float* out; unsigned long long* cyc; hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE, PRIO, SWAP, YIELD>), dim3(blocks), dim3(512), 0, 0, out, cyc, mfma_iters, other_iters);
    hipDeviceSynchronize();
    static unsigned long long h[256 * 8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, o = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : o) += (double)h[b * 8 + w];
    m /= blocks * 4; o /= blocks * 4;
    printf("%-34s mfma wave: %7.1f cyc/mfma | other wave: %9.0f cycles", name, m / (mfma_iters * 16.0), o);
    if (other_iters) printf(" = %6.1f cyc/op", o / (other_iters * other_ops_per_iter));
    printf("\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 0>("mfma alone", 2000, 0, 1);
    run<1, 0>("+ VALU wave (prio 0), short", 2000, 500, 16);
    run<1, 3>("+ VALU wave (prio 3), short", 2000, 500, 16);
    run<1, 0>("+ VALU wave (prio 0), long", 2000, 8000, 16);
    run<1, 3>("+ VALU wave (prio 3), long", 2000, 8000, 16);
    run<2, 0>("+ SALU wave (prio 0)", 2000, 2000, 16);
    run<2, 3>("+ SALU wave (prio 3)", 2000, 2000, 16);
    run<3, 0>("+ LDS read wave (prio 0)", 2000, 2000, 4);
    run<3, 3>("+ LDS read wave (prio 3)", 2000, 2000, 4);
    run<1, 0, 1>("older VALU wave (prio 0), short", 2000, 500, 16);
    run<1, 3, 1>("older VALU wave (prio 3), short", 2000, 500, 16);
    run<1, 0, 1>("older VALU wave (prio 0), long", 2000, 8000, 16);
    run<3, 0, 1>("older LDS read wave (prio 0)", 2000, 2000, 4);
    run<1, 0, 0, 1>("yield s_nop 0 per mfma; VALU short", 2000, 500, 16);
    run<1, 0, 0, 2>("yield s_sleep 1 per 16; VALU short", 2000, 500, 16);
    run<1, 0, 0, 3>("yield s_nop 7 per 16; VALU short", 2000, 500, 16);
    run<1, 0, 0, 4>("yield 2x s_nop 15 per 16; VALU short", 2000, 500, 16);
    run<1, 3, 0, 4>("yield 2x s_nop 15 per 16; VALU prio3", 2000, 500, 16);
    run<1, 3, 0, 5>("yield setprio dance; VALU prio3", 2000, 500, 16);
    run<1, 0, 0, 2>("yield s_sleep 1 per 16; VALU long", 2000, 8000, 16);
    return 0;
}
