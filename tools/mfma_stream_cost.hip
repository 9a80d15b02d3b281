// Diagnostic: cost of VALU / LDS instructions interleaved into one wave's MFMA stream (1 wave per SIMD).
// Each iteration: 16 independent-ish v_mfma_f32_16x16x4_f32 (4 accumulators x 4 k-steps) + KV v_add_u32 + KL ds_read_b128
// whose results are consumed by the MFMAs of the NEXT iteration (two-deep pipeline, like the conv item loop).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int KV, int KL, int WAITPOS, int PATTERN>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = 1.f;
    __syncthreads();
    v4f acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4f){0, 0, 0, 0};
    v4f cur[5], nxt[5];
    for (int i = 0; i < 5; ++i) cur[i] = (v4f){1.f, 2.f, 3.f, 4.f};
    int addr[8];
    // PATTERN 0: lane*16 (contiguous); 1: NHWC pixel-major cells, lane (k = lane>>4, n = lane&15) reads quad k of pixel n
    for (int i = 0; i < 8; ++i) addr[i] = (PATTERN == 0 ? lane * 16 : (lane & 15) * 64 + (lane >> 4) * 16) + i * 1024;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define GROUP(CUR, NXT)                                                                                           \
    do {                                                                                                          \
        _Pragma("unroll") for (int v = 0; v < KV; ++v) asm volatile("v_add_u32 %0, %1, %0" : "+v"(addr[v & 7]) : "s"(it & 0)); \
        _Pragma("unroll") for (int l = 0; l < KL; ++l) asm volatile("ds_read_b128 %0, %1" : "=v"(NXT[l]) : "v"(addr[l]));   \
        if (KL && WAITPOS == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(KL) : "memory");                      \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                             \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(CUR[0][r], CUR[1 + i][r], acc[i], 0, 0, 0);          \
        if (KL && WAITPOS == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                \
    } while (0)
    for (int it = 0; it < iters; it += 2) {
        GROUP(cur, nxt);
        GROUP(nxt, cur);
    }
#undef GROUP
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float res = 0;
    for (int i = 0; i < 4; ++i) res += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; ++i) res += addr[i];
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = res;
}
template <int KV, int KL, int WAITPOS = 1, int PATTERN = 0>
void run(int iters) {
    const int blocks = 256;
    float* out; unsigned long long* cyc; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<KV, KL, WAITPOS, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    static unsigned long long h[256 * 4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks * 4; ++i) m += (double)h[i];
    m /= blocks * 4;
    printf("%s wait %s: 16 mfma + %2d v_add + %d ds_read_b128: %7.1f cycles per group (%.1f over 512)\n", PATTERN ? "pixel-major" : "contiguous ", WAITPOS ? "after mfmas " : "before mfmas", KV, KL, m / iters, m / iters - 512.0);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 0>(4000); run<4, 0>(4000); run<8, 0>(4000); run<16, 0>(4000); run<32, 0>(4000);
    run<0, 1>(4000); run<0, 3>(4000); run<0, 5>(4000); run<4, 5>(4000);
    run<0, 1, 0>(4000); run<0, 3, 0>(4000); run<0, 5, 0>(4000); run<4, 5, 0>(4000);
    run<0, 1, 0, 1>(4000); run<0, 3, 0, 1>(4000); run<0, 5, 0, 1>(4000); run<4, 5, 0, 1>(4000);
    return 0;
}
