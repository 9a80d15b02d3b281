// Diagnostic: read bandwidth of (a) a classic grid-stride float4 stream and (b) the strip kernel's loader pattern
// (one workgroup per strip, 4 waves, each wave 8 x 1 KB pieces in flight, walking its strip sequentially).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void stream_read(const float4* __restrict__ x, size_t n4, float* out) {
    float4 a = make_float4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = x[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 12345.f) out[0] = 1;
}
template <int PU>
__global__ __launch_bounds__(256) void strip_read(const float4* __restrict__ x, size_t strip4, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4* base = x + (size_t)blockIdx.x * strip4;
    float4 a = make_float4(0, 0, 0, 0);
    const size_t pieces = strip4 / 64;
    for (size_t p0 = wave; p0 < pieces; p0 += 4 * PU) {
        float4 v[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) { size_t p = p0 + 4 * u; v[u] = base[(p < pieces ? p : pieces - 1) * 64 + lane]; }
#pragma unroll
        for (int u = 0; u < PU; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    if (a.x + a.y + a.z + a.w == 12345.f) out[0] = 1;
}
int main() {
    const size_t bytes = 68224000ull * 1;   // layer1 activation
    float4* x; float* out; hipMalloc(&x, bytes * 4); hipMalloc(&out, 4);
    hipMemset(x, 0, bytes * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch, double nbytes) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-44s %.1f us  %.2f TB/s\n", name, ms * 1e3, nbytes / ms / 1e9);
    };
    const size_t n4 = bytes / 16;
    timeit("stream 68 MB (fits MALL), 2048 blocks", [&] { hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, x, n4, out); }, bytes);
    timeit("stream 272 MB (beyond MALL), 2048 blocks", [&] { hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, x, n4 * 4, out); }, bytes * 4.0);
    timeit("strip pattern 68 MB, 256 WGs x 4 waves x 8", [&] { hipLaunchKernelGGL(strip_read<8>, dim3(256), dim3(256), 0, 0, x, n4 / 256, out); }, bytes);
    timeit("strip pattern 68 MB, 512 WGs x 4 waves x 8", [&] { hipLaunchKernelGGL(strip_read<8>, dim3(512), dim3(256), 0, 0, x, n4 / 512, out); }, bytes);
    timeit("strip pattern 68 MB, 256 WGs x 4 waves x 16", [&] { hipLaunchKernelGGL(strip_read<16>, dim3(256), dim3(256), 0, 0, x, n4 / 256, out); }, bytes);
    timeit("strip pattern 272 MB, 256 WGs x 4 waves x 16", [&] { hipLaunchKernelGGL(strip_read<16>, dim3(256), dim3(256), 0, 0, x, n4 * 4 / 256, out); }, bytes * 4.0);
    return 0;
}
