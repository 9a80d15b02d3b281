// tools/grid_barrier_probe.hip -- what does a device-scope hand-off between two phases of ONE persistent launch cost on MI355X,
// against the same phases as dependent launches of a hipGraph?  ("measure the persistent form instead of
// citing round 1" -- the first rung: the hand-off itself, with a phase body shaped like a deep-stage convolution's.)
//
//   hipcc -O3 --offload-arch=gfx950 tools/grid_barrier_probe.hip -o tools/grid_barrier_probe && tools/grid_barrier_probe
//
// Phase body: every workgroup (256 threads) issues `mfma` v_mfma_f32_16x16x4_f32 per wave, writes 4 KB of "activations" and, in
// the next phase, reads the 4 KB another workgroup wrote (so the hand-off really carries data through L2 / HBM).
// Persistent form: grid = CUs x per_cu workgroups, all co-resident; barrier = per-XCD arrival counters (workgroup b sits on XCD
// b % 8) -> one top counter -> a generation word every workgroup polls with s_sleep; every spin is bounded (a stuck barrier sets
// an error flag and the kernel drains).  Launch form: one kernel per phase, N dependent kernel nodes captured into a hipGraph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

struct Barrier {
    unsigned shard[8 * 32];     // one counter per XCD, 128 bytes apart
    unsigned top;
    unsigned pad0[31];
    unsigned gen;
    unsigned pad1[31];
    unsigned error;
};

__device__ __forceinline__ void phase_body(int phase, int mfma, const float* __restrict__ in, float* __restrict__ out, int nwg, v4f& acc) {
    const int tid = threadIdx.x, wg = blockIdx.x;
    // read what workgroup (wg + 1) % nwg wrote in the previous phase (4 KB), fold it into the MFMA operands
    const float x = in[(size_t)((wg + 1) % nwg) * 1024 + tid * 4 + (phase & 3)];
    float a = x * 1e-9f + 1.0f, b = 0.5f;
    for (int i = 0; i < mfma; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    float4 o = make_float4(acc.x, acc.y, acc.z, acc.w);
    *reinterpret_cast<float4*>(out + (size_t)wg * 1024 + tid * 4) = o;
}

// MODE 0: per-XCD shards -> top counter, pollers sleep between reads; 1: ONE counter, tight polling; 2: shards, tight polling
template <int MODE>
__device__ __forceinline__ bool grid_barrier(Barrier* bar, int nwg, unsigned& local_gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();                                         // this workgroup's stores are visible device-wide
        if (MODE == 1) {
            if (atomicAdd(&bar->top, 1u) == (unsigned)nwg - 1) {
                bar->top = 0;
                __threadfence();
                atomicAdd(&bar->gen, 1u);
            }
        } else {
            const int x = blockIdx.x & 7;
            const unsigned per_shard = (unsigned)((nwg - x + 7) / 8);
            if (atomicAdd(&bar->shard[x * 32], 1u) == per_shard - 1) {
                bar->shard[x * 32] = 0;
                if (atomicAdd(&bar->top, 1u) == 7u) {
                    bar->top = 0;
                    __threadfence();
                    atomicAdd(&bar->gen, 1u);
                }
            }
        }
        int spins = 0;
        while (__hip_atomic_load(&bar->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == local_gen) {
            if (MODE == 0) __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 24)) { bar->error = 1; ok = false; break; }       // bounded, then drain
        }
        ++local_gen;
    }
    __syncthreads();
    return ok;
}

template <int MODE>
__global__ __launch_bounds__(256) void persistent_kernel(Barrier* bar, int phases, int mfma, float* buf0, float* buf1, int nwg, float* sink) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    unsigned gen = 0;
    if (threadIdx.x == 0) gen = __hip_atomic_load(&bar->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    for (int p = 0; p < phases; ++p) {
        phase_body(p, mfma, (p & 1) ? buf1 : buf0, (p & 1) ? buf0 : buf1, nwg, acc);
        if (p + 1 < phases && !grid_barrier<MODE>(bar, nwg, gen)) break;
        if (__hip_atomic_load(&bar->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (acc.x == 12345.678f) sink[0] = acc.x;
}

__global__ __launch_bounds__(256) void phase_kernel(int p, int mfma, const float* in, float* out, int nwg, float* sink) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    phase_body(p, mfma, in, out, nwg, acc);
    if (acc.x == 12345.678f) sink[0] = acc.x;
}

int main() {
    int dev = 0, cus = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    Barrier* bar;
    float *b0, *b1, *sink;
    const int max_wg = cus * 4;
    CHECK(hipMalloc(&bar, sizeof(Barrier)));
    CHECK(hipMalloc(&b0, (size_t)max_wg * 4096));
    CHECK(hipMalloc(&b1, (size_t)max_wg * 4096));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(b0, 0, (size_t)max_wg * 4096));
    CHECK(hipMemset(b1, 0, (size_t)max_wg * 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int phases = 24;
    printf("MI355X: %d CUs; %d phases per run; us per phase (total / phases), median of 7 runs\n", cus, phases);
    printf("%-6s %-8s %-10s %-22s %-22s %-12s\n", "mode", "per_cu", "mfma/wave", "persistent + barrier", "hipGraph of launches", "difference");
    for (int mode = 0; mode < 3; ++mode)
    for (int per_cu = 1; per_cu <= 2; ++per_cu)
        for (int mfma : {0, 64, 256, 768}) {
            const int nwg = cus * per_cu;
            int occ = 0;
            CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persistent_kernel<0>, 256, 0));
            if (occ < per_cu) { printf("occupancy %d < %d: skipped\n", occ, per_cu); continue; }
            // launch chain captured into a graph
            hipGraph_t graph;
            hipGraphExec_t exec;
            CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int p = 0; p < phases; ++p)
                hipLaunchKernelGGL(phase_kernel, dim3(nwg), dim3(256), 0, st, p, mfma, (p & 1) ? b1 : b0, (p & 1) ? b0 : b1, nwg, sink);
            CHECK(hipStreamEndCapture(st, &graph));
            CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            std::vector<float> tp, tg;
            for (int r = 0; r < 9; ++r) {
                CHECK(hipMemsetAsync(bar, 0, sizeof(Barrier), st));
                CHECK(hipEventRecord(e0, st));
                if (mode == 0) hipLaunchKernelGGL(persistent_kernel<0>, dim3(nwg), dim3(256), 0, st, bar, phases, mfma, b0, b1, nwg, sink);
                else if (mode == 1) hipLaunchKernelGGL(persistent_kernel<1>, dim3(nwg), dim3(256), 0, st, bar, phases, mfma, b0, b1, nwg, sink);
                else hipLaunchKernelGGL(persistent_kernel<2>, dim3(nwg), dim3(256), 0, st, bar, phases, mfma, b0, b1, nwg, sink);
                CHECK(hipEventRecord(e1, st));
                CHECK(hipStreamSynchronize(st));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) tp.push_back(ms * 1e3f / phases);
                Barrier h;
                CHECK(hipMemcpy(&h, bar, sizeof(h), hipMemcpyDeviceToHost));
                if (h.error) { printf("barrier timed out (per_cu %d mfma %d)\n", per_cu, mfma); return 2; }
                CHECK(hipEventRecord(e0, st));
                CHECK(hipGraphLaunch(exec, st));
                CHECK(hipEventRecord(e1, st));
                CHECK(hipStreamSynchronize(st));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) tg.push_back(ms * 1e3f / phases);
            }
            auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            printf("%-6d %-8d %-10d %-22.2f %-22.2f %-12.2f\n", mode, per_cu, mfma, med(tp), med(tg), med(tg) - med(tp));
            CHECK(hipGraphExecDestroy(exec));
            CHECK(hipGraphDestroy(graph));
        }
    return 0;
}
