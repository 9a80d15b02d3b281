// Shared helpers for the gfx950 kernels of libdam_hip.so (device + launch side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dam_hip.h"

#define DAM_CHECK_LAUNCH()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return DAM_ERR_LAUNCH; \
    } while (0)

namespace dam {

constexpr int WAVE = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// 16-byte vector with 4-byte alignment: global_load_dwordx4 does not need more on gfx950.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef double f64x2_u __attribute__((ext_vector_type(2), aligned(8)));

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// Orders LDS traffic of ONE wave: earlier ds_writes of any lane are visible to later
// ds_reads of any lane of the same wave (LDS serves a wave's instructions in order).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The same ordering without the fences' side effects: a workgroup-scope release also waits for the wave's outstanding
// GLOBAL loads and stores (s_waitcnt vmcnt(0)), which a purely intra-wave LDS exchange does not need.  The LDS executes one
// wave's instructions in order, so later ds_reads see earlier ds_writes; what is left is keeping the compiler from moving
// LDS accesses across this point and draining the wave's own LDS queue.
__device__ __forceinline__ void wave_lds_order() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float wave_sum16(float v) {   // sum over the 16 lanes sharing lane>>4
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
    v = wave_sum16(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Launch-side caches.  A kernel's raised LDS limit (hipFuncSetAttribute), the CU count and an occupancy answer belong to ONE
// device: a process that drives a second GPU must not reuse what it learnt on the first.  PerDevice<T> is a zero-initialised
// slot per device ordinal, read through the calling thread's current device.
constexpr int MAX_DEVICES = 32;
static inline int device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d % MAX_DEVICES;
}
template <class T>
struct PerDevice {
    T v[MAX_DEVICES] = {};
    T& operator()() { return v[device_slot()]; }
};
static inline int device_cus() {
    static PerDevice<int> cus;
    int& n = cus();
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    }
    return n;
}

}  // namespace dam
