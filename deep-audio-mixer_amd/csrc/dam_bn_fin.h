// "The last workgroup finalizes": folds the tiny BatchNorm finalize launches (5-7 us each, ~60 per ResNet18 step) into
// the kernels that produce the partial records.
//
// Hand-off between workgroups of one launch without L2 write-back / invalidate fences (which cost ~125 us per kernel when
// every workgroup executes them: round 1 tried __threadfence()): the form the microarch guide lists as valid and cheap --
//   producer : partial records stored with `sc1` (agent-scope relaxed atomic stores: write-through, the line leaves the
//              XCD's L2), every storing wave drains them (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane does a returning
//              agent-scope atomic add on the arrival counter;
//   consumer : the workgroup whose add returned total-1 arrived last; it reads the records with `sc1` loads (agent-scope
//              relaxed atomic loads: never served from a stale L1 line) behind a workgroup barrier, and resets the counter
//              for the next launch on the stream.
// The counter is a zero-initialised device word owned by the caller; launches that share it must be stream-ordered.
#pragma once
#include "dam_common.h"

namespace dam {

// Records [parts][C][3] a statistics producer may emit (the size dam_bn_workspace_floats() provides for): the BatchNorm
// kernels' own passes use up to 1024, the loader-wave convolution up to one per (workgroup or tile, wave).
constexpr int BN_RECORDS_MAX = 2048;
constexpr int BN_BWD_RECORDS_MAX = 1024;      // records [..][C][2] that dam_bn_backward_f32 takes as partials_given

struct BnFinArgs {            // device pointers; the launch-side mirror of dam_bn_fin (include/dam_hip.h)
    const float* gamma;
    const float* beta;
    float* running_mean;      // may be null
    float* running_var;
    long long* num_batches;   // may be null
    float momentum, eps;
    float* save_mean;
    float* save_invstd;
    float* scale;
    float* shift;
    unsigned* counter;        // null: no in-kernel finalize
};

__device__ __forceinline__ void store_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_sc1(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by EVERY thread of the workgroup after its record stores.  Returns true in every thread of the one workgroup
// that arrived last.  `slot` is one word of LDS.
__device__ __forceinline__ bool block_arrive_last(unsigned* counter, unsigned total, unsigned* slot) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's sc1 stores have left
    __syncthreads();
    if (threadIdx.x == 0) *slot = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = *slot == total - 1;
    if (last && threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return last;
}

// Chan merge of (n, mean, M2) records in double.
struct Moments { double n, mean, m2; };
__device__ __forceinline__ void merge(Moments& a, double nb, double mb, double qb) {
    if (nb == 0) return;
    const double nn = a.n + nb, d = mb - a.mean;
    a.mean += d * (nb / nn);
    a.m2 += qb + d * d * (a.n * nb / nn);
    a.n = nn;
}

// Forward statistics: partial [parts][C][3] = (n, mean, M2) -> save_mean / save_invstd / scale / shift + running update.
// All `nthreads` threads of the calling workgroup take part; scratch: >= nthreads * 3 doubles of LDS (16-byte aligned).
// Thread layout: channel = t % C, slice = t / C; a slice merges parts slice, slice + S, ... then the slices are merged.
__device__ __forceinline__ void bn_stats_finalize_block(const float* partial, int parts, int C, const BnFinArgs& a,
                                                        double* scratch, int tid, int nthreads) {
    const int S = nthreads / C > 0 ? nthreads / C : 1;      // slices per channel (C <= nthreads required when S == 1)
    for (int c0 = 0; c0 < C; c0 += nthreads) {              // C > nthreads: several rounds of one slice each
        const int c = c0 + tid % (C < nthreads ? C : nthreads), sl = tid / (C < nthreads ? C : nthreads);
        Moments m{0.0, 0.0, 0.0};
        if (c < C && sl < S) {
            // the records were dropped from L2 by their sc1 stores: every load is a memory-side round trip, so a thread
            // requests eight records before it merges the first (one record at a time cost 10-15 us per launch)
            constexpr int U = 8;
            for (int p0 = sl; p0 < parts; p0 += S * U) {
                float rn[U], rm[U], rq[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = p0 + u * S;
                    const float* o = partial + ((size_t)(p < parts ? p : sl) * C + c) * 3;
                    rn[u] = load_sc1(o); rm[u] = load_sc1(o + 1); rq[u] = load_sc1(o + 2);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (p0 + u * S < parts) merge(m, (double)rn[u], (double)rm[u], (double)rq[u]);
            }
        }
        scratch[tid * 3] = m.n; scratch[tid * 3 + 1] = m.mean; scratch[tid * 3 + 2] = m.m2;
        __syncthreads();
        if (sl == 0 && c < C) {
            const int stride = C < nthreads ? C : nthreads;
            for (int s = 1; s < S; ++s) {
                const double* o = scratch + (size_t)(tid + s * stride) * 3;
                merge(m, o[0], o[1], o[2]);
            }
            const double var = m.m2 / m.n;
            const float mean = (float)m.mean;
            const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
            a.save_mean[c] = mean;
            a.save_invstd[c] = invstd;
            const float sc = a.gamma[c] * invstd;
            a.scale[c] = sc;
            a.shift[c] = a.beta[c] - mean * sc;
            if (a.running_mean) {
                const double unbiased = m.n > 1 ? m.m2 / (m.n - 1) : var;
                a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mean;
                a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
            }
        }
        __syncthreads();
    }
    if (tid == 0 && a.num_batches) *a.num_batches += 1;
}

}  // namespace dam
