// Shared LDS staging of an input patch for the implicit-GEMM convolution kernels (forward, dgrad, wgrad).
#pragma once
#include "dam_common.h"

namespace dam {

struct PatchGeo {
    int H, W, C;             // input tensor rows, cols, channel stride (planes if in_nchw)
    int s;                   // input stride per output pixel (1 or 2): stride-2 de-interleaves even/odd columns
    int c0;                  // input column of patch column 0
    int PR, PWin, PWs, PWT;  // patch rows, input columns covered, slots per parity, slots per row
    int in_nchw;             // 1: [C][H][W] planes with C <= 16 (first layer)
    int relu_in;             // with in_scale: relu(x*scale+shift)
};

// Stages rows ih0 .. ih0+PR-1, channels [chunk0*16, (chunk0+nch)*16) of image `ximg` into
// smem as [chunk][row][slot][16 floats] (powers of two for nch index by shifts, others by division); everything outside the tensor
// (spatially or beyond C) is zero.  Called by all 256 threads of the workgroup.
// Loads are issued in batches of STAGE_U per thread before any of them is consumed, so a workgroup exposes
// one or two HBM latencies per patch instead of one per item.
constexpr int STAGE_U = 8;

__device__ __forceinline__ int fast_div(int e, int d, float inv_d) {   // e / d for 0 <= e < 2^22, d >= 1
    int q = (int)((float)e * inv_d);
    if (q * d > e) --q;
    if ((q + 1) * d <= e) ++q;
    return q;
}

__device__ __forceinline__ void stage_patch(unsigned char* smem, int chunk_bytes, const float* __restrict__ ximg,
                                            const PatchGeo& g, int ih0, int chunk0, int nch,
                                            const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                            int tid) {
    if (!g.in_nchw) {
        const int qpp = nch * 4;                        // float4 quads per pixel
        const int qshift = 31 - __builtin_clz(qpp);
        const bool qp2 = (qpp & (qpp - 1)) == 0;        // three chunks per tile (48 channels): quad index by division
        const float inv_qpp = 1.0f / (float)qpp;
        const int ipr = g.PWin * qpp;                   // items per patch row
        const float inv_ipr = 1.0f / (float)ipr;
        const int total = g.PR * ipr;
        const int ch0 = chunk0 * 16;
        // (row, item-in-row) of this thread's items advance incrementally: one division per call, not one per item
        const int step_pr = 256 / ipr, step_rem = 256 - step_pr * ipr;
        int pr = fast_div(tid, ipr, inv_ipr), rem = tid - pr * ipr;
        for (int base = tid; base < total; base += 256 * STAGE_U) {
            float4 v[STAGE_U];
            int dst[STAGE_U];
#pragma unroll
            for (int u = 0; u < STAGE_U; ++u) {
                const int e = base + 256 * u;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                dst[u] = -1;
                if (e < total) {
                    const int pw = qp2 ? rem >> qshift : fast_div(rem, qpp, inv_qpp), cq = rem - pw * qpp;
                    const int ih = ih0 + pr, iw = g.c0 + pw;
                    const int slot = g.s == 1 ? pw : (pw & 1) * g.PWs + (pw >> 1);
                    dst[u] = (cq >> 2) * chunk_bytes + ((pr * g.PWT + slot) * 16 + (cq & 3) * 4) * 4;
                    if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W && ch0 + cq * 4 < g.C)
                        v[u] = *reinterpret_cast<const float4*>(ximg + (unsigned)((ih * g.W + iw) * g.C + ch0 + cq * 4));
                    else
                        dst[u] = -2 - dst[u];            // out of the tensor: stays zero, no BN prologue
                }
                rem += step_rem; pr += step_pr;
                if (rem >= ipr) { rem -= ipr; ++pr; }
            }
#pragma unroll
            for (int u = 0; u < STAGE_U; ++u) {
                if (dst[u] == -1) continue;
                float4 x = v[u];
                int d = dst[u];
                if (d < 0) {
                    d = -2 - d;
                } else if (in_scale) {
                    const int e = base + 256 * u;
                    const int r_ = e - fast_div(e, ipr, inv_ipr) * ipr;
                    const int cq = qp2 ? r_ & (qpp - 1) : r_ - fast_div(r_, qpp, inv_qpp) * qpp;
                    const float4 sc = *reinterpret_cast<const float4*>(in_scale + ch0 + cq * 4);
                    const float4 sh = *reinterpret_cast<const float4*>(in_shift + ch0 + cq * 4);
                    x.x = fmaf(x.x, sc.x, sh.x); x.y = fmaf(x.y, sc.y, sh.y);
                    x.z = fmaf(x.z, sc.z, sh.z); x.w = fmaf(x.w, sc.w, sh.w);
                    if (g.relu_in) {
                        x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
                    }
                }
                *reinterpret_cast<float4*>(smem + d) = x;
            }
        }
    } else {
        // first layer: a thread gathers 4 channel planes of one pixel (column fastest across lanes), zero-fills C..15
        const size_t plane = (size_t)g.H * g.W;
        const int ipr = g.PWin * 4;
        const float inv_ipr = 1.0f / (float)ipr, inv_pw = 1.0f / (float)g.PWin;
        const int total = g.PR * ipr;
        constexpr int U = 4;
        for (int base = tid; base < total; base += 256 * U) {
            float v[U][4];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = base + 256 * u;
                dst[u] = -1;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[u][r] = 0.f;
                if (e < total) {
                    const int pr = fast_div(e, ipr, inv_ipr), rem = e - pr * ipr;
                    const int q = fast_div(rem, g.PWin, inv_pw), pw = rem - q * g.PWin;
                    const int ih = ih0 + pr, iw = g.c0 + pw;
                    const int slot = g.s == 1 ? pw : (pw & 1) * g.PWs + (pw >> 1);
                    dst[u] = ((pr * g.PWT + slot) * 16 + q * 4) * 4;
                    if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ch = q * 4 + r;
                            if (ch < g.C) v[u][r] = ximg[ch * plane + (size_t)ih * g.W + iw];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) *reinterpret_cast<float4*>(smem + dst[u]) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
        }
    }
}

}  // namespace dam
