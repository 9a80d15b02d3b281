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
// smem as [chunk][row][slot][16 floats]; nch is a power of two; everything outside the tensor
// (spatially or beyond C) is zero.  Called by all 4 waves of the workgroup.
__device__ __forceinline__ void stage_patch(unsigned char* smem, int chunk_bytes, const float* __restrict__ ximg,
                                            const PatchGeo& g, int ih0, int chunk0, int nch,
                                            const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                            int lane, int wave) {
    if (!g.in_nchw) {
        const int qpp = nch * 4;                        // float4 quads per pixel
        const int qshift = 31 - __builtin_clz(qpp);
        const int items = g.PWin * qpp;
        const int ch0 = chunk0 * 16;
        for (int pr = wave; pr < g.PR; pr += 4) {
            const int ih = ih0 + pr;
            const bool row_ok = ih >= 0 && ih < g.H;
            const float* xr = ximg + (size_t)(row_ok ? ih : 0) * g.W * g.C + ch0;
            for (int e = lane; e < items; e += 64) {
                const int pw = e >> qshift, cq = e & (qpp - 1);
                const int iw = g.c0 + pw;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row_ok && iw >= 0 && iw < g.W && ch0 + cq * 4 < g.C) {
                    v = *reinterpret_cast<const float4*>(xr + (size_t)iw * g.C + cq * 4);
                    if (in_scale) {
                        const float4 sc = *reinterpret_cast<const float4*>(in_scale + ch0 + cq * 4);
                        const float4 sh = *reinterpret_cast<const float4*>(in_shift + ch0 + cq * 4);
                        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
                        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                        if (g.relu_in) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                    }
                }
                const int slot = g.s == 1 ? pw : (pw & 1) * g.PWs + (pw >> 1);
                *reinterpret_cast<float4*>(smem + (cq >> 2) * chunk_bytes + ((pr * g.PWT + slot) * 16 + (cq & 3) * 4) * 4) = v;
            }
        }
    } else {
        // first layer: lane gathers 4 channel planes of one pixel (column fastest across lanes), zero-fills C..15
        const size_t plane = (size_t)g.H * g.W;
        const int items = g.PWin * 4;
        for (int pr = wave; pr < g.PR; pr += 4) {
            const int ih = ih0 + pr;
            const bool row_ok = ih >= 0 && ih < g.H;
            for (int e = lane; e < items; e += 64) {
                const int q = e / g.PWin, pw = e - q * g.PWin;
                const int iw = g.c0 + pw;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (row_ok && iw >= 0 && iw < g.W) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ch = q * 4 + r;
                        if (ch < g.C) v[r] = ximg[ch * plane + (size_t)ih * g.W + iw];
                    }
                }
                const int slot = g.s == 1 ? pw : (pw & 1) * g.PWs + (pw >> 1);
                *reinterpret_cast<float4*>(smem + ((pr * g.PWT + slot) * 16 + q * 4) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

}  // namespace dam
