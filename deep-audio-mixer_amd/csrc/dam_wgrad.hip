// dam_wgrad.hip -- convolution weight gradient on the gfx950 fp32 matrix cores.
//
// Replaces the autograd weight-gradient of every nn.Conv2d on the path (models/model_resnet.py:11-21,64;
// models/model_scalar_1s.py:167-172; models/model_scalar_2s.py:25-30), reached from loss.backward() at
// model_trainer.py:36.
//
//   dW[n][k][kh][kw] = sum_{b, oh, ow} dY[b, oh, ow, n] * f(X[b, oh*s + kh*d - pad, ow*s + kw*d - pad, k])
//
// GEMM view per tap: D[n][k] += A[n][pix] * B[pix][k] with the PIXEL index as the MFMA reduction
// dimension (v_mfma_f32_16x16x4_f32, 4 pixels per step).  NHWC makes both operands plain ds_read_b32:
// lanes 0-15 read 16 consecutive channels of one pixel, the 4 lane groups read 4 consecutive pixels.
//   * a workgroup owns an output tile (16*TNB out-channels x 16*TKB in-channels x TA x TB taps) and walks a
//     strided subset of 256-pixel tiles (split-K over pixels); per tile it stages the dY pixels and the
//     input rows they touch (same patch layout as the forward kernel, optional fused BN-apply+ReLU);
//   * each of the 4 waves reduces its own quarter of the tile's pixels into the full output tile held in
//     registers; waves are combined through LDS once at the end, the workgroup writes ONE partial slab;
//   * a second kernel sums the slabs in a fixed order (deterministic, no float atomics) and writes
//     torch's [O][I][KH][KW] layout.
#include "dam_common.h"
#include "dam_conv_stage.h"

namespace dam {

struct WgradGeo {
    int B, H, W, C;          // input tensor
    int Ho, Wo, N;           // dY tensor [B][Ho][Wo][N]
    int s;                   // conv stride (1 or 2)
    int KH, KW;              // full kernel
    int off_h, step_h, off_w, step_w;   // tap (kh,kw) reads input (oh*s + off_h + kh*step_h, ...)
    int r0, c0;
    int PR, PWin, PWs, PWT;
    int nchunks;             // ceil(C/16) (1 if in_nchw)
    int nblk;                // N/16
    int tiles_n, tiles_k, tap_groups;
    int tiles_m, total_tiles, nsplit;
    int in_nchw, relu_in;
    int TMW;                 // pixels per tile (256 or 128): 4 waves x TMW/4 pixels
};

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

template <int TNB, int TKB, int TA, int TB>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradGeo g, const float* __restrict__ X,
                                                    const float* __restrict__ dY, const float* __restrict__ in_scale,
                                                    const float* __restrict__ in_shift, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBLK = TNB * TKB * TA * TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    int xt = blockIdx.x;
    const int tg = xt % g.tap_groups; xt /= g.tap_groups;
    const int tk = xt % g.tiles_k;
    const int tn = xt / g.tiles_k;
    const int a0 = tg * TA;
    const int HoWo = g.Ho * g.Wo;
    const int chunk_bytes = g.PR * g.PWT * 64;
    const int TMW = g.TMW, WP = g.TMW >> 2;   // pixels per wave
    unsigned char* dy_s = smem + TKB * chunk_bytes;        // [TNB][TMW][16] floats

    PatchGeo pg;
    pg.H = g.H; pg.W = g.W; pg.C = g.C; pg.s = g.s; pg.c0 = g.c0; pg.PR = g.PR; pg.PWin = g.PWin; pg.PWs = g.PWs;
    pg.PWT = g.PWT; pg.in_nchw = g.in_nchw; pg.relu_in = g.relu_in;

    v4f acc[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};

    const size_t img_elems = (size_t)g.H * g.W * g.C;
    for (int tile = blockIdx.y; tile < g.total_tiles; tile += g.nsplit) {
        const int img = tile / g.tiles_m;
        const int p0 = (tile - img * g.tiles_m) * TMW;
        const int oh_first = p0 / g.Wo;
        __syncthreads();
        stage_patch(smem, chunk_bytes, X + (size_t)img * img_elems, pg, oh_first * g.s + g.r0, tk * TKB, TKB,
                    in_scale, in_shift, tid);
        {   // dY pixels p0 .. p0+TMW-1, channels of this n tile; zero beyond the image / channel count
            constexpr int QPP = TNB * 4;
            const float* dyb = dY + ((size_t)img * HoWo) * g.N + tn * TNB * 16;
            constexpr int U = 4;
            for (int base = tid; base < TMW * QPP; base += 256 * U) {
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = base + 256 * u, pl = e / QPP, cq = e % QPP;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e < TMW * QPP && p0 + pl < HoWo && tn * TNB * 16 + cq * 4 < g.N)
                        v[u] = *reinterpret_cast<const float4*>(dyb + (size_t)(p0 + pl) * g.N + cq * 4);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = base + 256 * u, pl = e / QPP, cq = e % QPP;
                    if (e < TMW * QPP)
                        *reinterpret_cast<float4*>(dy_s + (((cq >> 2) * TMW + pl) * 16 + (cq & 3) * 4) * 4) = v[u];
                }
            }
        }
        __syncthreads();

        // this wave's 64 pixels, 4 per MFMA step; lane group kq owns pixel 4*t + kq
        int p = p0 + wave * WP + kq;
        int pc = p < HoWo ? p : HoWo - 1;
        int oh = pc / g.Wo, ow = pc - oh * g.Wo;
        for (int t = 0; t < (WP >> 2); ++t) {
            const int pl = wave * WP + 4 * t + kq;
            float av[TNB];
#pragma unroll
            for (int nb = 0; nb < TNB; ++nb)
                av[nb] = *reinterpret_cast<const float*>(dy_s + ((nb * TMW + pl) * 16 + j) * 4);
            const int base = (((oh - oh_first) * g.s) * g.PWT + ow) * 64 + j * 4;
#pragma unroll
            for (int ta = 0; ta < TA; ++ta) {
                const int roff = g.off_h + (a0 + ta) * g.step_h - g.r0;
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) {
                    const int coff = g.off_w + tb * g.step_w - g.c0;
                    const int slotoff = g.s == 1 ? coff : (coff & 1) * g.PWs + (coff >> 1);
                    const int toff = (roff * g.PWT + slotoff) * 64;
#pragma unroll
                    for (int kb = 0; kb < TKB; ++kb) {
                        const float bv = *reinterpret_cast<const float*>(smem + kb * chunk_bytes + base + toff);
#pragma unroll
                        for (int nb = 0; nb < TNB; ++nb) {
                            const int idx = ((nb * TKB + kb) * TA + ta) * TB + tb;
                            acc[idx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[nb], bv, acc[idx], 0, 0, 0);
                        }
                    }
                }
            }
            // advance this lane's pixel by 4 (clamped pixels past the image end carry dY == 0)
            p += 4;
            if (p < HoWo) {
                ow += 4;
                while (ow >= g.Wo) { ow -= g.Wo; ++oh; }
            }
        }
    }

    // combine the 4 waves through LDS (sequential adds: fixed order), then one slab per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NBLK; ++i) {
                float4* dst = reinterpret_cast<float4*>(red + (i * 64 + lane) * 4);
                if (w == 0) {
                    *dst = make_float4(acc[i].x, acc[i].y, acc[i].z, acc[i].w);
                } else {
                    float4 o = *dst;
                    o.x += acc[i].x; o.y += acc[i].y; o.z += acc[i].z; o.w += acc[i].w;
                    *dst = o;
                }
            }
        }
        __syncthreads();
    }
    float4* out = reinterpret_cast<float4*>(partial) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NBLK * 64;
    for (int e = tid; e < NBLK * 64; e += 256) out[e] = reinterpret_cast<const float4*>(red)[e];
}

// dW[n][k][kh][kw] = sum over splits of the slabs.  Block = 32 elements x 8 split lanes: every thread adds a
// strided subset of the splits, the 8 subtotals are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradGeo g, int TNB, int TKB, int TA, int TB, int n_real,
                                                           int k_real, const float* __restrict__ partial,
                                                           float* __restrict__ dw, int nx) {
    __shared__ float sub[8][32];
    const int nblk_tile = TNB * TKB * TA * TB;
    const int64_t per_split = (int64_t)nx * nblk_tile * 256;
    const int el = threadIdx.x & 31, ys = threadIdx.x >> 5;
    for (int64_t e0 = (int64_t)blockIdx.x * 32; e0 < per_split; e0 += (int64_t)gridDim.x * 32) {
        const int64_t e = e0 + el;
        float s = 0.f;
        if (e < per_split)
            for (int y = ys; y < g.nsplit; y += 8) s += partial[y * per_split + e];
        sub[ys][el] = s;
        __syncthreads();
        if (ys == 0 && e < per_split) {
            s = ((sub[0][el] + sub[1][el]) + (sub[2][el] + sub[3][el])) + ((sub[4][el] + sub[5][el]) + (sub[6][el] + sub[7][el]));
            const int r = e & 3, lane = (e >> 2) & 63;
            int64_t q = e >> 8;
            const int blk = q % nblk_tile;
            int xt = q / nblk_tile;
            const int tb = blk % TB, ta = (blk / TB) % TA, kb = (blk / (TB * TA)) % TKB, nb = blk / (TB * TA * TKB);
            const int tg = xt % g.tap_groups; xt /= g.tap_groups;
            const int tk = xt % g.tiles_k, tn = xt / g.tiles_k;
            const int n = (tn * TNB + nb) * 16 + (lane >> 4) * 4 + r;
            const int k = (tk * TKB + kb) * 16 + (lane & 15);
            const int kh = tg * TA + ta;
            if (n < n_real && k < k_real && kh < g.KH) dw[(((size_t)n * k_real + k) * g.KH + kh) * g.KW + tb] = s;
        }
        __syncthreads();
    }
}

template <int TNB, int TKB, int TA, int TB>
int launch_wgrad(WgradGeo& g, const float* X, const float* dY, const float* sc, const float* sh, float* partial,
                 int64_t ws_floats, float* dw, int n_real, int k_real, hipStream_t st) {
    constexpr int NBLK = TNB * TKB * TA * TB;
    g.tiles_n = (int)cdiv(g.nblk, TNB);
    g.tiles_k = (int)cdiv(g.nchunks, TKB);
    g.tap_groups = (int)cdiv(g.KH, TA);
    const int nx = g.tiles_n * g.tiles_k * g.tap_groups;
    int nsplit = (int)cdiv(512, nx);
    if (nsplit > g.total_tiles) nsplit = g.total_tiles;
    if (nsplit < 1) nsplit = 1;
    while (nsplit > 1 && (int64_t)nsplit * nx * NBLK * 256 > ws_floats) --nsplit;
    if ((int64_t)nsplit * nx * NBLK * 256 > ws_floats) return DAM_ERR_WORKSPACE;
    g.nsplit = nsplit;
    size_t lds = (size_t)TKB * g.PR * g.PWT * 64 + (size_t)TNB * g.TMW * 64;
    if (lds < (size_t)NBLK * 1024) lds = (size_t)NBLK * 1024;
    if (lds > 160 * 1024) return DAM_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<TNB, TKB, TA, TB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return DAM_ERR_LAUNCH;
            raised = true;
        }
    }
    hipLaunchKernelGGL((wgrad_kernel<TNB, TKB, TA, TB>), dim3(nx, nsplit), dim3(256), lds, st, g, X, dY, sc, sh, partial);
    DAM_CHECK_LAUNCH();
    const int64_t per_split = (int64_t)nx * NBLK * 256;
    const int rb = (int)(cdiv(per_split, 32) < 2048 ? cdiv(per_split, 32) : 2048);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb), dim3(256), 0, st, g, TNB, TKB, TA, TB, n_real, k_real, partial, dw, nx);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

}  // namespace
}  // namespace dam

extern "C" int64_t dam_conv2d_wgrad_workspace_floats(int n_out, int c_in, int kh, int kw) {
    if (n_out <= 0 || c_in <= 0 || kh <= 0 || kw <= 0) return 0;
    // enough for ~512 workgroup slabs of the largest tile, and at least 4 splits of the whole gradient
    const int64_t whole = dam::cdiv(n_out, 32) * dam::cdiv(c_in, 32) * 4 * 256 * (int64_t)kh * kw;
    const int64_t a = 512LL * 36 * 256 + 4 * whole;
    return a;
}

extern "C" int dam_conv2d_wgrad_f32(const float* x, int B, int H, int W, int C, int in_nchw, const float* in_scale,
                                    const float* in_shift, int relu_in, const float* dy, int Ho, int Wo, int n_chan,
                                    int n_out, int kh, int kw, int stride, int pad, int dil, float* dw,
                                    float* workspace, int64_t workspace_floats, void* stream) {
    using namespace dam;
    if (!x || !dy || !dw || !workspace || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return DAM_ERR_BAD_ARG;
    if (n_chan % 16 || n_out > n_chan || (stride != 1 && stride != 2)) return DAM_ERR_UNSUPPORTED;
    if (in_nchw ? C > 16 : C % 16) return DAM_ERR_UNSUPPORTED;
    if (in_scale && !in_shift) return DAM_ERR_BAD_ARG;
    WgradGeo g;
    g.B = B; g.H = H; g.W = W; g.C = C; g.Ho = Ho; g.Wo = Wo; g.N = n_chan; g.s = stride; g.KH = kh; g.KW = kw;
    g.off_h = -pad; g.step_h = dil; g.off_w = -pad; g.step_w = dil;
    g.r0 = -pad; g.c0 = -pad;
    g.PWin = (Wo - 1) * stride + (kw - 1) * dil + 1;
    g.PWs = (int)cdiv(g.PWin, stride);
    g.PWT = g.PWs * stride;
    auto set_tile = [&](int tmw) {
        int rows_out = (tmw + Wo - 2) / Wo + 1;
        if (rows_out > Ho) rows_out = Ho;
        g.TMW = tmw;
        g.PR = (rows_out - 1) * stride + (kh - 1) * dil + 1;
        g.tiles_m = (int)cdiv((int64_t)Ho * Wo, tmw);
        g.total_tiles = g.tiles_m * B;
    };
    auto lds_bytes = [&](int tnb, int tkb) { return (size_t)tkb * g.PR * g.PWT * 64 + (size_t)tnb * g.TMW * 64; };
    g.nchunks = in_nchw ? 1 : C / 16;
    g.nblk = n_chan / 16;
    g.in_nchw = in_nchw; g.relu_in = relu_in;
    const int k_real = C;
    hipStream_t st = (hipStream_t)stream;
    // tile choice: 2x2 channel blocks when both sides have them and LDS allows two workgroups per CU
    bool small = g.nchunks == 1 || g.nblk == 1;
    set_tile(256);
    if (!small && lds_bytes(2, 2) > 72 * 1024) {
        set_tile(128);
        if (lds_bytes(2, 2) > 72 * 1024) small = true;
    }
    if (small) {
        set_tile(256);
        if (lds_bytes(1, 1) > 72 * 1024) set_tile(128);
    }
#define DAM_WG(TN_, TK_, TA_, TB_) \
    return launch_wgrad<TN_, TK_, TA_, TB_>(g, x, dy, in_scale, in_shift, workspace, workspace_floats, dw, n_out, k_real, st)
    if (kw == 3 && kh == 3) { if (small) DAM_WG(1, 1, 3, 3); else DAM_WG(2, 2, 3, 3); }
    if (kw == 1 && kh == 1) { if (small) DAM_WG(1, 1, 1, 1); else DAM_WG(2, 2, 1, 1); }
    if (kw == 5) { if (small) DAM_WG(1, 1, 1, 5); else DAM_WG(2, 2, 1, 5); }
    if (kw == 7) { if (small) DAM_WG(1, 1, 1, 7); else DAM_WG(2, 2, 1, 7); }
    if (kw == 9) { if (small) DAM_WG(1, 1, 1, 9); else DAM_WG(2, 2, 1, 9); }
#undef DAM_WG
    return DAM_ERR_UNSUPPORTED;
}
