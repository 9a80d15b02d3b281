// dam_conv.hip -- 2-D convolution forward / data-gradient as an implicit GEMM on the gfx950
// fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fma chains, same numerics as a CPU conv).
//
// Replaces the nn.Conv2d calls of models/model_resnet.py:11-21,64 (3x3 pad 1 stride 1/2, 1x1 stride 2,
// no bias) and models/model_scalar_1s.py:167-172 / model_scalar_2s.py:25-30 (valid 3/5/7/9 kernels,
// bias, stride 2 / dilation 2 first block), forward and dgrad.
//
// Design (MI355X first, not a cuDNN-shaped port):
//   * activations are NHWC fp32 with C a multiple of 16; GEMM view  D[cout][pixel] = W[cout][k] * X[k][pixel],
//     so one lane ends up with 4 consecutive output channels of one pixel -> one 16-byte store, and a
//     16x16 output block is one contiguous 1 KB wave store when Cout == 16;
//   * the MFMA takes ONE float per operand per lane and is free in the K order, so K is ordered
//     "channel 4*(lane>>4)+s at step s": a single ds_read_b128 of a lane's 4 consecutive channels feeds four
//     MFMA steps, and the 64 lanes of a wave read 16 pixels x 64 B = 1 KB contiguous (conflict-free);
//   * a workgroup (4 waves) owns TM = 64*MB consecutive output pixels (flattened over rows, so odd widths
//     such as 130 waste nothing) x 16*NB output channels; the input rows those pixels touch are staged once
//     per channel group into LDS as [chunk][row][col slot][16 ch] (stride-2 convs de-interleave even/odd columns
//     so that consecutive output pixels stay contiguous), optionally applying the producer's
//     BatchNorm scale/shift + ReLU on the way in (fused BN-apply);
//   * weights are pre-packed [tap][k chunk][n block][lane] float4 and go global(L2) -> registers, 1 KB per wave load;
//   * a "tap grid" (nA x nB taps with signed input steps, weight-tap strides and an output stride/offset)
//     expresses forward, stride-1 dgrad (negative steps) and the four parity classes of a stride-2 dgrad
//     with the same kernel.
#include <cstdlib>
#include "dam_common.h"
#include "dam_conv_geo.h"
#include "dam_conv_stage.h"
#include "dam_bn_fin.h"

namespace dam {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// Up to four launches that differ only in their geometry (the parity classes of a strided data gradient: same tensors, same
// weights, same tile) run as ONE launch: blockIdx.z = class * B + image; a workgroup outside its class's grid leaves.
constexpr int GEO_PACK_MAX = 4;
struct ConvGeoPack {
    ConvGeo g[GEO_PACK_MAX];
    int n, B;
};

template <int MB, int NB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvGeoPack pk, const float* __restrict__ X,
                                                         const float4* __restrict__ Wp, const float* __restrict__ bias,
                                                         const float* __restrict__ in_scale,
                                                         const float* __restrict__ in_shift, float* __restrict__ Y,
                                                         const float* __restrict__ res, const float* __restrict__ res_mask,
                                                         float* __restrict__ splitk_ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    constexpr int MW = 16 * MB, TM = 4 * MW;
    const int cls = pk.n > 1 ? (int)blockIdx.z / pk.B : 0;
    const ConvGeo& g = pk.g[cls];
    const int HoWo = g.Ho * g.Wo;
    if (pk.n > 1 && ((int)blockIdx.x * TM >= HoWo || (int)blockIdx.y >= ((g.N / 16 + NB - 1) / NB) * g.ksplit)) return;
    const int ks = blockIdx.y % g.ksplit;                       // split-K slice of this workgroup
    const int img = (int)blockIdx.z - cls * pk.B, nb0 = (blockIdx.y / g.ksplit) * NB;
    const int p0 = blockIdx.x * TM;
    const float inv_wo = 1.0f / (float)g.Wo;
    const int oh_first = fast_div(p0, g.Wo, inv_wo);
    const int chunk_bytes = g.PR * g.PWT * 64;

    // per-lane LDS base of each M block (pixel of column j)
    int base_b[MB];
    int pix[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        int p = p0 + wave * MW + mb * 16 + j;
        pix[mb] = p;
        p = p < HoWo ? p : HoWo - 1;
        const int oh = fast_div(p, g.Wo, inv_wo), ow = p - oh * g.Wo;
        base_b[mb] = (((oh - oh_first) * g.s) * g.PWT + ow) * 64 + kq * 16;
    }

    v4f acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (v4f){0.f, 0.f, 0.f, 0.f};

    const int ih0 = oh_first * g.s + g.r0;
    PatchGeo pg;
    pg.H = g.H; pg.W = g.W; pg.C = g.C; pg.s = g.s; pg.c0 = g.c0; pg.PR = g.PR; pg.PWin = g.PWin; pg.PWs = g.PWs;
    pg.PWT = g.PWT; pg.in_nchw = g.in_nchw; pg.relu_in = g.relu_in;
    const int lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float4*>(Wp), 0, 0x7fffffff, 0x00020000);      // offsets are validated on the host (tap grid inside the packing)
    const int ngroups = g.nchunks / g.CG;
    const int cg_lo = ks * g.gps, cg_hi = cg_lo + g.gps < ngroups ? cg_lo + g.gps : ngroups;
    for (int cg = cg_lo; cg < cg_hi; ++cg) {
        if (cg != cg_lo) __syncthreads();
        // ---- stage the input patch of this channel group ----
        {
            const size_t img_elems = (size_t)g.H * g.W * g.C;
            stage_patch(smem, chunk_bytes, X + (size_t)img * img_elems, pg, ih0, cg * g.CG, g.CG, in_scale, in_shift, tid);
        }
        __syncthreads();

        // ---- MFMA over (tap a, tap b, chunk) items of the group.  Three operand sets rotate: the weights (L2, ~1-2 us
        //      under load, longer than the ~0.5 us of MFMAs of one item) and the LDS operands of item i+2 are requested
        //      before the MFMAs of item i issue.  Loads are unconditional (the load pointer stops at the last item) so
        //      that the compiler can count vmcnt; weights come through a buffer resource: scalar item offset + lane*16.
        {
            const int n_items = g.nA * g.nB * g.CG;
            float4 wa[3][NB], xv[3][MB];
            int la = 0, lb = 0, lc = 0;                      // item the next ISSUE loads (scalar)
#define DAM_TILE_ISSUE(S_)                                                                                                 \
    do {                                                                                                                   \
        const int roff_ = g.off_h + la * g.step_h - g.r0;                                                                  \
        const int coff_ = g.off_w + lb * g.step_w - g.c0;                                                                  \
        const int slotoff_ = g.s == 1 ? coff_ : (coff_ & 1) * g.PWs + (coff_ >> 1);                                        \
        const int lo_ = lc * chunk_bytes + (roff_ * g.PWT + slotoff_) * 64;                                                \
        const int tap_ = g.wt_base + la * g.wt_sa + lb * g.wt_sb;                                                          \
        const int ws_ = ((tap_ * g.nchunks + cg * g.CG + lc) * g.NBtot + nb0) * 1024;                                      \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                                  \
            wa[S_][nb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + nb * 1024, ws_, 0)); \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                  \
            xv[S_][mb] = *reinterpret_cast<const float4*>(smem + base_b[mb] + lo_);                                        \
        if (!(la == g.nA - 1 && lb == g.nB - 1 && lc == g.CG - 1)) {                                                       \
            if (++lc == g.CG) { lc = 0; if (++lb == g.nB) { lb = 0; ++la; } }                                              \
        }                                                                                                                  \
    } while (0)
#define DAM_TILE_MFMA(S_)                                                                                                  \
    do {                                                                                                                   \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                  \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                            \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[S_][nb].x, xv[S_][mb].x, acc[mb][nb], 0, 0, 0);      \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[S_][nb].y, xv[S_][mb].y, acc[mb][nb], 0, 0, 0);      \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[S_][nb].z, xv[S_][mb].z, acc[mb][nb], 0, 0, 0);      \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[S_][nb].w, xv[S_][mb].w, acc[mb][nb], 0, 0, 0);      \
            }                                                                                                              \
    } while (0)
            DAM_TILE_ISSUE(0);
            DAM_TILE_ISSUE(1);
            for (int it = 0; it < n_items; it += 3) {
                DAM_TILE_ISSUE(2);
                DAM_TILE_MFMA(0);
                DAM_TILE_ISSUE(0);
                if (it + 1 < n_items) DAM_TILE_MFMA(1);
                DAM_TILE_ISSUE(1);
                if (it + 2 < n_items) DAM_TILE_MFMA(2);
            }
#undef DAM_TILE_ISSUE
#undef DAM_TILE_MFMA
        }
    }

    // ---- epilogue: lane holds channels 4*kq..+3 of pixel j of every (mb, nb) block ----
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int p = pix[mb];
        if (p >= HoWo) continue;
        const int oh = fast_div(p, g.Wo, inv_wo), ow = p - oh * g.Wo;
        const size_t opix = ((size_t)img * g.OHt + (oh * g.os + g.oo_h)) * g.OWt + (ow * g.os + g.oo_w);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int ch = (nb0 + nb) * 16 + kq * 4;
            if (ch >= g.N) continue;
            v4f v = acc[mb][nb];
            const size_t o = opix * g.N + ch;
            if (g.ksplit > 1) {      // raw partial sums; bias / residual are applied by splitk_reduce_kernel
                *reinterpret_cast<float4*>(splitk_ws + (size_t)ks * ((size_t)g.B * g.OHt * g.OWt * g.N) + o) =
                    make_float4(v.x, v.y, v.z, v.w);
                continue;
            }
            if (bias) {
                const float4 bv = *reinterpret_cast<const float4*>(bias + ch);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if (res) {   // dgrad of a residual block: + dOut * (out > 0)
                const float4 rv = *reinterpret_cast<const float4*>(res + o);
                if (res_mask) {
                    const float4 mv = *reinterpret_cast<const float4*>(res_mask + o);
                    v.x += mv.x > 0.f ? rv.x : 0.f; v.y += mv.y > 0.f ? rv.y : 0.f;
                    v.z += mv.z > 0.f ? rv.z : 0.f; v.w += mv.w > 0.f ? rv.w : 0.f;
                } else {
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
            }
            if (g.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(Y + o) = make_float4(v.x, v.y, v.z, v.w);
        }
    }
}

// OIHW [O][I][KH][KW] -> packed [tap][kchunk][nblk][kq][i][s].
//   forward (transpose == 0): n = O index, k = I index;  dgrad (transpose == 1): n = I index, k = O index.
__global__ void pack_weights_kernel(const float* __restrict__ w, int O, int I, int KH, int KW, int transpose,
                                    int nchunks, int nblks, float* __restrict__ out, int64_t total) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int s = e & 3, i = (e >> 2) & 15, kq = (e >> 6) & 3;
        int64_t r = e >> 8;
        const int nblk = r % nblks; r /= nblks;
        const int chunk = r % nchunks;
        const int tap = r / nchunks;
        const int n = nblk * 16 + i, k = chunk * 16 + kq * 4 + s;
        const int o = transpose ? k : n, ii = transpose ? n : k;
        float v = 0.f;
        if (o < O && ii < I) v = w[((size_t)o * I + ii) * KH * KW + tap];
        out[e] = v;
    }
}

// All convolution weights of a model in ONE launch: desc[t] = {src, dst, O, I, KH, KW, transpose, total floats}.
// A thread owns one (n, k) position and walks its KH*KW taps: the taps of a weight are contiguous in OIHW (36 bytes for 3x3), so
// its loads hit one or two cache lines and a wave's loads of one tap are neighbours of its loads of the next -- the first version
// gave every (tap, n, k) its own thread, so each 128-byte line of a weight tensor was fetched by nine workgroups at nine different
// times (19.6 us per ResNet18 step for 25 MB of packed images); the stores of one tap stay coalesced (consecutive positions).
__global__ void pack_weights_multi_kernel(const long long* __restrict__ desc) {
    const long long* d = desc + (size_t)blockIdx.y * 8;
    const float* w = reinterpret_cast<const float*>(d[0]);
    float* out = reinterpret_cast<float*>(d[1]);
    const int O = (int)d[2], I = (int)d[3], KH = (int)d[4], KW = (int)d[5], transpose = (int)d[6];
    const int n = transpose ? I : O, k = transpose ? O : I;
    const int nchunks = (k + 15) / 16, nblks = (n + 15) / 16, taps = KH * KW;
    const int64_t plane = (int64_t)nchunks * nblks * 256;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < plane; e += (int64_t)gridDim.x * blockDim.x) {
        const int s = e & 3, i = (e >> 2) & 15, kq = (e >> 6) & 3;
        const int64_t r = e >> 8;
        const int nblk = (int)(r % nblks), chunk = (int)(r / nblks);
        const int nn = nblk * 16 + i, kk = chunk * 16 + kq * 4 + s;
        const int o = transpose ? kk : nn, ii = transpose ? nn : kk;
        const bool in = o < O && ii < I;
        const float* src = w + ((size_t)(in ? o : 0) * I + (in ? ii : 0)) * taps;
        int tap = 0;
        for (; tap + 9 <= taps; tap += 9) {             // nine loads in flight (a whole 3x3 kernel)
            float v[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) v[u] = src[tap + u];
#pragma unroll
            for (int u = 0; u < 9; ++u) out[(int64_t)(tap + u) * plane + e] = in ? v[u] : 0.f;
        }
        for (; tap < taps; ++tap) out[(int64_t)tap * plane + e] = in ? src[tap] : 0.f;
    }
}

// y = sum_ks slab[ks] + bias (+ res [* (mask > 0)]) for the split-K launches of the tile kernel (fixed order: deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int ksplit, int64_t n4, int Q, const float* __restrict__ bias,
                                     const float* __restrict__ res, const float* __restrict__ res_mask, int relu_out,
                                     float* __restrict__ y) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(ws)[e];
        for (int k = 1; k < ksplit; ++k) {
            const float4 b = reinterpret_cast<const float4*>(ws)[(int64_t)k * n4 + e];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (bias) {
            const float4 b = reinterpret_cast<const float4*>(bias)[e % Q];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (res) {
            const float4 r = reinterpret_cast<const float4*>(res)[e];
            if (res_mask) {
                const float4 m = reinterpret_cast<const float4*>(res_mask)[e];
                a.x += m.x > 0.f ? r.x : 0.f; a.y += m.y > 0.f ? r.y : 0.f; a.z += m.z > 0.f ? r.z : 0.f; a.w += m.w > 0.f ? r.w : 0.f;
            } else {
                a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
            }
        }
        if (relu_out) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
        reinterpret_cast<float4*>(y)[e] = a;
    }
}

template <int MB, int NB>
int launch_conv_pack(const ConvGeoPack& pk, size_t lds, const float* X, const float* Wp, const float* bias, const float* sc,
                     const float* sh, float* Y, const float* res, const float* res_mask, float* splitk_ws, hipStream_t st) {
    constexpr int TM = 64 * MB;
    if (lds > 64 * 1024) {
        static PerDevice<bool> raised_pd; bool& raised = raised_pd();
        if (!raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<MB, NB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return DAM_ERR_LAUNCH;
            raised = true;
        }
    }
    int64_t gx = 0, gy = 0;
    for (int i = 0; i < pk.n; ++i) {
        const ConvGeo& g = pk.g[i];
        const int64_t tx = cdiv((int64_t)g.Ho * g.Wo, TM), ty = cdiv(g.N / 16, NB) * g.ksplit;
        gx = tx > gx ? tx : gx; gy = ty > gy ? ty : gy;
    }
    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)(pk.B * pk.n));
    hipLaunchKernelGGL((conv_igemm_kernel<MB, NB>), grid, dim3(256), lds, st, pk, X, reinterpret_cast<const float4*>(Wp),
                       bias, sc, sh, Y, res, res_mask, splitk_ws);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

template <int MB, int NB>
int launch_conv(const ConvGeo& g, size_t lds, const float* X, const float* Wp, const float* bias, const float* sc,
                const float* sh, float* Y, const float* res, const float* res_mask, float* splitk_ws, hipStream_t st) {
    ConvGeoPack pk;
    pk.g[0] = g; pk.n = 1; pk.B = g.B;
    const int rc = launch_conv_pack<MB, NB>(pk, lds, X, Wp, bias, sc, sh, Y, res, res_mask, splitk_ws, st);
    if (rc != DAM_OK) return rc;
    if (g.ksplit > 1) {
        const int64_t n4 = (int64_t)g.B * g.OHt * g.OWt * g.N / 4;
        const int blocks = (int)(cdiv(n4, 256) < 2048 ? cdiv(n4, 256) : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, splitk_ws, g.ksplit, n4, g.N / 4, bias, res, res_mask, g.relu_out, Y);
        DAM_CHECK_LAUNCH();
    }
    return DAM_OK;
}

// Caller-owned batch of tile-kernel launches (include/dam_hip.h: dam_conv_batch_*): dam_conv2d_tapgrid_f32 records a launch that
// would take conv_igemm_kernel without split-K instead of making it; the flush packs consecutive records that differ only in
// geometry into one launch.
struct ConvLaunchRec {
    ConvGeo g;
    int MB, NB;
    size_t lds;
    const float *X, *Wp, *bias, *sc, *sh, *res, *res_mask;
    float *Y, *ws;
};
constexpr int CONV_BATCH_MAX = 8;
constexpr unsigned CONV_BATCH_MAGIC = 0x43424154u;      // "CBAT"
struct ConvBatch {
    unsigned magic;
    int n;
    ConvLaunchRec rec[CONV_BATCH_MAX];
};

int conv_pack_dispatch(const ConvGeoPack& pk, const ConvLaunchRec& r, hipStream_t st) {
#define DAM_CONV_PCASE(M_, N_) \
    if (r.MB == M_ && r.NB == N_) return launch_conv_pack<M_, N_>(pk, r.lds, r.X, r.Wp, r.bias, r.sc, r.sh, r.Y, r.res, r.res_mask, r.ws, st)
    DAM_CONV_PCASE(4, 4); DAM_CONV_PCASE(4, 2); DAM_CONV_PCASE(4, 1);
    DAM_CONV_PCASE(4, 3); DAM_CONV_PCASE(2, 3); DAM_CONV_PCASE(1, 3);
    DAM_CONV_PCASE(2, 4); DAM_CONV_PCASE(2, 2); DAM_CONV_PCASE(2, 1);
    DAM_CONV_PCASE(1, 4); DAM_CONV_PCASE(1, 2); DAM_CONV_PCASE(1, 1);
#undef DAM_CONV_PCASE
    return DAM_ERR_UNSUPPORTED;
}

int conv_batch_flush(ConvBatch& b, hipStream_t st) {
    int i = 0, rc = DAM_OK;
    while (i < b.n && rc == DAM_OK) {
        const ConvLaunchRec& r = b.rec[i];
        ConvGeoPack pk;
        pk.B = r.g.B; pk.n = 0;
        size_t lds = 0;
        int k = i;
        while (k < b.n && pk.n < GEO_PACK_MAX) {
            const ConvLaunchRec& q = b.rec[k];
            if (q.MB != r.MB || q.NB != r.NB || q.X != r.X || q.Wp != r.Wp || q.bias != r.bias || q.sc != r.sc || q.sh != r.sh ||
                q.Y != r.Y || q.res != r.res || q.res_mask != r.res_mask || q.g.B != r.g.B)
                break;
            pk.g[pk.n++] = q.g;
            lds = q.lds > lds ? q.lds : lds;
            ++k;
        }
        ConvLaunchRec merged = r;
        merged.lds = lds;
        rc = conv_pack_dispatch(pk, merged, st);
        i = k;
    }
    b.n = 0;
    return rc;
}

}  // namespace
}  // namespace dam

extern "C" int64_t dam_conv_packed_weight_count(int n_out, int k_in, int kh, int kw) {
    if (n_out <= 0 || k_in <= 0 || kh <= 0 || kw <= 0) return 0;
    return (int64_t)kh * kw * dam::cdiv(k_in, 16) * dam::cdiv(n_out, 16) * 256;
}

extern "C" int dam_conv_pack_weights_f32(const float* w_oihw, int O, int I, int KH, int KW, int transpose,
                                         float* packed, void* stream) {
    using namespace dam;
    if (!w_oihw || !packed || O <= 0 || I <= 0 || KH <= 0 || KW <= 0) return DAM_ERR_BAD_ARG;
    const int n = transpose ? I : O, k = transpose ? O : I;
    const int nchunks = (int)cdiv(k, 16), nblks = (int)cdiv(n, 16);
    const int64_t total = (int64_t)KH * KW * nchunks * nblks * 256;
    const int blocks = (int)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, O, I, KH, KW,
                       transpose, nchunks, nblks, packed, total);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_conv_pack_weights_multi_f32(const int64_t* desc_dev, int n_tensors, int64_t max_total, void* stream) {
    using namespace dam;
    if (!desc_dev || n_tensors <= 0 || max_total <= 0) return DAM_ERR_BAD_ARG;
    if (n_tensors > 65535) return DAM_ERR_UNSUPPORTED;
    int64_t bx = cdiv(max_total, 256 * 9);          // (a thread packs every tap of its position)
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(pack_weights_multi_kernel, dim3((unsigned)bx, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(desc_dev));
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

namespace dam {
namespace {

// Single-tap convolutions (the 1x1 strided shortcut of a down-sampling block, forward and its data gradient): no operand is
// reused across taps, so nothing is staged -- a wave owns 16 consecutive output pixels of one output row and NB output
// blocks, the input pixels (16 x 64-byte segments per 16-channel chunk) and the packed weights come straight from HBM / L2
// through buffer loads, all chunks in flight at once.  Memory- and latency-bound: 51 MB for the largest instance.
template <int NB, int NCHMAX>
__global__ __launch_bounds__(256) void conv1x1_direct_kernel(const ConvGeo g, const float* __restrict__ X,
                                                             const float4* __restrict__ Wp, const float* __restrict__ bias,
                                                             float* __restrict__ Y, const float* __restrict__ res,
                                                             const float* __restrict__ res_mask, int segs, int total_units) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int unit = blockIdx.x * 4 + wave;                  // (image, output row, 16-pixel segment)
    if (unit >= total_units) return;
    const int nb0 = blockIdx.y * NB;
    const int seg = unit % segs, row = unit / segs, img = row / g.Ho, oh = row - img * g.Ho;
    const int ow = seg * 16 + j;
    const bool valid = ow < g.Wo;
    const int ih = oh * g.s + g.off_h, iw = ow * g.s + g.off_w;          // single tap: always inside the tensor for pad 0;
    const bool inb = valid && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W; // checked anyway
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(Wp), 0, 0x7fffffff, 0x00020000);
    const int xoff = inb ? (((img * g.H + ih) * g.W + iw) * g.C + kq * 4) * 4 : 0x7fffffff;
    const int tap = g.wt_base;
    float4 xv[NCHMAX], wa[NCHMAX][NB];
#pragma unroll
    for (int c = 0; c < NCHMAX; ++c) {
        const bool cok = c < g.nchunks;
        xv[c] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, cok ? xoff : 0x7fffffff, c * 64, 0));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            wa[c][nb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                wr, cok ? lane * 16 : 0x7fffffff, (((tap * g.nchunks + c) * g.NBtot + nb0 + nb) * 64) * 16, 0));
    }
    v4f acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCHMAX; ++c)            // chunks past nchunks loaded zeros (out-of-range offsets)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][nb].x, xv[c].x, acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][nb].y, xv[c].y, acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][nb].z, xv[c].z, acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][nb].w, xv[c].w, acc[nb], 0, 0, 0);
        }
    if (!valid) return;
    const size_t opix = ((size_t)img * g.OHt + (oh * g.os + g.oo_h)) * g.OWt + (ow * g.os + g.oo_w);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int ch = (nb0 + nb) * 16 + kq * 4;
        v4f v = acc[nb];
        if (bias) v += *reinterpret_cast<const v4f*>(bias + ch);
        const size_t o = opix * g.N + ch;
        if (res) {
            const float4 rv = *reinterpret_cast<const float4*>(res + o);
            if (res_mask) {
                const float4 mv = *reinterpret_cast<const float4*>(res_mask + o);
                v.x += mv.x > 0.f ? rv.x : 0.f; v.y += mv.y > 0.f ? rv.y : 0.f;
                v.z += mv.z > 0.f ? rv.z : 0.f; v.w += mv.w > 0.f ? rv.w : 0.f;
            } else {
                v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            }
        }
        if (g.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(Y + o) = make_float4(v.x, v.y, v.z, v.w);
    }
}

// Two single-tap operators into the SAME output pixels in one launch: the data gradient of a down-sampling block's input gets, at its
// (even, even) pixels, the centre tap of conv1's 3x3 / stride-2 transposed operator AND the whole 1x1 / stride-2 shortcut operator
// (models/model_resnet.py:17-21,26: x feeds conv1 and the shortcut convolution) -- two direct launches of 7-16 us each, the second
// re-reading what the first wrote, become one: y = Wp1[tap1]^T-chunks . x1 + Wp2[tap2]^T-chunks . x2.  Both inputs have the
// geometry [B][H][W][C]; rounds of <= 8 chunks (all loads of a round in flight, then its MFMAs).
template <int NB, int R, bool BOTH>      // R: chunk slots per operand and round; BOTH: nch <= R, the two operands' loads go out together
__global__ __launch_bounds__(256) void conv1x1_pair_kernel(const float* __restrict__ X1, const float4* __restrict__ Wp1, int tap1,
                                                           const float* __restrict__ X2, const float4* __restrict__ Wp2, int tap2,
                                                           int B, int H, int W, int C, int N, float* __restrict__ Y, int OHt, int OWt,
                                                           int os, int oo_h, int oo_w, int segs, int total_units) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int unit = blockIdx.x * 4 + wave;                  // (image, row, 16-pixel segment)
    if (unit >= total_units) return;
    const int nb0 = blockIdx.y * NB, nch = C / 16, NBtot = N / 16;
    const int seg = unit % segs, row = unit / segs, img = row / H, oh = row - img * H;
    const int ow = seg * 16 + j;
    const bool valid = ow < W;
    const int xoff = valid ? (((img * H + oh) * W + ow) * C + kq * 4) * 4 : 0x7fffffff;
    v4f acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = (v4f){0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t xr1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X1), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(Wp1), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X2), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(Wp2), 0, 0x7fffffff, 0x00020000);
#define DAM_PAIR_LOAD(XR_, WR_, TAP_, C0_, XV_, WA_)                                                                          \
    _Pragma("unroll") for (int c = 0; c < R; ++c) {                                                                           \
        const bool cok = (C0_) + c < nch;                                                                                     \
        XV_[c] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(XR_, cok ? xoff : 0x7fffffff, ((C0_) + c) * 64, 0)); \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                                     \
            WA_[c][nb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(                                    \
                WR_, cok ? lane * 16 : 0x7fffffff, ((((TAP_) * nch + (C0_) + c) * NBtot + nb0 + nb) * 64) * 16, 0));          \
    }
#define DAM_PAIR_MFMA(XV_, WA_)                       /* chunks past the end loaded zeros (out-of-range offsets) */             \
    _Pragma("unroll") for (int c = 0; c < R; ++c)                                                                             \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                                   \
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(WA_[c][nb].x, XV_[c].x, acc[nb], 0, 0, 0);                         \
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(WA_[c][nb].y, XV_[c].y, acc[nb], 0, 0, 0);                         \
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(WA_[c][nb].z, XV_[c].z, acc[nb], 0, 0, 0);                         \
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(WA_[c][nb].w, XV_[c].w, acc[nb], 0, 0, 0);                         \
        }
    if constexpr (BOTH) {
        float4 xa[R], wa[R][NB], xb[R], wb[R][NB];
        DAM_PAIR_LOAD(xr1, wr1, tap1, 0, xa, wa)
        DAM_PAIR_LOAD(xr2, wr2, tap2, 0, xb, wb)
        DAM_PAIR_MFMA(xa, wa)
        DAM_PAIR_MFMA(xb, wb)
    } else {
        // two register sets: the next round (of either operand) is requested before the current one's MFMAs
        float4 xa[R], wa[R][NB], xb[R], wb[R][NB];
        const int rounds = (nch + R - 1) / R, total = 2 * rounds;
        DAM_PAIR_LOAD(xr1, wr1, tap1, 0, xa, wa)
#pragma unroll 1
        for (int r = 0; r < total; r += 2) {
            {   // round r + 1 into set b
                const int q = r + 1, c0 = (q >= rounds ? q - rounds : q) * R;
                if (q >= rounds) { DAM_PAIR_LOAD(xr2, wr2, tap2, c0, xb, wb) } else { DAM_PAIR_LOAD(xr1, wr1, tap1, c0, xb, wb) }
            }
            DAM_PAIR_MFMA(xa, wa)
            if (r + 2 < total) {
                const int q = r + 2, c0 = (q >= rounds ? q - rounds : q) * R;
                if (q >= rounds) { DAM_PAIR_LOAD(xr2, wr2, tap2, c0, xa, wa) } else { DAM_PAIR_LOAD(xr1, wr1, tap1, c0, xa, wa) }
            }
            DAM_PAIR_MFMA(xb, wb)
        }
    }
#undef DAM_PAIR_LOAD
#undef DAM_PAIR_MFMA
    if (!valid) return;
    const size_t opix = ((size_t)img * OHt + (oh * os + oo_h)) * OWt + (ow * os + oo_w);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int ch = (nb0 + nb) * 16 + kq * 4;
        *reinterpret_cast<float4*>(Y + opix * N + ch) = make_float4(acc[nb].x, acc[nb].y, acc[nb].z, acc[nb].w);
    }
}

template <int NB, int NCHMAX>
int launch_conv1x1(const ConvGeo& g, const float* X, const float* Wp, const float* bias, float* Y, const float* res,
                   const float* res_mask, hipStream_t st) {
    const int segs = (int)cdiv(g.Wo, 16);
    const int64_t units = (int64_t)g.B * g.Ho * segs;
    if (units >= (1ll << 30)) return DAM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv1x1_direct_kernel<NB, NCHMAX>), dim3((unsigned)cdiv(units, 4), (unsigned)(g.N / 16 / NB)), dim3(256), 0, st,
                       g, X, reinterpret_cast<const float4*>(Wp), bias, Y, res, res_mask, segs, (int)units);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

}  // namespace
}  // namespace dam

extern "C" int dam_conv1x1_pair_f32(const float* x1, const float* w1_packed, int tap1, const float* x2, const float* w2_packed,
                                    int tap2, int B, int H, int W, int C, int n_out, float* y, int OHt, int OWt, int out_stride,
                                    int out_off_h, int out_off_w, void* stream) {
    using namespace dam;
    if (!x1 || !w1_packed || !x2 || !w2_packed || !y || B <= 0 || H <= 0 || W <= 0 || tap1 < 0 || tap2 < 0 || out_stride < 1)
        return DAM_ERR_BAD_ARG;
    if (C % 16 || n_out % 16 || C <= 0 || n_out <= 0) return DAM_ERR_UNSUPPORTED;
    if ((H - 1) * out_stride + out_off_h >= OHt || (W - 1) * out_stride + out_off_w >= OWt || out_off_h < 0 || out_off_w < 0)
        return DAM_ERR_BAD_ARG;
    if ((int64_t)B * H * W * C * 4 >= (1ll << 31)) return DAM_ERR_UNSUPPORTED;
    const int segs = (int)cdiv(W, 16);
    const int64_t units = (int64_t)B * H * segs;
    if (units >= (1ll << 30)) return DAM_ERR_UNSUPPORTED;
    const int nblk = n_out / 16, nch = C / 16;
    hipStream_t st = (hipStream_t)stream;
#define DAM_PAIR_GO(NB_, R_, BOTH_)                                                                                           \
    hipLaunchKernelGGL((conv1x1_pair_kernel<NB_, R_, BOTH_>), dim3((unsigned)cdiv(units, 4), (unsigned)(nblk / NB_)), dim3(256), 0, st, x1, \
                       reinterpret_cast<const float4*>(w1_packed), tap1, x2, reinterpret_cast<const float4*>(w2_packed), tap2, B, H, W, \
                       C, n_out, y, OHt, OWt, out_stride, out_off_h, out_off_w, segs, (int)units)
    if (nblk % 2 == 0) {
        if (nch <= 2) DAM_PAIR_GO(2, 2, true); else if (nch <= 4) DAM_PAIR_GO(2, 4, true); else DAM_PAIR_GO(2, 4, false);
    } else {
        if (nch <= 2) DAM_PAIR_GO(1, 2, true); else if (nch <= 4) DAM_PAIR_GO(1, 4, true); else DAM_PAIR_GO(1, 8, false);
    }
#undef DAM_PAIR_GO
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

// Generic tap-grid convolution (see dam_hip.h).  The host wrapper derives the patch geometry and picks the tile.
extern "C" int64_t dam_conv_batch_bytes(void) { return (int64_t)sizeof(dam::ConvBatch); }

extern "C" int dam_conv_batch_init(void* batch) {
    if (!batch) return DAM_ERR_BAD_ARG;
    dam::ConvBatch* b = static_cast<dam::ConvBatch*>(batch);
    b->magic = dam::CONV_BATCH_MAGIC;
    b->n = 0;
    return DAM_OK;
}

extern "C" int dam_conv_batch_flush(void* batch, void* stream) {
    dam::ConvBatch* b = static_cast<dam::ConvBatch*>(batch);
    if (!b || b->magic != dam::CONV_BATCH_MAGIC) return DAM_ERR_BAD_ARG;
    return dam::conv_batch_flush(*b, (hipStream_t)stream);
}

extern "C" int dam_conv2d_tapgrid_f32(const float* x, int B, int H, int W, int C, int in_nchw, const float* w_packed,
                                      int k_chunks, int n_out, const float* bias, const float* in_scale,
                                      const float* in_shift, int relu_in, int relu_out, float* y, int OHt, int OWt, int Ho, int Wo,
                                      int out_stride, int out_off_h, int out_off_w, int in_stride, int nA, int nB,
                                      int off_h, int step_h, int off_w, int step_w, int wt_base, int wt_sa, int wt_sb,
                                      const float* res, const float* res_mask, float* bn_partial, int* bn_parts_host,
                                      const dam_bn_fin* bn_fin, const dam_bn_bwd_sums* bn_bwd, float* workspace,
                                      int64_t workspace_floats, void* batch, void* stream) {
    using namespace dam;
    if (bn_parts_host) *bn_parts_host = 0;
    BnBwdEpi bwd{};
    if (bn_bwd && bn_bwd->x) {
        if (!bn_partial || !bn_parts_host || bn_fin || !bn_bwd->mean || !bn_bwd->invstd) return DAM_ERR_BAD_ARG;
        if (bn_bwd->mask_bits ? (bn_bwd->mask_scale || bn_bwd->mask_shift || !res) : (!bn_bwd->mask_scale || !bn_bwd->mask_shift))
            return DAM_ERR_BAD_ARG;
        if (res && (!bn_bwd->res_mask_bits || !res_mask)) return DAM_ERR_BAD_ARG;      // bytes for the kernels that can, floats for the rest
        bwd = BnBwdEpi{bn_bwd->x, bn_bwd->mean, bn_bwd->invstd, bn_bwd->mask_scale, bn_bwd->mask_shift, bn_bwd->res_mask_bits,
                       bn_bwd->mask_bits};
    }
    BnFinArgs fin{};
    if (bn_fin && bn_partial) {
        if (!bn_fin->gamma || !bn_fin->beta || !bn_fin->save_mean || !bn_fin->save_invstd || !bn_fin->scale || !bn_fin->shift ||
            !bn_fin->counter) return DAM_ERR_BAD_ARG;
        fin = BnFinArgs{bn_fin->gamma, bn_fin->beta, bn_fin->running_mean, bn_fin->running_var,
                        (long long*)bn_fin->num_batches_tracked, bn_fin->momentum, bn_fin->eps, bn_fin->save_mean,
                        bn_fin->save_invstd, bn_fin->scale, bn_fin->shift, bn_fin->counter};
    }
    if (!x || !w_packed || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || nA <= 0 || nB <= 0)
        return DAM_ERR_BAD_ARG;
    if (n_out % 16 || (in_stride != 1 && in_stride != 2) || out_stride < 1) return DAM_ERR_UNSUPPORTED;
    if (in_nchw ? (C > 16 || k_chunks != 1) : (C % 16 || k_chunks != C / 16)) return DAM_ERR_UNSUPPORTED;
    if (in_scale && !in_shift) return DAM_ERR_BAD_ARG;
    if (B > 65535) return DAM_ERR_UNSUPPORTED;
    ConvGeo g;
    g.B = B; g.H = H; g.W = W; g.C = C; g.Ho = Ho; g.Wo = Wo; g.N = n_out; g.OHt = OHt; g.OWt = OWt;
    g.os = out_stride; g.oo_h = out_off_h; g.oo_w = out_off_w; g.s = in_stride; g.nA = nA; g.nB = nB;
    g.off_h = off_h; g.step_h = step_h; g.off_w = off_w; g.step_w = step_w;
    g.wt_base = wt_base; g.wt_sa = wt_sa; g.wt_sb = wt_sb;
    g.in_nchw = in_nchw; g.relu_in = relu_in; g.relu_out = relu_out != 0; g.nchunks = k_chunks; g.NBtot = n_out / 16;
    g.epi_bwd = 0;
    const int h_lo = off_h + (step_h < 0 ? (nA - 1) * step_h : 0), h_hi = off_h + (step_h > 0 ? (nA - 1) * step_h : 0);
    const int w_lo = off_w + (step_w < 0 ? (nB - 1) * step_w : 0), w_hi = off_w + (step_w > 0 ? (nB - 1) * step_w : 0);
    g.r0 = h_lo; g.c0 = w_lo;
    g.PWin = (Wo - 1) * in_stride + (w_hi - w_lo) + 1;
    g.PWs = (int)cdiv(g.PWin, in_stride);
    g.PWT = g.PWs * in_stride;

    hipStream_t st = (hipStream_t)stream;
    g.PR = 0; g.CG = 1; g.tiles_m = 0; g.ksplit = 1; g.gps = 1 << 30;
    // single tap (1x1 shortcut convolutions, forward and data gradient): direct kernel, nothing staged
    if (nA == 1 && nB == 1 && !in_nchw && !in_scale && k_chunks <= 8 &&
        (int64_t)B * H * W * C * 4 < (1ll << 31) && (int64_t)B * OHt * OWt * n_out * 4 < (1ll << 62)) {
        const int nblk = n_out / 16;
        if (k_chunks <= 2) return nblk % 2 == 0 ? launch_conv1x1<2, 2>(g, x, w_packed, bias, y, res, res_mask, st)
                                                : launch_conv1x1<1, 2>(g, x, w_packed, bias, y, res, res_mask, st);
        if (k_chunks <= 4) return nblk % 2 == 0 ? launch_conv1x1<2, 4>(g, x, w_packed, bias, y, res, res_mask, st)
                                                : launch_conv1x1<1, 4>(g, x, w_packed, bias, y, res, res_mask, st);
        return nblk % 2 == 0 ? launch_conv1x1<2, 8>(g, x, w_packed, bias, y, res, res_mask, st)
                             : launch_conv1x1<1, 8>(g, x, w_packed, bias, y, res, res_mask, st);
    }
    // persistent strip variant (LDS row ring fed by loader waves, optional fused BatchNorm statistics) when the layer fits it
    if (!in_nchw) {
        int parts = 0;
        const int rc = conv_strip_try(g, h_lo, h_hi, x, w_packed, bias, y, res, res_mask, bn_partial, &parts,
                                      fin.counter ? &fin : nullptr, in_scale, in_shift, bwd, st);
        if (rc == DAM_OK) {
            if (bn_partial && bn_parts_host) *bn_parts_host = parts;
            return DAM_OK;
        }
        if (rc != DAM_ERR_UNSUPPORTED) return rc;
        if (bn_partial) {       // maybe only the statistics did not fit: retry without them
            const int rc2 = conv_strip_try(g, h_lo, h_hi, x, w_packed, bias, y, res, res_mask, nullptr, nullptr, nullptr, in_scale, in_shift,
                                           BnBwdEpi{}, st);
            if (rc2 == DAM_OK) return DAM_OK;
            if (rc2 != DAM_ERR_UNSUPPORTED) return rc2;
        }
    }
    // thick 3x3 layers: the persistent variant whose patches are staged chunk by chunk by loader waves beside the MFMAs
    // (DAM_NO_PIPE: diagnostic switch for the A/B in tools/pipe_ab.sh)
    if (!getenv("DAM_NO_PIPE")) {
        int parts = 0;
        const int rc = conv_pipe_try(g, h_hi - h_lo, x, w_packed, bias, in_scale, in_shift, y, res, res_mask, workspace,
                                     (fin.counter || (bwd.x && res)) ? nullptr : bn_partial, &parts, (bwd.x && res) ? BnBwdEpi{} : bwd, st);
        if (rc == DAM_OK && bn_partial && bn_parts_host) *bn_parts_host = parts;
        if (rc != DAM_ERR_UNSUPPORTED) return rc;
        // Too wide for its patch buffers (the strided 16 -> 32 convolution on 1025x130: 131 input columns x 5 rows): the same
        // kernel on 2-4 column ranges of the output, one launch each -- a range is just another output sub-grid of this entry
        // point's geometry (width, output column offset, input column offset).  Not with an epilogue that leaves records.
        if (nA == 3 && nB == 3 && !in_nchw && !bn_partial && !getenv("DAM_PIPE_NO_COLSPLIT")) {
            for (int ns = 2; ns <= 4; ++ns) {
                const int wd0 = (int)cdiv(Wo, ns);
                if (wd0 < 16) break;
                int rc2 = DAM_OK, done = 0;
                for (int ow0 = 0; ow0 < Wo && rc2 == DAM_OK; ow0 += wd0, ++done) {
                    ConvGeo g2 = g;
                    g2.Wo = Wo - ow0 < wd0 ? Wo - ow0 : wd0;
                    g2.oo_w = out_off_w + ow0 * out_stride;
                    g2.off_w = off_w + ow0 * in_stride;
                    g2.c0 = w_lo + ow0 * in_stride;
                    g2.PWin = (g2.Wo - 1) * in_stride + (w_hi - w_lo) + 1;
                    g2.PWs = (int)cdiv(g2.PWin, in_stride);
                    g2.PWT = g2.PWs * in_stride;
                    rc2 = conv_pipe_try(g2, h_hi - h_lo, x, w_packed, bias, in_scale, in_shift, y, res, res_mask, workspace, nullptr,
                                        nullptr, BnBwdEpi{}, st);
                }
                if (rc2 == DAM_OK) return DAM_OK;
                if (rc2 != DAM_ERR_UNSUPPORTED || done > 1) return rc2 == DAM_ERR_UNSUPPORTED ? DAM_ERR_LAUNCH : rc2;   // a later range cannot fail where the first fitted
            }
        }
    }
    // tile choice.  With a workspace: keep big tiles (weights are streamed per workgroup: FLOPs per weight byte grow with
    // the tile) and get the workgroup count from split-K over channel groups; otherwise shrink tiles to fill the chip.
    const int64_t npix = (int64_t)Ho * Wo;
    const int nblk = n_out / 16;
    // NB must divide the block count.  Three-block tiles for the 48-channel layers of the scalar models (models/model_scalar_2s.py:
    // 64-77: the 7x7 data gradient 64 -> 48 ran the one-block tile, every LDS operand feeding ONE MFMA group: 371 us = 0.50 of peak)
    static const bool no_nb3 = getenv("DAM_CONV_NO_NB3") != nullptr;      // A/B switch
    int MB = 4, NB = nblk % 4 == 0 ? 4 : (nblk % 3 == 0 && !no_nb3 ? 3 : (nblk % 2 == 0 ? 2 : 1));
    auto wgs = [&](int mb, int nb) { return cdiv(npix, 64 * mb) * cdiv(nblk, nb) * B; };
    const bool can_split = workspace && out_stride == 1 && Ho == OHt && Wo == OWt && out_off_h == 0 && out_off_w == 0 &&
                           k_chunks >= 4;
    bool split = can_split;
    if (split) {
        // enough workgroups without splitting: the tile with the smallest makespan.  Every CU works through
        // ceil(workgroups / 256) tiles (co-resident workgroups share the matrix pipe, so it is the count that matters, not
        // the residency), a tile costs its MFMAs (~ mb * NB) plus a fixed staging / weight-streaming part (~ 1 unit).
        int64_t best = -1;
        for (int mb = 4; mb >= 1; mb >>= 1) {
            const int64_t w = wgs(mb, NB);
            if (w < 400) continue;
            const int64_t cost = cdiv(w, 256) * (mb * NB + 1);
            if (best < 0 || cost < best) { best = cost; MB = mb; split = false; }
        }
    }
    if (split) {
        int64_t best_pad = -1;
        for (int mb = 4; mb >= 1; mb >>= 1) {           // least padded pixels, larger tile on (near) ties
            const int64_t pad = cdiv(npix, 64 * mb) * 64 * mb;
            if (best_pad < 0 || pad * 10 < best_pad * 9) { best_pad = pad; MB = mb; }
        }
    } else if (!can_split) {
        while (wgs(MB, NB) < 512 && (MB > 1 || NB > 1)) {
            if (MB > 1 && (MB >= NB || NB == 1)) MB >>= 1; else NB >>= 1;
        }
    }
    // LDS: shrink the channel group, then the M tile, until the patch fits
    const size_t LDS_MAX = 64 * 1024;
    int CG = 1;
    while (CG * 2 <= 4 && k_chunks % (CG * 2) == 0) CG *= 2;
    auto patch_rows = [&](int mb) {
        const int tm = 64 * mb;
        int rows_out = (int)((tm + Wo - 2) / Wo + 1);
        if (rows_out > Ho) rows_out = Ho;
        return (rows_out - 1) * in_stride + (h_hi - h_lo) + 1;
    };
    for (;;) {
        g.PR = patch_rows(MB);
        const size_t bytes = (size_t)CG * g.PR * g.PWT * 64;
        if (bytes <= LDS_MAX) break;
        if (CG > 1) CG >>= 1;
        else if (MB > 1) MB >>= 1;
        else if (bytes <= 150 * 1024) break;
        else return DAM_ERR_UNSUPPORTED;
    }
    g.CG = CG;
    g.tiles_m = (int)cdiv(npix, 64 * MB);
    if (split) {
        const int ngroups = k_chunks / CG;
        const int64_t w = wgs(MB, NB), out_elems = (int64_t)B * OHt * OWt * n_out;
        int ksplit = (int)cdiv(512, w);
        if (ksplit > 8) ksplit = 8;
        if (ksplit > ngroups) ksplit = ngroups;
        while (ksplit > 1 && ksplit * out_elems > workspace_floats) --ksplit;
        if (ksplit > 1) {
            g.gps = (int)cdiv(ngroups, ksplit);
            g.ksplit = (int)cdiv(ngroups, g.gps);
        }
    }
    const size_t lds = (size_t)CG * g.PR * g.PWT * 64;
    if (batch && g.ksplit == 1) {       // recorded, launched (packed with its siblings) by dam_conv_batch_flush
        ConvBatch* cb = static_cast<ConvBatch*>(batch);
        if (cb->magic != CONV_BATCH_MAGIC) return DAM_ERR_BAD_ARG;
        if (cb->n == CONV_BATCH_MAX) {
            const int rc = conv_batch_flush(*cb, st);
            if (rc != DAM_OK) return rc;
        }
        ConvLaunchRec& r = cb->rec[cb->n++];
        r.g = g; r.MB = MB; r.NB = NB; r.lds = lds; r.X = x; r.Wp = w_packed; r.bias = bias; r.sc = in_scale; r.sh = in_shift;
        r.res = res; r.res_mask = res_mask; r.Y = y; r.ws = workspace;
        return DAM_OK;
    }
#define DAM_CONV_CASE(M_, N_) \
    if (MB == M_ && NB == N_) return launch_conv<M_, N_>(g, lds, x, w_packed, bias, in_scale, in_shift, y, res, res_mask, workspace, st)
    DAM_CONV_CASE(4, 4); DAM_CONV_CASE(4, 2); DAM_CONV_CASE(4, 1);
    DAM_CONV_CASE(4, 3); DAM_CONV_CASE(2, 3); DAM_CONV_CASE(1, 3);
    DAM_CONV_CASE(2, 4); DAM_CONV_CASE(2, 2); DAM_CONV_CASE(2, 1);
    DAM_CONV_CASE(1, 4); DAM_CONV_CASE(1, 2); DAM_CONV_CASE(1, 1);
#undef DAM_CONV_CASE
    return DAM_ERR_UNSUPPORTED;
}
