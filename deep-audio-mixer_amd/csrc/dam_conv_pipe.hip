// dam_conv_pipe.hip -- the tile convolution (dam_conv.hip: same GEMM mapping, packed weights, tap grid, epilogue) with the
// input staging taken OFF the compute waves, for the thick 3x3 / stride-1 layers (ResNet layer3-6 forward and data gradient:
// 64-256 channels on 33- to 5-pixel-wide rows).
//
// What the PMC passes on conv_igemm_kernel showed (profiles/r02_pmc_thick_layers.csv, layer3): the matrix pipe is busy 55 % of
// the launch; a workgroup stages its whole patch (all channel chunks, ~50 KB) before its first MFMA, every first-round
// workgroup does so at the same time, and with that much LDS only three fit a CU, so 1064 workgroups need a second, 39 %-full
// round that stages with nothing to overlap.  Here:
//   * a workgroup = 4 compute waves + 2 LOADER waves.  The patch is staged one 16-channel chunk at a time into two alternating
//     LDS buffers by the loader waves (own vmcnt queues: a compute wave that waited for its weights would otherwise also wait for
//     every older patch load) while the compute waves run the nine taps of the previous chunk; one barrier per chunk;
//   * the loaders' item geometry (patch row / column / channel quad -> global offset, LDS offset, in-tensor?) is chunk
//     independent and computed once per workgroup; a chunk costs PIPE_U loads + PIPE_U LDS writes per loader thread;
//   * two chunk buffers are 2 x 13 KB for layer3 instead of 54 KB: five workgroups per CU, every workgroup resident in one round;
//   * the weight pipeline (L2 -> registers, two items ahead, three rotating operand sets) runs ACROSS chunk boundaries: only the
//     LDS operand reads restart behind the barrier.
#include <cstdlib>
#include "dam_common.h"
#include "dam_conv_geo.h"
#include "dam_conv_stage.h"

namespace dam {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int PIPE_U = 8;                       // float4 items per loader thread and half chunk
constexpr int PIPE_H = 2;                       // a chunk is fetched as PIPE_H batches of PIPE_U items per loader thread

// Raw barrier: the wave's LDS traffic is drained, its global loads are NOT (__syncthreads() would also wait for the patch /
// weight loads that were just put in flight on purpose).
#define DAM_PIPE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// timing experiments (tools/build_variant.sh): results are wrong with any of these defined
#ifdef DAM_PIPE_DIAG_W0
#define DAM_PIPE_WOFF(x) ((x) & 0)              // every weight load hits the same 4 KB
#else
#define DAM_PIPE_WOFF(x) (x)
#endif

// Diagnostic build only (-DDAM_PIPE_STAMPS, launches without split-K): the workspace receives s_memtime stamps of phase
// boundaries, [workgroup][role: wave 0 / first loader wave][32] of (tag << 56 | time); read by tools/pipe_stamps_probe.py.
#ifdef DAM_PIPE_STAMPS
#define DAM_PSTAMP(role, tag)                                                                                         \
    do {                                                                                                              \
        if (lane == 0 && (wave == 0 || wave == 4) && stamp_n < 32) {                                                  \
            const unsigned long long t_ = __builtin_readcyclecounter();                                               \
            stamp_p[(role) * 32 + stamp_n++] = ((unsigned long long)(tag) << 56) | (t_ & ((1ull << 56) - 1));         \
        }                                                                                                             \
    } while (0)
#else
#define DAM_PSTAMP(role, tag) do { } while (0)
#endif

// NL = loader waves (1: patches up to 1024 float4 items per chunk, 2: up to 2048)
template <int MB, int NB, int NL>
__global__ __launch_bounds__(256 + 64 * NL) void conv_pipe_kernel(const ConvGeo g, const float* __restrict__ X,
                                                                 const float4* __restrict__ Wp, const float* __restrict__ bias,
                                                                 const float* __restrict__ in_scale,
                                                                 const float* __restrict__ in_shift, float* __restrict__ Y,
                                                                 const float* __restrict__ res, const float* __restrict__ res_mask,
                                                                 float* __restrict__ splitk_ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PIPE_LTHREADS = 64 * NL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    constexpr int MW = 16 * MB, TM = 4 * MW;
    const int ks = blockIdx.y % g.ksplit;
    const int img = blockIdx.z, nb0 = (blockIdx.y / g.ksplit) * NB;
    const int HoWo = g.Ho * g.Wo;
    const int p0 = blockIdx.x * TM;
    const float inv_wo = 1.0f / (float)g.Wo;
    const int oh_first = fast_div(p0, g.Wo, inv_wo);
    const int chunk_bytes = g.PR * g.PWT * 64;
    const int ih0 = oh_first * g.s + g.r0;
    const int cg_lo = ks * g.gps, cg_hi = cg_lo + g.gps < g.nchunks ? cg_lo + g.gps : g.nchunks;     // CG == 1: group = chunk
    const float* ximg = X + (size_t)img * g.H * g.W * g.C;
#ifdef DAM_PIPE_STAMPS
    unsigned long long* stamp_p = reinterpret_cast<unsigned long long*>(splitk_ws) +
                                  (size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64;
    int stamp_n = 0;
#endif
    DAM_PSTAMP(wave >> 2, 1);

    if (wave >= 4) {
        // ================= loader waves =================
        const int ltid = tid - 256;
        const int ipr = g.PWin * 4;                              // float4 items per patch row (one 16-channel chunk)
        const int total = g.PR * ipr;
        // items of this thread: e = ltid + PIPE_LTHREADS * u, u < nu (wave uniform: unused u cost nothing -- the geometry of an
        // item is ~25 vector instructions, and a full set of 16 was 2 us of set-up in front of every workgroup's first load)
        const int nu = (total + PIPE_LTHREADS - 1) / PIPE_LTHREADS;
        const int step_r = PIPE_LTHREADS / ipr, step_c = PIPE_LTHREADS - step_r * ipr;      // e += LTHREADS in (row, column)
        // global float index (chunk 0) or -1 = outside the tensor (zero); LDS byte offset (channel quad in bits 4-5) or -1
        int goff[PIPE_H][PIPE_U], dst[PIPE_H][PIPE_U];
        int pr = fast_div(ltid, ipr, 1.0f / (float)ipr), rem = ltid - pr * ipr;
#pragma unroll
        for (int h = 0; h < PIPE_H; ++h)
#pragma unroll
            for (int u = 0; u < PIPE_U; ++u) {
                goff[h][u] = -1; dst[h][u] = -1;
                if (h * PIPE_U + u < nu) {
                    if (pr < g.PR) {
                        const int pw = rem >> 2, cq = rem & 3;
                        const int ih = ih0 + pr, iw = g.c0 + pw;
                        const int slot = g.s == 1 ? pw : (pw & 1) * g.PWs + (pw >> 1);
                        dst[h][u] = ((pr * g.PWT + slot) * 16 + cq * 4) * 4;
                        if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) goff[h][u] = (ih * g.W + iw) * g.C + cq * 4;
                    }
                    pr += step_r; rem += step_c;
                    if (rem >= ipr) { rem -= ipr; ++pr; }
                }
            }
        float4 v[PIPE_U];
#define DAM_PIPE_ISSUE(H_, CHUNK_)                                                                                         \
    do {                                                                                                                   \
        _Pragma("unroll") for (int u = 0; u < PIPE_U; ++u) {                                                               \
            if ((H_) * PIPE_U + u >= nu) break;                                                                            \
            const int o_ = goff[H_][u] >= 0 ? goff[H_][u] + (CHUNK_) * 16 : 0;   /* unconditional load, clamped address */ \
            v[u] = *reinterpret_cast<const float4*>(ximg + (unsigned)o_);                                                  \
        }                                                                                                                  \
    } while (0)
#define DAM_PIPE_COMMIT(H_, CHUNK_, BUF_)                                                                                  \
    do {                                                                                                                   \
        _Pragma("unroll") for (int u = 0; u < PIPE_U; ++u) {                                                               \
            if ((H_) * PIPE_U + u >= nu) break;                                                                            \
            if (dst[H_][u] < 0) continue;                                                                                  \
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);                                                                    \
            if (goff[H_][u] >= 0) {                                                                                        \
                x = v[u];                                                                                                  \
                if (in_scale) {                                                                                            \
                    const int cq4_ = (dst[H_][u] >> 2) & 12;                                                               \
                    const float4 sc = *reinterpret_cast<const float4*>(in_scale + (CHUNK_) * 16 + cq4_);                   \
                    const float4 sh = *reinterpret_cast<const float4*>(in_shift + (CHUNK_) * 16 + cq4_);                   \
                    x.x = fmaf(x.x, sc.x, sh.x); x.y = fmaf(x.y, sc.y, sh.y);                                              \
                    x.z = fmaf(x.z, sc.z, sh.z); x.w = fmaf(x.w, sc.w, sh.w);                                              \
                    if (g.relu_in) {                                                                                       \
                        x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);        \
                    }                                                                                                      \
                }                                                                                                          \
            }                                                                                                              \
            *reinterpret_cast<float4*>(smem + (BUF_) + dst[H_][u]) = x;                                                    \
        }                                                                                                                  \
    } while (0)
        DAM_PIPE_ISSUE(0, cg_lo);
        // the slots of a stride-2 de-interleave that no item covers must read as zero: clear both buffers once (with stride 1
        // every slot a valid pixel reads is an item).  One loader wave's LDS writes execute in order, so its commits land
        // after its clear; two loader waves meet at barrier (0).
        if (g.s != 1)
            for (int e = ltid * 16; e < 2 * chunk_bytes; e += PIPE_LTHREADS * 16)
                *reinterpret_cast<float4*>(smem + e) = make_float4(0.f, 0.f, 0.f, 0.f);
        DAM_PSTAMP(1, 2);
        if (NL > 1) DAM_PIPE_BARRIER();             // (0)
        DAM_PSTAMP(1, 3);
        DAM_PIPE_COMMIT(0, cg_lo, 0);
        if (nu > PIPE_U) {
            DAM_PIPE_ISSUE(1, cg_lo);
            DAM_PIPE_COMMIT(1, cg_lo, 0);
        }
        if (cg_lo + 1 < cg_hi) DAM_PIPE_ISSUE(0, cg_lo + 1);
        DAM_PSTAMP(1, 4);
        DAM_PIPE_BARRIER();                         // (1) chunk cg_lo is staged
        DAM_PSTAMP(1, 5);
        for (int cg = cg_lo; cg < cg_hi; ++cg) {
            const int nbuf = (((cg - cg_lo) & 1) ^ 1) * chunk_bytes;
            if (cg + 1 < cg_hi) {
                DAM_PIPE_COMMIT(0, cg + 1, nbuf);
                if (nu > PIPE_U) {
                    DAM_PIPE_ISSUE(1, cg + 1);
                    DAM_PIPE_COMMIT(1, cg + 1, nbuf);
                }
                if (cg + 2 < cg_hi) DAM_PIPE_ISSUE(0, cg + 2);
            }
            DAM_PSTAMP(1, 6);
            DAM_PIPE_BARRIER();                     // (2 + i) chunk cg consumed, chunk cg + 1 staged
            DAM_PSTAMP(1, 7);
        }
#undef DAM_PIPE_ISSUE
#undef DAM_PIPE_COMMIT
        return;
    }

    // ================= compute waves =================
    int base_b[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        int p = p0 + wave * MW + mb * 16 + j;
        p = p < HoWo ? p : HoWo - 1;
        const int oh = fast_div(p, g.Wo, inv_wo), ow = p - oh * g.Wo;
        base_b[mb] = (((oh - oh_first) * g.s) * g.PWT + ow) * 64 + kq * 16;
    }
    v4f acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(Wp), 0, 0x7fffffff, 0x00020000);
    float4 wa[3][NB], xv[3][MB];
    // weight items in (chunk, tap a, tap b) order; the request pointer stops at the last item of this workgroup's K range
    int wc = cg_lo, wa_i = 0, wb_i = 0;
#ifdef DAM_PIPE_DIAG_NOW
#define DAM_PIPE_WGATE if (wgate++ < 3)
#else
#define DAM_PIPE_WGATE
#endif
#ifdef DAM_PIPE_DIAG_NOX
#define DAM_PIPE_XGATE if (xgate++ < 3)
#else
#define DAM_PIPE_XGATE
#endif
    int wgate = 0, xgate = 0;
    (void)wgate; (void)xgate;
#define DAM_PIPE_W(S_)                                                                                                     \
    DAM_PIPE_WGATE do {                                                                                                    \
        const int tap_ = g.wt_base + wa_i * g.wt_sa + wb_i * g.wt_sb;                                                      \
        const int ws_ = DAM_PIPE_WOFF(((tap_ * g.nchunks + wc) * g.NBtot + nb0) * 1024);                                   \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                                  \
            wa[S_][nb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + nb * 1024, ws_, 0)); \
        if (!(wc == cg_hi - 1 && wa_i == 2 && wb_i == 2)) {                                                                \
            if (++wb_i == 3) { wb_i = 0; if (++wa_i == 3) { wa_i = 0; ++wc; } }                                            \
        }                                                                                                                  \
    } while (0)
#define DAM_PIPE_X(S_, T_)                                                                                                 \
    DAM_PIPE_XGATE do {                                                                                                    \
        constexpr int a_ = (T_) / 3, b_ = (T_) % 3;                                                                        \
        const int roff_ = g.off_h + a_ * g.step_h - g.r0;                                                                  \
        const int coff_ = g.off_w + b_ * g.step_w - g.c0;                                                                  \
        const int slotoff_ = g.s == 1 ? coff_ : (coff_ & 1) * g.PWs + (coff_ >> 1);                                        \
        const int lo_ = boff + (roff_ * g.PWT + slotoff_) * 64;                                                            \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                  \
            xv[S_][mb] = *reinterpret_cast<const float4*>(smem + base_b[mb] + lo_);                                        \
    } while (0)
#ifdef DAM_PIPE_DIAG_NOMFMA
#define DAM_PIPE_MFMA(S_)                                                                                                  \
    do {                                                                                                                   \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                  \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                            \
                acc[mb][nb].x += wa[S_][nb].x * xv[S_][mb].x; acc[mb][nb].y += wa[S_][nb].y * xv[S_][mb].y;                \
                acc[mb][nb].z += wa[S_][nb].z * xv[S_][mb].z; acc[mb][nb].w += wa[S_][nb].w * xv[S_][mb].w;                \
            }                                                                                                              \
    } while (0)
#else
#define DAM_PIPE_MFMA(S_)                                                                                                  \
    do {                                                                                                                   \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                  \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                            \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[S_][nb].x, xv[S_][mb].x, acc[mb][nb], 0, 0, 0);      \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[S_][nb].y, xv[S_][mb].y, acc[mb][nb], 0, 0, 0);      \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[S_][nb].z, xv[S_][mb].z, acc[mb][nb], 0, 0, 0);      \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[S_][nb].w, xv[S_][mb].w, acc[mb][nb], 0, 0, 0);      \
            }                                                                                                              \
    } while (0)
#endif
    DAM_PIPE_W(0);                                  // weights of the first two items: nothing to wait for
    DAM_PIPE_W(1);
    DAM_PSTAMP(0, 2);
    if (NL > 1) DAM_PIPE_BARRIER();                 // (0)
    DAM_PSTAMP(0, 3);
    DAM_PIPE_BARRIER();                             // (1)
    DAM_PSTAMP(0, 5);
    for (int cg = cg_lo; cg < cg_hi; ++cg) {
        const int boff = ((cg - cg_lo) & 1) * chunk_bytes;
        DAM_PIPE_X(0, 0);
        DAM_PIPE_X(1, 1);
        DAM_PIPE_W(2); DAM_PIPE_X(2, 2); DAM_PIPE_MFMA(0);
        DAM_PIPE_W(0); DAM_PIPE_X(0, 3); DAM_PIPE_MFMA(1);
        DAM_PIPE_W(1); DAM_PIPE_X(1, 4); DAM_PIPE_MFMA(2);
        DAM_PIPE_W(2); DAM_PIPE_X(2, 5); DAM_PIPE_MFMA(0);
        DAM_PIPE_W(0); DAM_PIPE_X(0, 6); DAM_PIPE_MFMA(1);
        DAM_PIPE_W(1); DAM_PIPE_X(1, 7); DAM_PIPE_MFMA(2);
        DAM_PIPE_W(2); DAM_PIPE_X(2, 8); DAM_PIPE_MFMA(0);
        DAM_PIPE_W(0);                   DAM_PIPE_MFMA(1);      // weights of the next chunk's first two items
        DAM_PIPE_W(1);                   DAM_PIPE_MFMA(2);
        DAM_PSTAMP(0, 6);
        DAM_PIPE_BARRIER();                         // (2 + i)
        DAM_PSTAMP(0, 7);
    }
#undef DAM_PIPE_W
#undef DAM_PIPE_X
#undef DAM_PIPE_MFMA

    // ---- epilogue (as conv_igemm_kernel): lane holds channels 4*kq..+3 of pixel j of every (mb, nb) block ----
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int p = p0 + wave * MW + mb * 16 + j;
        if (p >= HoWo) continue;
        const int oh = fast_div(p, g.Wo, inv_wo), ow = p - oh * g.Wo;
        const size_t opix = ((size_t)img * g.OHt + (oh * g.os + g.oo_h)) * g.OWt + (ow * g.os + g.oo_w);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int ch = (nb0 + nb) * 16 + kq * 4;
            if (ch >= g.N) continue;
            v4f v = acc[mb][nb];
            const size_t o = opix * g.N + ch;
            if (g.ksplit > 1) {
                *reinterpret_cast<float4*>(splitk_ws + (size_t)ks * ((size_t)g.B * g.OHt * g.OWt * g.N) + o) =
                    make_float4(v.x, v.y, v.z, v.w);
                continue;
            }
            if (bias) {
                const float4 bv = *reinterpret_cast<const float4*>(bias + ch);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
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
            *reinterpret_cast<float4*>(Y + o) = make_float4(v.x, v.y, v.z, v.w);
        }
    }
    DAM_PSTAMP(0, 8);
}

template <int MB, int NB, int NL>
int launch_pipe(const ConvGeo& g, size_t lds, const float* X, const float* Wp, const float* bias, const float* sc, const float* sh,
                float* Y, const float* res, const float* res_mask, float* splitk_ws, hipStream_t st) {
    constexpr int TM = 64 * MB;
    dim3 grid((unsigned)cdiv((int64_t)g.Ho * g.Wo, TM), (unsigned)(cdiv(g.N / 16, NB) * g.ksplit), (unsigned)g.B);
    hipLaunchKernelGGL((conv_pipe_kernel<MB, NB, NL>), grid, dim3(256 + 64 * NL), lds, st, g, X, reinterpret_cast<const float4*>(Wp),
                       bias, sc, sh, Y, res, res_mask, splitk_ws);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

}  // namespace

// Returns DAM_ERR_UNSUPPORTED when the layer / tile does not fit this variant (the caller launches conv_igemm_kernel).
// g: fully set up by dam_conv2d_tapgrid_f32 (tile MB x NB chosen, PR, ksplit / gps) EXCEPT the channel-group size, which is 1 here.
int conv_pipe_try(ConvGeo g, int MB, int NB, const float* X, const float* Wp, const float* bias, const float* sc, const float* sh,
                  float* Y, const float* res, const float* res_mask, float* splitk_ws, hipStream_t st) {
    if (g.nA != 3 || g.nB != 3 || g.in_nchw || g.nchunks < 2) return DAM_ERR_UNSUPPORTED;
    const int total = g.PR * g.PWin * 4;
    if (total > 128 * PIPE_U * PIPE_H) return DAM_ERR_UNSUPPORTED;
    // two loader waves measured faster than one on every thick layer even where one would reach (the first chunk's staging is
    // exposed, and two waves halve it); DAM_PIPE_NL=1 is the diagnostic switch for the comparison
    int NL = 2;
    if (const char* e = getenv("DAM_PIPE_NL")) NL = atoi(e) == 1 && total <= 64 * PIPE_U * PIPE_H ? 1 : 2;
    const size_t lds = (size_t)2 * g.PR * g.PWT * 64;
    if (lds > 64 * 1024) return DAM_ERR_UNSUPPORTED;
    // split-K bookkeeping in chunks: gps was computed in groups of the caller's CG
    g.gps = g.ksplit > 1 ? g.gps * g.CG : g.nchunks;
    g.CG = 1;
#define DAM_PIPE_CASE(M_, N_)                                                                                        \
    if (MB == M_ && NB == N_)                                                                                        \
        return NL == 2 ? launch_pipe<M_, N_, 2>(g, lds, X, Wp, bias, sc, sh, Y, res, res_mask, splitk_ws, st)        \
                       : launch_pipe<M_, N_, 1>(g, lds, X, Wp, bias, sc, sh, Y, res, res_mask, splitk_ws, st)
    DAM_PIPE_CASE(4, 4); DAM_PIPE_CASE(4, 2); DAM_PIPE_CASE(4, 1);
    DAM_PIPE_CASE(2, 4); DAM_PIPE_CASE(2, 2); DAM_PIPE_CASE(2, 1);
    DAM_PIPE_CASE(1, 4); DAM_PIPE_CASE(1, 2); DAM_PIPE_CASE(1, 1);
#undef DAM_PIPE_CASE
    return DAM_ERR_UNSUPPORTED;
}

}  // namespace dam
