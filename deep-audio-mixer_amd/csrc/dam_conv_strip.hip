// dam_conv_strip.hip -- persistent "strip" variant of the implicit-GEMM convolution (forward / dgrad) for layers whose
// input rows, with ALL channels, fit an LDS ring: the HBM-bound full-resolution layers of ResNet18.
//
// Same GEMM mapping, packed weights and tap-grid semantics as conv_igemm_kernel (dam_conv.hip); what changes is how the
// input reaches LDS and how much of it is re-read:
//   * a workgroup walks `tpw` CONSECUTIVE 64*MB-pixel tiles of one image.  Input rows live in an LDS ring indexed by
//     (absolute row & (NR-1)); a new tile only fetches the rows the previous tiles did not (halo re-reads drop from
//     2.5x to ~1.1x of the input for 3x3 convs on 130-wide images);
//   * rows are fetched by dedicated LOADER waves (waves 4..7: one wave sustains only ~1 LDS-DMA piece per 1k cycles)
//     with global_load_lds_dwordx4 (LDS-DMA: no VGPR round trip,
//     1 KB per instruction) for tile k+1 while the 4 compute waves run the MFMAs of tile k; one workgroup barrier per
//     tile.  Compute waves never wait on HBM: their only global loads are the L2-resident packed weights;
//   * LDS image: [chunk][ring row][column slot][16 ch] with the same stride-2 column de-interleave as the tile kernel;
//     border slots (zero padding) are zeroed once, the DMA only writes in-tensor pixels;
//   * optional epilogue: per-channel BatchNorm partial statistics (n, mean, M2) of the produced tiles, one record per
//     workgroup, merged later by bn_stats_finalize -- removes the separate statistics pass over the conv output.
#include "dam_common.h"
#include "dam_conv_geo.h"

namespace dam {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DDAM_STAMPS): the `stats` buffer receives s_memtime stamps of phase boundaries instead.
#ifdef DAM_STAMPS
#define DAM_STAMP(slot)                                                                                   \
    do {                                                                                                  \
        if (lane == 0 && (wave == 0 || wave == 4) && stamp_i < 30)                                        \
            reinterpret_cast<unsigned long long*>(stats)[(((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 2 + (wave == 4)) * 32 + \
                                                         (stamp_i++)] = __builtin_amdgcn_s_memtime() | ((unsigned long long)(slot) << 56); \
    } while (0)
#else
#define DAM_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ int fdiv(int e, int d, float inv_d) {   // e / d, 0 <= e < 2^22
    int q = (int)((float)e * inv_d);
    if (q * d > e) --q;
    if ((q + 1) * d <= e) ++q;
    return q;
}

constexpr int STRIP_LOADERS = 4;                        // loader waves per workgroup (waves 4..7)
constexpr int STRIP_THREADS = 256 + 64 * STRIP_LOADERS;

template <int MB, int NB>
__global__ __launch_bounds__(STRIP_THREADS) void conv_strip_kernel(const ConvGeo g, const StripGeo sg, const float* __restrict__ X,
                                                         const float4* __restrict__ Wp, const float* __restrict__ bias,
                                                         float* __restrict__ Y, const float* __restrict__ res,
                                                         const float* __restrict__ res_mask, float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    constexpr int MW = 16 * MB, TM = 4 * MW;
    const int img = blockIdx.z, nb0 = blockIdx.y * NB;
    const int HoWo = g.Ho * g.Wo;
    const int t_begin = blockIdx.x * sg.tpw;
    int t_end = t_begin + sg.tpw;
    if (t_end > sg.tiles_m) t_end = sg.tiles_m;
    const int RB = g.PWT * 64;                 // bytes of one ring row of one chunk plane
    const int CHB = sg.NR * RB;                // bytes of one chunk plane
    const int RH = sg.RH;                      // input rows touched by one output row
    const float inv_wo = 1.0f / (float)g.Wo;
    const float* ximg = X + (size_t)img * g.H * g.W * g.C;
#ifdef DAM_STAMPS
    int stamp_i = 0;
#endif
    DAM_STAMP(1);

    // zero the whole ring once: border slots stay zero for the lifetime of the workgroup
    for (int e = tid * 16; e < CHB * g.nchunks; e += STRIP_THREADS * 16)
        *reinterpret_cast<float4*>(smem + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    // thin layers: the packed weights of this N tile are small; keep them in LDS so that the MFMA loop never waits on L2
    const int w_base = CHB * g.nchunks;       // always address LDS as smem + integer offset (a derived pointer variable
                                              // degrades to flat_load, which is slower and also counts on vmcnt)
    if (sg.w_lds) {
        const int n4 = sg.w_taps * g.nchunks * NB * 64;          // float4 count: [tap][chunk][nb][lane]
        for (int e = tid; e < n4; e += STRIP_THREADS) {
            const int ln = e & 63, nb = (e >> 6) % NB, tc = (e >> 6) / NB;
            *reinterpret_cast<float4*>(smem + w_base + e * 16) = Wp[((size_t)tc * g.NBtot + nb0 + nb) * 64 + ln];
        }
    }
    __syncthreads();
    DAM_STAMP(2);

    // rows [lo, hi] of the input needed by tile t
    auto tile_rows = [&](int t, int& lo, int& hi) {
        const int p0 = t * TM;
        int p1 = p0 + TM - 1;
        if (p1 > HoWo - 1) p1 = HoWo - 1;
        lo = fdiv(p0, g.Wo, inv_wo) * g.s + g.r0;
        hi = fdiv(p1, g.Wo, inv_wo) * g.s + g.r0 + RH - 1;
    };
    // fetch input rows [lo, hi] into the ring.  The lane -> (column slot, channel quad) mapping of a DMA group does not
    // depend on the row, so it is computed once per group; per row and chunk a piece then costs an M0 write, a scalar
    // base update and the global_load_lds itself.  `part`/`nparts`: the rows are dealt round-robin to `nparts` waves.
    const int groups_per_plane = (g.PWT * 4 + 63) >> 6;
    auto load_rows = [&](int lo, int hi, int part, int nparts) {
        for (int gi = 0; gi < groups_per_plane; ++gi) {
            const int L = gi * 64 + lane;                    // float4 index inside a row plane
            const int slot = L >> 2, quad = L & 3;
            int pw = slot;
            if (g.s != 1) pw = slot < g.PWs ? 2 * slot : 2 * (slot - g.PWs) + 1;
            const int iw = pw + g.c0;
            const bool ok = slot < g.PWT && pw < g.PWin && iw >= 0 && iw < g.W;
            const unsigned lane_off = (unsigned)((iw * g.C + quad * 4) * 4);      // bytes inside an input row
            for (int ih = lo + part; ih <= hi; ih += nparts) {
                const int slot_row = (ih + sg.ring_off) & (sg.NR - 1);
                const bool row_ok = ih >= 0 && ih < g.H;
                const char* rowp = reinterpret_cast<const char*>(ximg) + (size_t)(row_ok ? ih : 0) * g.W * g.C * 4;
                for (int cc = 0; cc < g.nchunks; ++cc) {
                    unsigned char* plane = smem + cc * CHB + slot_row * RB;
                    if (row_ok) {
                        if (ok)
                            __builtin_amdgcn_global_load_lds(
                                (const __attribute__((address_space(1))) void*)(rowp + cc * 64 + lane_off),
                                (__attribute__((address_space(3))) void*)(plane + gi * 1024), 16, 0, 0);
                    } else if (ok) {                                 // row outside the image: zeros
                        *reinterpret_cast<float4*>(plane + L * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        }
    };

    int loaded_hi;
    {
        int lo, hi;
        tile_rows(t_begin, lo, hi);
        load_rows(lo, hi, wave, 4 + STRIP_LOADERS);          // first tile: every wave fetches (the compute waves have nothing else to do)
        loaded_hi = hi;
    }
    __syncthreads();      // (compiler drains vmcnt before the barrier: the DMA has landed)
    DAM_STAMP(3);
    // the loader shares its SIMD with compute waves that keep the issue port busy: without priority it is starved
    if (wave >= 4) __builtin_amdgcn_s_setprio(3);

    // BatchNorm partial statistics of this workgroup's outputs (shifted sums per lane, channels 4*kq..+3 of block nb)
    float st_k[NB][4], st_s1[NB][4], st_s2[NB][4];
    int st_n = 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st_k[nb][r] = 0.f; st_s1[nb][r] = 0.f; st_s2[nb][r] = 0.f; }

    for (int t = t_begin; t < t_end; ++t) {
        if (wave >= 4) {
            if (t + 1 < t_end) {
                int lo, hi;
                tile_rows(t + 1, lo, hi);
                if (hi > loaded_hi) load_rows(loaded_hi + 1 > lo ? loaded_hi + 1 : lo, hi, wave - 4, STRIP_LOADERS);
            }
            DAM_STAMP(4);
        } else {
            const int p0 = t * TM;
            int colbase[MB], ohs[MB], pix[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                int p = p0 + wave * MW + mb * 16 + j;
                pix[mb] = p;
                p = p < HoWo ? p : HoWo - 1;
                const int oh = fdiv(p, g.Wo, inv_wo), ow = p - oh * g.Wo;
                ohs[mb] = oh * g.s + sg.ring_off;
                colbase[mb] = ow * 64 + kq * 16;
            }
            v4f acc[MB][NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (v4f){0.f, 0.f, 0.f, 0.f};

            const int n_items = g.nA * g.nB * g.nchunks;
            int ia = 0, ib = 0, ic = 0;
            auto item = [&](int a, int b, int cc, int& aoff, int& coloff, size_t& w_off) {
                aoff = g.off_h + a * g.step_h;                        // input row offset of the tap (ring index added per lane)
                const int coff = g.off_w + b * g.step_w - g.c0;
                const int slotoff = g.s == 1 ? coff : (coff & 1) * g.PWs + (coff >> 1);
                coloff = cc * CHB + slotoff * 64;
                const int tap = g.wt_base + a * g.wt_sa + b * g.wt_sb;
                w_off = sg.w_lds ? ((size_t)(tap * g.nchunks + cc) * NB) * 64 + lane
                                 : ((size_t)(tap * g.nchunks + cc) * g.NBtot + nb0) * 64 + lane;
            };
            auto fetch = [&](int aoff, int coloff, size_t w_off, float4 (&wa)[NB], float4 (&xv)[MB]) {
                if (sg.w_lds) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) wa[nb] = *reinterpret_cast<const float4*>(smem + w_base + (int)(w_off + nb * 64) * 16);
                } else {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) wa[nb] = Wp[w_off + nb * 64];
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int row = (ohs[mb] + aoff) & (sg.NR - 1);
                    xv[mb] = *reinterpret_cast<const float4*>(smem + row * RB + colbase[mb] + coloff);
                }
            };
            float4 wa_n[NB], xv_n[MB];
            {
                int aoff, coloff; size_t wo;
                item(0, 0, 0, aoff, coloff, wo);
                fetch(aoff, coloff, wo, wa_n, xv_n);
            }
            for (int it = 0; it < n_items; ++it) {
                float4 wa[NB], xv[MB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wa[nb] = wa_n[nb];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) xv[mb] = xv_n[mb];
                if (++ic == g.nchunks) { ic = 0; if (++ib == g.nB) { ib = 0; ++ia; } }
                if (it + 1 < n_items) {
                    int aoff, coloff; size_t wo;
                    item(ia, ib, ic, aoff, coloff, wo);
                    fetch(aoff, coloff, wo, wa_n, xv_n);
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nb].x, xv[mb].x, acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nb].y, xv[mb].y, acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nb].z, xv[mb].z, acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nb].w, xv[mb].w, acc[mb][nb], 0, 0, 0);
                    }
            }

            DAM_STAMP(5);
            // epilogue: lane holds channels 4*kq..+3 of pixel j of every (mb, nb) block
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int p = pix[mb];
                if (p >= HoWo) continue;
                const int oh = fdiv(p, g.Wo, inv_wo), ow = p - oh * g.Wo;
                const size_t opix = ((size_t)img * g.OHt + (oh * g.os + g.oo_h)) * g.OWt + (ow * g.os + g.oo_w);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int ch = (nb0 + nb) * 16 + kq * 4;
                    if (ch >= g.N) continue;
                    v4f v = acc[mb][nb];
                    if (bias) {
                        const float4 bv = *reinterpret_cast<const float4*>(bias + ch);
                        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    }
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
                    *reinterpret_cast<float4*>(Y + o) = make_float4(v.x, v.y, v.z, v.w);
#ifndef DAM_STAMPS
                    if (stats) {
                        const float e[4] = {v.x, v.y, v.z, v.w};
                        if (st_n == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) st_k[nb][r] = e[r];
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float d = e[r] - st_k[nb][r];
                            st_s1[nb][r] += d;
                            st_s2[nb][r] = fmaf(d, d, st_s2[nb][r]);
                        }
                    }
#endif
                }
                if (stats) ++st_n;
            }
        }
        if (t + 1 < t_end) {
            int lo, hi;
            tile_rows(t + 1, lo, hi);
            if (hi > loaded_hi) loaded_hi = hi;
        }
        // tile boundary: the loader's DMA must have landed, the compute waves' LDS reads are already consumed by their
        // MFMAs.  A raw s_barrier (not __syncthreads) so that the compute waves do NOT drain their output stores here.
        DAM_STAMP(6);
        if (wave >= 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        DAM_STAMP(7);
    }

#ifdef DAM_STAMPS
    DAM_STAMP(8);
    return;
#endif
    if (stats) {
        // (n, mean, M2) per lane -> Chan merge over the 16 pixel lanes, then over the 4 compute waves through LDS
        float* sm = reinterpret_cast<float*>(smem);        // ring no longer needed: [4 waves][NB*16 ch][3]
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float n = (float)st_n;
                const float md = st_n ? st_s1[nb][r] / n : 0.f;
                float mean = st_k[nb][r] + md;
                float m2 = st_n ? fmaxf(st_s2[nb][r] - st_s1[nb][r] * md, 0.f) : 0.f;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const float nb_ = __shfl_xor(n, off), mb_ = __shfl_xor(mean, off), qb_ = __shfl_xor(m2, off);
                    const float nn = n + nb_;
                    if (nn > 0.f) {
                        const float d = mb_ - mean;
                        mean += d * (nb_ / nn);
                        m2 += qb_ + d * d * (n * nb_ / nn);
                    }
                    n = nn;
                }
                if (j == 0 && wave < 4) {
                    float* o = sm + ((wave * NB * 16) + nb * 16 + kq * 4 + r) * 3;
                    o[0] = n; o[1] = mean; o[2] = m2;
                }
            }
        __syncthreads();
        if (tid < NB * 16) {
            float n = 0.f, mean = 0.f, m2 = 0.f;
            for (int w = 0; w < 4; ++w) {
                const float* o = sm + ((w * NB * 16) + tid) * 3;
                const float nb_ = o[0];
                if (nb_ == 0.f) continue;
                const float nn = n + nb_, d = o[1] - mean;
                mean += d * (nb_ / nn);
                m2 += o[2] + d * d * (n * nb_ / nn);
                n = nn;
            }
            const int ch = nb0 * 16 + tid;
            if (ch < g.N) {
                const size_t part = (size_t)blockIdx.z * gridDim.x + blockIdx.x;
                float* o = stats + (part * g.N + ch) * 3;
                o[0] = n; o[1] = mean; o[2] = m2;
            }
        }
    }
}

}  // namespace

// Returns DAM_OK if launched, DAM_ERR_UNSUPPORTED if the layer does not fit this variant (caller falls back).
template <int MB, int NB>
static int launch_strip(ConvGeo& g, StripGeo& sg, size_t lds, const float* X, const float* Wp, const float* bias, float* Y,
                        const float* res, const float* res_mask, float* stats, hipStream_t st) {
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_strip_kernel<MB, NB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return DAM_ERR_LAUNCH;
            raised = true;
        }
    }
    dim3 grid((unsigned)sg.strips, (unsigned)cdiv(g.N / 16, NB), (unsigned)g.B);
    hipLaunchKernelGGL((conv_strip_kernel<MB, NB>), grid, dim3(STRIP_THREADS), lds, st, g, sg, X, reinterpret_cast<const float4*>(Wp), bias,
                       Y, res, res_mask, stats);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

int conv_strip_try(ConvGeo& g, int h_lo, int h_hi, const float* X, const float* Wp, const float* bias, float* Y,
                   const float* res, const float* res_mask, float* stats, int* stats_parts, hipStream_t st) {
    if (g.in_nchw || g.B > 65535) return DAM_ERR_UNSUPPORTED;
    const int nblk = g.N / 16;
    const int64_t npix = (int64_t)g.Ho * g.Wo;
    if (npix >= (1 << 22)) return DAM_ERR_UNSUPPORTED;
    const int NB = nblk % 4 == 0 ? 4 : (nblk % 2 == 0 ? 2 : 1);
    if (stats && cdiv(nblk, NB) != 1) return DAM_ERR_UNSUPPORTED;        // statistics need all channels in one workgroup
    StripGeo sg;
    sg.RH = h_hi - h_lo + 1;
    const size_t LDS_MAX = 72 * 1024;
    int MB = 4;
    size_t lds = 0;
    for (;; MB >>= 1) {
        const int tm = 64 * MB;
        int rows_out = (int)((tm + g.Wo - 2) / g.Wo + 1);
        if (rows_out > g.Ho) rows_out = g.Ho;
        const int rows_tile = (rows_out - 1) * g.s + sg.RH;       // rows one tile reads
        const int rows_new = rows_out * g.s;                      // rows the next tile can add
        int nr = 1;
        while (nr < rows_tile + rows_new) nr <<= 1;
        lds = (size_t)nr * g.PWT * 64 * g.nchunks;
        if (lds <= LDS_MAX) { sg.NR = nr; break; }
        if (MB == 2) return DAM_ERR_UNSUPPORTED;     // 64-pixel strips measured slower than the tile kernel (MB = 1)
    }
    const int tm = 64 * MB;
    sg.tiles_m = (int)cdiv(npix, tm);
    const int64_t total = (int64_t)sg.tiles_m * g.B * cdiv(nblk, NB);
    // one resident round: <= 512 workgroups (2 per CU) when a strip of <= 32 tiles allows it, else many small ones
    int tpw = (int)cdiv(total, 512);
    if (tpw > 32) tpw = 32;
    if (tpw > sg.tiles_m) tpw = sg.tiles_m;
    sg.tpw = tpw;
    sg.strips = (int)cdiv(sg.tiles_m, tpw);
    sg.ring_off = sg.NR * 64;           // keeps (row + ring_off) non-negative for row >= -64*NR
    sg.w_taps = g.wt_base + (g.nA - 1) * g.wt_sa + (g.nB - 1) * g.wt_sb + 1;
    const size_t w_bytes = (size_t)sg.w_taps * g.nchunks * NB * 1024;
    sg.w_lds = (w_bytes <= 40 * 1024 && lds + w_bytes <= 80 * 1024) ? 1 : 0;
    if (sg.w_lds) lds += w_bytes;
    if (stats_parts) *stats_parts = sg.strips * g.B;
    if (stats && (int64_t)sg.strips * g.B > 1024) return DAM_ERR_UNSUPPORTED;
    if (lds < (size_t)4 * NB * 16 * 3 * sizeof(float)) lds = (size_t)4 * NB * 16 * 3 * sizeof(float);
#define DAM_STRIP_CASE(M_, N_) \
    if (MB == M_ && NB == N_) return launch_strip<M_, N_>(g, sg, lds, X, Wp, bias, Y, res, res_mask, stats, st)
    DAM_STRIP_CASE(4, 4); DAM_STRIP_CASE(4, 2); DAM_STRIP_CASE(4, 1);
    DAM_STRIP_CASE(2, 4); DAM_STRIP_CASE(2, 2); DAM_STRIP_CASE(2, 1);
    DAM_STRIP_CASE(1, 4); DAM_STRIP_CASE(1, 2); DAM_STRIP_CASE(1, 1);
#undef DAM_STRIP_CASE
    return DAM_ERR_UNSUPPORTED;
}

}  // namespace dam
