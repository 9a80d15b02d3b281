// dam_conv_s2.hip -- forward of the two convolutions a down-sampling block applies to its input, in one launch
// (models/model_resnet.py:17-21,24-26 of the reference: conv1 = 3x3 / stride 2 / pad 1 and the shortcut's 1x1 / stride 2, both
// followed by a training-mode BatchNorm):
//
//     c1[i, j] = sum_{a, b} W1[a][b] x[2i + a - 1, 2j + b - 1]          cs[i, j] = Wsc x[2i, 2j]   (= the centre tap's operand)
//
// plus the BatchNorm statistics records (n, mean, M2) of both outputs.  Rounds 1-3 ran this as two or three launches of the
// loader-wave kernel (its column ranges) + a direct 1x1 launch + a statistics pass over both outputs: 85 us for the 16 -> 32
// channel block of the ResNet at 1025 x 130, 60 us for 32 -> 64, the convolutions at 0.30-0.42 of the fp32 matrix peak.  Here
// the workgroups are persistent and keep BOTH packed weight images in LDS (20 / 80 KB); a wave takes 16 consecutive output
// pixels (rows run on into each other) per unit, loads the nine stride-2 views of x straight into the MFMA's B operand (one
// 16-byte load per lane and tap: the taps of neighbouring pixels overlap in L1), runs 9 * NB * 4 MFMAs per 16-channel chunk and
// NB * 4 more for the shortcut from the centre operand it already holds, and stores both outputs as full pixel rows.  Two
// operand sets: the next chunk's (or next unit's) nine loads are in flight under the current MFMAs.  Statistics: every lane
// keeps shifted sums (pivot = its first value) of the 4 * NB channels it owns across all its units, the sixteen pixel lanes are
// merged once at the end (Chan), then the four waves: one record per workgroup -- no pass over c1 / cs.
#include "dam_common.h"
#include "dam_bn_fin.h"
#include <cstdlib>

namespace dam {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// NB: output-channel blocks of 16 (Co / 16), NCH: input chunks of 16 (Ci / 16)
// WAVES: waves per workgroup (4; 8 where the weight images leave room for one workgroup per CU only -- two waves per SIMD cover each
// other's waits: a wave's stores sit in the same in-order counter as its operand loads, every unit starts by waiting for them)
template <int NB, int NCH, bool STATS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void conv_s2_pair_kernel(const float* __restrict__ X, unsigned x_bytes, const float4* __restrict__ Wp,
                                                           const float4* __restrict__ Wp2, int Hd, int Wd, int H, int W,
                                                           float* __restrict__ Y, float* __restrict__ Ys, float* __restrict__ P1,
                                                           float* __restrict__ P2, int total_px, int total_units, int xcd_aware) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [9 + 1][NCH][NB][64 lanes] float4
    constexpr int Ci = 16 * NCH, Co = 16 * NB;
    static_assert(NCH == 1 || NCH == 2, "the operand sets alternate per chunk (NCH == 2) or per unit (NCH == 1)");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, x_bytes, 0x00020000);
    const int w_lane = lane * 16;
    float4 xs[2][9], wa[2][NB];
    v4f acc1[NB], accs[NB];
    float k1[STATS ? NB : 1][4], s11[STATS ? NB : 1][4], s21[STATS ? NB : 1][4];       // conv1: pivot, sum d, sum d^2
    float ks[STATS ? NB : 1][4], s1s[STATS ? NB : 1][4], s2s[STATS ? NB : 1][4];       // shortcut
    int n_px = 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        acc1[nb] = (v4f){0.f, 0.f, 0.f, 0.f};
        accs[nb] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (STATS) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) { k1[nb][q] = s11[nb][q] = s21[nb][q] = 0.f; ks[nb][q] = s1s[nb][q] = s2s[nb][q] = 0.f; }
    }
    // this lane's output pixel of unit U_: byte offset of x[2i][2j] (its channel quad) and which neighbours exist
    //   flag bits: 1 pixel live, 2 row 2i-1 exists, 4 row 2i+1 exists, 8 column 2j-1 exists, 16 column 2j+1 exists
#define DAM_CS2_OFFS(U_, BASE_, FL_)                                                                                          \
    do {                                                                                                                      \
        const int p_ = (U_) * 16 + j;                                                                                         \
        const int row_ = p_ / Wd, col_ = p_ - row_ * Wd, img_ = row_ / Hd, i_ = row_ - img_ * Hd;                             \
        BASE_ = (((img_ * H + 2 * i_) * W + 2 * col_) * Ci + kq * 4) * 4;                                                     \
        FL_ = ((U_) < unit_end && p_ < total_px)                                                                             \
                  ? (1 | (i_ > 0 ? 2 : 0) | (2 * i_ + 1 < H ? 4 : 0) | (col_ > 0 ? 8 : 0) | (2 * col_ + 1 < W ? 16 : 0))      \
                  : 0;                                                                                                        \
    } while (0)
    // the nine taps' operands of chunk CH_ (a tap outside the image: an offset the range check rejects -> zeros)
#define DAM_CS2_LOAD(S_, BASE_, FL_, CH_)                                                                                     \
    _Pragma("unroll") for (int a = 0; a < 3; ++a) {                                                                           \
        _Pragma("unroll") for (int b = 0; b < 3; ++b) {                                                                       \
            const int need = 1 | (a == 0 ? 2 : a == 2 ? 4 : 0) | (b == 0 ? 8 : b == 2 ? 16 : 0);                              \
            const int off = ((FL_) & need) == need ? (BASE_) + ((a - 1) * W + (b - 1)) * (Ci * 4) : 0x7fffffff;               \
            xs[S_][a * 3 + b] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, (CH_) * 64, 0));    \
        }                                                                                                                     \
    }
#define DAM_CS2_MFMA4(ACC_, WA_, XV_)                                                                                         \
    do {                                                                                                                      \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x4f32(WA_.x, XV_.x, ACC_, 0, 0, 0);                                             \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x4f32(WA_.y, XV_.y, ACC_, 0, 0, 0);                                             \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x4f32(WA_.z, XV_.z, ACC_, 0, 0, 0);                                             \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x4f32(WA_.w, XV_.w, ACC_, 0, 0, 0);                                             \
    } while (0)
    // tap T_'s NB weight fragments (T_ == 9: the shortcut's image behind the nine-tap one) -- requested ONE TAP AHEAD of their MFMAs:
    // left to itself the compiler reads them right in front of their use and every eight MFMAs wait a full LDS round trip
#define DAM_CS2_WREAD(BUF_, T_, CH_)                                                                                          \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                                         \
        wa[BUF_][nb] = *reinterpret_cast<const float4*>(smem + w_lane + (((T_) * NCH + (CH_)) * NB + nb) * 1024);
#define DAM_CS2_COMPUTE(S_, CH_)                                                                                              \
    do {                                                                                                                      \
        DAM_CS2_WREAD(0, 0, CH_)                                                                                              \
        _Pragma("unroll") for (int t = 0; t < 10; ++t) {                                                                      \
            if (t < 9) { DAM_CS2_WREAD((t + 1) & 1, t + 1, CH_) }                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
            /* component-major: consecutive MFMAs go to different accumulators (NB independent chains) */                     \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                   \
                _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                           \
                    const float wq = q == 0 ? wa[t & 1][nb].x : q == 1 ? wa[t & 1][nb].y : q == 2 ? wa[t & 1][nb].z : wa[t & 1][nb].w; \
                    const float4 xv = xs[S_][t < 9 ? t : 4];                                                                  \
                    const float xq = q == 0 ? xv.x : q == 1 ? xv.y : q == 2 ? xv.z : xv.w;                                    \
                    if (t < 9) acc1[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq, xq, acc1[nb], 0, 0, 0);                    \
                    else accs[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq, xq, accs[nb], 0, 0, 0);                          \
                }                                                                                                             \
            }                                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
        }                                                                                                                     \
    } while (0)
    // write-out of unit U_ (lane: channels 4 kq .. +3 of every block of its pixel), statistics, accumulators back to zero
#define DAM_CS2_EPILOGUE(U_)                                                                                                  \
    do {                                                                                                                      \
        const int p_ = (U_) * 16 + j;                                                                                         \
        if (p_ < total_px) {                                                                                                  \
            float* o1 = Y + (size_t)p_ * Co + kq * 4;                                                                         \
            float* o2 = Ys + (size_t)p_ * Co + kq * 4;                                                                        \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                               \
                *reinterpret_cast<v4f*>(o1 + nb * 16) = acc1[nb];                                                             \
                *reinterpret_cast<v4f*>(o2 + nb * 16) = accs[nb];                                                             \
            }                                                                                                                 \
            if constexpr (STATS) {                                                                                            \
                _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                           \
                    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
                        if (n_px == 0) { k1[nb][q] = acc1[nb][q]; ks[nb][q] = accs[nb][q]; }                                  \
                        const float d1 = acc1[nb][q] - k1[nb][q], d2 = accs[nb][q] - ks[nb][q];                               \
                        s11[nb][q] += d1; s21[nb][q] = fmaf(d1, d1, s21[nb][q]);                                              \
                        s1s[nb][q] += d2; s2s[nb][q] = fmaf(d2, d2, s2s[nb][q]);                                              \
                    }                                                                                                         \
                }                                                                                                             \
                ++n_px;                                                                                                       \
            }                                                                                                                 \
        }                                                                                                                     \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                                   \
            acc1[nb] = (v4f){0.f, 0.f, 0.f, 0.f};                                                                             \
            accs[nb] = (v4f){0.f, 0.f, 0.f, 0.f};                                                                             \
        }                                                                                                                     \
    } while (0)

    // XCD-aware unit order: workgroup b runs on XCD b % 8 and every XCD has its own L2.  Each XCD takes one CONTIGUOUS eighth of
    // the units, so the rows that neighbouring units share (the taps reach one row up and down) are fetched into ONE L2 instead of
    // up to three; inside an XCD wave-major: the units left over after the last full round go to ONE wave each of different
    // workgroups (SIMDs).  DAM_S2_NO_XCD=1 (read by the launcher: `xcd_aware`): the plain interleaved order (A/B).
    const int n_xcd = (xcd_aware && (gridDim.x & 7) == 0) ? 8 : 1;
    const int wg_per_xcd = gridDim.x / n_xcd, per_xcd = (total_units + n_xcd - 1) / n_xcd;
    const int u_lo = (blockIdx.x % n_xcd) * per_xcd;
    const int unit_end = u_lo + per_xcd < total_units ? u_lo + per_xcd : total_units;
    const int ustride = wg_per_xcd * WAVES;
    int unit = u_lo + wave * wg_per_xcd + blockIdx.x / n_xcd;
    int base0, fl0, base1, fl1;
    DAM_CS2_OFFS(unit, base0, fl0);
    DAM_CS2_LOAD(0, base0, fl0, 0)
    // (the weight images go to LDS BEHIND the first unit's operand requests: their round trips overlap)
    {
        constexpr int N4 = 9 * NCH * NB * 64, N4P = NCH * NB * 64;
        for (int e = tid; e < N4; e += 64 * WAVES) reinterpret_cast<float4*>(smem)[e] = Wp[e];
        for (int e = tid; e < N4P; e += 64 * WAVES) reinterpret_cast<float4*>(smem)[N4 + e] = Wp2[e];
    }
    __syncthreads();
    while (unit < unit_end) {                                            // wave-uniform
        if constexpr (NCH == 2) {
            DAM_CS2_LOAD(1, base0, fl0, 1)
            DAM_CS2_OFFS(unit + ustride, base1, fl1);
            __builtin_amdgcn_sched_barrier(0);
            DAM_CS2_COMPUTE(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            DAM_CS2_LOAD(0, base1, fl1, 0)                               // (no next unit: flags 0, every load reads zeros)
            __builtin_amdgcn_sched_barrier(0);
            DAM_CS2_COMPUTE(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            DAM_CS2_EPILOGUE(unit);
            unit += ustride;
            base0 = base1; fl0 = fl1;
        } else {
            const int u1 = unit + ustride, u2 = u1 + ustride;
            DAM_CS2_OFFS(u1, base1, fl1);
            DAM_CS2_LOAD(1, base1, fl1, 0)
            __builtin_amdgcn_sched_barrier(0);
            DAM_CS2_COMPUTE(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            DAM_CS2_EPILOGUE(unit);
            DAM_CS2_OFFS(u2, base0, fl0);
            DAM_CS2_LOAD(0, base0, fl0, 0)
            __builtin_amdgcn_sched_barrier(0);
            if (u1 < unit_end) {
                DAM_CS2_COMPUTE(1, 0);
                __builtin_amdgcn_sched_barrier(0);
                DAM_CS2_EPILOGUE(u1);
            }
            unit = u2;
        }
    }
#undef DAM_CS2_OFFS
#undef DAM_CS2_LOAD
#undef DAM_CS2_MFMA4
#undef DAM_CS2_COMPUTE
#undef DAM_CS2_WREAD
#undef DAM_CS2_EPILOGUE
    if constexpr (STATS) {
        // lane -> (n, mean, M2) of its pixels per channel; Chan merge over the sixteen pixel lanes of a channel quad (xor 1, 2, 4, 8:
        // every lane ends with a full merge); waves without pixels contribute n = 0
        float cnt = (float)n_px;
        const float inv = n_px ? 1.0f / cnt : 0.f;
        float mean1[NB][4], m21[NB][4], means[NB][4], m2s[NB][4];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = s11[nb][q] * inv, b = s1s[nb][q] * inv;
                mean1[nb][q] = n_px ? k1[nb][q] + a : 0.f;
                m21[nb][q] = n_px ? fmaxf(s21[nb][q] - s11[nb][q] * a, 0.f) : 0.f;
                means[nb][q] = n_px ? ks[nb][q] + b : 0.f;
                m2s[nb][q] = n_px ? fmaxf(s2s[nb][q] - s1s[nb][q] * b, 0.f) : 0.f;
            }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const float cb = __shfl_xor(cnt, off), nn = cnt + cb;
            const float r = nn > 0.f ? cb / nn : 0.f, f = cnt * r;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float mb = __shfl_xor(mean1[nb][q], off), qb = __shfl_xor(m21[nb][q], off), d = mb - mean1[nb][q];
                    mean1[nb][q] = fmaf(d, r, mean1[nb][q]);
                    m21[nb][q] += qb + d * d * f;
                    mb = __shfl_xor(means[nb][q], off); qb = __shfl_xor(m2s[nb][q], off); d = mb - means[nb][q];
                    means[nb][q] = fmaf(d, r, means[nb][q]);
                    m2s[nb][q] += qb + d * d * f;
                }
            cnt = nn;
        }
        // the four waves' records -> one per workgroup (through the LDS the weights no longer need): 2048 records made the
        // finalize launch that follows 12 us instead of 5
        __syncthreads();
        float* rec = reinterpret_cast<float*>(smem);                 // [WAVES][2 outputs][Co][3]
        if (j == 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float* o1 = rec + ((wave * 2 + 0) * Co + nb * 16 + kq * 4 + q) * 3;
                    float* o2 = rec + ((wave * 2 + 1) * Co + nb * 16 + kq * 4 + q) * 3;
                    o1[0] = cnt; o1[1] = mean1[nb][q]; o1[2] = m21[nb][q];
                    o2[0] = cnt; o2[1] = means[nb][q]; o2[2] = m2s[nb][q];
                }
        }
        __syncthreads();
        if (tid < 2 * Co) {
            const int which = tid / Co, c = tid - which * Co;
            float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const float* r = rec + ((w * 2 + which) * Co + c) * 3;
                const float nb_ = r[0], nn = n + nb_, rr = nn > 0.f ? nb_ / nn : 0.f, d = r[1] - mean;
                mean = fmaf(d, rr, mean);
                m2 += r[2] + d * d * (n * rr);
                n = nn;
            }
            float* o = (which ? P2 : P1) + ((size_t)blockIdx.x * Co + c) * 3;
            o[0] = n; o[1] = mean; o[2] = m2;
        }
    }
}

template <int NB, int NCH, int WAVES>
int launch_conv_s2_pair(const float* x, const float* wp, const float* wp2, int B, int H, int W, float* y, float* ys, float* p1,
                        float* p2, int* parts_host, hipStream_t st) {
    const int Hd = (H + 1) / 2, Wd = (W + 1) / 2;
    const int64_t px = (int64_t)B * Hd * Wd, xb = (int64_t)B * H * W * NCH * 64;
    if (px >= (1ll << 26) || xb >= (1ll << 31)) return DAM_ERR_UNSUPPORTED;      // byte offsets of the range-checked loads
    const int64_t units = cdiv(px, 16);
    const size_t lds = (size_t)10 * NCH * NB * 1024;
    const int cus = device_cus();
    // Workgroups per CU by makespan, as in dam_dgrad_s2.hip: n resident waves per SIMD share its MFMA pipe; more waves hide the operand
    // latency better, which decides while the costs are within ~15 %.  One statistics record per workgroup, <= BN_RECORDS_MAX.
    const bool stats = p1 != nullptr;
    static PerDevice<int> occ[2];       // resident workgroups per CU of this instantiation (registers and LDS; asked once per device)
    int& oc = occ[stats ? 1 : 0]();
    if (!oc) {
        int n = 0;
        const hipError_t e = stats ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_s2_pair_kernel<NB, NCH, true, WAVES>, 64 * WAVES, lds)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_s2_pair_kernel<NB, NCH, false, WAVES>, 64 * WAVES, lds);
        oc = (e == hipSuccess && n >= 1) ? n : 1;
    }
    int max_per_cu = oc;
    if (max_per_cu > 3) max_per_cu = 3;
    if (WAVES == 8) max_per_cu = 1;
    if (max_per_cu < 1) max_per_cu = 1;
    while (max_per_cu > 1 && (int64_t)cus * max_per_cu > BN_RECORDS_MAX) --max_per_cu;
    if ((int64_t)cus > BN_RECORDS_MAX) return DAM_ERR_UNSUPPORTED;
    static const int forced = [] { const char* e = getenv("DAM_CS2_PER_CU"); return e ? atoi(e) : 0; }();      // A/B knob
    int per_cu = 1;
    int64_t best = 0;
    for (int n = 1; n <= max_per_cu; ++n) {
        const int64_t cost = cdiv(units, (int64_t)WAVES * cus * n) * n * 100;
        if (n == 1 || cost * 100 <= best * 115) { best = n == 1 ? cost : (cost < best ? cost : best); per_cu = n; }
    }
    if (forced >= 1 && forced <= max_per_cu) per_cu = forced;
    int64_t wgs = (int64_t)cus * per_cu;
    if (wgs > cdiv(units, WAVES)) wgs = cdiv(units, WAVES);
    static const int xcd_aware = getenv("DAM_S2_NO_XCD") ? 0 : 1;      // A/B knob
    if (parts_host) *parts_host = stats ? (int)wgs : 0;
    static PerDevice<bool> raised_pd[2];
    bool& raised = raised_pd[stats ? 1 : 0]();
    if (lds > 64 * 1024 && !raised) {
        const void* fn = stats ? reinterpret_cast<const void*>(&conv_s2_pair_kernel<NB, NCH, true, WAVES>)
                               : reinterpret_cast<const void*>(&conv_s2_pair_kernel<NB, NCH, false, WAVES>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DAM_ERR_LAUNCH;
        raised = true;
    }
    if (stats)
        hipLaunchKernelGGL((conv_s2_pair_kernel<NB, NCH, true, WAVES>), dim3((unsigned)wgs), dim3(64 * WAVES), lds, st, x, (unsigned)xb,
                           reinterpret_cast<const float4*>(wp), reinterpret_cast<const float4*>(wp2), Hd, Wd, H, W, y, ys, p1, p2, (int)px,
                           (int)units, xcd_aware);
    else
        hipLaunchKernelGGL((conv_s2_pair_kernel<NB, NCH, false, WAVES>), dim3((unsigned)wgs), dim3(64 * WAVES), lds, st, x, (unsigned)xb,
                           reinterpret_cast<const float4*>(wp), reinterpret_cast<const float4*>(wp2), Hd, Wd, H, W, y, ys, p1, p2, (int)px,
                           (int)units, xcd_aware);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}


// ---- the wide down-sampling blocks (64 -> 96 ... 128 -> 256 channels): the two weight images (240 KB ... 1.3 MB) do not fit a
// workgroup's LDS and the outputs are small (<= 17 k pixels).  As in dam_dgrad_s2.hip's streaming form a wave takes exactly ONE unit --
// 16 flattened output pixels x ONE 16-channel block of both outputs; two operand register sets keep the next chunk's nine loads in
// flight under the current chunk's 40 MFMAs.  The four waves of a workgroup take four consecutive pixel blocks of the SAME channel
// block: the ten weight fragments of a chunk are fetched once per workgroup into LDS (two chunk buffers, one barrier per chunk), and
// the waves' statistics merge through LDS into the workgroup's 16 channels of record blockIdx.x / NB: the NB workgroups of a pixel
// group fill one record between them.
// MB: pixel blocks of 16 per wave (1 is what ships; 2 = twice the MFMAs under every round of operand requests, measured slower)
template <bool STATS, int MB>
__global__ __launch_bounds__(256) void conv_s2_pair_stream_kernel(const float* __restrict__ X, unsigned x_bytes, const float4* __restrict__ Wp,
                                                                  const float4* __restrict__ Wp2, int Hd, int Wd, int H, int W, int NCH,
                                                                  int NB, float* __restrict__ Y, float* __restrict__ Ys,
                                                                  float* __restrict__ P1, float* __restrict__ P2, int total_px) {
    __shared__ float rec[4][2][16][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    // XCD-aware order: workgroup b runs on XCD b % 8 (round-robin dispatch; every XCD has its own L2) and the NB workgroups of a
    // pixel group read the same x -- dealt b -> (group, block) in plain order they sat on NB different XCDs and FETCH_SIZE was 6.8 x
    // the bytes of x on the 64 -> 96 block.  Here the groups are dealt to the XCDs (group g on XCD g % 8) and an XCD's workgroups
    // walk (group, block) pairs in order: the grid is padded to whole rounds of eight groups, the surplus workgroups leave.
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int nb = seq % NB, pgroup = (seq / NB) * 8 + xcd;
    if (pgroup * (64 * MB) >= total_px) return;                       // (whole workgroup: no barrier is left waiting)
    const int Ci = 16 * NCH, Co = 16 * NB;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, x_bytes, 0x00020000);
    int pix[MB], off[MB][9];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int p = ((pgroup * 4 + wave) * MB + mb) * 16 + j;       // this lane's output pixels (flattened [B * Hd * Wd])
        pix[mb] = p;
        const int row = p / Wd, col = p - row * Wd, img = row / Hd, i = row - img * Hd;
        const int base = (((img * H + 2 * i) * W + 2 * col) * Ci + kq * 4) * 4;
        const int fl = p < total_px ? (1 | (i > 0 ? 2 : 0) | (2 * i + 1 < H ? 4 : 0) | (col > 0 ? 8 : 0) | (2 * col + 1 < W ? 16 : 0)) : 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int need = 1 | (a == 0 ? 2 : a == 2 ? 4 : 0) | (b == 0 ? 8 : b == 2 ? 16 : 0);
                off[mb][a * 3 + b] = (fl & need) == need ? base + ((a - 1) * W + (b - 1)) * (Ci * 4) : 0x7fffffff;
            }
    }
    // The ten weight fragments of a chunk are the same for the four waves (one channel block per workgroup): each wave fetches a
    // quarter of them (640 quads over 256 threads) and they meet in LDS, two chunk buffers, one barrier per chunk -- every wave
    // loading all ten itself kept a CU's texture path ~90 % busy (nineteen 1 KB loads per 40 MFMAs).
    __shared__ float4 wbuf[2][10 * 64];
    float4 xs[2][MB][9], wa[2], wr0, wr1, wr2;
    const int we0 = tid, we1 = tid + 256, we2 = tid < 128 ? tid + 512 : 0;      // this thread's quads of a chunk's 640
    // weight quad e of chunk CH_: e < 576 -> tap e / 64 of the 3x3 image, else the shortcut's
#define DAM_CS2S_WSRC(E_, CH_)                                                                                               \
    ((E_) < 576 ? Wp + ((size_t)(((E_) >> 6) * NCH + (CH_)) * NB + nb) * 64 + ((E_) & 63)                                     \
                : Wp2 + ((size_t)(CH_) * NB + nb) * 64 + ((E_) & 63))
#define DAM_CS2S_WFETCH(CH_)                                                                                                  \
    do {                                                                                                                      \
        wr0 = *DAM_CS2S_WSRC(we0, CH_); wr1 = *DAM_CS2S_WSRC(we1, CH_); wr2 = *DAM_CS2S_WSRC(we2, CH_);                       \
    } while (0)
#define DAM_CS2S_WSTORE(BUF_)                                                                                                 \
    do {                                                                                                                      \
        wbuf[BUF_][we0] = wr0; wbuf[BUF_][we1] = wr1;                                                                         \
        if (tid < 128) wbuf[BUF_][we2] = wr2;                                                                                 \
    } while (0)
    // (the chunk rides in the scalar offset, which the range check ignores: the callers keep CH_ < NCH)
#define DAM_CS2S_XLOAD(S_, CH_)                                                                                               \
    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                         \
        _Pragma("unroll") for (int t = 0; t < 9; ++t)                                                                         \
            xs[S_][mb][t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, off[mb][t], (CH_) * 64, 0));
    // the MFMAs of one chunk: tap t's fragment is read from LDS one tap ahead of its use (tap 9 = the shortcut's)
#define DAM_CS2S_CHUNK(S_, BUF_)                                                                                              \
    do {                                                                                                                      \
        wa[0] = wbuf[BUF_][lane];                                                                                             \
        _Pragma("unroll") for (int t = 0; t < 10; ++t) {                                                                      \
            if (t < 9) wa[(t + 1) & 1] = wbuf[BUF_][(t + 1) * 64 + lane];                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
            const float4 w_ = wa[t & 1];                                                                                      \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                                               \
                const float4 x_ = xs[S_][mb][t < 9 ? t : 4];                                                                  \
                if (t < 9) {                                                                                                  \
                    acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.x, x_.x, acc1[mb], 0, 0, 0);                           \
                    acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.y, x_.y, acc1[mb], 0, 0, 0);                           \
                    acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.z, x_.z, acc1[mb], 0, 0, 0);                           \
                    acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.w, x_.w, acc1[mb], 0, 0, 0);                           \
                } else {                                                                                                      \
                    accs[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.x, x_.x, accs[mb], 0, 0, 0);                           \
                    accs[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.y, x_.y, accs[mb], 0, 0, 0);                           \
                    accs[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.z, x_.z, accs[mb], 0, 0, 0);                           \
                    accs[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.w, x_.w, accs[mb], 0, 0, 0);                           \
                }                                                                                                             \
            }                                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
        }                                                                                                                     \
    } while (0)
    v4f acc1[MB], accs[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) { acc1[mb] = (v4f){0.f, 0.f, 0.f, 0.f}; accs[mb] = (v4f){0.f, 0.f, 0.f, 0.f}; }
    DAM_CS2S_XLOAD(0, 0)
    DAM_CS2S_WFETCH(0);
    DAM_CS2S_WSTORE(0);
    __syncthreads();
    for (int c = 0; c < NCH; c += 2) {                                   // NCH is even (the entry point checks)
        DAM_CS2S_XLOAD(1, c + 1)
        DAM_CS2S_WFETCH(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        DAM_CS2S_CHUNK(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        DAM_CS2S_WSTORE(1);
        __syncthreads();
        if (c + 2 < NCH) {
            DAM_CS2S_XLOAD(0, c + 2)
            DAM_CS2S_WFETCH(c + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        DAM_CS2S_CHUNK(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < NCH) { DAM_CS2S_WSTORE(0); }
        __syncthreads();
    }
#undef DAM_CS2S_WSRC
#undef DAM_CS2S_WFETCH
#undef DAM_CS2S_WSTORE
#undef DAM_CS2S_XLOAD
#undef DAM_CS2S_CHUNK
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
        if (pix[mb] < total_px) {
            *reinterpret_cast<v4f*>(Y + (size_t)pix[mb] * Co + nb * 16 + kq * 4) = acc1[mb];
            *reinterpret_cast<v4f*>(Ys + (size_t)pix[mb] * Co + nb * 16 + kq * 4) = accs[mb];
        }
    if constexpr (STATS) {
        // <= MB values per lane and channel: the wave's (n, mean, M2) by two shuffle sums over its sixteen pixel lanes (mean first,
        // then the squared deviations from it), the four waves merged by Chan through LDS
        float cnt = 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) cnt += pix[mb] < total_px ? 1.f : 0.f;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) cnt += __shfl_xor(cnt, o);
        const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
        float mean[2][4], m2[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                if (pix[mb] < total_px) { s1 += acc1[mb][q]; s2 += accs[mb][q]; }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            mean[0][q] = s1 * inv; mean[1][q] = s2 * inv;
            float d1 = 0.f, d2 = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                if (pix[mb] < total_px) {
                    const float e1 = acc1[mb][q] - mean[0][q], e2 = accs[mb][q] - mean[1][q];
                    d1 = fmaf(e1, e1, d1); d2 = fmaf(e2, e2, d2);
                }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); }
            m2[0][q] = d1; m2[1][q] = d2;
        }
        if (j == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rec[wave][0][kq * 4 + q][0] = cnt; rec[wave][0][kq * 4 + q][1] = mean[0][q]; rec[wave][0][kq * 4 + q][2] = m2[0][q];
                rec[wave][1][kq * 4 + q][0] = cnt; rec[wave][1][kq * 4 + q][1] = mean[1][q]; rec[wave][1][kq * 4 + q][2] = m2[1][q];
            }
        }
        __syncthreads();
        if (tid < 32) {
            const int which = tid >> 4, c = tid & 15;
            float n = 0.f, mu = 0.f, q2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float nw = rec[w][which][c][0], nn = n + nw, rr = nn > 0.f ? nw / nn : 0.f, d = rec[w][which][c][1] - mu;
                mu = fmaf(d, rr, mu);
                q2 += rec[w][which][c][2] + d * d * (n * rr);
                n = nn;
            }
            float* o = (which ? P2 : P1) + ((size_t)pgroup * Co + nb * 16 + c) * 3;
            o[0] = n; o[1] = mu; o[2] = q2;
        }
    }
}

int launch_conv_s2_pair_stream(const float* x, const float* wp, const float* wp2, int B, int H, int W, int Ci, int Co, float* y, float* ys,
                               float* p1, float* p2, int* parts_host, hipStream_t st) {
    const int Hd = (H + 1) / 2, Wd = (W + 1) / 2, NCH = Ci / 16, NB = Co / 16;
    const int64_t px = (int64_t)B * Hd * Wd, xb = (int64_t)B * H * W * Ci * 4;
    // one pixel block per wave; two (DAM_CS2_STREAM_MB=2: A/B) measured slower on the one layer with the units for it: 48.9 us
    // against 41.8 on the 64 -> 96 block (230 VGPRs: two waves per SIMD instead of three)
    static const int mb_forced = [] { const char* e = getenv("DAM_CS2_STREAM_MB"); return e ? atoi(e) : 0; }();
    const int mb = mb_forced == 2 ? 2 : 1;
    const int64_t groups = cdiv(px, 64 * mb);
    if (px >= (1ll << 26) || xb >= (1ll << 31) || groups * NB >= (1ll << 31) || (int64_t)9 * Ci * Co * 4 >= (1ll << 31)) return DAM_ERR_UNSUPPORTED;
    const bool stats = p1 != nullptr;
    if (stats && groups > BN_RECORDS_MAX) return DAM_ERR_UNSUPPORTED;      // (one record per group of 64 * mb pixels)
    if (parts_host) *parts_host = stats ? (int)groups : 0;
    const dim3 grid((unsigned)(cdiv(groups, 8) * 8 * NB)), block(256);        // whole rounds of eight groups (one per XCD)
#define DAM_CS2S_GO(ST_, MB_)                                                                                                   \
    hipLaunchKernelGGL((conv_s2_pair_stream_kernel<ST_, MB_>), grid, block, 0, st, x, (unsigned)xb, reinterpret_cast<const float4*>(wp), \
                       reinterpret_cast<const float4*>(wp2), Hd, Wd, H, W, NCH, NB, y, ys, p1, p2, (int)px)
    if (stats) { if (mb == 2) DAM_CS2S_GO(true, 2); else DAM_CS2S_GO(true, 1); }
    else { if (mb == 2) DAM_CS2S_GO(false, 2); else DAM_CS2S_GO(false, 1); }
#undef DAM_CS2S_GO
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

}  // namespace
}  // namespace dam

// include/dam_hip.h.  DAM_ERR_UNSUPPORTED: not a layer this kernel takes (the caller runs the separate launches).
extern "C" int dam_conv_s2_pair_fwd_f32(const float* x, const float* w_packed, const float* wsc_packed, int B, int H, int W, int Ci,
                                        int Co, float* y, float* ysc, float* partial, float* partial_sc, int* parts_host,
                                        void* stream) {
    using namespace dam;
    if (!x || !w_packed || !wsc_packed || !y || !ysc || B <= 0 || H <= 0 || W <= 0) return DAM_ERR_BAD_ARG;
    if ((partial != nullptr) != (partial_sc != nullptr)) return DAM_ERR_BAD_ARG;
    if (partial && !parts_host) return DAM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (Ci == 16 && Co == 32) return launch_conv_s2_pair<2, 1, 4>(x, w_packed, wsc_packed, B, H, W, y, ysc, partial, partial_sc, parts_host, st);
    static const int w4 = [] { const char* e = getenv("DAM_CS2_WAVES4"); return e ? atoi(e) : 0; }();          // A/B knob
    if (Ci == 32 && Co == 64 && w4) return launch_conv_s2_pair<4, 2, 4>(x, w_packed, wsc_packed, B, H, W, y, ysc, partial, partial_sc, parts_host, st);
    if (Ci == 32 && Co == 64) return launch_conv_s2_pair<4, 2, 8>(x, w_packed, wsc_packed, B, H, W, y, ysc, partial, partial_sc, parts_host, st);
    // the wide blocks: weight fragments streamed from L2 (an even chunk count: the two register sets alternate)
    static const int no_stream = getenv("DAM_CS2_NO_STREAM") ? 1 : 0;        // A/B knob
    if (!no_stream && Ci % 32 == 0 && Co % 16 == 0)
        return launch_conv_s2_pair_stream(x, w_packed, wsc_packed, B, H, W, Ci, Co, y, ysc, partial, partial_sc, parts_host, st);
    return DAM_ERR_UNSUPPORTED;
}
