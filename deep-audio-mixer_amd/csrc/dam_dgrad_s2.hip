// dam_dgrad_s2.hip -- the data gradient of a down-sampling block's INPUT in one launch (models/model_resnet.py:17-21,26 of the
// reference, reached from loss.backward() at model_trainer.py:36): x feeds conv1 (3x3, stride 2, pad 1) and the shortcut convolution
// (1x1, stride 2), so
//
//     dx[2i  , 2j  ] = W[1][1]^T dc[i,j]                                                              (+ Wsc^T ds[i,j])
//     dx[2i  , 2j+1] = W[1][0]^T dc[i,j+1] + W[1][2]^T dc[i,j]
//     dx[2i+1, 2j  ] = W[0][1]^T dc[i+1,j] + W[2][1]^T dc[i,j]
//     dx[2i+1, 2j+1] = W[0][0]^T dc[i+1,j+1] + W[0][2]^T dc[i+1,j] + W[2][0]^T dc[i,j+1] + W[2][2]^T dc[i,j]
//
// with dc = d conv1 output, ds = d shortcut output (both [B][Hd][Wd][Co]) and zero beyond their edges.  Round 3 ran this as the
// four output-parity classes of the transposed operator: three launches of the tile / row-ring kernels (each staging the same dc
// again) plus one or two single-tap launches, 0.22-0.29 of the fp32 matrix peak, and every class writing 64-byte pieces at a
// 128-byte pitch.  Here one wave owns 16 * MB consecutive pixels of dc (the flattened [B * Hd * Wd] index: rows run on into each
// other) and produces ALL FOUR classes of them -- the 2 x 2 output pixels under each input pixel -- from four shifted operand
// loads per 16-channel chunk (the shifts re-read neighbours: L1 hits), nine weight taps and 9 * 4 * MB * NB MFMAs per chunk; the two
// classes of an output row are stored back to back, so every 128-byte line of dx leaves complete.  Two kernels:
//   dgrad_s2_kernel         the thin blocks (16 <- 32, 32 <- 64 channels): persistent workgroups keep the WHOLE packed transposed
//                           weight image (9 taps + the shortcut's: 20 / 80 KB) in LDS, read from L2 once per workgroup;
//   dgrad_s2_stream_kernel  the wide blocks (image 240 KB ... 1.3 MB): one (pixel block, channel block) unit per wave, the weight
//                           fragments go straight from L2 into the MFMA operand.
// DESIGN.md section 4.2a has the measurements; profiles/r04_s2_kernels_ab.txt the steps that led here.
#include "dam_common.h"
#include "dam_bn_fin.h"
#include <cstdlib>

namespace dam {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// NB: output-channel blocks of 16 (Ci / 16), NCH: input chunks of 16 (Co / 16), MB: pixel blocks of 16 per wave unit
// WAVES: waves per workgroup (4; 8 where the weight image leaves room for one workgroup per CU only -- two waves per SIMD cover each
// other's waits: a wave's stores sit in the same in-order counter as its operand loads, every unit starts by waiting for them)
// SUMS: dx is the gradient reaching y = relu(bn(u) + shortcut) of the block in FRONT of this one (models/model_resnet.py:26-27; its
// ReLU mask as the sign bytes bn_apply left): the two per-channel sums of that BatchNorm's backward pass, sum(dz) and sum(dz * uhat)
// with dz = dx * (y > 0), uhat = (u - mean) * invstd, are taken here from the output registers and one read of u and the sign
// bytes -- instead of a pass over dx and u (bn_bwd_partial_kernel<3>: 28 / 16 us on the two thin stages).  One record per workgroup.
struct S2Sums {
    const float* u; const unsigned char* bits; const float* mean; const float* invstd; float* rec;      // rec: [workgroups][Ci][2]
};
template <int NB, int NCH, int MB, bool PAIR, int WAVES, bool SUMS>
__global__ __launch_bounds__(64 * WAVES) void dgrad_s2_kernel(const float* __restrict__ DC, const float4* __restrict__ Wp,
                                                       const float* __restrict__ DS, const float4* __restrict__ Wp2, int B, int Hd,
                                                       int Wd, float* __restrict__ DX, int H, int W, int total_px, int total_units, int xcd_aware,
                                                       const S2Sums sums) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [9 (+1)][NCH][NB][64 lanes] float4
    constexpr int Co = 16 * NCH, Ci = 16 * NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DC), 0, (unsigned)((size_t)B * Hd * Wd * Co * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PAIR ? DS : DC), 0, (unsigned)((size_t)B * Hd * Wd * Co * 4), 0x00020000);
    const int w_lane = lane * 16;
    float su1[SUMS ? NB : 1][4], su2[SUMS ? NB : 1][4];
    float4 umu[SUMS ? NB : 1], uis[SUMS ? NB : 1];
    if constexpr (SUMS) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            umu[nb] = *reinterpret_cast<const float4*>(sums.mean + nb * 16 + kq * 4);
            uis[nb] = *reinterpret_cast<const float4*>(sums.invstd + nb * 16 + kq * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) su1[nb][q] = su2[nb][q] = 0.f;
        }
    }
    static_assert(NCH % 2 == 0, "chunk c of every unit uses operand set c & 1");
    // Two operand sets: the loads of the NEXT chunk of the stream -- the unit's next chunk, or chunk 0 of the wave's next unit -- are
    // requested before the MFMAs of the current one (the first version waited a full memory round trip in front of every chunk:
    // 75 / 80 us per launch where the launches it replaced took 82 / 61).
    float4 xo[2][PAIR ? 5 : 4][MB], wa[2][NB];      // operands: 0 = dc[i][j], 1 = dc[i][j+1], 2 = dc[i+1][j], 3 = dc[i+1][j+1], 4 = ds[i][j]
    int o00[MB], o01[MB], o10[MB], o11[MB], n00[MB], n01[MB], n10[MB], n11[MB];
    // byte offsets of this lane's pixel in the four shifted views (out of the tensor: an offset the range check rejects -> 0)
#define DAM_S2_OFFSETS(U_, A_, B_, C_, D_)                                                                                    \
    do {                                                                                                                      \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                                                   \
            const int p_ = (U_) * (16 * MB) + mb * 16 + j;          /* pixel of the flattened [B * Hd * Wd] index space */    \
            const int row_ = p_ / Wd, col_ = p_ - row_ * Wd, i_ = row_ % Hd;                                                  \
            const int base = (p_ * Co + kq * 4) * 4;                                                                          \
            const bool c0 = (U_) < unit_end && p_ < total_px,    c1 = c0 && col_ + 1 < Wd, row1_ = i_ + 1 < Hd;               \
            A_[mb] = c0 ? base : 0x7fffffff;                                                                                  \
            B_[mb] = c1 ? base + Co * 4 : 0x7fffffff;                                                                         \
            C_[mb] = (c0 && row1_) ? base + Wd * Co * 4 : 0x7fffffff;                                                         \
            D_[mb] = (c1 && row1_) ? base + (Wd + 1) * Co * 4 : 0x7fffffff;                                                   \
        }                                                                                                                     \
    } while (0)
#define DAM_S2_LOAD(S_, A_, B_, C_, D_, CH_)                                                                                  \
    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                                                       \
        xo[S_][0][mb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr, A_[mb], (CH_) * 64, 0));         \
        xo[S_][1][mb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr, B_[mb], (CH_) * 64, 0));         \
        xo[S_][2][mb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr, C_[mb], (CH_) * 64, 0));         \
        xo[S_][3][mb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr, D_[mb], (CH_) * 64, 0));         \
        if constexpr (PAIR) xo[S_][4][mb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sr, A_[mb], (CH_) * 64, 0)); \
    }
    // The taps of the transposed operator in issue order: {weight tap (9 = the shortcut's), output class 2 p + q, operand}.
    // The NB weight fragments of a tap are requested ONE TAP AHEAD of its MFMAs (left to itself the compiler reads them right in
    // front of their use and every 4 * MB * NB MFMAs wait a full LDS round trip).
    constexpr int NT = PAIR ? 10 : 9;
    constexpr int TAPS[10][3] = {{4, 0, 0}, {3, 1, 1}, {5, 1, 0}, {1, 2, 2}, {7, 2, 0}, {0, 3, 3}, {2, 3, 2}, {6, 3, 1}, {8, 3, 0}, {9, 0, 4}};
#define DAM_S2_WREAD(BUF_, T_, CH_)                                                                                           \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                                         \
        wa[BUF_][nb] = *reinterpret_cast<const float4*>(smem + w_lane + (((T_) * NCH + (CH_)) * NB + nb) * 1024);
#define DAM_S2_CHUNK(S_, CH_)                                                                                                 \
    do {                                                                                                                      \
        DAM_S2_WREAD(0, TAPS[0][0], CH_)                                                                                      \
        _Pragma("unroll") for (int s = 0; s < NT; ++s) {                                                                      \
            if (s + 1 < NT) { DAM_S2_WREAD((s + 1) & 1, TAPS[s + 1 < NT ? s + 1 : 0][0], CH_) }                               \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
            /* component-major: consecutive MFMAs go to different accumulators (NB * MB independent chains) */               \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                   \
                _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                           \
                    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                                       \
                        v4f& a_ = acc[TAPS[s][1]][mb][nb];                                                                    \
                        const float4 w_ = wa[s & 1][nb], x_ = xo[S_][TAPS[s][2]][mb];                                         \
                        const float wq = q == 0 ? w_.x : q == 1 ? w_.y : q == 2 ? w_.z : w_.w;                                \
                        const float xq = q == 0 ? x_.x : q == 1 ? x_.y : q == 2 ? x_.z : x_.w;                                \
                        a_ = __builtin_amdgcn_mfma_f32_16x16x4f32(wq, xq, a_, 0, 0, 0);                                       \
                    }                                                                                                         \
                }                                                                                                             \
            }                                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
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
    DAM_S2_OFFSETS(unit, o00, o01, o10, o11);
    DAM_S2_LOAD(0, o00, o01, o10, o11, 0)
    // (the weight image goes to LDS BEHIND the first unit's operand requests: their round trips overlap)
    {   // the weight image: one pass, all loads of a thread in flight
        constexpr int N4 = 9 * NCH * NB * 64, N4P = PAIR ? NCH * NB * 64 : 0;
        for (int e = tid; e < N4; e += 64 * WAVES) reinterpret_cast<float4*>(smem)[e] = Wp[e];
        if constexpr (PAIR)
            for (int e = tid; e < N4P; e += 64 * WAVES) reinterpret_cast<float4*>(smem)[N4 + e] = Wp2[e];
    }
    __syncthreads();
    while (unit < unit_end) {
        v4f acc[4][MB][NB];                 // class (p, q) = 2 p + q
#pragma unroll
        for (int cl = 0; cl < 4; ++cl)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[cl][mb][nb] = (v4f){0.f, 0.f, 0.f, 0.f};
        DAM_S2_OFFSETS(unit + ustride, n00, n01, n10, n11);      // (no next unit: every offset out of range, the loads read 0)
        float4 uv[SUMS ? MB : 1][4][NB];
        unsigned ub[SUMS ? MB : 1][4][NB];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) { DAM_S2_LOAD((c + 1) & 1, o00, o01, o10, o11, c + 1) }
            else { DAM_S2_LOAD(0, n00, n01, n10, n11, 0) }
            if constexpr (SUMS) {
                if (c == NCH - 1) {
                    // the write-out's u quads and sign bytes, requested under the unit's last chunk of MFMAs (asked for at the
                    // write-out itself they cost the launch 8-12 us of exposed round trips)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const int p = unit * (16 * MB) + mb * 16 + j;
                        const int row = p / Wd, col = p - row * Wd, img = row / Hd, i = row - img * Hd;
                        const bool live = p < total_px, orow1 = 2 * i + 1 < H, ocol1 = 2 * col + 1 < W;
                        const size_t px00 = ((size_t)img * H + 2 * i) * W + 2 * col;
#pragma unroll
                        for (int cl = 0; cl < 4; ++cl) {
                            const bool ok = live && (cl & 1 ? ocol1 : true) && (cl & 2 ? orow1 : true);
                            const size_t px = px00 + (cl & 1 ? 1 : 0) + (cl & 2 ? (size_t)W : 0);
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) {
                                uv[mb][cl][nb] = ok ? *reinterpret_cast<const float4*>(sums.u + px * Ci + nb * 16 + kq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                                ub[mb][cl][nb] = ok ? sums.bits[px * (Ci / 4) + nb * 4 + kq] : 0u;
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            DAM_S2_CHUNK(c & 1, c);
            __builtin_amdgcn_sched_barrier(0);
        }
        // write-out: lane holds channels 4 kq .. +3 of block nb of its pixel; the two classes of an output row go out back to back
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int p = unit * (16 * MB) + mb * 16 + j;
            if (p >= total_px) continue;
            const int row = p / Wd, col = p - row * Wd, img = row / Hd, i = row - img * Hd;
            const bool orow1 = 2 * i + 1 < H, ocol1 = 2 * col + 1 < W;
            const size_t px00 = ((size_t)img * H + 2 * i) * W + 2 * col;            // output pixel of class (0, 0)
            const size_t p00 = px00 * Ci + kq * 4;
            if constexpr (SUMS) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float* o = DX + p00 + nb * 16;
                    *reinterpret_cast<v4f*>(o) = acc[0][mb][nb];
                    if (ocol1) *reinterpret_cast<v4f*>(o + Ci) = acc[1][mb][nb];
                    if (orow1) {
                        *reinterpret_cast<v4f*>(o + (size_t)W * Ci) = acc[2][mb][nb];
                        if (ocol1) *reinterpret_cast<v4f*>(o + (size_t)W * Ci + Ci) = acc[3][mb][nb];
                    }
                }
#pragma unroll
                for (int cl = 0; cl < 4; ++cl)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const float uq[4] = {uv[mb][cl][nb].x, uv[mb][cl][nb].y, uv[mb][cl][nb].z, uv[mb][cl][nb].w};
                        const float mq[4] = {umu[nb].x, umu[nb].y, umu[nb].z, umu[nb].w}, iq[4] = {uis[nb].x, uis[nb].y, uis[nb].z, uis[nb].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float dz = (ub[mb][cl][nb] >> q) & 1u ? acc[cl][mb][nb][q] : 0.f;       // (a class outside the image: bits 0)
                            su1[nb][q] += dz;
                            su2[nb][q] = fmaf(dz, (uq[q] - mq[q]) * iq[q], su2[nb][q]);
                        }
                    }
            } else {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float* o = DX + p00 + nb * 16;
                    *reinterpret_cast<v4f*>(o) = acc[0][mb][nb];
                    if (ocol1) *reinterpret_cast<v4f*>(o + Ci) = acc[1][mb][nb];
                    if (orow1) {
                        *reinterpret_cast<v4f*>(o + (size_t)W * Ci) = acc[2][mb][nb];
                        if (ocol1) *reinterpret_cast<v4f*>(o + (size_t)W * Ci + Ci) = acc[3][mb][nb];
                    }
                }
            }
        }
        unit += ustride;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) { o00[mb] = n00[mb]; o01[mb] = n01[mb]; o10[mb] = n10[mb]; o11[mb] = n11[mb]; }
    }
    if constexpr (SUMS) {
        // lane sums -> the sixteen pixel lanes of a channel quad (fixed order) -> the workgroup's waves through LDS: one record
#pragma unroll
        for (int off = 1; off < 16; off <<= 1)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) { su1[nb][q] += __shfl_xor(su1[nb][q], off); su2[nb][q] += __shfl_xor(su2[nb][q], off); }
        __syncthreads();                                               // every wave is done with the weights
        float* red = reinterpret_cast<float*>(smem);                 // [WAVES][Ci][2]
        if (j == 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    red[(wave * Ci + nb * 16 + kq * 4 + q) * 2] = su1[nb][q];
                    red[(wave * Ci + nb * 16 + kq * 4 + q) * 2 + 1] = su2[nb][q];
                }
        }
        __syncthreads();
        if (tid < 2 * Ci) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += red[w * Ci * 2 + tid];
            sums.rec[(size_t)blockIdx.x * Ci * 2 + tid] = t;
        }
    }
#undef DAM_S2_OFFSETS
#undef DAM_S2_LOAD
#undef DAM_S2_WREAD
#undef DAM_S2_CHUNK
}

template <int NB, int NCH, int MB, int WAVES>
int launch_dgrad_s2(const float* dc, const float* wpt, const float* ds, const float* wpt2, int B, int Hd, int Wd, float* dx, int H,
                    int W, const S2Sums& sums, int* parts_host, hipStream_t st) {
    // Units are 16 * MB pixels of the FLATTENED [B * Hd * Wd] index space (a unit per row segment left the last segment of every row
    // nearly empty: Wd = 33 -> half the MFMAs on padding), each lane finds its row / column by division.
    const int64_t px = (int64_t)B * Hd * Wd;
    if (px >= (1ll << 30)) return DAM_ERR_UNSUPPORTED;
    const int64_t units = cdiv(px, 16 * MB);
    const size_t lds = (size_t)(9 + (ds ? 1 : 0)) * NCH * NB * 1024;
    const int cus = device_cus();
    // resident workgroups per CU of THIS instantiation (registers and LDS; asked once): a grid beyond it would run a second, partial round
    static PerDevice<int> occ[2][2];
    int& oc = occ[ds ? 1 : 0][sums.u ? 1 : 0]();
    if (!oc) {
        int n = 0;
        hipError_t e;
        if (ds) e = sums.u ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dgrad_s2_kernel<NB, NCH, MB, true, WAVES, true>, 64 * WAVES, lds)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dgrad_s2_kernel<NB, NCH, MB, true, WAVES, false>, 64 * WAVES, lds);
        else e = sums.u ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dgrad_s2_kernel<NB, NCH, MB, false, WAVES, true>, 64 * WAVES, lds)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dgrad_s2_kernel<NB, NCH, MB, false, WAVES, false>, 64 * WAVES, lds);
        oc = (e == hipSuccess && n >= 1) ? n : 1;
    }
    int max_per_cu = oc;
    if (max_per_cu > 3) max_per_cu = 3;
    if (WAVES == 8) max_per_cu = 1;
    static const int xcd_aware = getenv("DAM_S2_NO_XCD") ? 0 : 1;      // A/B knob
    static const int forced = [] { const char* e = getenv("DAM_S2_PER_CU"); return e ? atoi(e) : 0; }();      // A/B knob
    // Workgroups per CU by makespan: n resident waves per SIMD share its MFMA pipe, so a SIMD's time is (units per wave) * n unit
    // times; more waves hide the operand latency better, which decides when the costs are within ~15 %.
    int per_cu = 1;
    int64_t best = 0;
    for (int n = 1; n <= max_per_cu; ++n) {
        const int64_t cost = cdiv(units, (int64_t)WAVES * cus * n) * n * 100;
        if (n == 1 || cost * 100 <= best * 115) { best = n == 1 ? cost : (cost < best ? cost : best); per_cu = n; }
    }
    if (forced >= 1 && forced <= max_per_cu) per_cu = forced;
    int64_t wgs = (int64_t)cus * per_cu;
    if (wgs > cdiv(units, WAVES)) wgs = cdiv(units, WAVES);
    if (sums.u && wgs > BN_BWD_RECORDS_MAX) return DAM_ERR_UNSUPPORTED;
#define DAM_S2_GO(PAIR_, SUMS_)                                                                                                   \
    do {                                                                                                                          \
        static PerDevice<bool> raised_pd; bool& raised = raised_pd();\
        if (!raised && lds > 64 * 1024) {                                                                                         \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_s2_kernel<NB, NCH, MB, PAIR_, WAVES, SUMS_>),              \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DAM_ERR_LAUNCH;     \
            raised = true;                                                                                                        \
        }                                                                                                                         \
        hipLaunchKernelGGL((dgrad_s2_kernel<NB, NCH, MB, PAIR_, WAVES, SUMS_>), dim3((unsigned)wgs), dim3(64 * WAVES), lds, st, dc, \
                           reinterpret_cast<const float4*>(wpt), ds, reinterpret_cast<const float4*>(wpt2), B, Hd, Wd, dx, H, W,      \
                           (int)px, (int)units, xcd_aware, sums);                                                                 \
    } while (0)
    if (ds) { if (sums.u) DAM_S2_GO(true, true); else DAM_S2_GO(true, false); }
    else { if (sums.u) DAM_S2_GO(false, true); else DAM_S2_GO(false, false); }
#undef DAM_S2_GO
    if (parts_host) *parts_host = sums.u ? (int)wgs : 0;
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}


// ---- the wide down-sampling blocks (96 <- 64 ... 256 <- 128 channels): the packed weight image (240 KB ... 1.3 MB) does not fit a
// workgroup's LDS and the tensors are small (<= 17 k pixels of dc), so a unit is one (16 * MB pixels, one 16-channel block of dx) pair
// and a wave takes exactly one: its ten weight fragments per 16-channel chunk of dc come straight from L2 into the MFMA's A operand
// (64 lanes x 16 bytes = one packed 1 KB block per load; every wave with the same channel block reads the same image), the operand
// loads are the four shifted views of the kernel above, and two register sets keep the next chunk's fifteen loads in flight under the
// current chunk's 40 * MB MFMAs.  No LDS, no barrier: several workgroups per CU cover each other's waits.
template <int MB, bool PAIR>
__global__ __launch_bounds__(256) void dgrad_s2_stream_kernel(const float* __restrict__ DC, const float4* __restrict__ Wp,
                                                              const float* __restrict__ DS, const float4* __restrict__ Wp2, int Hd, int Wd,
                                                              float* __restrict__ DX, int H, int W, int total_px, int NCH, int NB,
                                                              int total_units) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int unit = blockIdx.x * 4 + wave;
    if (unit >= total_units) return;                                   // (no barrier below)
    const int nb = unit % NB, pg = unit / NB;                         // the waves of a workgroup mostly share the pixels: L1 hits
    const int j = lane & 15, kq = lane >> 4;
    const int Co = 16 * NCH, Ci = 16 * NB;
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DC), 0, (unsigned)((size_t)total_px * Co * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PAIR ? DS : DC), 0, (unsigned)((size_t)total_px * Co * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(Wp), 0, (unsigned)((size_t)9 * NCH * NB * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(PAIR ? Wp2 : Wp), 0, (unsigned)((size_t)NCH * NB * 1024), 0x00020000);
    int o00[MB], o01[MB], o10[MB], o11[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int p = pg * (16 * MB) + mb * 16 + j;
        const int row = p / Wd, col = p - row * Wd, i = row % Hd;
        const int base = (p * Co + kq * 4) * 4;
        const bool c0 = p < total_px, c1 = c0 && col + 1 < Wd, row1 = i + 1 < Hd;
        o00[mb] = c0 ? base : 0x7fffffff;
        o01[mb] = c1 ? base + Co * 4 : 0x7fffffff;
        o10[mb] = (c0 && row1) ? base + Wd * Co * 4 : 0x7fffffff;
        o11[mb] = (c1 && row1) ? base + (Wd + 1) * Co * 4 : 0x7fffffff;
    }
    float4 x00[2][MB], x01[2][MB], x10[2][MB], x11[2][MB], s00[2][PAIR ? MB : 1], wt[2][PAIR ? 10 : 9];
    // (the chunk rides in the scalar offset, which the range check ignores: the caller of this macro keeps CH_ < NCH)
#define DAM_S2S_LOAD(S_, CH_)                                                                                                 \
    do {                                                                                                                      \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                                                   \
            x00[S_][mb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr, o00[mb], (CH_) * 64, 0));      \
            x01[S_][mb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr, o01[mb], (CH_) * 64, 0));      \
            x10[S_][mb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr, o10[mb], (CH_) * 64, 0));      \
            x11[S_][mb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr, o11[mb], (CH_) * 64, 0));      \
            if constexpr (PAIR) s00[S_][mb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sr, o00[mb], (CH_) * 64, 0)); \
        }                                                                                                                     \
        _Pragma("unroll") for (int t = 0; t < 9; ++t)                                                                         \
            wt[S_][t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, ((t * NCH + (CH_)) * NB + nb) * 1024, 0)); \
        if constexpr (PAIR)                                                                                                   \
            wt[S_][9] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w2r, lane * 16, ((CH_) * NB + nb) * 1024, 0)); \
    } while (0)
#define DAM_S2S_TAP(T_, CL_, X_, S_)                                                                                          \
    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                                                       \
        acc[CL_][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[S_][T_].x, X_[S_][mb].x, acc[CL_][mb], 0, 0, 0);               \
        acc[CL_][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[S_][T_].y, X_[S_][mb].y, acc[CL_][mb], 0, 0, 0);               \
        acc[CL_][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[S_][T_].z, X_[S_][mb].z, acc[CL_][mb], 0, 0, 0);               \
        acc[CL_][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[S_][T_].w, X_[S_][mb].w, acc[CL_][mb], 0, 0, 0);               \
    }
#define DAM_S2S_CHUNK(S_)                                                                                                     \
    do {                                                                                                                      \
        DAM_S2S_TAP(4, 0, x00, S_)                                                                                            \
        if constexpr (PAIR) { DAM_S2S_TAP(9, 0, s00, S_) }                                                                    \
        DAM_S2S_TAP(3, 1, x01, S_) DAM_S2S_TAP(5, 1, x00, S_)                                                                 \
        DAM_S2S_TAP(1, 2, x10, S_) DAM_S2S_TAP(7, 2, x00, S_)                                                                 \
        DAM_S2S_TAP(0, 3, x11, S_) DAM_S2S_TAP(2, 3, x10, S_) DAM_S2S_TAP(6, 3, x01, S_) DAM_S2S_TAP(8, 3, x00, S_)           \
    } while (0)
    v4f acc[4][MB];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[cl][mb] = (v4f){0.f, 0.f, 0.f, 0.f};
    DAM_S2S_LOAD(0, 0);
    for (int c = 0; c < NCH; c += 2) {                                 // NCH is even (the entry point checks)
        DAM_S2S_LOAD(1, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        DAM_S2S_CHUNK(0);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < NCH) DAM_S2S_LOAD(0, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        DAM_S2S_CHUNK(1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int p = pg * (16 * MB) + mb * 16 + j;
        if (p >= total_px) continue;
        const int row = p / Wd, col = p - row * Wd, img = row / Hd, i = row - img * Hd;
        const bool orow1 = 2 * i + 1 < H, ocol1 = 2 * col + 1 < W;
        float* o = DX + (((size_t)img * H + 2 * i) * W + 2 * col) * Ci + nb * 16 + kq * 4;
        *reinterpret_cast<v4f*>(o) = acc[0][mb];
        if (ocol1) *reinterpret_cast<v4f*>(o + Ci) = acc[1][mb];
        if (orow1) {
            *reinterpret_cast<v4f*>(o + (size_t)W * Ci) = acc[2][mb];
            if (ocol1) *reinterpret_cast<v4f*>(o + (size_t)W * Ci + Ci) = acc[3][mb];
        }
    }
#undef DAM_S2S_LOAD
#undef DAM_S2S_TAP
#undef DAM_S2S_CHUNK
}

// The same with a chunk's weight fragments fetched ONCE per workgroup into LDS: the four waves of a workgroup take four consecutive
// pixel blocks of the SAME channel block of dx, every wave fetches a quarter of the chunk's 9 (+ 1) x 64 quads, two chunk buffers, one
// barrier per chunk (each wave loading all ten itself kept a CU's texture path ~75 % busy).  One pixel block per wave.
template <bool PAIR>
__global__ __launch_bounds__(256) void dgrad_s2_stream_lds_kernel(const float* __restrict__ DC, const float4* __restrict__ Wp,
                                                                  const float* __restrict__ DS, const float4* __restrict__ Wp2, int Hd,
                                                                  int Wd, float* __restrict__ DX, int H, int W, int total_px, int NCH,
                                                                  int NB) {
    constexpr int NT = PAIR ? 10 : 9, NQ = NT * 64;
    __shared__ float4 wbuf[2][NQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware order (as in dam_conv_s2.hip's streaming kernel): the NB workgroups of a 64-pixel group read the same dc / ds, so a
    // group's workgroups stay on ONE XCD (group g on XCD g % 8; the grid is padded to whole rounds of eight groups)
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int nb = seq % NB, grp = (seq / NB) * 8 + xcd;
    if (grp * 64 >= total_px) return;                                  // (whole workgroup: no barrier is left waiting)
    const int pg = grp * 4 + wave;
    const int j = lane & 15, kq = lane >> 4;
    const int Co = 16 * NCH, Ci = 16 * NB;
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DC), 0, (unsigned)((size_t)total_px * Co * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PAIR ? DS : DC), 0, (unsigned)((size_t)total_px * Co * 4), 0x00020000);
    const int p = pg * 16 + j;
    const int row = p / Wd, col = p - row * Wd, img = row / Hd, i = row - img * Hd;
    int ov[4];
    {
        const int base = (p * Co + kq * 4) * 4;
        const bool c0 = p < total_px, c1 = c0 && col + 1 < Wd, row1 = i + 1 < Hd;
        ov[0] = c0 ? base : 0x7fffffff;
        ov[1] = c1 ? base + Co * 4 : 0x7fffffff;
        ov[2] = (c0 && row1) ? base + Wd * Co * 4 : 0x7fffffff;
        ov[3] = (c1 && row1) ? base + (Wd + 1) * Co * 4 : 0x7fffffff;
    }
    float4 xo[2][PAIR ? 5 : 4], wa[2], wr0, wr1, wr2;       // operands: 0 = dc[i][j], 1 = dc[i][j+1], 2 = dc[i+1][j], 3 = dc[i+1][j+1], 4 = ds[i][j]
    const int we0 = tid, we1 = tid + 256, we2 = tid + 512 < NQ ? tid + 512 : 0;        // this thread's quads of a chunk
#define DAM_S2L_WSRC(E_, CH_)                                                                                                \
    ((E_) < 576 ? Wp + ((size_t)(((E_) >> 6) * NCH + (CH_)) * NB + nb) * 64 + ((E_) & 63)                                     \
                : Wp2 + ((size_t)(CH_) * NB + nb) * 64 + ((E_) & 63))
#define DAM_S2L_WFETCH(CH_)                                                                                                   \
    do {                                                                                                                      \
        wr0 = *DAM_S2L_WSRC(we0, CH_); wr1 = *DAM_S2L_WSRC(we1, CH_); wr2 = *DAM_S2L_WSRC(we2, CH_);                          \
    } while (0)
#define DAM_S2L_WSTORE(BUF_)                                                                                                  \
    do {                                                                                                                      \
        wbuf[BUF_][we0] = wr0; wbuf[BUF_][we1] = wr1;                                                                         \
        if (tid + 512 < NQ) wbuf[BUF_][we2] = wr2;                                                                            \
    } while (0)
    // (the chunk rides in the scalar offset, which the range check ignores: the callers keep CH_ < NCH)
#define DAM_S2L_XLOAD(S_, CH_)                                                                                                \
    do {                                                                                                                      \
        _Pragma("unroll") for (int v = 0; v < 4; ++v)                                                                         \
            xo[S_][v] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cr, ov[v], (CH_) * 64, 0));          \
        if constexpr (PAIR) xo[S_][4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sr, ov[0], (CH_) * 64, 0)); \
    } while (0)
    // taps in issue order: {weight tap (9 = the shortcut's), output class 2 p + q, operand}; fragments read one tap ahead
    constexpr int TAPS[10][3] = {{4, 0, 0}, {3, 1, 1}, {5, 1, 0}, {1, 2, 2}, {7, 2, 0}, {0, 3, 3}, {2, 3, 2}, {6, 3, 1}, {8, 3, 0}, {9, 0, 4}};
#define DAM_S2L_CHUNK(S_, BUF_)                                                                                               \
    do {                                                                                                                      \
        wa[0] = wbuf[BUF_][TAPS[0][0] * 64 + lane];                                                                           \
        _Pragma("unroll") for (int s = 0; s < NT; ++s) {                                                                      \
            if (s + 1 < NT) wa[(s + 1) & 1] = wbuf[BUF_][TAPS[s + 1 < NT ? s + 1 : 0][0] * 64 + lane];                        \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
            const float4 w_ = wa[s & 1], x_ = xo[S_][TAPS[s][2]];                                                             \
            v4f& a_ = acc[TAPS[s][1]];                                                                                        \
            a_ = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.x, x_.x, a_, 0, 0, 0);                                               \
            a_ = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.y, x_.y, a_, 0, 0, 0);                                               \
            a_ = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.z, x_.z, a_, 0, 0, 0);                                               \
            a_ = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.w, x_.w, a_, 0, 0, 0);                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
        }                                                                                                                     \
    } while (0)
    v4f acc[4];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl) acc[cl] = (v4f){0.f, 0.f, 0.f, 0.f};
    DAM_S2L_XLOAD(0, 0);
    DAM_S2L_WFETCH(0);
    DAM_S2L_WSTORE(0);
    __syncthreads();
    for (int c = 0; c < NCH; c += 2) {                                 // NCH is even (the entry point checks)
        DAM_S2L_XLOAD(1, c + 1);
        DAM_S2L_WFETCH(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        DAM_S2L_CHUNK(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        DAM_S2L_WSTORE(1);
        __syncthreads();
        if (c + 2 < NCH) {
            DAM_S2L_XLOAD(0, c + 2);
            DAM_S2L_WFETCH(c + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        DAM_S2L_CHUNK(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < NCH) { DAM_S2L_WSTORE(0); }
        __syncthreads();
    }
#undef DAM_S2L_WSRC
#undef DAM_S2L_WFETCH
#undef DAM_S2L_WSTORE
#undef DAM_S2L_XLOAD
#undef DAM_S2L_CHUNK
    if (p < total_px) {
        const bool orow1 = 2 * i + 1 < H, ocol1 = 2 * col + 1 < W;
        float* o = DX + (((size_t)img * H + 2 * i) * W + 2 * col) * Ci + nb * 16 + kq * 4;
        *reinterpret_cast<v4f*>(o) = acc[0];
        if (ocol1) *reinterpret_cast<v4f*>(o + Ci) = acc[1];
        if (orow1) {
            *reinterpret_cast<v4f*>(o + (size_t)W * Ci) = acc[2];
            if (ocol1) *reinterpret_cast<v4f*>(o + (size_t)W * Ci + Ci) = acc[3];
        }
    }
}

template <int MB>
int launch_dgrad_s2_stream(const float* dc, const float* wpt, const float* ds, const float* wpt2, int B, int Hd, int Wd, int Co, int Ci,
                           float* dx, int H, int W, hipStream_t st) {
    const int64_t px = (int64_t)B * Hd * Wd;
    const int NCH = Co / 16, NB = Ci / 16;
    const int64_t units = cdiv(px, 16 * MB) * NB;
    if (px >= (1ll << 26) || units >= (1ll << 30)) return DAM_ERR_UNSUPPORTED;
    static const int no_lds = getenv("DAM_S2_STREAM_NO_LDS") ? 1 : 0;       // A/B knob: every wave loads all its weight fragments
    if (MB == 1 && !no_lds) {
        const dim3 grid_l((unsigned)(cdiv(cdiv(px, 64), 8) * 8 * NB)), block_l(256);      // whole rounds of eight groups (one per XCD)
        if (ds)
            hipLaunchKernelGGL((dgrad_s2_stream_lds_kernel<true>), grid_l, block_l, 0, st, dc, reinterpret_cast<const float4*>(wpt), ds,
                               reinterpret_cast<const float4*>(wpt2), Hd, Wd, dx, H, W, (int)px, NCH, NB);
        else
            hipLaunchKernelGGL((dgrad_s2_stream_lds_kernel<false>), grid_l, block_l, 0, st, dc, reinterpret_cast<const float4*>(wpt),
                               (const float*)nullptr, (const float4*)nullptr, Hd, Wd, dx, H, W, (int)px, NCH, NB);
        DAM_CHECK_LAUNCH();
        return DAM_OK;
    }
    const dim3 grid((unsigned)cdiv(units, 4)), block(256);
    if (ds)
        hipLaunchKernelGGL((dgrad_s2_stream_kernel<MB, true>), grid, block, 0, st, dc, reinterpret_cast<const float4*>(wpt), ds,
                           reinterpret_cast<const float4*>(wpt2), Hd, Wd, dx, H, W, (int)px, NCH, NB, (int)units);
    else
        hipLaunchKernelGGL((dgrad_s2_stream_kernel<MB, false>), grid, block, 0, st, dc, reinterpret_cast<const float4*>(wpt),
                           (const float*)nullptr, (const float4*)nullptr, Hd, Wd, dx, H, W, (int)px, NCH, NB, (int)units);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

}  // namespace
}  // namespace dam

// include/dam_hip.h.  DAM_ERR_UNSUPPORTED: the layer is not one this kernel takes (the caller runs the parity classes).
extern "C" int dam_dgrad_s2_3x3_f32(const float* dy, const float* w_packed_t, const float* dy_pair, const float* w_pair_packed_t,
                                    int B, int Hd, int Wd, int Co, int Ci, float* dx, int H, int W, const dam_bn_bwd_sums* bn_bwd,
                                    float* bn_partial, int* bn_parts_host, void* stream) {
    using namespace dam;
    if (!dy || !w_packed_t || !dx || B <= 0 || Hd <= 0 || Wd <= 0 || H <= 0 || W <= 0) return DAM_ERR_BAD_ARG;
    if ((dy_pair != nullptr) != (w_pair_packed_t != nullptr)) return DAM_ERR_BAD_ARG;
    if (Hd != (H + 1) / 2 || Wd != (W + 1) / 2) return DAM_ERR_BAD_ARG;             // 3x3 / stride 2 / pad 1 geometry
    if ((int64_t)B * Hd * Wd * Co * 4 >= (1ll << 31)) return DAM_ERR_UNSUPPORTED;   // byte offsets of the range-checked loads
    hipStream_t st = (hipStream_t)stream;
    if (bn_parts_host) *bn_parts_host = 0;
    // the upstream BatchNorm's backward sums (mask as sign bytes only): taken by the persistent kernels, "not produced" elsewhere
    S2Sums sums{nullptr, nullptr, nullptr, nullptr, nullptr};
    static const int no_sums = getenv("DAM_S2_NO_SUMS") ? 1 : 0;          // A/B knob
    if (bn_bwd && bn_bwd->x && bn_bwd->mask_bits && bn_bwd->mean && bn_bwd->invstd && bn_partial && bn_parts_host && !no_sums)
        sums = S2Sums{bn_bwd->x, bn_bwd->mask_bits, bn_bwd->mean, bn_bwd->invstd, bn_partial};
    static const int mb1 = [] { const char* e = getenv("DAM_S2_MB1"); return e ? atoi(e) : 0; }();           // A/B knob
    if (Co == 32 && Ci == 16 && !(mb1 & 1)) return launch_dgrad_s2<1, 2, 2, 4>(dy, w_packed_t, dy_pair, w_pair_packed_t, B, Hd, Wd, dx, H, W, sums, bn_parts_host, st);
    if (Co == 32 && Ci == 16 && (mb1 & 1)) return launch_dgrad_s2<1, 2, 1, 4>(dy, w_packed_t, dy_pair, w_pair_packed_t, B, Hd, Wd, dx, H, W, sums, bn_parts_host, st);
    if (Co == 64 && Ci == 32) return launch_dgrad_s2<2, 4, 1, 8>(dy, w_packed_t, dy_pair, w_pair_packed_t, B, Hd, Wd, dx, H, W, sums, bn_parts_host, st);
    // wider layers: the weight image streams from L2 (even chunk count: the two register sets alternate)
    static const int no_stream = [] { const char* e = getenv("DAM_S2_NO_STREAM"); return e ? atoi(e) : 0; }();       // A/B knob
    if (!no_stream && Co % 32 == 0 && Ci % 16 == 0 && (int64_t)9 * Co * Ci * 4 < (1ll << 31)) {
        const int64_t units1 = cdiv((int64_t)B * Hd * Wd, 16) * (Ci / 16);
        static const int mb_forced = [] { const char* e = getenv("DAM_S2_STREAM_MB"); return e ? atoi(e) : 0; }();   // A/B knob
        const bool two = mb_forced ? mb_forced == 2 : units1 >= 8 * 1024;          // two pixel blocks per wave once the chip is full
        return two ? launch_dgrad_s2_stream<2>(dy, w_packed_t, dy_pair, w_pair_packed_t, B, Hd, Wd, Co, Ci, dx, H, W, st)
                   : launch_dgrad_s2_stream<1>(dy, w_packed_t, dy_pair, w_pair_packed_t, B, Hd, Wd, Co, Ci, dx, H, W, st);
    }
    return DAM_ERR_UNSUPPORTED;
}
