// dam_conv_pipe.hip -- the tile convolution (dam_conv.hip: same GEMM mapping, packed weights, tap grid, epilogue) with the
// input staging taken OFF the compute waves, for the thick 3x3 layers (ResNet layer3-6 forward and data gradient: 64-256
// channels on 33- to 5-pixel-wide rows).
//
// What the measurements on conv_igemm_kernel and on the first version of this kernel showed (profiles/r02_pmc_thick_layers.csv,
// tools/pipe_stamps_probe.py, tools/mfma_peak.hip):
//   * v_mfma_f32_16x16x4_f32 sustains 32 clocks per instruction per SIMD at 2.37 GHz on the whole chip (155 TFLOP/s), dependent
//     accumulator chains included -- the matrix pipe itself is not the limit;
//   * a workgroup that stages its whole patch before its first MFMA idles the pipe for 3-4 us, and the next workgroup the
//     dispatcher starts in its place does so again: with ~4 tiles per CU and two resident workgroups that was a third of the launch;
//   * a wave issues one instruction per ~4 clocks, so the ~40 scalar instructions and two branches of per-item bookkeeping
//     (which packed-weight block is next?) cost the pipe ~7 clocks per MFMA.
// Here:
//   * a workgroup = 4 compute waves + 4 LOADER waves (one per SIMD), and it is PERSISTENT: it walks work units (tile x
//     output-block group) of its XCD's contiguous eighth of the units, side by side with the XCD's other workgroups (round 4; before:
//     blockIdx.x, blockIdx.x + gridDim.x, ...).  The stream of (unit, 16-channel chunk) stages flows through
//     two alternating LDS buffers without a break at unit boundaries: while the compute waves run the last chunk of a unit and
//     write its tile out, the loaders already stage the first chunks of the next one.  One raw barrier per chunk;
//   * a loader thread's items (LDS offset, byte offset in the image) are computed once per launch; everything that changes from
//     unit to unit is scalar, and the padding comes from the range check of the buffer loads.  A chunk costs a loader thread
//     <= 5 x (add, load, LDS write);
//   * the item grid is static: item (chunk, tap T) requests the packed weights of item T+2 (of the next chunk, or of the next
//     unit's first chunk, behind the chunk boundary) at `tap constant + chunk scalar` -- two scalar adds, no branch -- and the
//     weight pipeline (L2 -> registers, two items ahead, three rotating operand sets) runs across chunk and unit boundaries.
#include <climits>
#include <cstdlib>
#include "dam_common.h"
#include "dam_conv_geo.h"
#include "dam_conv_stage.h"
#include "dam_bn_fin.h"

namespace dam {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int PIPE_LT = 256;                    // loader threads: 4 waves, one per SIMD beside a compute wave (a wave streaming MFMAs
                                                // leaves its SIMD's other waves few issue slots: two loader waves on two SIMDs held
                                                // back the two compute waves they shared with, and the barrier made all four wait)
constexpr int PIPE_THREADS = 256 + PIPE_LT;
// float4 items per loader thread and chunk = template parameter PU: 3 (the K-split form of the small stages: every item a thread does
// not have still costs its commit a slot of the SIMD it shares with a compute wave), 5 (a chunk of the patch <= 1280 items: every stride-1 stage,
// also 64 pixels of 54-pixel rows at the reference's 216 frames) or 8 (<= 2048 items: the strided 3x3 convolutions at the head of
// a down-sampling block, whose patch covers twice the rows and columns)
constexpr int PIPE_BUF1 = 32768;                // LDS byte offset of the second chunk buffer: a compile-time constant, so the
                                                // loaders' LDS writes carry it as an immediate

// Raw barrier: the wave's LDS traffic is drained, its global loads are NOT (__syncthreads() would also wait for the patch /
// weight loads that were just put in flight on purpose).
#define DAM_PIPE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// timing experiments (tools/build_variant.sh): results are wrong with any of these defined
#ifdef DAM_PIPE_DIAG_W0
#define DAM_PIPE_WOFF(x) ((x) & 0)              // every weight load hits the same 4 KB
#else
#define DAM_PIPE_WOFF(x) (x)
#endif

// Diagnostic build only (-DDAM_PIPE_STAMPS): the caller's workspace (`stamps`, otherwise unused here) receives s_memtime stamps of phase
// boundaries, [workgroup][role: wave 0 / first loader wave][64] of (tag << 56 | time); read by tools/pipe_stamps_probe.py.
#ifdef DAM_PIPE_STAMPS
#define DAM_PSTAMP(role, tag)                                                                                         \
    do {                                                                                                              \
        if (lane == 0 && (wave == 0 || wave == 4) && stamp_n < 64) {                                                  \
            const unsigned long long t_ = __builtin_readcyclecounter();                                               \
            stamp_p[(role) * 64 + stamp_n++] = ((unsigned long long)(tag) << 56) | (t_ & ((1ull << 56) - 1));         \
        }                                                                                                             \
    } while (0)
#else
#define DAM_PSTAMP(role, tag) do { } while (0)
#endif

struct PipeUnit {            // one work unit, decoded (wave uniform)
    int p0, img, nb0, oh_first;
};

template <int UPX, int NB>                    // UPX: output pixels of a unit
__device__ __forceinline__ PipeUnit pipe_decode(const ConvGeo& g, int unit, int nby, float inv_wo) {
    PipeUnit u;
    const int r = unit / g.tiles_m, tm = unit - r * g.tiles_m;
    u.img = r / nby;
    u.nb0 = (r - u.img * nby) * NB;
    u.p0 = tm * UPX;
    u.oh_first = fast_div(u.p0, g.Wo, inv_wo);
    return u;
}

// The second launch bound is the register budget (waves per SIMD): without it the compiler hoists every loop-invariant address
// of the loader's 16 segments into registers (198 VGPRs, one workgroup per CU).
// STATS: the launch also emits BatchNorm partial statistics (a separate instantiation: the epilogue code costs every launch
// 1-2 us through register allocation even when it does not run).
// KS (1 or 2): K split over the compute waves.  The matrix pipe's atom is one wave's 16 x 16 block over the whole K; the 33 x 5 /
// 65 x 9 stages have 1408 / 2368 of them for 1024 SIMDs, and with 64-pixel units the CUs that get two of the 384 units set the
// launch.  KS = 2: a unit is 32 * MB pixels, a STAGE of the stream is TWO 16-channel chunks, waves 0 / 1 take the two pixel blocks
// on the first chunk of every stage and waves 2 / 3 the same blocks on the second; waves 2 / 3 leave their accumulators in LDS in
// front of the unit's last barrier and waves 0 / 1 add them and run the epilogue -- 768 / 1216 units, no extra barrier.
template <int MB, int NB, int PIPE_U, int STATS, int KS>   // STATS: 0 none, 1 forward statistics, 2 BatchNorm-backward sums (BnBwdEpi)
__global__ __launch_bounds__(PIPE_THREADS) void conv_pipe_kernel(const ConvGeo g, const int nunits, const float* __restrict__ X,
                                                                 const float4* __restrict__ Wp, const float* __restrict__ bias,
                                                                 const float* __restrict__ in_scale,
                                                                 const float* __restrict__ in_shift, float* __restrict__ Y,
                                                                 const float* __restrict__ res, const float* __restrict__ res_mask,
                                                                 float* __restrict__ stamps, float* __restrict__ stats,
                                                                 const BnBwdEpi bwd, const int stats_acc, const int xcd_aware) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int MW = 16 * MB;
    constexpr int MT = 4 / KS;                  // pixel blocks (waves) of a unit
    constexpr int UPX = MW * MT;                // output pixels of a unit
    constexpr int QM = 4 * KS - 1;              // channel quads of a stage - 1
    const int HoWo = g.Ho * g.Wo;
    const float inv_wo = 1.0f / (float)g.Wo;
    const int chunk_bytes = g.PR * g.PWT * 64;
    const int stage_bytes = chunk_bytes * KS;
    const int nst = g.nchunks / KS;             // stages of a unit
    const int nby = g.N / 16 / NB;
    // XCD-aware unit walk: workgroup b runs on XCD b % 8 and every XCD has its own L2.  Each XCD takes one CONTIGUOUS eighth of the
    // units and its workgroups walk it side by side, so that neighbouring tiles (they share halo rows) and the channel-block groups
    // of one tile meet in ONE L2 (plain stride walk: FETCH_SIZE 2.6 x the input on the 257 x 33 stage, 6 x on 33 x 5).
    const bool xa = xcd_aware && (gridDim.x & 7) == 0;
    const int G = xa ? (int)gridDim.x >> 3 : (int)gridDim.x;
    const int per_xcd = xa ? (nunits + 7) >> 3 : nunits;
    const int u_lo = xa ? (int)(blockIdx.x & 7) * per_xcd : 0;
    const int unit_end = u_lo + per_xcd < nunits ? u_lo + per_xcd : nunits;
    const int unit0 = u_lo + (xa ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    if (unit0 >= unit_end) return;                 // (the whole workgroup, in front of its first barrier)
    const size_t img_floats = (size_t)g.H * g.W * g.C;
#ifdef DAM_PIPE_STAMPS
    unsigned long long* stamp_p = reinterpret_cast<unsigned long long*>(stamps) + (size_t)blockIdx.x * 128;
    int stamp_n = 0;
#endif
    DAM_PSTAMP(wave >> 2, 1);

    if (wave >= 4) {
        // ================= loader waves =================
        // An item = one pixel's channel quad of the chunk (a float4); thread t of the 256 takes items t, t + 256, ... of the
        // patch (<= PIPE_U of them).  Per item, computed once per launch: its LDS byte offset and its BYTE offset in the image
        // relative to (patch row 0, column 0).  Everything that changes from unit to unit is scalar: a request adds the unit's
        // row base and the chunk to the offsets and loads through a buffer resource that spans exactly the image -- rows above /
        // below the tensor and the items of columns outside it (offset INT_MIN) fail the range check and read as zero, which
        // is the padding.
        const int ltid = tid - 256;
        const int WC = g.W * g.C;
        const int ipr = g.PWin * 4 * KS;                         // items per patch row
        const int total = g.PR * ipr;
        const int tab_off = PIPE_BUF1 + stage_bytes;             // scale / shift tables (2 * C floats, only with in_scale)
        const int dump_off = tab_off + (in_scale ? 2 * g.C * 4 : 0);    // 4 KB that absorb the writes of threads without an item
        int dst[PIPE_U], gcol[PIPE_U];
        {
            const int step_r = PIPE_LT / ipr, step_c = PIPE_LT - step_r * ipr;  // e += PIPE_LT in (row, item of the row)
            int pr = fast_div(ltid, ipr, 1.0f / (float)ipr), rem = ltid - pr * ipr;
#pragma unroll
            for (int u = 0; u < PIPE_U; ++u) {
                dst[u] = dump_off + ltid * 16; gcol[u] = INT_MIN;
                if (ltid + PIPE_LT * u < total) {
                    const int pw = rem / (4 * KS), cq = rem & QM;        // cq >> 2: the chunk of the stage
                    const int slot = g.s == 1 ? pw : (pw & 1) * g.PWs + (pw >> 1);
                    dst[u] = ((pr * g.PWT + slot) * 16 + (cq & 3) * 4) * 4 + (cq >> 2) * chunk_bytes;
                    const int iw = g.c0 + pw;
                    if (iw >= 0 && iw < g.W) gcol[u] = (pr * WC + iw * g.C + cq * 4) * 4;
                }
                pr += step_r; rem += step_c;
                if (rem >= ipr) { rem -= ipr; ++pr; }
            }
        }
        const int cq4 = (lane & QM) * 4;                         // an item's channel quad is its lane's (256 and ipr are multiples of 4 * KS)
        // Two register sets hold two DIFFERENT chunks of the stream, so a chunk's loads are in flight for two compute periods
        // (one measured period is ~2.3 us at one workgroup per CU, about the latency of the load itself: with one chunk in
        // flight the loaders, not the MFMAs, set the pace).  A round = commit the older set, refill it from the head of the
        // stream, barrier; a refill is ALWAYS PIPE_U loads (behind the end of the stream at an offset that fails the range
        // check: no memory traffic), so the wait in front of a commit is the compile-time vmcnt(PIPE_U), not 0.
        const float relu_lo = g.relu_in ? 0.f : -__builtin_inff();
        float4 v[2][PIPE_U];
        int unit = unit0;                                        // head of the stream: the next chunk to request
        PipeUnit cur = pipe_decode<UPX, NB>(g, unit, nby, inv_wo);
        int icg = 0;
        int rowb = (cur.oh_first * g.s + g.r0) * WC * 4;         // byte offset of patch row 0 in the image (negative above it)
        const unsigned img_bytes = (unsigned)img_floats * 4u;
        __amdgpu_buffer_rsrc_t xrsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + (size_t)cur.img * img_floats), 0, img_bytes, 0x00020000);
        bool head = true;
        int s_sb[2] = {0, 0}, s_cg[2] = {0, 0};                  // per register set: byte base and chunk of what it holds
        bool s_live[2] = {false, false};
#define DAM_PIPE_REFILL(S_)                                                                                                \
    do {                                                                                                                   \
        const int sb_ = head ? rowb + icg * (64 * KS) : INT_MIN;                                                                \
        _Pragma("unroll") for (int u = 0; u < PIPE_U; ++u)                                                                 \
            v[S_][u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, gcol[u] + sb_, 0, 0));      \
        s_sb[S_] = sb_; s_cg[S_] = icg; s_live[S_] = head;                                                                 \
        if (head) {                                         /* advance the head */                                         \
            if (icg + 1 < nst) {                                                                                           \
                ++icg;                                                                                                     \
            } else if (unit + G < unit_end) {                                                                                \
                unit += G;                                                                                                 \
                cur = pipe_decode<UPX, NB>(g, unit, nby, inv_wo);                                                          \
                icg = 0;                                                                                                   \
                rowb = (cur.oh_first * g.s + g.r0) * WC * 4;                                                               \
                xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + (size_t)cur.img * img_floats), 0, img_bytes, \
                                                          0x00020000);                                                     \
            } else {                                                                                                       \
                head = false;                                                                                              \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)
#define DAM_PIPE_COMMIT(S_)                                 /* set S_ -> buffer S_ (round i: chunk i, set and buffer i & 1) */ \
    do {                                                                                                                   \
        constexpr int boff_ = (S_) * PIPE_BUF1;                                                                            \
        if (in_scale) {                                     /* relu(x * scale + shift) of what is inside the tensor */     \
            const float4 sc = *reinterpret_cast<const float4*>(smem + tab_off + (s_cg[S_] * (16 * KS) + cq4) * 4);         \
            const float4 sh = *reinterpret_cast<const float4*>(smem + tab_off + (g.C + s_cg[S_] * (16 * KS) + cq4) * 4);   \
            _Pragma("unroll") for (int u = 0; u < PIPE_U; ++u) {                                                           \
                float4 x = v[S_][u];                                                                                       \
                x.x = fmaf(x.x, sc.x, sh.x); x.y = fmaf(x.y, sc.y, sh.y);                                                  \
                x.z = fmaf(x.z, sc.z, sh.z); x.w = fmaf(x.w, sc.w, sh.w);                                                  \
                x.x = fmaxf(x.x, relu_lo); x.y = fmaxf(x.y, relu_lo); x.z = fmaxf(x.z, relu_lo); x.w = fmaxf(x.w, relu_lo); /* -inf: no ReLU */ \
                if ((unsigned)(gcol[u] + s_sb[S_]) >= img_bytes) x = make_float4(0.f, 0.f, 0.f, 0.f);                      \
                *reinterpret_cast<float4*>(smem + boff_ + dst[u]) = x;                                                     \
            }                                                                                                              \
        } else {                                                                                                           \
            _Pragma("unroll") for (int u = 0; u < PIPE_U; ++u)                                                             \
                *reinterpret_cast<float4*>(smem + boff_ + dst[u]) = v[S_][u];                                              \
        }                                                                                                                  \
    } while (0)
#define DAM_PIPE_ROUND_END()                                /* (1 + i): chunk i of the stream is staged, chunk i - 1 consumed */ \
    do {                                                                                                                   \
        DAM_PSTAMP(1, 6);                                                                                                  \
        DAM_PIPE_BARRIER();                                                                                                \
        DAM_PSTAMP(1, 7);                                                                                                  \
    } while (0)
        // (the table copy comes first: its loads would otherwise sit between the requests and the first commit, and the
        // compiler's wait in front of every commit becomes vmcnt(0))
        if (in_scale)
            for (int e = ltid * 4; e < g.C; e += PIPE_LT * 4) {
                *reinterpret_cast<float4*>(smem + tab_off + e * 4) = *reinterpret_cast<const float4*>(in_scale + e);
                *reinterpret_cast<float4*>(smem + tab_off + (g.C + e) * 4) = *reinterpret_cast<const float4*>(in_shift + e);
            }
        // the slots of a stride-2 de-interleave that no item covers must read as zero: clear both buffers once (with stride 1
        // every slot a valid pixel reads is an item); the two loader waves meet at barrier (0) before either commits
        if (g.s != 1)
            for (int e = ltid * 16; e < stage_bytes; e += PIPE_LT * 16) {
                *reinterpret_cast<float4*>(smem + e) = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(smem + PIPE_BUF1 + e) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        DAM_PIPE_REFILL(0);
        __builtin_amdgcn_sched_barrier(0);          // set 0 is the OLDER request, as in every round of the loop
        DAM_PIPE_REFILL(1);
        __builtin_amdgcn_sched_barrier(0);
        DAM_PSTAMP(1, 2);
        DAM_PIPE_BARRIER();                         // (0)
        DAM_PSTAMP(1, 3);
        for (;;) {
            DAM_PIPE_COMMIT(0);
            DAM_PIPE_REFILL(0);
            DAM_PIPE_ROUND_END();
            if (!s_live[1]) break;
            DAM_PIPE_COMMIT(1);
            DAM_PIPE_REFILL(1);
            DAM_PIPE_ROUND_END();
            if (!s_live[0]) break;
        }
        DAM_PIPE_BARRIER();                         // the last chunk is consumed
        if (STATS && stats_acc) DAM_PIPE_BARRIER(); // the compute waves' statistics are in LDS (their merge follows, see below)
#undef DAM_PIPE_REFILL
#undef DAM_PIPE_COMMIT
#undef DAM_PIPE_ROUND_END
        return;
    }

    // ================= compute waves =================
    const int j = lane & 15, kq = lane >> 4;
    const int mt = wave & (MT - 1), kh = wave / MT;        // pixel block of the unit, K half (KS == 2)
    const int part_off = PIPE_BUF1 + stage_bytes + (in_scale ? 2 * g.C * 4 : 0) + PIPE_LT * 16;   // behind the loaders' dump zone
    // stats_acc (one channel-block group per image row of units: every unit of the launch has the SAME channels): the waves'
    // statistics accumulate over the units this workgroup walks -- in LDS, [wave][channel][k, s1, s2] + [wave] n -- and the
    // workgroup leaves ONE record per channel at the end instead of one per wave and unit (257 x 33 stage: 512 instead of 4256)
    const int sacc_off = part_off + (KS == 2 ? 2 * MB * NB * 1024 : 0);
    float* const sacc = reinterpret_cast<float*>(smem + sacc_off) + wave * (g.N * 3 + 4);
    bool sacc_first = true;
    // tap constants: packed-weight byte offset of (tap, chunk 0, block 0) and LDS byte offset of the tap inside the patch
    int tapw[9], tapx[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int a = t / 3, b = t % 3;
        tapw[t] = (g.wt_base + a * g.wt_sa + b * g.wt_sb) * g.nchunks * g.NBtot * 1024;
        const int roff = g.off_h + a * g.step_h - g.r0, coff = g.off_w + b * g.step_w - g.c0;
        const int slotoff = g.s == 1 ? coff : (coff & 1) * g.PWs + (coff >> 1);
        tapx[t] = (roff * g.PWT + slotoff) * 64;
    }
    const int lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(Wp), 0, 0x7fffffff, 0x00020000);
    // Weight pipeline depth: an item is 4 * MB * NB MFMAs = 128 * MB * NB clocks of the matrix pipe.  Two items ahead (three
    // rotating sets) cover an L2 hit (~700 clocks) only from four-block tiles up; the one-block tile of the small deep stages
    // takes NINE sets, one per tap: item T's MFMAs are followed by the request of tap T of the NEXT chunk into the same set, so
    // every request has a whole chunk (1152 clocks) to land.  (Measured: no difference on those stages -- a build whose weight
    // loads all hit one 4 KB block is no faster either, the weights are not what they wait for -- so the two-block tiles, where
    // nine sets cost 48 registers, keep three.)
#ifdef DAM_PIPE_NO_W9
    constexpr bool W9 = false;
#else
    constexpr bool W9 = MB * NB == 1;
#endif
    constexpr int WSETS = W9 ? 9 : 3;
    v4f acc[MB][NB];
    float4 wa[WSETS][NB], xv[3][MB];
    int base_b[MB];
    int unit = unit0;
    PipeUnit cur = pipe_decode<UPX, NB>(g, unit, nby, inv_wo);
    int sbuf = 0;
#define DAM_PIPE_W(S_, CPART_, T_)                                                                                         \
    do {                                                                                                                   \
        const int ws_ = DAM_PIPE_WOFF(tapw[T_] + (CPART_));                                                                \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                                  \
            wa[S_][nb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + nb * 1024, ws_, 0)); \
    } while (0)
#define DAM_PIPE_X(S_, T_)                                                                                                 \
    do {                                                                                                                   \
        const int lo_ = boff + tapx[T_];                                                                                   \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                  \
            xv[S_][mb] = *reinterpret_cast<const float4*>(smem + base_b[mb] + lo_);                                        \
    } while (0)
#ifdef DAM_PIPE_DIAG_NOMFMA
#define DAM_PIPE_MFMA2(SW_, SX_)                                                                                           \
    do {                                                                                                                   \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                  \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                            \
                acc[mb][nb].x += wa[SW_][nb].x * xv[SX_][mb].x; acc[mb][nb].y += wa[SW_][nb].y * xv[SX_][mb].y;            \
                acc[mb][nb].z += wa[SW_][nb].z * xv[SX_][mb].z; acc[mb][nb].w += wa[SW_][nb].w * xv[SX_][mb].w;            \
            }                                                                                                              \
    } while (0)
#else
#define DAM_PIPE_MFMA2(SW_, SX_)                                                                                           \
    do {                                                                                                                   \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                  \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                            \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[SW_][nb].x, xv[SX_][mb].x, acc[mb][nb], 0, 0, 0);    \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[SW_][nb].y, xv[SX_][mb].y, acc[mb][nb], 0, 0, 0);    \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[SW_][nb].z, xv[SX_][mb].z, acc[mb][nb], 0, 0, 0);    \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[SW_][nb].w, xv[SX_][mb].w, acc[mb][nb], 0, 0, 0);    \
            }                                                                                                              \
    } while (0)
#endif
#define DAM_PIPE_MFMA(S_) DAM_PIPE_MFMA2(S_, S_)
    {
        const int cfirst = (kh * g.NBtot + cur.nb0) * 1024;
        DAM_PIPE_W(0, cfirst, 0);                   // weights of the first two items: nothing to wait for
        DAM_PIPE_W(1, cfirst, 1);
        if constexpr (W9) {                         // ... of the whole first chunk
            DAM_PIPE_W(2 % WSETS, cfirst, 2); DAM_PIPE_W(3 % WSETS, cfirst, 3); DAM_PIPE_W(4 % WSETS, cfirst, 4);
            DAM_PIPE_W(5 % WSETS, cfirst, 5); DAM_PIPE_W(6 % WSETS, cfirst, 6); DAM_PIPE_W(7 % WSETS, cfirst, 7);
            DAM_PIPE_W(8 % WSETS, cfirst, 8);
        }
    }
    DAM_PSTAMP(0, 2);
    DAM_PIPE_BARRIER();                             // (0)
    DAM_PSTAMP(0, 3);
    DAM_PIPE_BARRIER();                             // (1) the first chunk is staged
    DAM_PSTAMP(0, 5);
    for (;;) {
        const bool has_next = unit + G < unit_end;
        const PipeUnit nxt = has_next ? pipe_decode<UPX, NB>(g, unit + G, nby, inv_wo) : cur;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            int p = cur.p0 + mt * MW + mb * 16 + j;
            p = p < HoWo ? p : HoWo - 1;
            const int oh = fast_div(p, g.Wo, inv_wo), ow = p - oh * g.Wo;
            base_b[mb] = (((oh - cur.oh_first) * g.s) * g.PWT + ow) * 64 + kq * 16;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
        for (int cg = 0; cg < nst; ++cg) {
            const int boff = (sbuf & 1) * PIPE_BUF1 + kh * chunk_bytes;
            ++sbuf;
            const int ccur = ((cg * KS + kh) * g.NBtot + cur.nb0) * 1024;
            // the chunk behind this one in the stream (the last chunk of the last unit re-requests its own first blocks)
            const int cnext = cg + 1 < nst ? ccur + KS * g.NBtot * 1024 : (kh * g.NBtot + nxt.nb0) * 1024;
            // the order below IS the software pipeline: without the scheduling barriers the compiler sinks every request to
            // just in front of its first use (fewer live registers) and the MFMAs wait for L2 on every item
// Four-block items (five requests per 16 MFMAs): each request sits behind one of the item's first MFMAs instead of all five in
// front of them -- 257 x 33 stage 53.8 -> 52.9 us over four runs; smaller items measured no different and keep the plain order.
#define DAM_PIPE_SCHED_A()                                                                                                 \
    do { if constexpr (NB < 4) __builtin_amdgcn_sched_barrier(0); } while (0)
#define DAM_PIPE_SCHED_B()                                                                                                 \
    do {                                                                                                                   \
        if constexpr (NB >= 4) {                                                                                           \
            _Pragma("unroll") for (int i_ = 0; i_ < NB; ++i_) {                                                            \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                         \
            }                                                                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
            __builtin_amdgcn_sched_group_barrier(0x100, MB, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * MB * NB, 0);                                                   \
        }                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)
#define DAM_PIPE_ITEM(SW_, CP_, TW_, SX_, TX_, SM_)                                                                        \
    do {                                                                                                                   \
        DAM_PIPE_W(SW_, CP_, TW_);                                                                                         \
        if ((TX_) >= 0) DAM_PIPE_X(SX_, (TX_) < 0 ? 0 : (TX_));                                                            \
        DAM_PIPE_SCHED_A();                                                                                                \
        DAM_PIPE_MFMA(SM_);                                                                                                \
        DAM_PIPE_SCHED_B();                                                                                                \
    } while (0)
#define DAM_PIPE_ITEM9(T_)                                  /* X two taps ahead, MFMAs of tap T_, its set refilled from the next chunk */ \
    do {                                                                                                                   \
        if ((T_) + 2 < 9) DAM_PIPE_X(((T_) + 2) % 3, (T_) + 2 < 9 ? (T_) + 2 : 0);                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        DAM_PIPE_MFMA2((T_) % WSETS, (T_) % 3);                                                                            \
        DAM_PIPE_W((T_) % WSETS, cnext, T_);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)
            DAM_PIPE_X(0, 0);
            DAM_PIPE_X(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (W9) {
                DAM_PIPE_ITEM9(0); DAM_PIPE_ITEM9(1); DAM_PIPE_ITEM9(2); DAM_PIPE_ITEM9(3); DAM_PIPE_ITEM9(4);
                DAM_PIPE_ITEM9(5); DAM_PIPE_ITEM9(6); DAM_PIPE_ITEM9(7); DAM_PIPE_ITEM9(8);
            } else {
            DAM_PIPE_ITEM(2, ccur, 2, 2, 2, 0);
            DAM_PIPE_ITEM(0, ccur, 3, 0, 3, 1);
            DAM_PIPE_ITEM(1, ccur, 4, 1, 4, 2);
            DAM_PIPE_ITEM(2, ccur, 5, 2, 5, 0);
            DAM_PIPE_ITEM(0, ccur, 6, 0, 6, 1);
            DAM_PIPE_ITEM(1, ccur, 7, 1, 7, 2);
            DAM_PIPE_ITEM(2, ccur, 8, 2, 8, 0);
            DAM_PIPE_ITEM(0, cnext, 0, 0, -1, 1);
            DAM_PIPE_ITEM(1, cnext, 1, 0, -1, 2);
            }
#undef DAM_PIPE_ITEM
#undef DAM_PIPE_ITEM9
            DAM_PSTAMP(0, 6);
            if (KS == 2 && kh == 1 && cg + 1 == nst) {          // the second K half's sums, for the wave that holds the first
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        *reinterpret_cast<v4f*>(smem + part_off + ((mt * MB + mb) * NB + nb) * 1024 + lane * 16) = acc[mb][nb];
            }
            DAM_PIPE_BARRIER();                     // (2 + i)
            DAM_PSTAMP(0, 7);
        }
        if (KS == 2) {
            if (kh == 1) {                          // no epilogue: on to the next unit (the same barriers as everybody)
                if (!has_next) break;
                unit += G;
                cur = nxt;
                continue;
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] += *reinterpret_cast<const v4f*>(smem + part_off + ((mt * MB + mb) * NB + nb) * 1024 + lane * 16);
        }

        // ---- write the tile out (as conv_igemm_kernel): lane holds channels 4*kq..+3 of pixel j of every (mb, nb) block ----
        // BatchNorm partial statistics of what this WAVE writes (optional), flushed below as one (n, mean, M2) record per
        // channel, wave and unit -- no cross-wave step, so no extra barrier with the loaders.  The 16 pixel lanes of a channel
        // share ONE shift (the value of the row's first lane: real data even behind the image's last pixel, whose lanes compute
        // that pixel again), so the lanes' sum(v - shift) and sum((v - shift)^2) simply add: DPP row rotations, no LDS.
        float st_k[NB][4], st_s1[NB][4], st_s2[NB][4];
        float st_n = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float first = reinterpret_cast<const float*>(&acc[0][nb])[r] +
                                    (bias ? bias[(cur.nb0 + nb) * 16 + kq * 4 + r] : 0.f);
                st_k[nb][r] = STATS == 1 ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, first), 0x150, 0xF, 0xF, false))
                                    : 0.f;                                     // row_newbcast:0 -- lane 0 of the row
                // accumulating form: ONE shift per channel for everything this wave sees (its first unit's, kept in LDS)
                if (STATS == 1 && stats_acc && !sacc_first) st_k[nb][r] = sacc[((cur.nb0 + nb) * 16 + kq * 4 + r) * 3];
                st_s1[nb][r] = 0.f; st_s2[nb][r] = 0.f;
            }
        // STATS == 2: `res` is the BatchNorm's input x (nothing is added); per channel quad the two affine maps
        // mask = x*mscale + mshift > 0 and xhat = x*invstd - mean*invstd, fetched per unit (L2 hits) so that they cost no
        // register outside the epilogue; st_s1 / st_s2 take sum(dz) / sum(dz * xhat)
        v4f bw_ms[STATS == 2 ? NB : 1], bw_mh[STATS == 2 ? NB : 1], bw_k1[STATS == 2 ? NB : 1], bw_k2[STATS == 2 ? NB : 1];
#pragma unroll
        for (int nb = 0; nb < (STATS == 2 ? NB : 0); ++nb) {
            const int ch = (cur.nb0 + nb) * 16 + kq * 4;
            bw_ms[nb] = *reinterpret_cast<const v4f*>(bwd.mscale + ch);
            bw_mh[nb] = *reinterpret_cast<const v4f*>(bwd.mshift + ch);
            bw_k1[nb] = *reinterpret_cast<const v4f*>(bwd.invstd + ch);
            bw_k2[nb] = -(*reinterpret_cast<const v4f*>(bwd.mean + ch)) * bw_k1[nb];
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int p = cur.p0 + mt * MW + mb * 16 + j;
            if (p >= HoWo) continue;
            const int oh = fast_div(p, g.Wo, inv_wo), ow = p - oh * g.Wo;
            const size_t opix = ((size_t)cur.img * g.OHt + (oh * g.os + g.oo_h)) * g.OWt + (ow * g.os + g.oo_w);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int ch = (cur.nb0 + nb) * 16 + kq * 4;
                if (ch >= g.N) continue;
                v4f v = acc[mb][nb];
                const size_t o = opix * g.N + ch;
                if (bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias + ch);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                }
                if (STATS == 2) {
                    const v4f rv = *reinterpret_cast<const v4f*>(res + o);
                    const v4f m = __builtin_elementwise_fma(rv, bw_ms[nb], bw_mh[nb]);
                    const v4f xh = __builtin_elementwise_fma(rv, bw_k1[nb], bw_k2[nb]);
                    const float dz[4] = {m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f};
                    const float xe[4] = {xh.x, xh.y, xh.z, xh.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        st_s1[nb][r] += dz[r];
                        st_s2[nb][r] = fmaf(dz[r], xe[r], st_s2[nb][r]);
                    }
                } else if (res) {
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
                if (STATS == 1) {
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float d = e[r] - st_k[nb][r];
                        st_s1[nb][r] += d;
                        st_s2[nb][r] = fmaf(d, d, st_s2[nb][r]);
                    }
                }
            }
            if (STATS == 1) st_n += 1.f;
        }
        if (STATS) {
            // row sums by rotation (ror 8, 4, 2, 1: every lane ends up with the total), then lane 0 of the row writes
#define DAM_ROW_SUM(V_)                                                                                                     \
    do {                                                                                                                    \
        V_ += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V_), 0x128, 0xF, 0xF, false)); \
        V_ += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V_), 0x124, 0xF, 0xF, false)); \
        V_ += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V_), 0x122, 0xF, 0xF, false)); \
        V_ += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V_), 0x121, 0xF, 0xF, false)); \
    } while (0)
            const size_t rec = ((size_t)cur.img * g.tiles_m + (cur.p0 / UPX)) * MT + mt;
            float n = st_n;
            DAM_ROW_SUM(n);
            if (stats_acc && j == 0 && kq == 0) sacc[g.N * 3] = sacc_first ? n : sacc[g.N * 3] + n;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s1 = st_s1[nb][r], s2 = st_s2[nb][r];
                    DAM_ROW_SUM(s1);
                    DAM_ROW_SUM(s2);
                    const int ch = (cur.nb0 + nb) * 16 + kq * 4 + r;
                    if (stats_acc) {            // this wave's own LDS cells: plain read-add-write, fixed order
                        if (j == 0 && ch < g.N) {
                            float* o = sacc + ch * 3;
                            if (sacc_first) { o[0] = st_k[nb][r]; o[1] = s1; o[2] = s2; }
                            else { o[1] += s1; o[2] += s2; }
                        }
                    } else if (STATS == 2) {
                        if (j == 0 && ch < g.N) { float* o = stats + (rec * g.N + ch) * 2; o[0] = s1; o[1] = s2; }
                    } else if (j == 0 && ch < g.N) {
                        const float md = n > 0.f ? s1 / n : 0.f;
                        float* o = stats + (rec * g.N + ch) * 3;
                        o[0] = n; o[1] = st_k[nb][r] + md; o[2] = n > 0.f ? fmaxf(s2 - s1 * md, 0.f) : 0.f;
                    }
                }
#undef DAM_ROW_SUM
            sacc_first = false;
        }
        DAM_PSTAMP(0, 8);
        if (!has_next) break;
        unit += G;
        cur = nxt;
    }
    if (STATS && stats_acc) {
        // every wave's sums are in LDS: one barrier (the loaders join it, see their last line), then the first N lanes of the
        // workgroup merge the MT waves of a channel in wave order and write the workgroup's record
        DAM_PIPE_BARRIER();
        const int c = tid;
        if (c < g.N) {
            const float* base = reinterpret_cast<const float*>(smem + sacc_off);
            const int wstride = g.N * 3 + 4;
            if (STATS == 2) {
                float s1 = 0.f, s2 = 0.f;
                for (int w = 0; w < MT; ++w) { s1 += base[w * wstride + c * 3 + 1]; s2 += base[w * wstride + c * 3 + 2]; }
                float* o = stats + ((size_t)blockIdx.x * g.N + c) * 2;
                o[0] = s1; o[1] = s2;
            } else {
                float na = 0.f, ma = 0.f, qa = 0.f;          // Chan merge of the waves' (n, mean, M2), fixed order
                for (int w = 0; w < MT; ++w) {
                    const float nb_ = base[w * wstride + g.N * 3];
                    if (nb_ <= 0.f) continue;
                    const float k_ = base[w * wstride + c * 3], s1 = base[w * wstride + c * 3 + 1], s2 = base[w * wstride + c * 3 + 2];
                    const float md = s1 / nb_, mb_ = k_ + md, qb = fmaxf(s2 - s1 * md, 0.f);
                    const float nn = na + nb_, d = mb_ - ma;
                    ma += d * (nb_ / nn);
                    qa += qb + d * d * (na * nb_ / nn);
                    na = nn;
                }
                float* o = stats + ((size_t)blockIdx.x * g.N + c) * 3;
                o[0] = na; o[1] = ma; o[2] = qa;
            }
        }
    }
#undef DAM_PIPE_W
#undef DAM_PIPE_X
#undef DAM_PIPE_MFMA
}

// resident workgroups per launch: occupancy x CUs, asked once per instance and LDS size
template <int MB, int NB, int PU, int STATS, int KS>
int pipe_slots(size_t lds) {
    const int cus = device_cus();
    static PerDevice<size_t> cached_lds_pd;          // (0 = nothing cached yet: the kernel always asks for LDS)
    static PerDevice<int> cached_pd;
    size_t& cached_lds = cached_lds_pd();
    int& cached = cached_pd();
    if (cached_lds != lds || !cached) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&conv_pipe_kernel<MB, NB, PU, STATS, KS>), PIPE_THREADS,
                                                         lds) != hipSuccess || per_cu < 1)
            per_cu = 1;
        cached = per_cu * cus;
        cached_lds = lds;
    }
    return cached;
}

template <int MB, int NB, int PU, int KS>
int launch_pipe(const ConvGeo& g, size_t lds, const float* X, const float* Wp, const float* bias, const float* sc, const float* sh,
                float* Y, const float* res, const float* res_mask, float* workspace, float* stats, int* stats_parts,
                const BnBwdEpi& bwd, hipStream_t st) {
    const int nunits = g.tiles_m * (g.N / 16 / NB) * g.B;
    // statistics records: one per (image tile, wave), each unit fills its own channels of it
    int mode = 0, acc = 0;
    if (stats) {
        const int64_t parts = (int64_t)g.B * g.tiles_m * (4 / KS);
        mode = bwd.x ? 2 : 1;
        // one channel-block group per unit row (every unit has all the layer's channels: the 1 x 4 tile of a 64-channel stage):
        // the workgroups accumulate over their units and leave ONE record each (kernel: stats_acc).  BUILT, PARITY-GREEN, MEASURED
        // SLOWER, OFF (DAM_PIPE_STATS_ACC=1 turns it on for the A/B): it removes the 257 x 33 stage's five separate statistics /
        // backward-sums passes (3 x 8.6 + 2 x 11 us), but the 1 x 4 tile's instantiations WITH an epilogue run 61.9 us (forward
        // statistics, 512 workgroups) and 73.8 us (backward sums: 256 workgroups, the register budget of two per CU is gone)
        // against 52.6 us without -- C3 step 4.456 -> 4.483 ms on one box (gpurun_out/r4, profiles/r04_pipe_stats_acc_ab.txt)
        static const bool acc_ok = [] { const char* e = getenv("DAM_PIPE_STATS_ACC"); return e && e[0] == '1'; }();
        if (acc_ok && g.N == 16 * NB && g.N <= PIPE_THREADS - PIPE_LT) acc = 1;
        else if (parts > (bwd.x ? BN_BWD_RECORDS_MAX : BN_RECORDS_MAX)) { stats = nullptr; mode = 0; }   // the caller runs the separate pass
        else if (stats_parts) *stats_parts = (int)parts;
    }
    if (bwd.x) res = mode == 2 ? bwd.x : nullptr;               // the sums epilogue reads x through the residual operand
    if (acc) lds += (size_t)(4 / KS) * (g.N * 3 + 4) * sizeof(float);
    int wgs = mode == 2 ? pipe_slots<MB, NB, PU, 2, KS>(lds) : (mode == 1 ? pipe_slots<MB, NB, PU, 1, KS>(lds) : pipe_slots<MB, NB, PU, 0, KS>(lds));
    if (const char* e = getenv("DAM_PIPE_WGS")) wgs = atoi(e);          // diagnostic
    if (wgs < 1) wgs = 1;
    if (wgs > nunits) wgs = nunits;
    if (acc && stats_parts) *stats_parts = wgs;
    static const int xcd_aware = getenv("DAM_PIPE_NO_XCD") ? 0 : 1;      // A/B knob
#define DAM_PIPE_LAUNCH(S_)                                                                                                 \
    hipLaunchKernelGGL((conv_pipe_kernel<MB, NB, PU, S_, KS>), dim3((unsigned)wgs), dim3(PIPE_THREADS), lds, st, g, nunits, X,  \
                       reinterpret_cast<const float4*>(Wp), bias, sc, sh, Y, res, res_mask, workspace, stats, bwd, acc, xcd_aware)
    if (mode == 2) DAM_PIPE_LAUNCH(2); else if (mode == 1) DAM_PIPE_LAUNCH(1); else DAM_PIPE_LAUNCH(0);
#undef DAM_PIPE_LAUNCH
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

}  // namespace

// Returns DAM_ERR_UNSUPPORTED when the layer does not fit this variant (the caller goes on to conv_igemm_kernel).
// g: set up by dam_conv2d_tapgrid_f32 up to the tile-independent part (patch columns PWin / PWs / PWT, tap grid);
// row_span = h_hi - h_lo of the tap grid.  The tile is chosen here: no split-K -- small tiles give this kernel its
// workgroups, their patches are staged beside the MFMAs and cost the matrix pipe nothing.
int conv_pipe_try(ConvGeo g, int row_span, const float* X, const float* Wp, const float* bias, const float* sc, const float* sh,
                  float* Y, const float* res, const float* res_mask, float* workspace, float* stats, int* stats_parts,
                  const BnBwdEpi& bwd, hipStream_t st) {
    if (bwd.x && (res || !stats)) return DAM_ERR_BAD_ARG;
    if (g.nA != 3 || g.nB != 3 || g.in_nchw) return DAM_ERR_UNSUPPORTED;     // (16-channel stride-1 layers never get here: row-ring kernel)
    if ((size_t)g.H * g.W * g.C * 4 >= ((size_t)1 << 30)) return DAM_ERR_UNSUPPORTED;      // offsets of the range-checked loads
    const int64_t npix = (int64_t)g.Ho * g.Wo;
    const int nblk = g.N / 16;
    auto patch_rows_px = [&](int px) {            // patch rows under `px` consecutive output pixels
        int rows_out = (int)((px + g.Wo - 2) / g.Wo + 1);
        if (rows_out > g.Ho) rows_out = g.Ho;
        return (rows_out - 1) * g.s + row_span + 1;
    };
    auto patch_rows = [&](int mb) { return patch_rows_px(64 * mb); };
    auto fits = [&](int mb, int nb) {
        const int pr = patch_rows(mb);
        return nblk % nb == 0 && pr * g.PWin * 4 <= PIPE_LT * 8 && (size_t)pr * g.PWT * 64 <= PIPE_BUF1;
    };
    // Tile: the largest of the <= 4-block tiles (<= 128 VGPRs: two workgroups per CU, three for the smallest) that still yields
    // >= 384 work units, else the one with the most units.  Measured on the ResNet's thick layers (tools/pipe_ab.sh):
    // layer3 1x4, layer4 2x2, layer5 / layer6 1x1 -- larger tiles lose more to idle CUs than they save in weight traffic.
    static const int cand[5][2] = {{1, 4}, {2, 2}, {1, 2}, {2, 1}, {1, 1}};
    int MB = 0, NB = 0;
    int64_t most = -1;
    for (int i = 0; i < 5; ++i) {
        const int mb = cand[i][0], nb = cand[i][1];
        if (!fits(mb, nb)) continue;
        const int64_t units = cdiv(npix, 64 * mb) * (nblk / nb) * g.B;
        if (units >= 384) { MB = mb; NB = nb; break; }
        if (units > most) { most = units; MB = mb; NB = nb; }
    }
    if (const char* e = getenv("DAM_TILE")) {          // diagnostic: "MBxNB"
        MB = e[0] - '0'; NB = e[2] - '0';
        if (MB < 1 || NB < 1 || !fits(MB, NB)) return DAM_ERR_UNSUPPORTED;
    }
    if (!MB) return DAM_ERR_UNSUPPORTED;
    // K split over the wave pairs (kernel comment): for one-block tiles that leave the chip fewer than three units per CU
    int KS = 1;
    // (>= 2 stages per unit: the hand-over area of unit u + 1 is written in front of that unit's LAST barrier, which the reading
    // waves of unit u pass only after their read -- with one stage per unit that barrier would also be the first)
    if (MB == 1 && NB == 1 && g.nchunks % 2 == 0 && g.nchunks >= 4 && cdiv(npix, 64) * nblk * g.B < 768) {
        const int pr = patch_rows_px(32);
        if (pr * g.PWin * 8 <= PIPE_LT * 3 && (size_t)pr * g.PWT * 128 <= PIPE_BUF1) KS = 2;     // (three items per loader thread)
    }
    if (const char* e = getenv("DAM_PIPE_KS")) { if (atoi(e) == 1) KS = 1; }     // diagnostic
    const int upx = 64 * MB / KS;
    g.PR = patch_rows_px(upx);
    g.tiles_m = (int)cdiv(npix, upx);
    g.CG = 1; g.ksplit = 1; g.gps = g.nchunks;
    // stage buffer 0 (padded to PIPE_BUF1), stage buffer 1, scale / shift tables, the dump zone, the K halves' hand-over
    const size_t lds = PIPE_BUF1 + (size_t)KS * g.PR * g.PWT * 64 + (sc ? (size_t)2 * g.C * 4 : 0) + PIPE_LT * 16 +
                       (KS == 2 ? (size_t)2 * MB * NB * 1024 : 0);
    const bool big = g.PR * g.PWin * 4 * KS > PIPE_LT * 5; // items per loader thread: 5 or 8
    if (KS == 2) return launch_pipe<1, 1, 3, 2>(g, lds, X, Wp, bias, sc, sh, Y, res, res_mask, workspace, stats, stats_parts, bwd, st);
#define DAM_PIPE_CASE(M_, N_)                                                                                              \
    if (MB == M_ && NB == N_)                                                                                              \
        return big ? launch_pipe<M_, N_, 8, 1>(g, lds, X, Wp, bias, sc, sh, Y, res, res_mask, workspace, stats, stats_parts, bwd, st) \
                   : launch_pipe<M_, N_, 5, 1>(g, lds, X, Wp, bias, sc, sh, Y, res, res_mask, workspace, stats, stats_parts, bwd, st)
    DAM_PIPE_CASE(1, 4); DAM_PIPE_CASE(2, 2); DAM_PIPE_CASE(1, 2); DAM_PIPE_CASE(2, 1); DAM_PIPE_CASE(1, 1);
#undef DAM_PIPE_CASE
    return DAM_ERR_UNSUPPORTED;
}

}  // namespace dam
