// dam_conv_strip.hip -- persistent "strip" variant of the implicit-GEMM convolution (forward / dgrad) for THIN layers:
// input rows with ALL channels fit an LDS ring and the packed weights of the output-channel tile fit beside it
// (ResNet layer1/2 and their data gradients).  These layers sit at the MFMA/HBM ridge.
//
// Same GEMM mapping, packed weights and tap-grid semantics as conv_igemm_kernel (dam_conv.hip).  One workgroup per CU,
// 12 waves with fixed roles, built from what in-kernel stamps and the issue-port probes showed (profiles/README.md,
// tools/mfma_valu_mix.hip, tools/mfma_stream_cost.hip):
//   * the workgroup walks `tpw` CONSECUTIVE 64*MB-pixel tiles of one image; input rows live in an LDS ring indexed by
//     (absolute row & (NR-1)), so a tile fetches only the rows its predecessors did not (HBM traffic 1.03x algorithmic);
//   * compute group A (waves 0-3) and B (waves 4-7) PING-PONG: in slot s one group runs the MFMAs of tile s while the
//     other writes out tile s-1 (bias, residual / mask, BatchNorm partial statistics) and prepares its next tile.  Each
//     SIMD hosts one wave of each group, so its matrix pipe always has exactly one MFMA stream;
//   * a streaming MFMA wave owns its SIMD's vector issue port, so everything wave-uniform runs on the scalar ALU: loader
//     waves 8-11 work in whole (row, chunk) planes (scalar plane arithmetic, per-lane column pattern computed once,
//     buffer_load + EXEC-masked ds_write: no VALU per piece; optional fused input affine = 8 VALU per piece), two slots
//     ahead in two register sets; pixel geometry by s_mul_hi; outputs through buffer_store;
//   * 3x3 / stride 1: compile-time item grid, operands = register + immediate, two-deep software pipeline; weights are LDS
//     resident in canonical [3a+b][chunk][nb][lane] order, so compute waves issue no vector-memory loads in the loop;
//   * one raw s_barrier per slot; compute waves do not drain their output stores at it;
//   * LDS image: [chunk][ring row][column slot][16 ch] with the same stride-2 column de-interleave as the tile kernel;
//     border slots (zero padding) are zeroed once, only in-tensor pixels are ever written;
//   * optional epilogue: per-channel BatchNorm partial statistics (n, mean, M2), one record per workgroup, merged by
//     bn_stats_finalize -- removes the statistics pass over the conv output.
#include <algorithm>
#include <cstdint>
#include "dam_common.h"
#include "dam_conv_geo.h"
#include "dam_bn_fin.h"

namespace dam {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// Diagnostic build only (-DDAM_STRIP_DIAG_TAGS, libdam_hip_diag.so; tests/test_strip_diag_gpu.py): every geometry-table entry the
// loader waves of the self-overlapped form write carries the index of the tile it describes, and every compute wave checks, for
// every pixel block of every tile, that the entry it is about to turn into addresses carries the tile it is about to compute.
// {checks made, mismatches seen}, read and reset by dam_strip_diag_counters().
#ifdef DAM_STRIP_DIAG_TAGS
}  // namespace
__device__ unsigned strip_diag_counters[2];
namespace {
#endif

// Diagnostic build only (-DDAM_STAMPS): the `stats` buffer receives s_memtime stamps of phase boundaries instead.
#ifdef DAM_STAMPS
#define DAM_STAMP(slot)                                                                                   \
    do {                                                                                                  \
        if (lane == 0 && (wave == 0 || wave == NCW) && stamp_i < 30)                                        \
            reinterpret_cast<unsigned long long*>(stats)[(((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 2 + (wave == NCW)) * 32 + \
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

constexpr int STRIP_LOADERS = 4;                        // loader waves per workgroup (waves 8..11)
constexpr int STRIP_THREADS = 512 + 64 * STRIP_LOADERS; // 2 compute groups of 4 waves + loaders
constexpr int STRIP_PU = 8;                             // 1 KB pieces in flight per loader wave

struct RowLoad {           // everything the row loader needs, by value (no closures over the kernel's arrays)
    const float* ximg;
    int H, W, C, s, c0, PWs, PWin, PWT, nchunks, RB, CHB, NR, ring_off, ppr, gpp;
    float inv_ppr, inv_gpp;
    int m_ppr, m_gpp;      // floor(65536 / d) + 1: q / d == (q * m) >> 16 for q < 65536 / d (piece indices are a few hundred)
};

// Requests pieces first + part + nparts*u (u < STRIP_PU) of input rows [lo, hi] into registers: unconditional loads from
// clamped addresses, unused / out-of-tensor lanes zeroed by select (a conditional around a load costs an s_waitcnt each).
__device__ __forceinline__ void rows_issue(const RowLoad& r, int lo, int hi, int first, int part, int nparts, int lane,
                                           float4 (&lv)[STRIP_PU], int (&ldst)[STRIP_PU], int (&laff)[STRIP_PU]) {
    const int total = (hi - lo + 1) * r.ppr;
#pragma unroll
    for (int u = 0; u < STRIP_PU; ++u) {
        const int q = first + part + nparts * u;
        const bool used = q < total;
        const int qc = used ? q : total - 1;
        // qc is wave-uniform (first, part, nparts are): integer magic division stays on the scalar ALU
        const int row = (qc * r.m_ppr) >> 16, rem = qc - row * r.ppr;
        const int cc = (rem * r.m_gpp) >> 16, gi = rem - cc * r.gpp;
        const int ih = lo + row;
        const int L = gi * 64 + lane, slot = L >> 2, quad = L & 3;
        int pw = slot;
        if (r.s != 1) pw = slot < r.PWs ? 2 * slot : 2 * (slot - r.PWs) + 1;
        const int iw = pw + r.c0;
        const bool col_ok = slot < r.PWT && pw < r.PWin && iw >= 0 && iw < r.W;
        const bool inb = used && col_ok && ih >= 0 && ih < r.H;
        const int ihc = ih < 0 ? 0 : (ih >= r.H ? r.H - 1 : ih), iwc = iw < 0 ? 0 : (iw >= r.W ? r.W - 1 : iw);
        const float4 x = *reinterpret_cast<const float4*>(r.ximg + (unsigned)((ihc * r.W + iwc) * r.C + cc * 16 + quad * 4));
        lv[u] = inb ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        ldst[u] = (used && col_ok) ? cc * r.CHB + ((ih + r.ring_off) & (r.NR - 1)) * r.RB + L * 16 : -1;
        laff[u] = inb ? cc : -1;               // real data of chunk cc (gets the fused input affine), else padding zeros
    }
}
// y = [relu](x * scale + shift) on the real-data lanes of a piece (the producer's BatchNorm + ReLU fused into the load)
template <int NCH>
__device__ __forceinline__ float4 in_affine(float4 v, int cc, const v4f (&scq)[NCH], const v4f (&shq)[NCH], bool relu) {
    const v4f sc = (NCH > 1 && cc > 0) ? scq[NCH - 1] : scq[0], sh = (NCH > 1 && cc > 0) ? shq[NCH - 1] : shq[0];
    float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    return o;
}
template <int NCH>
__device__ __forceinline__ void rows_commit(unsigned char* smem, const float4 (&lv)[STRIP_PU], int (&ldst)[STRIP_PU],
                                            const int (&laff)[STRIP_PU], bool has_aff, const v4f (&scq)[NCH],
                                            const v4f (&shq)[NCH], bool relu) {
#pragma unroll
    for (int u = 0; u < STRIP_PU; ++u) {
        if (ldst[u] >= 0) {
            float4 v = lv[u];
            if (has_aff && laff[u] >= 0) v = in_affine<NCH>(v, laff[u], scq, shq, relu);
            *reinterpret_cast<float4*>(smem + ldst[u]) = v;
        }
        ldst[u] = -1;
    }
}

// NCH: 16-channel input chunks (compile time so that every MFMA operand offset of the 3x3xNCH item grid is a scalar)
// LW: loader shape -- false: rows of up to 144 (16 ch) / 80 (32 ch) cells, true: up to 224 / 112 cells (the reference's native
// 216-frame spectrograms and their half-resolution stage)
// EPI: the BatchNorm-backward sums epilogue (BnBwdEpi) is compiled in -- its per-lane constants cost 16*NB registers, so only
// the data-gradient shapes that use it are instantiated with it.  1: the launch has no residual, x rides in the residual
// operand.  2: the launch carries the identity shortcut (residual operand, its ReLU mask as sign BYTES in `res_mask`) and the
// sums are those of the BatchNorm UPSTREAM of the block input (its x is a third prefetched operand).  3: as 2, the upstream ReLU
// mask comes as sign bytes as well (bwd.mask_bits: the upstream layer is relu(bn(x) + shortcut), a residual block's bn2)
// SO: 0 = the ping-pong form above (two compute groups alternate MFMA and write-out slots).  > 0 = SELF-OVERLAPPED form (T33
// only, 8 waves): ONE compute group; every compute wave writes tile s-1 out, computes the geometry of tile s+1 and requests the
// residual operands of tile s INSIDE its own MFMA stream of tile s.  The ablation ladder (profiles/r03_strip_ladder.txt) showed
// why: a wave that streams MFMAs leaves the other waves of its SIMD no vector issue slots, so the ping-pong partner's write-out
// ran AFTER the stream, not beside it (slot = 4.6 k cycles of MFMAs + 1.0-1.9 k of tail; the MFMA-only build still needed the
// yield nops' 7 us); one wave's own stream, on the other hand, has 24 of every 32 MFMA cycles free for 4-cycle instructions
// (MI355X_MICROARCH.md, vector-instruction issue cost).  The slot body is branch-free (one scheduling region per item, fillers
// pinned between the MFMAs with sched_group_barrier), so what the write-out does is compile time: with EPI == 0, SO = 1 raw
// store, 2 raw + BatchNorm partial statistics, 3 + bias, max(., floor) [folded inference convolution], 4 as 3 + residual,
// 5 + residual * (mask > 0); with EPI >= 1 the sums epilogues as above (SO = 1).
// DAM_STRIP_2WG (experiment, profiles/r05_strip_two_per_cu.txt): the self-overlapped 16-channel forms compiled for FOUR waves per
// SIMD (<= 128 registers; the second __launch_bounds__ argument is HIP's minimum waves per execution unit) so that two 8-wave
// workgroups share a CU -- one's prologue, loader stalls and tail under the other's MFMA stream.
#ifdef DAM_STRIP_2WG
#define DAM_STRIP_WAVES_PER_EU(SO_, EPI_, NCH_) (((SO_) && (EPI_) <= DAM_STRIP_2WG && (NCH_) == 1) ? 4 : 1)
#else
#define DAM_STRIP_WAVES_PER_EU(SO_, EPI_, NCH_) 1
#endif
template <int MB, int NB, int NCH, bool T33, bool LW, int EPI, int SO>
__global__ __launch_bounds__(SO ? 512 : STRIP_THREADS, DAM_STRIP_WAVES_PER_EU(SO, EPI, NCH)) void conv_strip_kernel(const ConvGeo g, const StripGeo sg, const float* __restrict__ X,
                                                         const float4* __restrict__ Wp, const float* __restrict__ bias,
                                                         float* __restrict__ Y, const float* __restrict__ res,
                                                         const float* __restrict__ res_mask, float* __restrict__ stats,
                                                         const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                         const BnFinArgs fin, const BnBwdEpi bwd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: role tests and loader arithmetic on the scalar ALU
    const int j = lane & 15, kq = lane >> 4;
    constexpr int MW = 16 * MB, TM = 4 * MW;
    constexpr int NT = SO ? 512 : STRIP_THREADS;       // threads; compute waves; role index of the loader waves
    constexpr int NCW = SO ? 4 : 8, LGRP = SO ? 1 : 2;
    const int img = blockIdx.z, nb0 = blockIdx.y * NB;
    const int HoWo = g.Ho * g.Wo;
    const int t_begin = blockIdx.x * sg.tpw;
    int t_end = t_begin + sg.tpw;
    if (t_end > sg.tiles_m) t_end = sg.tiles_m;
    const int RB = g.PWT * 64;                 // bytes of one ring row of one chunk plane
    const int CHB = sg.NR * RB;                // bytes of one chunk plane
    const int RH = sg.RH;                      // input rows touched by one output row
    const float* ximg = X + (size_t)img * g.H * g.W * g.C;
    // fused input affine (the producer's BatchNorm + ReLU applied while the rows are written to LDS): a lane always holds
    // channel quad (lane & 3) of a cell, so its scale / shift values are two registers per chunk for the whole kernel
    const bool has_aff = in_scale != nullptr, relu_in = g.relu_in != 0, relu_out = g.relu_out != 0;
    v4f scq[NCH], shq[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        scq[c] = has_aff ? *reinterpret_cast<const v4f*>(in_scale + c * 16 + (lane & 3) * 4) : (v4f){1.f, 1.f, 1.f, 1.f};
        shq[c] = has_aff ? *reinterpret_cast<const v4f*>(in_shift + c * 16 + (lane & 3) * 4) : (v4f){0.f, 0.f, 0.f, 0.f};
    }
#ifdef DAM_STAMPS
    int stamp_i = 0;
#endif
    DAM_STAMP(1);


    // rows [lo, hi] of the input needed by tile t
    auto tile_rows = [&](int t, int& lo, int& hi) {
        const int p0 = t * TM;
        int p1 = p0 + TM - 1;
        if (p1 > HoWo - 1) p1 = HoWo - 1;
        lo = (int)__umulhi((unsigned)p0, sg.wo_magic) * g.s + g.r0;
        hi = (int)__umulhi((unsigned)p1, sg.wo_magic) * g.s + g.r0 + RH - 1;
    };
    // Fetch input rows [lo, hi] into the ring with plain vector loads: pieces first + part, + nparts, ... (one piece =
    // one wave-wide 1 KB request = 16 column slots of one (row, chunk) plane), STRIP_PU pieces requested before the
    // first is written.  Loads are unconditional from clamped addresses; unused / out-of-tensor lanes become zeros by
    // select (a conditional around a load costs an s_waitcnt per piece).
    const int groups_per_plane = (g.PWT * 4 + 63) >> 6;
    const int pieces_per_row = g.nchunks * groups_per_plane;
    const float inv_ppr = 1.0f / (float)pieces_per_row, inv_gpp = 1.0f / (float)groups_per_plane;
    float4 lv[STRIP_PU];
    int ldst[STRIP_PU], laff[STRIP_PU];
#pragma unroll
    for (int u = 0; u < STRIP_PU; ++u) ldst[u] = -1;
    RowLoad rl;
    rl.ximg = ximg; rl.H = g.H; rl.W = g.W; rl.C = g.C; rl.s = g.s; rl.c0 = g.c0; rl.PWs = g.PWs; rl.PWin = g.PWin; rl.PWT = g.PWT;
    rl.nchunks = g.nchunks; rl.RB = RB; rl.CHB = CHB; rl.NR = sg.NR; rl.ring_off = sg.ring_off;
    rl.ppr = pieces_per_row; rl.gpp = groups_per_plane; rl.inv_ppr = inv_ppr; rl.inv_gpp = inv_gpp;
    rl.m_ppr = (int)(65536.0f * inv_ppr) + 1; rl.m_gpp = (int)(65536.0f * inv_gpp) + 1;

    const int grp = wave >> 2, cw = wave & 3;          // role: 0/1 = compute group A/B, 2 = loader; wave index inside the role
    const int n_tiles = t_end - t_begin;
    const int n_slots = (n_tiles + 2) & ~1;            // tile s is computed in slot s and written out in slot s+1; padded to even
    // ---- row loader: per-lane constants, register sets and the request / commit macros (used by the loader waves' slot loop
    //      and, in the self-overlapped form, by every wave for the workgroup's first tile)
    int lo0, hi0;
    tile_rows(t_begin, lo0, hi0);
    // HS = 2 (16-channel layers, self-overlapped form): a tile adds ~2 planes, so with whole planes two of the four loader waves
    // carried all the pieces -- and all the VALU of a fused input affine (72 instructions per slot on the SIMDs of two compute
    // waves, the other two idle: +8 us per launch).  A loader's unit is then HALF a plane: the even or the odd 1 KB pieces
    // (wave parity), every wave the same 4-5 pieces.
#ifdef DAM_STRIP_NO_HS          // timing A/B (tools/build_variant.sh)
    constexpr int HS = 1;
#else
    constexpr int HS = (NCH == 1 && !LW && SO != 0) ? 2 : 1;
#endif
    constexpr int KP = (NCH == 1 ? 1 : (LW ? 2 : 3)) * HS;    // units (planes / half planes) per loader wave per tile
    constexpr int GPPF = NCH == 1 ? (LW ? 14 : 9) : (LW ? 7 : 5);  // 1 KB pieces per plane (host: groups_per_plane <= GPPF)
    constexpr int GPP = (GPPF + HS - 1) / HS;                 // ... per unit
    const int hsel = HS == 2 ? (wave & 1) : 0;
    int loffb[GPP];
    unsigned long long cmask[GPP];
#pragma unroll
    for (int gu = 0; gu < GPP; ++gu) {
        const int gi = gu * HS + hsel;
        const int L = gi * 64 + lane, slot = L >> 2, quad = L & 3;
        int pw = slot;
        if (g.s != 1) pw = slot < g.PWs ? 2 * slot : 2 * (slot - g.PWs) + 1;
        const int iw = pw + g.c0;
        const bool ok = gi < groups_per_plane && slot < g.PWT && pw < g.PWin && iw >= 0 && iw < g.W;
        const int iwc = iw < 0 ? 0 : (iw >= g.W ? g.W - 1 : iw);
        loffb[gu] = (iwc * g.C + quad * 4) * 4;
        cmask[gu] = __ballot(ok);
    }
    // buffer addressing: scalar resource (this image) + scalar plane offset + per-lane column offset, no VALU
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, g.H * g.W * g.C * 4, 0x00020000);
    const int lane16 = lane * 16 + hsel * 1024;
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
    const float relu_lo = relu_in ? 0.f : -__builtin_inff();       // (a run-time `if (relu_in)` became four v_cndmask per piece on the loaders)
    const v4f relu_lo4 = {relu_lo, relu_lo, relu_lo, relu_lo};
    v4f lvA[KP][GPP], lvB[KP][GPP];
    int dstA[KP], dstB[KP];                          // scalar: ring byte offset of the plane | 1 << 30 if the row is
    int ccA[KP], ccB[KP];                            // outside the tensor (zeros are written), -1 = nothing to write; chunk
#pragma unroll
    for (int k = 0; k < KP; ++k) { dstA[k] = -1; dstB[k] = -1; ccA[k] = 0; ccB[k] = 0; }
    // TIMING EXPERIMENT ONLY (-DDAM_DIAG_DXHAT=1|2, tools/dxhat_ladder.py; results are wrong): what it would cost this kernel to form
    // its INPUT operand dc = a * (dy . mask) + b * c + k (BatchNorm backward: the bn_bwd_apply launch folded into the loaders)
    // itself -- every plane is accompanied by a second plane from another tensor (read through the `bias` pointer: real HBM
    // traffic); 1 = the loads and one fma per quad, 2 = the full arithmetic with the mask recomputed from the second stream.
#ifdef DAM_DIAG_DXHAT
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(bias) + (size_t)img * g.H * g.W * g.C, 0, g.H * g.W * g.C * 4, 0x00020000);
    v4f lvcA[KP][GPP], lvcB[KP][GPP];
#define DAM_SDX_ON 1
#define DAM_SDX_REQ(LVC_, K_, SOFF_)                                                                                       \
    _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi)                                                                     \
        LVC_[K_][gi] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(crsrc, loffb[gi], SOFF_, 0))
#if DAM_DIAG_DXHAT == 1
#define DAM_SDX_COMBINE(LV_, LVC_, K_, GI_) __builtin_elementwise_fma(LVC_[K_][GI_], shq[0], LV_[K_][GI_])
#else
#define DAM_SDX_COMBINE(LV_, LVC_, K_, GI_)                                                                                \
    ([&]() {                                                                                                               \
        const v4f c_ = LVC_[K_][GI_], dy_ = LV_[K_][GI_];                                                                  \
        const v4f m_ = __builtin_elementwise_fma(c_, scq[0], shq[0]);                                                      \
        v4f dz_;                                                                                                           \
        dz_.x = m_.x > 0.f ? dy_.x : 0.f; dz_.y = m_.y > 0.f ? dy_.y : 0.f;                                                \
        dz_.z = m_.z > 0.f ? dy_.z : 0.f; dz_.w = m_.w > 0.f ? dy_.w : 0.f;                                                \
        return __builtin_elementwise_fma(dz_, scq[0], __builtin_elementwise_fma(c_, shq[0], relu_lo4));                    \
    }())
#endif
#else
#define DAM_SDX_ON 0
#define DAM_SDX_REQ(LVC_, K_, SOFF_) do { } while (0)
#define DAM_SDX_COMBINE(LV_, LVC_, K_, GI_) (LV_[K_][GI_])
#define lvcA lvA
#define lvcB lvB
#endif
    const int chs = NCH == 1 ? 0 : 1;
    int loaded_hi;
    { int lo; tile_rows(t_begin, lo, loaded_hi); }
#define DAM_STRIP_REQUEST(K_, LV_, DST_, CC_, LVC_)                                                                          \
    do {                                                                                                                   \
        int first_ = 0, planes_ = 0;                                                                                       \
        if ((K_) < n_tiles) {                                                                                              \
            int lo_, hi_;                                                                                                  \
            tile_rows(t_begin + (K_), lo_, hi_);                                                                           \
            first_ = loaded_hi + 1 > lo_ ? loaded_hi + 1 : lo_;                                                            \
            if (hi_ >= first_) { planes_ = (hi_ - first_ + 1) << chs; loaded_hi = hi_; }                                   \
        }                                                                                                                  \
        _Pragma("unroll") for (int k = 0; k < KP; ++k) {                                                                   \
            const int pl_ = (cw + STRIP_LOADERS * k) / HS;                                                                 \
            const bool used_ = pl_ < planes_;                                                                              \
            const int cc_ = pl_ & (NCH - 1), ih_ = first_ + (pl_ >> chs);                                                  \
            const bool rowok_ = used_ && ih_ >= 0 && ih_ < g.H;                                                            \
            const int soff_ = ((rowok_ ? ih_ : 0) * g.W * g.C + (used_ ? cc_ : 0) * 16) * 4;                               \
            _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi)                                                             \
                LV_[k][gi] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, loffb[gi], soff_, 0));   \
            DAM_SDX_REQ(LVC_, k, soff_);                                                                                   \
            DST_[k] = used_ ? ((cc_ * CHB + ((ih_ + sg.ring_off) & (sg.NR - 1)) * RB) | (rowok_ ? 0 : 1 << 30)) : -1;     \
            CC_[k] = cc_;                                                                                                  \
        }                                                                                                                  \
    } while (0)
#define DAM_STRIP_WRITE(ADDR_, DATA_, GI_)                                                                                 \
    asm volatile("s_mov_b64 exec, %2\n\tds_write_b128 %0, %1 offset:%3\n\ts_mov_b64 exec, -1"                              \
                 : : "v"(ADDR_), "v"(DATA_), "s"(cmask[GI_]), "n"((GI_) * 1024 * HS) : "memory")
#define DAM_STRIP_COMMIT(LV_, DST_, CC_, LVC_) DAM_STRIP_COMMIT_N(KP, LV_, DST_, CC_, LVC_)
#define DAM_STRIP_COMMIT_N(KPX_, LV_, DST_, CC_, LVC_)                                                                        \
    do {                                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < (KPX_); ++k) {                                                                   \
            if (DST_[k] >= 0) {                                                                                            \
                const int va_ = lane16 + (DST_[k] & 0x3fffffff);                                                           \
                if (!(DST_[k] >> 30)) {                                                                                    \
                    if (has_aff) {             /* 8 VALU per piece (4 fma + 4 max), ~40 per loader wave and slot */        \
                        const v4f sc_ = (NCH > 1 && CC_[k] > 0) ? scq[NCH - 1] : scq[0];                                   \
                        const v4f sh_ = (NCH > 1 && CC_[k] > 0) ? shq[NCH - 1] : shq[0];                                   \
                        _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi) {                                               \
                            v4f v_ = __builtin_elementwise_fma(LV_[k][gi], sc_, sh_);                                      \
                            v_ = __builtin_elementwise_max(v_, relu_lo4);   /* max(., -inf) when there is no ReLU: no selects */ \
                            DAM_STRIP_WRITE(va_, v_, gi);                                                                  \
                        }                                                                                                  \
                    } else if (DAM_SDX_ON) {                                                                               \
                        _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi) {                                               \
                            const v4f v_ = DAM_SDX_COMBINE(LV_, LVC_, k, gi);                                              \
                            DAM_STRIP_WRITE(va_, v_, gi);                                                                  \
                        }                                                                                                  \
                    } else {                                                                                               \
                        _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi) DAM_STRIP_WRITE(va_, LV_[k][gi], gi);           \
                    }                                                                                                      \
                } else {                                                                                                   \
                    _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi) DAM_STRIP_WRITE(va_, zero4, gi);                    \
                }                                                                                                          \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)
#ifdef DAM_DIAG_NO_LOAD        // timing experiments only (results are wrong)
#undef DAM_STRIP_REQUEST
#define DAM_STRIP_REQUEST(K_, LV_, DST_, CC_, LVC_) do { } while (0)
#endif
    // SO prologue: the rows of the workgroup's FIRST tile as whole planes too, spread over all NT / 64 waves (plane = wave + k * NT/64)
    constexpr int KP0 = (NCH == 1 ? 1 : 2) * HS;
#define DAM_STRIP_REQUEST0(LV_, DST_, CC_, LVC_)                                                                             \
    do {                                                                                                                   \
        const int planes_ = (hi0 - lo0 + 1) << chs;                                                                        \
        _Pragma("unroll") for (int k = 0; k < KP0; ++k) {                                                                  \
            const int pl_ = (wave + (NT / 64) * k) / HS;                                                                   \
            const bool used_ = pl_ < planes_;                                                                              \
            const int cc_ = pl_ & (NCH - 1), ih_ = lo0 + (pl_ >> chs);                                                     \
            const bool rowok_ = used_ && ih_ >= 0 && ih_ < g.H;                                                            \
            const int soff_ = ((rowok_ ? ih_ : 0) * g.W * g.C + (used_ ? cc_ : 0) * 16) * 4;                               \
            _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi)                                                             \
                LV_[k][gi] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, loffb[gi], soff_, 0));   \
            DAM_SDX_REQ(LVC_, k, soff_);                                                                                   \
            DST_[k] = used_ ? ((cc_ * CHB + ((ih_ + sg.ring_off) & (sg.NR - 1)) * RB) | (rowok_ ? 0 : 1 << 30)) : -1;     \
            CC_[k] = cc_;                                                                                                  \
        }                                                                                                                  \
    } while (0)

    // Prologue: the rows of the first tile are requested first (every wave, into registers), the ring is zeroed and the
    // weights copied while they are in flight, then the rows are written.
    const int total0 = (hi0 - lo0 + 1) * pieces_per_row;
    // thin layers: the packed weights of this N tile are small; keep them in LDS so that the MFMA loop never waits on L2
    const int w_base = CHB * g.nchunks;       // always address LDS as smem + integer offset (a derived pointer variable
                                              // degrades to flat_load, which is slower and also counts on vmcnt)
    // canonical order [a*3+b][chunk][nb][lane] whatever the layer's tap numbering: operand offsets become immediates.
    // All of a thread's weight loads are requested FIRST (clamped addresses, no load inside a conditional), then the rows of
    // the first tile, then the ring is zeroed while both are in flight: one memory latency for the prologue, not three.
    constexpr int WPT = (9 * NCH * NB * 64 + NT - 1) / NT;
    const int n4 = 9 * g.nchunks * NB * 64;
    float4 wreg[WPT];
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const int e = tid + u * NT, ec = e < n4 ? e : n4 - 1;
        const int ln = ec & 63, nb = (ec >> 6) % NB, tc = (ec >> 6) / NB;
        const int ct = tc / g.nchunks, cc = tc - ct * g.nchunks, a = ct / 3, b = ct - a * 3;
        const bool tap_ok = a < g.nA && b < g.nB;
        const int tap = tap_ok ? g.wt_base + a * g.wt_sa + b * g.wt_sb : g.wt_base;
        wreg[u] = Wp[((size_t)(tap * g.nchunks + cc) * g.NBtot + nb0 + nb) * 64 + ln];
        if (!tap_ok) wreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    DAM_STAMP(9);
    v4f lv0[SO ? KP0 : 1][SO ? GPP : 1];
#ifdef DAM_DIAG_DXHAT
    v4f lvc0[SO ? KP0 : 1][SO ? GPP : 1];
#else
#define lvc0 lv0
#endif
    int dst0[KP0], cc0[KP0];
    if constexpr (SO != 0) {
        // rows of the first tile: whole planes over all waves (the per-piece path of the ping-pong form costs 3.4-5 k cycles
        // of address arithmetic per wave); the loader waves' requests for tiles 1 and 2 follow at once, so that tile 1's rows
        // are there long before slot 0 ends (requested after the prologue, the first barrier waited 1.3-3.4 k cycles for them)
        DAM_STRIP_REQUEST0(lv0, dst0, cc0, lvc0);
        if (grp == LGRP) {
#ifdef DAM_STRIP_ONE_SET       // experiment: ONE register set, the rows of tile s+2 are requested in slot s (one slot ahead, not two)
            DAM_STRIP_REQUEST(1, lvA, dstA, ccA, lvcA);
#else
            DAM_STRIP_REQUEST(1, lvB, dstB, ccB, lvcB);
            DAM_STRIP_REQUEST(2, lvA, dstA, ccA, lvcA);
#endif
        }
    } else {
        rows_issue(rl, lo0, hi0, 0, wave, NT / 64, lane, lv, ldst, laff);
    }
    DAM_STAMP(10);
    // zero the whole ring once: border slots stay zero for the lifetime of the workgroup
    for (int e = tid * 16; e < CHB * g.nchunks; e += NT * 16)
        *reinterpret_cast<float4*>(smem + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    DAM_STAMP(11);
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const int e = tid + u * NT;
        if (e < n4) *reinterpret_cast<float4*>(smem + w_base + e * 16) = wreg[u];
    }
    DAM_STAMP(12);
    __syncthreads();
    DAM_STAMP(2);
    if constexpr (SO != 0) {
        DAM_STRIP_COMMIT_N(KP0, lv0, dst0, cc0, lvc0);
    } else {
        rows_commit<NCH>(smem, lv, ldst, laff, has_aff, scq, shq, relu_in);
        for (int base = (NT / 64) * STRIP_PU; base < total0; base += (NT / 64) * STRIP_PU) {
            rows_issue(rl, lo0, hi0, base, wave, NT / 64, lane, lv, ldst, laff);
            rows_commit<NCH>(smem, lv, ldst, laff, has_aff, scq, shq, relu_in);
        }
    }
    // SO: per-tile geometry tables.  A compute wave's scalar pixel decode (magic division, ring-row masks, multiplies: ~90
    // dependent scalar instructions per tile) sits in its own in-order stream in front of its MFMAs -- measured 800 of a 6.0 k
    // cycle slot.  The loader wave paired with compute wave cw does that arithmetic instead, all (pixel block, field) pairs at
    // once in its lanes (~25 vector instructions per tile), and leaves 8 ints per pixel block in LDS: {scalar part of the output
    // offset, lanes-on-this-row threshold, three operand row bases for lanes on the block's first output row, three for lanes
    // wrapped to the next}; the compute wave reads them as two broadcast ds_read_b128.  FOUR buffers (tile & 3): the loaders
    // write tile s+2 during slot s while the compute waves read tile s+1 -- and tile 0 BEFORE slot 0, beside the loaders' first
    // write (with two buffers that write landed in tile 0's entries: a late compute wave, e.g. on a cold instruction cache,
    // read tile 2's geometry for its first tile -- single pixel blocks wrong on first launches).
    //
    // WHY `tile & 3` CANNOT ALIAS (the slip bound).  Every wave of the workgroup executes the SAME sequence of workgroup barriers:
    // the two __syncthreads() of the prologue (P1 behind the ring clear / weight copy, P2 behind the first tile's rows and the
    // tables of tiles 0 and 1), then exactly ONE raw s_barrier per slot -- the loaders' paired loop runs n_slots of them, a
    // compute wave n_tiles SO_SLOTs / SO_LAST plus the padding barriers up to n_slots.  A wave leaves barrier k only when all
    // eight have arrived at it, so at any instant all waves are inside the same slot sigma (between slot barrier sigma - 1 and
    // sigma, P2 counting as barrier -1) or have finished it and wait at its barrier: NO wave is ever a slot ahead of another.
    // Every wave drains its LDS traffic (s_waitcnt lgkmcnt(0)) in front of each barrier, so a table read or write belongs
    // entirely to the slot it was issued in.  Inside slot sigma:
    //     the loader paired with compute wave cw WRITES the entry of tile sigma + 2           (SO_TABLE(s + 2) / SO_TABLE(s + 3)),
    //     compute wave cw READS the entry of tile sigma + 1 (fillers: SO_TABREAD(sa_ + 1)) and, in slot 0 only, of tile 0
    //     (the SO_TABREAD(0, mb) in front of the first SO_SLOT; tiles 0 and 1 were written before P2).
    // Concurrent accesses therefore touch tiles {sigma + 2} and {sigma + 1} (+ {0} when sigma = 0): three distinct values mod 4
    // in slot 0, two in every other slot -- never the same buffer.  With TWO buffers slot 0's write of tile 2 hit tile 0's entry
    // while a compute wave that left P2 late (cold instruction cache on a first launch) had not read it yet: the failure of
    // round 3 (gpurun_out/r3_t8.log); three buffers would do, four keep the index a mask.  The loaders' row REQUESTS for tiles
    // 1 and 2 in the prologue are register loads from the read-only input: they write no LDS, so their being early changes when
    // data travels, not what any wave reads -- the ring writes (commits) of tile sigma + 1 stay in slot sigma, where the host's
    // ring sizing (StripGeo::NR: the rows of two consecutive tiles never alias) covers them exactly as in the ping-pong form.
    // tests/test_strip_diag_gpu.py runs the layer1 / layer2 shapes of the C3 step through a build that checks this on the device.
    const int tab_base = CHB * g.nchunks + 9 * g.nchunks * NB * 1024;
#ifdef DAM_STRIP_DIAG_TAGS
#define SO_TAG(T_) ((((T_) + 1) & 0x7fff) << 16)       /* rides in the upper half of the lanes-on-this-row field (< 2^16) */
#else
#define SO_TAG(T_) 0
#endif
#define SO_TABLE(T_)                                                                                                      \
    do {                                                                                                                  \
        if (lane < 8 * MB) {                                                                                              \
            const int mbt_ = lane >> 3, f_ = lane & 7;                                                                    \
            const int pm_ = (t_begin + (T_)) * TM + (wave & 3) * MW + mbt_ * 16;                                          \
            const int oh_ = (int)__umulhi((unsigned)pm_, sg.wo_magic), ow_ = pm_ - oh_ * g.Wo;                            \
            const bool k1_ = f_ >= 5;                                                                                     \
            const int a_ = k1_ ? f_ - 5 : f_ - 2;                                                                         \
            const int rr_ = oh_ * g.s + sg.ring_off + g.off_h + a_ * g.step_h + (k1_ ? g.s : 0);                          \
            const int kv_ = (rr_ & (sg.NR - 1)) * RB + (ow_ + g.off_w - g.c0 - (k1_ ? g.Wo : 0)) * 64;                    \
            const int so_ = oh_ * (g.os * g.OWt * g.N * 4) + ow_ * (g.os * g.N * 4);                                      \
            const int val_ = f_ == 0 ? so_ : (f_ == 1 ? ((g.Wo - ow_) | SO_TAG(T_)) : kv_);                               \
            *reinterpret_cast<int*>(smem + tab_base + ((((wave & 3) * 4 + ((T_) & 3)) * MB * 8 + lane) << 2)) = val_;     \
        }                                                                                                                 \
    } while (0)
    if constexpr (SO != 0) {
        if (grp == LGRP) { SO_TABLE(0); SO_TABLE(1); }
    }
    __syncthreads();
    DAM_STAMP(3);

    if (grp == LGRP) {
        // ================= loader waves: their own slot loop (same number of barriers as the compute waves) =============
        // Slot s writes the rows tile s+1 adds (requested two slots earlier: HBM latency under this load is ~4 us, a full
        // slot) and requests those of tile s+3.  Two register sets alternate; tile k uses set k & 1.
        //
        // A wave that streams MFMAs owns the SIMD's vector issue port (tools/mfma_valu_mix.hip: a co-resident wave's VALU
        // instructions make no progress at all until the MFMA stream pauses, whatever s_setprio says), so every VALU
        // instruction of a loader is paid for in MFMA time.  The loader therefore works in whole (row, chunk) planes:
        // everything that depends on the plane is wave-uniform and lives on the scalar ALU, everything that depends on the
        // lane (column pattern of piece gi: byte offset inside a row, which lanes are real columns) is computed once, and
        // a piece costs no VALU instruction at all: buffer_load with scalar base + per-lane offset, ds_write with the
        // piece's column mask in EXEC and an immediate offset.
        if constexpr (SO == 0) {
            DAM_STRIP_REQUEST(1, lvB, dstB, ccB, lvcB);
            DAM_STRIP_REQUEST(2, lvA, dstA, ccA, lvcA);
        }
        // slots come in pairs (n_slots is even) so that no load sits inside a conditional: the compiler then knows that the
        // set being written is the older of the two in flight and waits with vmcnt(pieces of the other set), not vmcnt(0)
#ifdef DAM_STRIP_ONE_SET
        if constexpr (SO != 0) {
        for (int s = 0; s < n_slots; s += 2) {
            DAM_STRIP_COMMIT(lvA, dstA, ccA, lvcA);       // tile s+1
            DAM_STRIP_REQUEST(s + 2, lvA, dstA, ccA, lvcA);
            SO_TABLE(s + 2);
            DAM_STAMP(4);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            DAM_STRIP_COMMIT(lvA, dstA, ccA, lvcA);       // tile s+2
            DAM_STRIP_REQUEST(s + 3, lvA, dstA, ccA, lvcA);
            SO_TABLE(s + 3);
            DAM_STAMP(4);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        } else
#endif
        {
        for (int s = 0; s < n_slots; s += 2) {
            DAM_STRIP_COMMIT(lvB, dstB, ccB, lvcB);       // tile s+1
            DAM_STRIP_REQUEST(s + 3, lvB, dstB, ccB, lvcB);
            if constexpr (SO != 0) SO_TABLE(s + 2);  // read by the compute waves in slot s+1
            DAM_STAMP(4);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            DAM_STRIP_COMMIT(lvA, dstA, ccA, lvcA);       // tile s+2
            DAM_STRIP_REQUEST(s + 4, lvA, dstA, ccA, lvcA);
            if constexpr (SO != 0) SO_TABLE(s + 3);
            DAM_STAMP(4);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        }
#undef DAM_STRIP_REQUEST
#undef DAM_STRIP_REQUEST0
#undef DAM_STRIP_WRITE
#undef DAM_STRIP_COMMIT
#undef DAM_STRIP_COMMIT_N
#undef DAM_SDX_ON
#undef DAM_SDX_REQ
#undef DAM_SDX_COMBINE
#ifndef DAM_DIAG_DXHAT
#undef lvcA
#undef lvcB
#undef lvc0
#endif
    }

    // BatchNorm partial statistics of this wave's outputs: shifted sums per lane (channels 4*kq..+3 of block nb), kept as
    // float pairs so that the three updates per value are v_pk_add / v_pk_add / v_pk_fma (two values per instruction)
    v2f st_nk[NB][2], st_s1[NB][2], st_s2[NB][2];       // -shift, sum(v - shift), sum((v - shift)^2)
    int st_n = 0;
    bool st_have = false;                               // wave-uniform: the shift has been taken from the first outputs
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h) { st_nk[nb][h] = (v2f){0.f, 0.f}; st_s1[nb][h] = (v2f){0.f, 0.f}; st_s2[nb][h] = (v2f){0.f, 0.f}; }

    v4f acc[MB][NB];            // results of this group's current tile: produced in one slot, written out in the next
    int wflag[MB];              // 1 where this lane's pixel of block mb lies on the output row after the block's first pixel
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) wflag[mb] = 0;

    // Pixel geometry is decoded on the scalar ALU (an MFMA stream owns the vector issue port, see the loader comment):
    // block mb of this wave starts at pixel pm = (oh_m, ow_m), lane j holds pixel pm + j, which wraps to the next output
    // row at most once (Wo >= 16).  Every per-lane address is  lane constant + scalar + wflag * scalar.
    const int lane_x = j * 64 + kq * 16;                                           // LDS: column slot, channel quad
    const int oA = g.os * g.OWt * g.N * 4, oB = g.os * g.N * 4;                    // output bytes per output row / column
    const int oD = oA - g.Wo * oB;
    const int lane_o = (j * g.os * g.N + kq * 4 + nb0 * 16) * 4 + (g.oo_h * g.OWt + g.oo_w) * g.N * 4;
    const int img_bytes_o = g.OHt * g.OWt * g.N * 4;
    const __amdgpu_buffer_rsrc_t yrsrc =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(Y) + (size_t)img * img_bytes_o, 0, img_bytes_o, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(res)) + (res ? (size_t)img * img_bytes_o : 0), 0, res ? img_bytes_o : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(res_mask)) + (res_mask ? (size_t)img * img_bytes_o : 0), 0, res_mask ? img_bytes_o : 0, 0x00020000);
    // EPI == 2: the sums' x (same geometry as the output) and the residual's sign bytes (one per channel quad)
    const __amdgpu_buffer_rsrc_t x2rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(bwd.x)) + (EPI >= 2 ? (size_t)img * img_bytes_o : 0), 0, EPI >= 2 ? img_bytes_o : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t bbrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(res_mask)) + (EPI >= 2 ? (size_t)img * (img_bytes_o >> 4) : 0), 0,
        EPI >= 2 ? (img_bytes_o >> 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ubrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(bwd.mask_bits)) + (EPI == 3 ? (size_t)img * (img_bytes_o >> 4) : 0), 0,
        EPI == 3 ? (img_bytes_o >> 4) : 0, 0x00020000);
    v4f bias4[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        bias4[nb] = bias ? *reinterpret_cast<const v4f*>(bias + (nb0 + nb) * 16 + kq * 4) : (v4f){0.f, 0.f, 0.f, 0.f};

    // BatchNorm-backward sums epilogue (BnBwdEpi): `res` is the BatchNorm's input x; per lane the two affine maps of its
    // channel quad: mask = x*mscale + mshift > 0, xhat = x*invstd - mean*invstd.  The sums live in st_s1 / st_s2.
    constexpr bool epi_bwd = EPI != 0;
    v4f bw_ms[EPI ? NB : 1], bw_mh[EPI ? NB : 1], bw_k1[EPI ? NB : 1], bw_k2[EPI ? NB : 1];
#pragma unroll
    for (int nb = 0; nb < (EPI ? NB : 0); ++nb) {
        const int ch = (nb0 + nb) * 16 + kq * 4;
        if constexpr (EPI != 3) {
            bw_ms[nb] = *reinterpret_cast<const v4f*>(bwd.mscale + ch);
            bw_mh[nb] = *reinterpret_cast<const v4f*>(bwd.mshift + ch);
        }
        bw_k1[nb] = *reinterpret_cast<const v4f*>(bwd.invstd + ch);
        bw_k2[nb] = -(*reinterpret_cast<const v4f*>(bwd.mean + ch)) * bw_k1[nb];
    }

    // T33: operand row bases of this group's NEXT tile.  They are computed in the group's write-out slot, where the wave
    // otherwise waits for the other group's MFMAs: ~200 scalar instructions that would sit in front of the MFMA stream.
    int base_a[3][MB];
    const int w_lane = w_base + lane * 16;
    // A wave whose next instruction is a VALU one sits behind the other group's queued MFMAs (in-order issue), so this
    // runs last in the slot, after the write-out: 4 scalar instructions + 13 VALU per pixel block (row index and column
    // offset per lane, three ring rows).
#define DAM_STRIP_GEOM(T_)                                                                                                \
    do {                                                                                                                  \
        const int p0g_ = (t_begin + (T_)) * TM + cw * MW;                                                                 \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                                               \
            const int pm_ = p0g_ + mb * 16;                                                                               \
            const int oh_ = (int)__umulhi((unsigned)pm_, sg.wo_magic), ow_ = pm_ - oh_ * g.Wo;                            \
            wflag[mb] = j >= g.Wo - ow_ ? 1 : 0;                                                                          \
            const int col_ = lane_x + (ow_ + g.off_w - g.c0) * 64 - wflag[mb] * (g.Wo * 64);                              \
            const int row_ = wflag[mb] * g.s + (oh_ * g.s + sg.ring_off + g.off_h);                                       \
            _Pragma("unroll") for (int a = 0; a < 3; ++a)                                                                 \
                base_a[a][mb] = ((row_ + a * g.step_h) & (sg.NR - 1)) * RB + col_;                                        \
        }                                                                                                                 \
    } while (0)
    if constexpr (T33) {
        if (grp == 0) DAM_STRIP_GEOM(0);
    }

    // residual / mask operands of the write-out (dgrad of a residual block): requested by the group right after its MFMAs,
    // one slot before they are needed -- in the write-out slot the group's vector instructions only get to issue once the
    // other group's MFMA stream has drained, and a load issued then would put its whole latency into the slot's tail
    constexpr bool RES_PF = MB * NB <= 4;
    v4f res_pf[RES_PF ? MB : 1][RES_PF ? NB : 1], msk_pf[RES_PF && EPI < 2 ? MB : 1][RES_PF && EPI < 2 ? NB : 1];
    v4f x_pf[EPI >= 2 ? MB : 1][EPI >= 2 ? NB : 1];
    int mskb_pf[EPI >= 2 ? MB : 1][EPI >= 2 ? NB : 1], upb_pf[EPI == 3 ? MB : 1][EPI == 3 ? NB : 1];
    static_assert(EPI < 2 || RES_PF, "the residual + sums epilogue is built on the prefetch path");

    for (int s = 0; SO == 0 && grp < 2 && s < n_slots; ++s) {
        if (false) {
        } else if (grp == (s & 1)) {
            // ---------------- MFMA slot of this group: tile s ----------------
#ifdef DAM_DIAG_NO_MFMA
            if (false) {
#else
            if (s < n_tiles) {
#endif
                int ohm[MB], owm[MB];                                 // scalars (generic path: decoded in the slot)
                if constexpr (!T33) {
                    const int p0w = (t_begin + s) * TM + cw * MW;     // scalar: first pixel of this wave
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const int pm = p0w + mb * 16;
                        ohm[mb] = (int)__umulhi((unsigned)pm, sg.wo_magic);
                        owm[mb] = pm - ohm[mb] * g.Wo;
                        wflag[mb] = j >= g.Wo - owm[mb] ? 1 : 0;
                    }
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (v4f){0.f, 0.f, 0.f, 0.f};

                if constexpr (T33) {
                    // 3x3, stride 1, unit column step: 9*NCH items, each MB + NB ds_read_b128 feeding 4*MB*NB MFMAs, every
                    // operand address = per-(tap row, pixel block) register (computed one slot ahead, see DAM_STRIP_GEOM)
                    // + immediate.  Two-deep software pipeline: the operands of item i+1 are requested before the MFMAs of
                    // item i are issued (sched_barrier pins that order).
                    constexpr int NI = 9 * NCH;
                    float4 wa[2][NB], xv[2][MB];
#define DAM_STRIP_LOAD(I_, BUF_)                                                                                          \
    do {                                                                                                                  \
        constexpr int a_ = (I_) / (3 * NCH), b_ = ((I_) / NCH) % 3, cc_ = (I_) % NCH;                                     \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                                 \
            wa[BUF_][nb] = *reinterpret_cast<const float4*>(smem + w_lane + ((((a_ * 3 + b_) * NCH + cc_) * NB + nb) * 1024)); \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                 \
            xv[BUF_][mb] = *reinterpret_cast<const float4*>(smem + (cc_ ? base_a[a_][mb] + CHB : base_a[a_][mb]) + b_ * 64); \
    } while (0)
#define DAM_STRIP_MFMA(BUF_)                                                                                              \
    do {                                                                                                                  \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                                     \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                             \
                _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                         \
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(reinterpret_cast<const float*>(&wa[BUF_][nb])[r],  \
                                                                       reinterpret_cast<const float*>(&xv[BUF_][mb])[r], acc[mb][nb], 0, 0, 0); \
    } while (0)
#ifndef DAM_STRIP_YIELD
#define DAM_STRIP_YIELD asm volatile("s_nop 15\n\ts_nop 15")
#endif
#define DAM_STRIP_ITEM(I_)                                                                                                \
    do {                                                                                                                  \
        if constexpr ((I_) + 1 < NI) DAM_STRIP_LOAD(((I_) + 1 < NI ? (I_) + 1 : 0), ((I_) + 1) & 1);                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        DAM_STRIP_MFMA((I_) & 1);                                                                                         \
        DAM_STRIP_YIELD;                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    } while (0)
                    DAM_STRIP_LOAD(0, 0);
                    DAM_STRIP_ITEM(0); DAM_STRIP_ITEM(1); DAM_STRIP_ITEM(2); DAM_STRIP_ITEM(3); DAM_STRIP_ITEM(4);
                    DAM_STRIP_ITEM(5); DAM_STRIP_ITEM(6); DAM_STRIP_ITEM(7); DAM_STRIP_ITEM(8);
                    if constexpr (NCH == 2) {
                        DAM_STRIP_ITEM(9); DAM_STRIP_ITEM(10); DAM_STRIP_ITEM(11); DAM_STRIP_ITEM(12); DAM_STRIP_ITEM(13);
                        DAM_STRIP_ITEM(14); DAM_STRIP_ITEM(15); DAM_STRIP_ITEM(16); DAM_STRIP_ITEM(17);
                    }
#undef DAM_STRIP_ITEM
#undef DAM_STRIP_MFMA
#undef DAM_STRIP_LOAD
                } else {
                // The whole (tap row a, tap column b, chunk) grid is unrolled (nA, nB <= 3, NCH compile time): an LDS operand
                // address is "per-(pixel, a) base + wave-uniform offset", a weight address is a wave-uniform offset, so an item
                // costs MB v_add + (MB + NB) ds_read_b128 next to its 16*MB*NB/4 MFMAs and the scheduler can slide the reads
                // of the next item under the MFMAs of the current one.  sched_barrier between tap rows bounds the hoisting.
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    if (a < g.nA) {
                        int base_a[MB];
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) {
                            const int r0 = ohm[mb] * g.s + sg.ring_off + g.off_h + a * g.step_h;      // scalar
                            const int R0 = (r0 & (sg.NR - 1)) * RB + owm[mb] * 64;
                            const int R1 = ((r0 + g.s) & (sg.NR - 1)) * RB + (owm[mb] - g.Wo) * 64;
                            base_a[mb] = R0 + (lane_x + wflag[mb] * (R1 - R0));
                        }
#pragma unroll
                        for (int b = 0; b < 3; ++b) {
                            if (b < g.nB) {
                                const int coff = g.off_w + b * g.step_w - g.c0;
                                const int slotoff = g.s == 1 ? coff : (coff & 1) * g.PWs + (coff >> 1);
                                const int tap = a * 3 + b;             // canonical order in LDS
#pragma unroll
                                for (int cc = 0; cc < NCH; ++cc) {
                                    const int co = cc * CHB + slotoff * 64;
                                    const int wo = w_base + (((tap * NCH + cc) * NB) * 64 + lane) * 16;
                                    float4 wa[NB], xv[MB];
#pragma unroll
                                    for (int nb = 0; nb < NB; ++nb) wa[nb] = *reinterpret_cast<const float4*>(smem + wo + nb * 1024);
#pragma unroll
                                    for (int mb = 0; mb < MB; ++mb) xv[mb] = *reinterpret_cast<const float4*>(smem + base_a[mb] + co);
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
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            }
            if constexpr (RES_PF) {
                if (res && s < n_tiles) {
                    const int p0r = (t_begin + s) * TM + cw * MW;
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const int pm = p0r + mb * 16;
                        const int oh_m = (int)__umulhi((unsigned)pm, sg.wo_magic), ow_m = pm - oh_m * g.Wo;
                        const int voff = __builtin_amdgcn_readfirstlane(oh_m * oA + ow_m * oB) + (lane_o + wflag[mb] * oD);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {       // pixels past the image: out of the buffer's range, reads 0
                            res_pf[mb][nb] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, voff + nb * 64, 0, 0));
                            if constexpr (EPI >= 2) {
                                mskb_pf[mb][nb] = (int)__builtin_amdgcn_raw_buffer_load_b8(bbrsrc, (voff + nb * 64) >> 4, 0, 0);
                                x_pf[mb][nb] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(x2rsrc, voff + nb * 64, 0, 0));
                                if constexpr (EPI == 3)
                                    upb_pf[mb][nb] = (int)__builtin_amdgcn_raw_buffer_load_b8(ubrsrc, (voff + nb * 64) >> 4, 0, 0);
                            } else if (res_mask)
                                msk_pf[mb][nb] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(mrsrc, voff + nb * 64, 0, 0));
                        }
                    }
                }
            }
            DAM_STAMP(5);
        } else {
            if (s >= 1 && s <= n_tiles) {
            // ---------------- write-out slot of this group: tile s-1 (computed in the previous slot) ----------------
            const int p0w = (t_begin + s - 1) * TM + cw * MW;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int pm = p0w + mb * 16;                    // scalar
#ifdef DAM_DIAG_NO_WRITEOUT
                const int nvalid = g.B == 12345 ? HoWo - pm : 0;     // runtime-false: keeps the MFMAs alive
#else
                const int nvalid = HoWo - pm;                    // lanes j < nvalid hold pixels of this image
#endif
                if (nvalid <= 0) continue;
                const int oh_m = (int)__umulhi((unsigned)pm, sg.wo_magic), ow_m = pm - oh_m * g.Wo;
                const int voff = __builtin_amdgcn_readfirstlane(oh_m * oA + ow_m * oB) + (lane_o + wflag[mb] * oD);
                if (j < nvalid) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        v4f v = acc[mb][nb] + bias4[nb];
                        if (res) {
                            v4f rv, mv;
                            if constexpr (EPI >= 2) { rv = res_pf[mb][nb]; }
                            else if constexpr (RES_PF) { rv = res_pf[mb][nb]; mv = msk_pf[mb][nb]; }
                            else {
                                rv = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, voff + nb * 64, 0, 0));
                                if (res_mask) mv = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(mrsrc, voff + nb * 64, 0, 0));
                            }
                            if constexpr (EPI >= 2) {   // identity shortcut: + res where the block output was positive (sign bytes)
                                const int b8 = mskb_pf[mb][nb];
                                v.x += (b8 & 1) ? rv.x : 0.f; v.y += (b8 & 2) ? rv.y : 0.f;
                                v.z += (b8 & 4) ? rv.z : 0.f; v.w += (b8 & 8) ? rv.w : 0.f;
                            } else if constexpr (EPI == 1) {    // rv = the BatchNorm's input at this pixel: sums only, v goes out as it is
                                const v4f m = __builtin_elementwise_fma(rv, bw_ms[nb], bw_mh[nb]);
                                const v4f xh = __builtin_elementwise_fma(rv, bw_k1[nb], bw_k2[nb]);
                                v4f dz;
                                dz.x = m.x > 0.f ? v.x : 0.f; dz.y = m.y > 0.f ? v.y : 0.f;
                                dz.z = m.z > 0.f ? v.z : 0.f; dz.w = m.w > 0.f ? v.w : 0.f;
                                st_s1[nb][0] += dz.xy; st_s1[nb][1] += dz.zw;
                                st_s2[nb][0] = __builtin_elementwise_fma(dz.xy, xh.xy, st_s2[nb][0]);
                                st_s2[nb][1] = __builtin_elementwise_fma(dz.zw, xh.zw, st_s2[nb][1]);
                            } else if (res_mask) {
                                v.x += mv.x > 0.f ? rv.x : 0.f; v.y += mv.y > 0.f ? rv.y : 0.f;
                                v.z += mv.z > 0.f ? rv.z : 0.f; v.w += mv.w > 0.f ? rv.w : 0.f;
                            } else {
                                v += rv;
                            }
                        }
                        if (relu_out) v = __builtin_elementwise_max(v, (v4f){0.f, 0.f, 0.f, 0.f});
#ifdef DAM_DIAG_NO_STORE     // timing experiments only: the write-out's VALU work without its stores
                        if (g.B == 12345)
#endif
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), yrsrc, voff + nb * 64, 0, 0);
                        if constexpr (EPI >= 2) {       // v is the gradient reaching relu(bn(x) [+ ..]) of the block input: its two sums
                            const v4f xq = x_pf[mb][nb];
                            const v4f xh = __builtin_elementwise_fma(xq, bw_k1[nb], bw_k2[nb]);
                            v4f dz;
                            if constexpr (EPI == 3) {
                                const int ub = upb_pf[mb][nb];
                                dz.x = (ub & 1) ? v.x : 0.f; dz.y = (ub & 2) ? v.y : 0.f;
                                dz.z = (ub & 4) ? v.z : 0.f; dz.w = (ub & 8) ? v.w : 0.f;
                            } else {
                                const v4f m = __builtin_elementwise_fma(xq, bw_ms[nb], bw_mh[nb]);
                                dz.x = m.x > 0.f ? v.x : 0.f; dz.y = m.y > 0.f ? v.y : 0.f;
                                dz.z = m.z > 0.f ? v.z : 0.f; dz.w = m.w > 0.f ? v.w : 0.f;
                            }
                            st_s1[nb][0] += dz.xy; st_s1[nb][1] += dz.zw;
                            st_s2[nb][0] = __builtin_elementwise_fma(dz.xy, xh.xy, st_s2[nb][0]);
                            st_s2[nb][1] = __builtin_elementwise_fma(dz.zw, xh.zw, st_s2[nb][1]);
                        }
#if !defined(DAM_STAMPS) && !defined(DAM_DIAG_NO_STATS)
                        if (stats && !epi_bwd) {
                            if (!st_have) { st_nk[nb][0] = -v.xy; st_nk[nb][1] = -v.zw; }
                            const v2f d0 = v.xy + st_nk[nb][0], d1 = v.zw + st_nk[nb][1];
                            st_s1[nb][0] += d0; st_s1[nb][1] += d1;
                            st_s2[nb][0] = __builtin_elementwise_fma(d0, d0, st_s2[nb][0]);
                            st_s2[nb][1] = __builtin_elementwise_fma(d1, d1, st_s2[nb][1]);
                        }
#endif
                    }
                    if (stats) ++st_n;
                }
                st_have = true;
            }
            }
#ifndef DAM_DIAG_NO_GEOM
            if constexpr (T33) {
                if (s + 1 < n_tiles) DAM_STRIP_GEOM(s + 1);        // this group's next MFMA slot
            }
#endif
            DAM_STAMP(6);
        }
        // slot boundary: the loaders' rows are in LDS (their ds_writes waited on the loads), the MFMA group's LDS reads
        // are consumed.  Raw s_barrier: the write-out group does not drain its output stores here.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        DAM_STAMP(7);
    }

    // =========================== self-overlapped compute waves (SO != 0; see the template comment) ===========================
    if constexpr (SO != 0) {
        static_assert(T33, "the self-overlapped form is built on the compile-time 3x3 item grid");
        if (grp == 0) {
            constexpr bool ST = EPI == 0 && SO == 2, BIAS = EPI == 0 && (SO == 3 || SO == 4), RADD = EPI == 0 && SO == 4,
                           RMSK = EPI == 0 && SO == 5, NEED_R = EPI != 0 || RADD || RMSK;
            constexpr int NI = 9 * NCH;
#ifdef DAM_DIAG_SO_FROZEN          // timing experiments only: the fillers see tile 1 in every slot -- their scalar pixel decode
            constexpr bool SO_FROZEN = true;      // becomes loop invariant (results are wrong)
#else
            constexpr bool SO_FROZEN = false;
#endif
#ifdef DAM_DIAG_SO_NO_STATS_UNIT   // timing experiments only: no statistics arithmetic in the write-out units
            constexpr bool SO_NOSTU = true;
#else
            constexpr bool SO_NOSTU = false;
#endif
#ifdef DAM_SO_LATE_PREFETCH        // A/B build: the write-out operands of a tile requested in one filler at the end of its slot
            constexpr bool SO_LATE_PF = true;
#else
            constexpr bool SO_LATE_PF = false;
#endif
#ifdef DAM_DIAG_SO_NOFILL          // timing experiments only: the bare MFMA + operand-read stream (results are wrong)
            constexpr bool SO_NOFILL = true;
#else
            constexpr bool SO_NOFILL = false;
#endif
            static_assert(NI >= 2 * MB + 1, "one filler per item: MB write-out units, MB geometry units, the operand prefetch");
            v4f accX[2][MB][NB];            // [parity of the tile]: produced in one slot, written out inside the next
            int voffX[2][MB];               // lane part of the output byte offset of pixel block mb (row wrap folded in)
            int baseX[2][3][MB];            // LDS operand row bases
            v4f rpf[NEED_R ? MB : 1][NEED_R ? NB : 1], mpf[RMSK ? MB : 1][RMSK ? NB : 1], xpf[EPI >= 2 ? MB : 1][EPI >= 2 ? NB : 1];
            int mbpf[EPI >= 2 ? MB : 1][EPI >= 2 ? NB : 1], ubpf[EPI == 3 ? MB : 1][EPI == 3 ? NB : 1];
            const float floor_out = relu_out ? 0.f : -3.0e38f;
            const int lane_oD = lane_o + oD;
            float4 wa[2][NB], xv[2][MB];

            // geometry of pixel block MBI_ of tile T_ (parity P_) from the loader's table: SO_TABREAD requests the 8 ints (two
            // broadcast reads, placed a few items ahead of their use), SO_GEOM turns them into per-lane offsets: one compare and
            // eight select / add instructions, no scalar arithmetic
            v4i tq[MB][2];
#ifdef DAM_STRIP_DIAG_TAGS
            int tq_want[MB];                // the tag the entry requested by SO_TABREAD must carry
#define SO_TAG_EXPECT(T_, MBI_) tq_want[MBI_] = SO_TAG(T_)
#define SO_TAG_CHECK(QA_, MBI_)                                                                                           \
    do {                                                                                                                  \
        if (lane == 0) {                                                                                                  \
            atomicAdd(&strip_diag_counters[0], 1u);                                                                       \
            if (((QA_).y & 0x7fff0000) != tq_want[MBI_]) atomicAdd(&strip_diag_counters[1], 1u);                          \
        }                                                                                                                 \
        (QA_).y &= 0xffff;                                                                                                \
    } while (0)
#else
#define SO_TAG_EXPECT(T_, MBI_) do { } while (0)
#define SO_TAG_CHECK(QA_, MBI_) do { } while (0)
#endif
#define SO_TABREAD(T_, MBI_)                                                                                              \
    do {                                                                                                                  \
        const int ta_ = tab_base + ((cw * 4 + ((T_) & 3)) * MB + (MBI_)) * 32;                                            \
        tq[MBI_][0] = *reinterpret_cast<const v4i*>(smem + ta_);                                                          \
        tq[MBI_][1] = *reinterpret_cast<const v4i*>(smem + ta_ + 16);                                                     \
        SO_TAG_EXPECT(T_, MBI_);                                                                                          \
    } while (0)
#define SO_GEOM(P_, MBI_)                                                                                                 \
    do {                                                                                                                  \
        v4i qa_ = tq[MBI_][0];                                                                                            \
        const v4i qb_ = tq[MBI_][1];                                                                                      \
        SO_TAG_CHECK(qa_, MBI_);                                                                                          \
        const bool nx_ = j >= qa_.y;                                                                                      \
        voffX[P_][MBI_] = (nx_ ? lane_oD : lane_o) + qa_.x;                                                               \
        baseX[P_][0][MBI_] = lane_x + (nx_ ? qb_.y : qa_.z);                                                              \
        baseX[P_][1][MBI_] = lane_x + (nx_ ? qb_.z : qa_.w);                                                              \
        baseX[P_][2][MBI_] = lane_x + (nx_ ? qb_.w : qb_.x);                                                              \
    } while (0)
#define SO_LOAD(P_, I_, BUF_)                                                                                             \
    do {                                                                                                                  \
        constexpr int a_ = (I_) / (3 * NCH), b_ = ((I_) / NCH) % 3, cc_ = (I_) % NCH;                                     \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                                 \
            wa[BUF_][nb] = *reinterpret_cast<const float4*>(smem + w_lane + ((((a_ * 3 + b_) * NCH + cc_) * NB + nb) * 1024)); \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                 \
            xv[BUF_][mb] = *reinterpret_cast<const float4*>(smem + (cc_ ? baseX[P_][a_][mb] + CHB : baseX[P_][a_][mb]) + b_ * 64); \
    } while (0)
#ifdef DAM_DIAG_NO_MFMA        // timing experiments only (results are wrong)
#define SO_MFMA(P_, I_, BUF_) do { if ((I_) == 0) { _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) accX[P_][mb][nb] = __builtin_bit_cast(v4f, wa[BUF_][nb]) + __builtin_bit_cast(v4f, xv[BUF_][mb]); } } while (0)
#else
#define SO_MFMA(P_, I_, BUF_)                                                                                             \
    do {                                                                                                                  \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                                     \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                             \
                _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                         \
                    accX[P_][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(                                              \
                        reinterpret_cast<const float*>(&wa[BUF_][nb])[r], reinterpret_cast<const float*>(&xv[BUF_][mb])[r], \
                        ((I_) == 0 && r == 0) ? (v4f){0.f, 0.f, 0.f, 0.f} : accX[P_][mb][nb], 0, 0, 0);                   \
    } while (0)
#endif
            // operands of the write-out of tile T_ (residual / mask / the BatchNorm's x), requested a slot ahead -- PER PIXEL BLOCK, in the
            // filler that has just written that block of the previous tile out (its operand registers are free from there on).
            // Round 3 requested all of a tile's blocks in ONE filler at item 2 * MB, the slot's last, and consumed them from item 0
            // of the next slot on: ~1 k cycles of lead for loads that take 4-5 k under this load.  PMC (profiles/r04_pmc_strip_epi.txt):
            // SQ_WAIT_ANY 12.4 M wave-cycles per forward launch, 26.4 M with the sums epilogue, 36.3 M with residual + upstream sums --
            // the compute waves stood in front of their write-out units.  DAM_SO_LATE_PREFETCH keeps the old placement (A/B build).
#define SO_PREFETCH1(P_, MBI_)                                                                                            \
    do {                                                                                                                  \
        if constexpr (NEED_R) {                                                                                           \
            const int vo_ = voffX[P_][MBI_];                                                                              \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {       /* pixels past the image fail the range check: 0 */  \
                rpf[MBI_][nb] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, vo_ + nb * 64, 0, 0)); \
                if constexpr (RMSK)                                                                                       \
                    mpf[MBI_][nb] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(mrsrc, vo_ + nb * 64, 0, 0)); \
                if constexpr (EPI >= 2) {                                                                                 \
                    mbpf[MBI_][nb] = (int)__builtin_amdgcn_raw_buffer_load_b8(bbrsrc, (vo_ >> 4) + nb * 4, 0, 0);         \
                    xpf[MBI_][nb] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(x2rsrc, vo_ + nb * 64, 0, 0)); \
                }                                                                                                         \
                if constexpr (EPI == 3)                                                                                   \
                    ubpf[MBI_][nb] = (int)__builtin_amdgcn_raw_buffer_load_b8(ubrsrc, (vo_ >> 4) + nb * 4, 0, 0);         \
            }                                                                                                             \
        }                                                                                                                 \
    } while (0)
#define SO_PREFETCH(P_)                                                                                                   \
    do {                                                                                                                  \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) SO_PREFETCH1(P_, mb);                                           \
    } while (0)
            // write-out of pixel block MBI_ of tile T_ (parity Q_).  PRED_: per-lane validity (the image's last tile only)
#define SO_UNIT(Q_, T_, MBI_, PRED_)                                                                                      \
    do {                                                                                                                  \
        /* the WHOLE byte offset is a vector register, soffset = 0.  (With a register in the store's soffset field the */     \
        /* compiler assumes the 128-bit data may be overwritten at once and places no wait state; on gfx950 the next VALU */  \
        /* write to those registers then reached the store: stale components in single lanes.) */                            \
        const int vo_ = voffX[Q_][MBI_];                                                                                  \
        if (!(PRED_) || j < HoWo - ((t_begin + (T_)) * TM + cw * MW + (MBI_) * 16)) {                                     \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                           \
                v4f v = accX[Q_][MBI_][nb];                                                                               \
                if constexpr (BIAS) {                                                                                     \
                    v += bias4[nb];                                                                                       \
                    if constexpr (RADD) v += rpf[MBI_][nb];                                                               \
                    v.x = fmaxf(v.x, floor_out); v.y = fmaxf(v.y, floor_out); v.z = fmaxf(v.z, floor_out); v.w = fmaxf(v.w, floor_out); \
                }                                                                                                         \
                if constexpr (RMSK) {                                                                                     \
                    const v4f rv = rpf[MBI_][nb], mv = mpf[MBI_][nb];                                                     \
                    v.x += mv.x > 0.f ? rv.x : 0.f; v.y += mv.y > 0.f ? rv.y : 0.f;                                       \
                    v.z += mv.z > 0.f ? rv.z : 0.f; v.w += mv.w > 0.f ? rv.w : 0.f;                                       \
                }                                                                                                         \
                if constexpr (EPI >= 2) {       /* identity shortcut: + res where the block output was positive */       \
                    const v4f rv = rpf[MBI_][nb];                                                                         \
                    const int b8 = mbpf[MBI_][nb];                                                                        \
                    v.x += (b8 & 1) ? rv.x : 0.f; v.y += (b8 & 2) ? rv.y : 0.f;                                           \
                    v.z += (b8 & 4) ? rv.z : 0.f; v.w += (b8 & 8) ? rv.w : 0.f;                                           \
                }                                                                                                         \
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), yrsrc, vo_ + nb * 64, 0, 0);           \
                if constexpr (EPI != 0) {       /* v reaches relu(bn(x) [+ ..]): the BatchNorm's two backward sums */     \
                    const v4f xq = EPI == 1 ? rpf[MBI_][nb] : xpf[MBI_][nb];                                              \
                    const v4f xh = __builtin_elementwise_fma(xq, bw_k1[nb], bw_k2[nb]);                                   \
                    v4f dz;                                                                                               \
                    if constexpr (EPI == 3) {                                                                             \
                        const int ub = ubpf[MBI_][nb];                                                                    \
                        dz.x = (ub & 1) ? v.x : 0.f; dz.y = (ub & 2) ? v.y : 0.f;                                         \
                        dz.z = (ub & 4) ? v.z : 0.f; dz.w = (ub & 8) ? v.w : 0.f;                                         \
                    } else {                                                                                              \
                        const v4f m = __builtin_elementwise_fma(xq, bw_ms[nb], bw_mh[nb]);                                \
                        dz.x = m.x > 0.f ? v.x : 0.f; dz.y = m.y > 0.f ? v.y : 0.f;                                       \
                        dz.z = m.z > 0.f ? v.z : 0.f; dz.w = m.w > 0.f ? v.w : 0.f;                                       \
                    }                                                                                                     \
                    st_s1[nb][0] += dz.xy; st_s1[nb][1] += dz.zw;                                                         \
                    st_s2[nb][0] = __builtin_elementwise_fma(dz.xy, xh.xy, st_s2[nb][0]);                                 \
                    st_s2[nb][1] = __builtin_elementwise_fma(dz.zw, xh.zw, st_s2[nb][1]);                                 \
                }                                                                                                         \
                if constexpr (ST && !SO_NOSTU) {                                                                          \
                    const v2f d0 = v.xy + st_nk[nb][0], d1 = v.zw + st_nk[nb][1];                                         \
                    st_s1[nb][0] += d0; st_s1[nb][1] += d1;                                                               \
                    st_s2[nb][0] = __builtin_elementwise_fma(d0, d0, st_s2[nb][0]);                                       \
                    st_s2[nb][1] = __builtin_elementwise_fma(d1, d1, st_s2[nb][1]);                                       \
                }                                                                                                         \
            }                                                                                                             \
            if constexpr (ST) ++st_n;                                                                                     \
        }                                                                                                                 \
    } while (0)
            // the filler of item I_ in the slot that computes tile S_ (parity P_): write-out units of tile S_-1 first, then the
            // geometry of tile S_+1 (into the parity tile S_-1 is leaving), then the operand prefetch of tile S_
#define SO_FILL(P_, S_, I_, WO_)                                                                                          \
    do {                                                                                                                  \
        /* the tile index through an opaque volatile copy: the scalar pixel decode below must stay in THIS item's region */  \
        /* (free-floating, the instruction selector lines all of a slot's decodes up in front of the slot's first MFMA) */ \
        int sa_ = SO_FROZEN ? 1 : (S_);                                                                                   \
        if constexpr (!SO_FROZEN) asm volatile("" : "+s"(sa_));                                                           \
        if constexpr (SO_NOFILL) { }                                                                                      \
        else if constexpr ((I_) < MB) {                                                                                   \
            if constexpr (WO_) SO_UNIT(1 - (P_), sa_ - 1, ((I_) < MB ? (I_) : 0), 0);                                     \
            SO_TABREAD(sa_ + 1, ((I_) < MB ? (I_) : 0));                                                                  \
            if constexpr (!SO_LATE_PF) SO_PREFETCH1(P_, ((I_) < MB ? (I_) : 0));                                          \
        } else if constexpr ((I_) < 2 * MB) SO_GEOM(1 - (P_), ((I_) < 2 * MB && (I_) >= MB ? (I_) - MB : 0));             \
        else if constexpr ((I_) == 2 * MB && SO_LATE_PF) SO_PREFETCH(P_);                                                 \
    } while (0)
            // one MFMA gap = 32 cycles of which the MFMA holds the issue port for 8: room for ~5 four-cycle instructions
#ifndef DAM_SO_PAT_VALU
#define DAM_SO_PAT_VALU 3
#endif
#ifndef DAM_SO_PAT_SALU
#define DAM_SO_PAT_SALU 2
#endif
#ifdef DAM_SO_NOPATTERN
#define SO_PATTERN() do { } while (0)
#else
#define SO_PATTERN()                                                                                                      \
    do {                                                                                                                  \
        _Pragma("unroll") for (int q_ = 0; q_ < 4 * MB * NB; ++q_) {                                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x002, DAM_SO_PAT_VALU, 0);                                              \
            __builtin_amdgcn_sched_group_barrier(0x004, DAM_SO_PAT_SALU, 0);                                              \
            __builtin_amdgcn_sched_group_barrier(0x030, 1, 0);                                                            \
        }                                                                                                                 \
    } while (0)
#endif
#define SO_ITEM(P_, S_, I_, WO_)                                                                                          \
    do {                                                                                                                  \
        if constexpr ((I_) + 1 < NI) SO_LOAD(P_, ((I_) + 1 < NI ? (I_) + 1 : 0), ((I_) + 1) & 1);                         \
        SO_MFMA(P_, I_, (I_) & 1);                                                                                        \
        if constexpr ((I_) != 0) SO_FILL(P_, S_, I_, WO_);                                                                \
        SO_PATTERN();                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    } while (0)
#define SO_SLOT(P_, S_, WO_)                                                                                              \
    do {                                                                                                                  \
        /* the first operands are requested, then the first filler runs while they travel (its pixel block's last MFMA */  \
        /* was issued >= 3 MFMAs = 96 cycles before the barrier: past the 40-cycle MFMA -> VALU distance) */               \
        SO_LOAD(P_, 0, 0);                                                                                                \
        SO_FILL(P_, S_, 0, WO_);                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        SO_ITEM(P_, S_, 0, WO_); SO_ITEM(P_, S_, 1, WO_); SO_ITEM(P_, S_, 2, WO_); SO_ITEM(P_, S_, 3, WO_); SO_ITEM(P_, S_, 4, WO_); \
        SO_ITEM(P_, S_, 5, WO_); SO_ITEM(P_, S_, 6, WO_); SO_ITEM(P_, S_, 7, WO_); SO_ITEM(P_, S_, 8, WO_);               \
        if constexpr (NCH == 2) {                                                                                         \
            SO_ITEM(P_, S_, 9, WO_); SO_ITEM(P_, S_, 10, WO_); SO_ITEM(P_, S_, 11, WO_); SO_ITEM(P_, S_, 12, WO_); SO_ITEM(P_, S_, 13, WO_); \
            SO_ITEM(P_, S_, 14, WO_); SO_ITEM(P_, S_, 15, WO_); SO_ITEM(P_, S_, 16, WO_); SO_ITEM(P_, S_, 17, WO_);       \
        }                                                                                                                 \
        DAM_STAMP(5);                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                   \
        DAM_STAMP(7);                                                                                                     \
    } while (0)
            // the last tile of the strip: nothing left to overlap with; per-lane validity (the image's last tile may be ragged)
#define SO_LAST(Q_, T_)                                                                                                   \
    do {                                                                                                                  \
        /* The accumulators were written by the MFMAs just in front of the barrier.  A VALU read of an MFMA result has no */ \
        /* hardware interlock; the compiler pads such pairs with s_nop, but not across the barrier's inline asm (seen on */   \
        /* hardware: the last-written register of a block stale in one lane column).  16x16x4 f32: 40 cycles to the result. */ \
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                                \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                                               \
            if (HoWo - ((t_begin + (T_)) * TM + cw * MW + mb * 16) > 0) SO_UNIT(Q_, T_, mb, 1);                           \
        }                                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                   \
        DAM_STAMP(7);                                                                                                     \
    } while (0)

#pragma unroll
            for (int mb = 0; mb < MB; ++mb) { SO_TABREAD(0, mb); SO_GEOM(0, mb); }
            // slot 0: tile 0 alone (the fillers: geometry of tile 1, operands of tile 0's write-out)
            SO_SLOT(0, 0, 0);
            if constexpr (ST) {       // statistics shift = the wave's first outputs (any value near the data would do)
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // MFMA result -> VALU read behind inline asm: see SO_LAST
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) { st_nk[nb][0] = -accX[0][0][nb].xy; st_nk[nb][1] = -accX[0][0][nb].zw; }
            }
            int s = 1;
            for (; s + 1 < n_tiles; s += 2) {
                SO_SLOT(1, s, 1);
                SO_SLOT(0, s + 1, 1);
            }
            if (s < n_tiles) {
                SO_SLOT(1, s, 1);
                ++s;
            }
            // s == n_tiles: the last tile's write-out, then the padding slots of the loaders' paired loop
            if ((n_tiles - 1) & 1) SO_LAST(1, n_tiles - 1);
            else SO_LAST(0, n_tiles - 1);
            for (++s; s < n_slots; ++s) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                DAM_STAMP(7);
            }
#undef SO_GEOM
#undef SO_TABREAD
#undef SO_TAG_EXPECT
#undef SO_TAG_CHECK
#undef SO_LOAD
#undef SO_MFMA
#undef SO_PREFETCH
#undef SO_PREFETCH1
#undef SO_UNIT
#undef SO_FILL
#undef SO_PATTERN
#undef SO_ITEM
#undef SO_SLOT
#undef SO_LAST
        }
    }

#undef SO_TABLE
#undef SO_TAG
#undef DAM_STRIP_GEOM
#ifdef DAM_STAMPS
    DAM_STAMP(8);
    return;
#endif
    if constexpr (epi_bwd) {
        // (sum dz, sum dz*xhat) per lane -> the 16 pixel lanes (butterfly), then the 8 compute waves through LDS in wave order
        __syncthreads();                                   // the ring is no longer needed
        float* sm = reinterpret_cast<float*>(smem);        // [8 compute waves][NB*16 ch][2]
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = st_s1[nb][r >> 1][r & 1], b = st_s2[nb][r >> 1][r & 1];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
                if (j == 0 && wave < NCW) {
                    float* o = sm + ((wave * NB * 16) + nb * 16 + kq * 4 + r) * 2;
                    o[0] = a; o[1] = b;
                }
            }
        __syncthreads();
        if (tid < NB * 16) {
            float a = 0.f, b = 0.f;
            for (int w = 0; w < NCW; ++w) { a += sm[((w * NB * 16) + tid) * 2]; b += sm[((w * NB * 16) + tid) * 2 + 1]; }
            const int ch = nb0 * 16 + tid;
            if (ch < g.N) {
                const size_t part = (size_t)blockIdx.z * gridDim.x + blockIdx.x;
                store_sc1(stats + (part * g.N + ch) * 2, a);
                store_sc1(stats + (part * g.N + ch) * 2 + 1, b);
            }
        }
        return;
    }
#ifdef DAM_DIAG_SO_NO_STATS_EPI    // timing experiments only: no merge of the statistics at the end
    if (stats && g.B == 12345) {
#else
    if (stats) {
#endif
        // (n, mean, M2) per lane -> Chan merge over the 16 pixel lanes, then over the 4 compute waves through LDS
        float* sm = reinterpret_cast<float*>(smem);        // ring no longer needed: [8 compute waves][NB*16 ch][3]
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float n = (float)st_n;
                const float s1 = st_s1[nb][r >> 1][r & 1], s2 = st_s2[nb][r >> 1][r & 1];
                const float md = st_n ? s1 / n : 0.f;
                float mean = md - st_nk[nb][r >> 1][r & 1];
                float m2 = st_n ? fmaxf(s2 - s1 * md, 0.f) : 0.f;
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
                if (j == 0 && wave < NCW) {
                    float* o = sm + ((wave * NB * 16) + nb * 16 + kq * 4 + r) * 3;
                    o[0] = n; o[1] = mean; o[2] = m2;
                }
            }
        __syncthreads();
        if (tid < NB * 16) {
            float n = 0.f, mean = 0.f, m2 = 0.f;
            for (int w = 0; w < NCW; ++w) {
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
                store_sc1(o, n); store_sc1(o + 1, mean); store_sc1(o + 2, m2);
            }
        }
        if (fin.counter) {      // the last workgroup to arrive merges all records (dam_bn_fin.h): no finalize launch
            unsigned* ticket = reinterpret_cast<unsigned*>(smem + 16 * 1024);
            const unsigned total = gridDim.x * gridDim.z;       // statistics launches have gridDim.y == 1 (host check)
            if (block_arrive_last(fin.counter, total, ticket))
                bn_stats_finalize_block(stats, (int)total, g.N, fin, reinterpret_cast<double*>(smem + 32 * 1024), tid, NT);
        }
    }
}

}  // namespace

// Returns DAM_OK if launched, DAM_ERR_UNSUPPORTED if the layer does not fit this variant (caller falls back).
template <int MB, int NB, int NCH, bool T33, bool LW = false, int EPI = 0, int SO = 0>
static int launch_strip(ConvGeo& g, StripGeo& sg, size_t lds, const float* X, const float* Wp, const float* bias, float* Y,
                        const float* res, const float* res_mask, float* stats, const float* in_scale, const float* in_shift,
                        const BnFinArgs& fin, const BnBwdEpi& bwd, hipStream_t st) {
    if (lds > 64 * 1024) {
        static PerDevice<bool> raised_pd; bool& raised = raised_pd();
        if (!raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_strip_kernel<MB, NB, NCH, T33, LW, EPI, SO>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return DAM_ERR_LAUNCH;
            raised = true;
        }
    }
    dim3 grid((unsigned)sg.strips, (unsigned)cdiv(g.N / 16, NB), (unsigned)g.B);
    hipLaunchKernelGGL((conv_strip_kernel<MB, NB, NCH, T33, LW, EPI, SO>), grid, dim3(SO ? 512 : STRIP_THREADS), lds, st, g, sg, X, reinterpret_cast<const float4*>(Wp), bias,
                       Y, res, res_mask, stats, in_scale, in_shift, fin, bwd);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

int conv_strip_try(ConvGeo& g_in, int h_lo, int h_hi, const float* X, const float* Wp, const float* bias, float* Y,
                   const float* res, const float* res_mask, float* stats, int* stats_parts, const BnFinArgs* fin_in,
                   const float* in_scale, const float* in_shift, const BnBwdEpi& bwd, hipStream_t st) {
    BnFinArgs fin{};
    if (fin_in && stats && !bwd.x) fin = *fin_in;
    if (bwd.x && !stats) return DAM_ERR_BAD_ARG;
    if (bwd.x && res && !bwd.res_bits) return DAM_ERR_UNSUPPORTED;     // with a residual the sums need its mask as sign bytes
    ConvGeo g = g_in;                 // the caller's copy stays as it is for the tile kernel
    g.epi_bwd = bwd.x ? (res ? (bwd.mask_bits ? 3 : 2) : 1) : 0;
    if (bwd.x && bwd.mask_bits && !res) return DAM_ERR_UNSUPPORTED;    // byte-masked sums exist with the residual epilogue only
    if (g.nB == 3 && g.step_w < 0) {  // same taps walked left to right: column step becomes +1, weight taps are re-indexed
        g.off_w += 2 * g.step_w; g.step_w = -g.step_w;
        g.wt_base += 2 * g.wt_sb; g.wt_sb = -g.wt_sb;
    }
    if (g.in_nchw || g.B > 65535 || g.nA > 3 || g.nB > 3 || g.nchunks > 2) return DAM_ERR_UNSUPPORTED;
    const int nblk = g.N / 16;
    const int64_t npix = (int64_t)g.Ho * g.Wo;
    if (npix >= (1 << 22)) return DAM_ERR_UNSUPPORTED;
    const int NB = nblk % 2 == 0 ? 2 : 1;
    if (stats && cdiv(nblk, NB) != 1) return DAM_ERR_UNSUPPORTED;        // statistics need all channels in one workgroup
    StripGeo sg;
    sg.RH = h_hi - h_lo + 1;
    sg.w_taps = 9;                                                         // canonical [a*3+b] order in LDS
    const size_t w_bytes = (size_t)9 * g.nchunks * NB * 1024;      // resident packed weights of the N tile
    if (w_bytes > 48 * 1024) return DAM_ERR_UNSUPPORTED;                    // thick layers: tile kernel (weights from L2)
    sg.w_lds = 1;
    const size_t LDS_MAX = 150 * 1024;                                      // one workgroup (12 waves) per CU
    int MB = 4;
    size_t lds = 0;
    bool wide = false;
    for (;; MB >>= 1) {
        const int tm = 64 * MB;
        int rows_out = (int)((tm + g.Wo - 2) / g.Wo + 1);
        if (rows_out > g.Ho) rows_out = g.Ho;
        const int rows_tile = (rows_out - 1) * g.s + sg.RH;       // rows one tile reads
        const int rows_new = rows_out * g.s;                      // rows each following tile can add
        int nr = 1;
        while (nr < rows_tile + rows_new) nr <<= 1;               // tile s in use, the rows tile s+1 adds being written
        // loader waves take whole (row, chunk) planes: KP planes of <= GPP pieces each per wave per tile (see the kernel)
        const int gpp = (g.PWT * 4 + 63) / 64;
        wide = gpp > (g.nchunks == 1 ? 9 : 5);                    // wide-row loader shape (LW): instantiated for two tiles only
        const int kp = g.nchunks == 1 ? 1 : (wide ? 2 : 3), gpp_max = g.nchunks == 1 ? (wide ? 14 : 9) : (wide ? 7 : 5);
        if (gpp > gpp_max) return DAM_ERR_UNSUPPORTED;
        if (wide && MB != (g.nchunks == 1 ? 4 : 2)) { if (MB == 2) return DAM_ERR_UNSUPPORTED; continue; }
        if (rows_new * g.nchunks > STRIP_LOADERS * kp) { if (MB == 2) return DAM_ERR_UNSUPPORTED; continue; }
        lds = (size_t)nr * g.PWT * 64 * g.nchunks + w_bytes;
        if (lds <= LDS_MAX) { sg.NR = nr; break; }
        if (MB == 2) return DAM_ERR_UNSUPPORTED;     // 64-pixel strips are not worth it: tile kernel
    }
    const int tm = 64 * MB;
    sg.tiles_m = (int)cdiv(npix, tm);
    // Strips per image by MAKESPAN: one workgroup per CU is resident (LDS), so a launch of w workgroups runs in ceil(w / 256)
    // rounds of (tiles per workgroup + ~2 tiles' worth of prologue and drain).  The first rule (cdiv(total, 256) tiles per
    // workgroup, at most 64) gave C5's batch of 59 chunks 295 and 531 workgroups -- a second and third round for a fraction of
    // the chip: 512 and 384 us per launch where 236 workgroups of 66 tiles / 472 of 66 need 290 / 270.
    int tpw = sg.tiles_m;
    int64_t slots = 256;
#ifdef DAM_STRIP_2WG
    // two co-resident workgroups per CU for the instantiations compiled that way (16 channels, self-overlapped, LDS <= 80 KB)
    if (g.nchunks == 1 && MB == 4 && NB == 1 && g.nA == 3 && g.nB == 3 && g.s == 1 && g.step_w == 1 && !wide && !getenv("DAM_STRIP_PINGPONG") &&
        (lds + 2048) * 2 <= 160 * 1024 && (bwd.x ? (res ? (bwd.mask_bits ? 3 : 2) : 1) : 0) <= DAM_STRIP_2WG)
        slots = 512;
#endif
    {
        const int64_t per_image = cdiv(nblk, NB);
        int64_t best = INT64_MAX;
        for (int strips = 1; strips <= sg.tiles_m; ++strips) {
            const int t = (int)cdiv(sg.tiles_m, strips);
            if (t < 2 && sg.tiles_m >= 2) break;                      // the loaders' slot loop wants two tiles per workgroup
            const int64_t wgs = (int64_t)cdiv(sg.tiles_m, t) * g.B * per_image;
            const int64_t cost = cdiv(wgs, slots) * (t + 2);
            if (cost < best) { best = cost; tpw = t; }
        }
    }
    sg.tpw = tpw;
    sg.strips = (int)cdiv(sg.tiles_m, tpw);
    if (g.Wo > 1024 || g.Wo < 16) return DAM_ERR_UNSUPPORTED;   // scalar pixel decode: 16 consecutive pixels span <= 2 output rows
    sg.wo_magic = (unsigned)((1ull << 32) / (unsigned)g.Wo) + 1u;
    sg.ring_off = sg.NR * 64;           // keeps (row + ring_off) non-negative for row >= -64*NR
    if (stats_parts) *stats_parts = sg.strips * g.B;
    if (stats && (int64_t)sg.strips * g.B > 1024) return DAM_ERR_UNSUPPORTED;
    if (lds < (size_t)8 * NB * 16 * 3 * sizeof(float)) lds = (size_t)8 * NB * 16 * 3 * sizeof(float);
    if (fin.counter && lds < (size_t)32 * 1024 + STRIP_THREADS * 3 * sizeof(double)) lds = (size_t)32 * 1024 + STRIP_THREADS * 3 * sizeof(double);
    // 3x3 taps, stride 1, unit column step: compile-time item grid with immediate operand offsets
    const bool t33 = g.nA == 3 && g.nB == 3 && g.s == 1 && g.step_w == 1;
#define DAM_STRIP_ARGS g, sg, lds, X, Wp, bias, Y, (bwd.x && !res ? bwd.x : res), (bwd.x && res ? reinterpret_cast<const float*>(bwd.res_bits) : res_mask), stats, in_scale, in_shift, fin, bwd, st
    // Self-overlapped form (template comment of the kernel) for the 3x3 / stride-1 shapes of the 16- and 32-channel stages at
    // ordinary row widths; DAM_STRIP_PINGPONG=1 keeps the ping-pong form (A/B switch).  The write-out variant is compile time.
    static const bool pingpong = getenv("DAM_STRIP_PINGPONG") != nullptr;
    const int rows_out_so = std::min(g.Ho, (64 * MB + g.Wo - 2) / g.Wo + 1), rows_tile_so = (rows_out_so - 1) * g.s + sg.RH;
    if (t33 && !wide && !pingpong && !fin.counter && rows_tile_so * g.nchunks <= 8 * (g.nchunks == 1 ? 1 : 2) &&
        ((g.nchunks == 1 && MB == 4 && NB == 1) || (g.nchunks == 2 && MB == 2 && NB == 2))) {
        const bool c16 = g.nchunks == 1;
        lds += 2048;                        // geometry tables: 4 compute waves x 4 buffers x 8 ints x <= 4 pixel blocks
#define DAM_SO_CASE(E_, S_)                                                                                              \
    return c16 ? launch_strip<4, 1, 1, true, false, E_, S_>(DAM_STRIP_ARGS) : launch_strip<2, 2, 2, true, false, E_, S_>(DAM_STRIP_ARGS)
        if (bwd.x && res) {
            if (c16) {
                if (bwd.mask_bits) return launch_strip<4, 1, 1, true, false, 3, 1>(DAM_STRIP_ARGS);
                return launch_strip<4, 1, 1, true, false, 2, 1>(DAM_STRIP_ARGS);
            }
        } else if (bwd.x) {
            DAM_SO_CASE(1, 1);
        } else if (stats) {
            if (!bias && !res && !g.relu_out) DAM_SO_CASE(0, 2);
        } else if (res && res_mask) {
            if (!bias && !g.relu_out) DAM_SO_CASE(0, 5);
        } else if (res) {
            DAM_SO_CASE(0, 4);
        } else if (bias || g.relu_out) {
            DAM_SO_CASE(0, 3);
        } else {
            DAM_SO_CASE(0, 1);
        }
#undef DAM_SO_CASE
    }
    if (bwd.x && res) { // residual + upstream sums: the 16-channel full-resolution data gradient only (stem <- first block)
        if (!t33 || g.nchunks != 1 || MB != 4 || NB != 1) return DAM_ERR_UNSUPPORTED;
        if (bwd.mask_bits)
            return wide ? launch_strip<4, 1, 1, true, true, 3>(DAM_STRIP_ARGS) : launch_strip<4, 1, 1, true, false, 3>(DAM_STRIP_ARGS);
        return wide ? launch_strip<4, 1, 1, true, true, 2>(DAM_STRIP_ARGS) : launch_strip<4, 1, 1, true, false, 2>(DAM_STRIP_ARGS);
    }
    if (bwd.x) {        // sums epilogue: the 3x3 / stride-1 data gradients of the 16- and 32-channel stages only
        if (!t33) return DAM_ERR_UNSUPPORTED;
        if (g.nchunks == 1 && MB == 4 && NB == 1)
            return wide ? launch_strip<4, 1, 1, true, true, 1>(DAM_STRIP_ARGS) : launch_strip<4, 1, 1, true, false, 1>(DAM_STRIP_ARGS);
        if (g.nchunks == 2 && MB == 2 && NB == 2)
            return wide ? launch_strip<2, 2, 2, true, true, 1>(DAM_STRIP_ARGS) : launch_strip<2, 2, 2, true, false, 1>(DAM_STRIP_ARGS);
        return DAM_ERR_UNSUPPORTED;
    }
#define DAM_STRIP_CASE(M_, N_)                                                                                           \
    if (MB == M_ && NB == N_) {                                                                                             \
        if (t33) return g.nchunks == 1 ? launch_strip<M_, N_, 1, true>(DAM_STRIP_ARGS) : launch_strip<M_, N_, 2, true>(DAM_STRIP_ARGS); \
        return g.nchunks == 1 ? launch_strip<M_, N_, 1, false>(DAM_STRIP_ARGS) : launch_strip<M_, N_, 2, false>(DAM_STRIP_ARGS); \
    }
    if (wide) {         // 16 ch: <4, 1>, 32 ch: <2, 2> only
        if (g.nchunks == 1 && MB == 4 && NB == 1)
            return t33 ? launch_strip<4, 1, 1, true, true>(DAM_STRIP_ARGS) : launch_strip<4, 1, 1, false, true>(DAM_STRIP_ARGS);
        if (g.nchunks == 2 && MB == 2 && NB == 2)
            return t33 ? launch_strip<2, 2, 2, true, true>(DAM_STRIP_ARGS) : launch_strip<2, 2, 2, false, true>(DAM_STRIP_ARGS);
        return DAM_ERR_UNSUPPORTED;
    }
    DAM_STRIP_CASE(4, 2); DAM_STRIP_CASE(4, 1);
    DAM_STRIP_CASE(2, 2); DAM_STRIP_CASE(2, 1);
#undef DAM_STRIP_CASE
#undef DAM_STRIP_ARGS
    return DAM_ERR_UNSUPPORTED;
}

}  // namespace dam

// include/dam_hip.h: {checks, mismatches} of the geometry-table tags since the last reset -- diagnostic builds
// (-DDAM_STRIP_DIAG_TAGS) only, DAM_ERR_UNSUPPORTED in the shipped library.  Synchronises the device.
extern "C" int dam_strip_diag_counters(uint32_t* out2_host, int reset) {
#ifdef DAM_STRIP_DIAG_TAGS
    if (!out2_host) return DAM_ERR_BAD_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return DAM_ERR_LAUNCH;
    unsigned v[2] = {0, 0};
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(dam::strip_diag_counters), sizeof(v)) != hipSuccess) return DAM_ERR_LAUNCH;
    out2_host[0] = v[0]; out2_host[1] = v[1];
    if (reset) {
        const unsigned z[2] = {0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(dam::strip_diag_counters), z, sizeof(z)) != hipSuccess) return DAM_ERR_LAUNCH;
    }
    return DAM_OK;
#else
    (void)out2_host; (void)reset;
    return DAM_ERR_UNSUPPORTED;
#endif
}
