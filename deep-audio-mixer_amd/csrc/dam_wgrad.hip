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
#include <cstdlib>
#include <string.h>
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
    int TMW;                 // pixels per tile (a multiple of 16, <= 256): 4 waves x TMW/4 pixels
};

// Weight gradients of ONE geometry batched into one launch of the tile kernel (blockIdx.z = job): the three equal convolutions of
// a deep ResNet stage (models/model_resnet.py:14-21 at 256 / 128 channels) are 20-28 us launches of which ~13 us are launch and
// pipeline fill; one launch of three jobs pays that once and needs a third of the pixel splits (slabs) to fill the chip.
constexpr int WG_MAX_JOBS = 4;
struct WgradJobs {
    const float* X[WG_MAX_JOBS];
    const float* dY[WG_MAX_JOBS];
    const float* sc[WG_MAX_JOBS];
    const float* sh[WG_MAX_JOBS];
    float* partial[WG_MAX_JOBS];
    int relu_in[WG_MAX_JOBS];             // per job (with sc): a block's conv2 reads relu(bn1(c1)), its conv1 the plain block input
};

// Jobs of a batched launch of the direct kernel (see wgrad_direct_batch_kernel): each with its own geometry and grid share.
constexpr int WD_MAX_JOBS = 8;
struct DirectJob {
    const float* X; const float* dY; float* partial;
    int rows, Ho, Wo, H, W, C, N, s, pad, dil, tiles_k;
    int nx, nsplit, wg0;                     // the job's (nx, nsplit) grid and its first workgroup in the flat launch
};
struct DirectJobs { DirectJob job[WD_MAX_JOBS]; int njobs; };

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

// LW (loader waves; experiment of round 5, off by default -- see launch_wgrad_jobs): 512 threads -- waves 0-3 compute exactly as in
// the plain form, waves 4-7 stage the NEXT tile's X patch and dY pixels into the other of two LDS images while the compute waves run
// the MFMAs of this one (one barrier per tile).  The plain form stages and computes in turn and relies on a second co-resident
// workgroup to fill the gaps: its matrix pipe was 58 % busy on the scalar models' 9x9 layer (0.585 of peak) and a 176-pixel tile of
// the 33 x 5 stage cost 19 us for 5.3 us of MFMAs -- and still beats this form.
template <int TNB, int TKB, int TA, int TB, bool LW>
__global__ __launch_bounds__(LW ? 512 : 256) void wgrad_kernel(const WgradGeo g, const WgradJobs jobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const float* __restrict__ X = jobs.X[blockIdx.z];
    const float* __restrict__ dY = jobs.dY[blockIdx.z];
    const float* __restrict__ in_scale = jobs.sc[blockIdx.z];
    const float* __restrict__ in_shift = jobs.sh[blockIdx.z];
    float* __restrict__ partial = jobs.partial[blockIdx.z];
    constexpr int NBLK = TNB * TKB * TA * TB;
    const int tid = threadIdx.x & 255, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) & 3;          // index inside the role
    const bool loader = LW && __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) != 0;
    const int j = lane & 15, kq = lane >> 4;
    int xt = blockIdx.x;
    const int tg = xt % g.tap_groups; xt /= g.tap_groups;
    const int tk = xt % g.tiles_k;
    const int tn = xt / g.tiles_k;
    const int a0 = tg * TA;
    const int HoWo = g.Ho * g.Wo;
    const int chunk_bytes = g.PR * g.PWT * 64;
    const int TMW = g.TMW, WP = g.TMW >> 2;   // pixels per wave
    const int img_bytes = TKB * chunk_bytes + TNB * TMW * 64;      // one LDS image: [TKB chunks of the X patch][TNB][TMW][16] floats of dY

    PatchGeo pg;
    pg.H = g.H; pg.W = g.W; pg.C = g.C; pg.s = g.s; pg.c0 = g.c0; pg.PR = g.PR; pg.PWin = g.PWin; pg.PWs = g.PWs;
    pg.PWT = g.PWT; pg.in_nchw = g.in_nchw; pg.relu_in = jobs.relu_in[blockIdx.z];

    const size_t img_elems = (size_t)g.H * g.W * g.C;
    // stage tile `tile` into the LDS image at `smem` (all 256 threads of the staging role)
    auto stage_tile = [&](int tile, unsigned char* smem) {
        unsigned char* dy_s = smem + TKB * chunk_bytes;
        const int img = tile / g.tiles_m;
        const int p0 = (tile - img * g.tiles_m) * TMW;
        const int oh_first = p0 / g.Wo;
        // only the input rows the TA kernel rows of this tap group touch (a 1 x KW group needs no vertical halo at all)
        stage_patch(smem, chunk_bytes, X + (size_t)img * img_elems, pg, oh_first * g.s + g.off_h + a0 * g.step_h, tk * TKB, TKB,
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
    };
    if constexpr (LW) {
        // ---- loader waves: their own loop with the SAME number of barriers as the compute waves' path below (one after the first
        //      image, one per tile, five in the cross-wave reduction); the accumulators do not exist on this path
        if (loader) {
            int tile = blockIdx.y;
            if (tile < g.total_tiles) stage_tile(tile, smem_all);
            __syncthreads();
            for (int k = 0; tile < g.total_tiles; tile += g.nsplit, ++k) {
                if (tile + g.nsplit < g.total_tiles) stage_tile(tile + g.nsplit, smem_all + ((k + 1) & 1) * img_bytes);
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) __syncthreads();
            return;
        }
    }
    v4f acc[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};

    int toffs[TA * TB];                      // LDS byte offset of tap (ta, tb) relative to the lane's pixel of the staged patch
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int coff = g.off_w + tb * g.step_w - g.c0;
            const int slotoff = g.s == 1 ? coff : (coff & 1) * g.PWs + (coff >> 1);
            toffs[ta * TB + tb] = (ta * g.step_h * g.PWT + slotoff) * 64;
        }
    // the MFMAs of tile `tile` from the LDS image at `smem` (the four compute waves)
    auto compute_tile = [&](int tile, const unsigned char* smem) {
        const unsigned char* dy_s = smem + TKB * chunk_bytes;
        const int img = tile / g.tiles_m;
        const int p0 = (tile - img * g.tiles_m) * TMW;
        const int oh_first = p0 / g.Wo;
        (void)img;
        // this wave's WP pixels, 4 per MFMA step; lane group kq owns pixel 4*t + kq.  Two operand sets: the TNB + TA*TB*TKB
        // ds_read_b32 of step t + 1 are requested before the MFMAs of step t (one wave per SIMD here -- 80-odd KB of LDS per
        // workgroup -- so nothing else hides the LDS latency: the single-set loop ran the 9x9 / 64 -> 128 layer of the scalar
        // models at 0.41 of the matrix peak; DAM_WG_NO_PIPELINE keeps it for the A/B)
        int p = p0 + wave * WP + kq;
        int pc = p < HoWo ? p : HoWo - 1;
        int oh = pc / g.Wo, ow = pc - oh * g.Wo;
        const int nsteps = WP >> 2;
#ifdef DAM_WG_NO_PIPELINE
        for (int t = 0; t < nsteps; ++t) {
            const int pl = wave * WP + 4 * t + kq;
            float av[TNB];
#pragma unroll
            for (int nb = 0; nb < TNB; ++nb)
                av[nb] = *reinterpret_cast<const float*>(dy_s + ((nb * TMW + pl) * 16 + j) * 4);
            const int base = (((oh - oh_first) * g.s) * g.PWT + ow) * 64 + j * 4;
#pragma unroll
            for (int ta = 0; ta < TA; ++ta) {
                const int roff = ta * g.step_h;                 // relative to the first staged row (tap row a0)
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
#else
        float av[2][TNB], bv[2][TA * TB * TKB];
#define DAM_WG_LOAD(BUF_, T_)                                                                                               \
    do {                                                                                                                  \
        const int pl_ = wave * WP + 4 * (T_) + kq;                                                                        \
        _Pragma("unroll") for (int nb = 0; nb < TNB; ++nb)                                                                \
            av[BUF_][nb] = *reinterpret_cast<const float*>(dy_s + ((nb * TMW + pl_) * 16 + j) * 4);                       \
        const int base_ = (((oh - oh_first) * g.s) * g.PWT + ow) * 64 + j * 4;                                            \
        _Pragma("unroll") for (int ta = 0; ta < TA; ++ta)                                                                 \
            _Pragma("unroll") for (int tb = 0; tb < TB; ++tb)                                                             \
                _Pragma("unroll") for (int kb = 0; kb < TKB; ++kb)                                                        \
                    bv[BUF_][(ta * TB + tb) * TKB + kb] =                                                                 \
                        *reinterpret_cast<const float*>(smem + kb * chunk_bytes + base_ + toffs[ta * TB + tb]);           \
        p += 4;                                       /* this lane's pixel of the NEXT step (past the image: dY == 0) */  \
        if (p < HoWo) {                                                                                                   \
            ow += 4;                                                                                                      \
            while (ow >= g.Wo) { ow -= g.Wo; ++oh; }                                                                      \
        }                                                                                                                 \
    } while (0)
#define DAM_WG_MFMA(BUF_)                                                                                                   \
    do {                                                                                                                  \
        _Pragma("unroll") for (int ta = 0; ta < TA; ++ta)                                                                 \
            _Pragma("unroll") for (int tb = 0; tb < TB; ++tb)                                                             \
                _Pragma("unroll") for (int kb = 0; kb < TKB; ++kb)                                                        \
                    _Pragma("unroll") for (int nb = 0; nb < TNB; ++nb) {                                                  \
                        const int idx = ((nb * TKB + kb) * TA + ta) * TB + tb;                                            \
                        acc[idx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[BUF_][nb], bv[BUF_][(ta * TB + tb) * TKB + kb], acc[idx], 0, 0, 0); \
                    }                                                                                                     \
    } while (0)
        DAM_WG_LOAD(0, 0);
        int t = 0;
        for (; t + 2 <= nsteps; t += 2) {
            DAM_WG_LOAD(1, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            DAM_WG_MFMA(0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nsteps) DAM_WG_LOAD(0, t + 2);
            __builtin_amdgcn_sched_barrier(0);
            DAM_WG_MFMA(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t < nsteps) DAM_WG_MFMA(0);
#undef DAM_WG_LOAD
#undef DAM_WG_MFMA
#endif
    };
    if constexpr (LW) {
        int tile = blockIdx.y;
        __syncthreads();                                  // the loaders have staged the first tile
        for (int k = 0; tile < g.total_tiles; tile += g.nsplit, ++k) {
            compute_tile(tile, smem_all + (k & 1) * img_bytes);
            __syncthreads();                              // image k & 1 may be refilled; image (k + 1) & 1 is complete
        }
    } else {
        for (int tile = blockIdx.y; tile < g.total_tiles; tile += g.nsplit) {
            __syncthreads();
            stage_tile(tile, smem_all);
            __syncthreads();
            compute_tile(tile, smem_all);
        }
    }

    // combine the 4 waves through LDS (sequential adds: fixed order), then one slab per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_all);
    for (int w = 0; w < 4; ++w) {       // (five barriers from here to the end: the loader waves' path counts them)
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

// dW[n][k][kh][kw] = sum over splits of the slabs, for a BATCH of weight gradients in one launch: a reduction is 5-10 us of
// mostly launch latency, a ResNet18 step has 20 of them and nothing but the optimizer waits for any (dam_wgrad_queue_*).
// Block = EL elements x SL split lanes (256 x 1 for few big slabs, 32 x 8 for few small ones, 8 x 32 for many): a thread adds
// a strided subset of the splits (4 independent chains, all loads in flight), the SL subtotals are combined by a fixed tree
// (deterministic).
struct ReduceJob {
    const float* partial;
    float* dw;
    int nsplit, nx, TNB, TKB, TA, TB, n_real, k_real, KH, KW, tap_groups, tiles_k;
    int sl;                  // split lanes: 32, 8 or 1
    int blocks;              // workgroups of the launch that work on this job
};
constexpr int REDUCE_MAX_JOBS = 40;       // one flush per ResNet18 backward (30 weight gradients); 40 x 72 B of kernel arguments
struct ReduceBatch {
    ReduceJob job[REDUCE_MAX_JOBS];
    int njobs;
};

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ReduceBatch batch) {
    __shared__ float4 sub[256];
    int ji = 0, b0 = blockIdx.x;
    while (ji + 1 < batch.njobs && b0 >= batch.job[ji].blocks) { b0 -= batch.job[ji].blocks; ++ji; }
    const ReduceJob& j = batch.job[ji];
    const int SL = j.sl, EL = 256 / SL;
    const int nblk_tile = j.TNB * j.TKB * j.TA * j.TB;
    const int64_t per_split4 = (int64_t)j.nx * nblk_tile * 64;       // float4 per slab
    const float4* p4 = reinterpret_cast<const float4*>(j.partial);
    const int el = threadIdx.x % EL, ys = threadIdx.x / EL;
    for (int64_t e0 = (int64_t)b0 * EL; e0 < per_split4; e0 += (int64_t)j.blocks * EL) {
        const int64_t e = e0 + el;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        if (e < per_split4) {
            int y = ys;
            for (; y + 3 * SL < j.nsplit; y += 4 * SL) {
                const float4 a = p4[y * per_split4 + e], b = p4[(y + SL) * per_split4 + e];
                const float4 c = p4[(y + 2 * SL) * per_split4 + e], d = p4[(y + 3 * SL) * per_split4 + e];
                s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
                s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w; s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
            }
            for (; y < j.nsplit; y += SL) { const float4 a = p4[y * per_split4 + e]; s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; }
        }
        sub[ys * EL + el] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                                        (s0.w + s1.w) + (s2.w + s3.w));
        __syncthreads();
        for (int stride = SL / 2; stride >= 1; stride >>= 1) {
            if (ys < stride) {
                float4 a = sub[ys * EL + el];
                const float4 b = sub[(ys + stride) * EL + el];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                sub[ys * EL + el] = a;
            }
            __syncthreads();
        }
        if (ys == 0 && e < per_split4) {
            const float4 s = sub[el];
            const float sv[4] = {s.x, s.y, s.z, s.w};
            const int lane = e & 63;
            int64_t q = e >> 6;
            const int blk = q % nblk_tile;
            int xt = q / nblk_tile;
            const int tb = blk % j.TB, ta = (blk / j.TB) % j.TA, kb = (blk / (j.TB * j.TA)) % j.TKB, nb = blk / (j.TB * j.TA * j.TKB);
            const int tg = xt % j.tap_groups; xt /= j.tap_groups;
            const int tk = xt % j.tiles_k, tn = xt / j.tiles_k;
            const int k = (tk * j.TKB + kb) * 16 + (lane & 15);
            const int kh = tg * j.TA + ta;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = (tn * j.TNB + nb) * 16 + (lane >> 4) * 4 + r;
                if (n < j.n_real && k < j.k_real && kh < j.KH) j.dw[(((size_t)n * j.k_real + k) * j.KH + kh) * j.KW + tb] = sv[r];
            }
        }
        __syncthreads();
    }
}

// Host side of a caller-owned queue of deferred reductions (include/dam_hip.h: dam_wgrad_queue_*).
struct WgradQueue;
// tile-kernel launches of one geometry recorded, not yet launched (dam_wgrad_queue_set_batching)
struct PendingTile {
    int njobs;                               // 0: nothing pending
    int sig;                                 // the instantiation: TNB | TKB << 4 | TA << 8 | TB << 12
    WgradGeo g;                              // shared geometry, nsplit = 0 (decided at launch from the number of jobs)
    WgradJobs jobs;
    float* dw[WG_MAX_JOBS];
    int64_t ws_floats[WG_MAX_JOBS];
    int n_real[WG_MAX_JOBS], k_real[WG_MAX_JOBS];
    int (*launch)(WgradQueue*, hipStream_t);
};
// direct-kernel launches of one instantiation recorded, not yet launched (any geometry: every job carries its own)
struct PendingDirect {
    int sig;                                 // TNB | TKB << 4 | KH << 8 | KW << 12
    DirectJobs jobs;                         // jobs.njobs == 0: nothing pending
    float* dw[WD_MAX_JOBS];
    int n_real[WD_MAX_JOBS], k_real[WD_MAX_JOBS];
    int (*launch)(WgradQueue*, hipStream_t);
};
struct WgradQueue {
    unsigned magic;
    int batching;                            // 1: same-geometry tile launches may wait for each other until the flush
    ReduceBatch batch;
    PendingTile pend;
    PendingDirect pend_direct;
};
constexpr unsigned WGRAD_QUEUE_MAGIC = 0x57475251u;     // "WGRQ"

int reduce_flush(ReduceBatch& b, hipStream_t st) {
    if (b.njobs <= 0) return DAM_OK;
    int blocks = 0;
    for (int i = 0; i < b.njobs; ++i) blocks += b.job[i].blocks;
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks), dim3(256), 0, st, b);
    b.njobs = 0;
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

// The slab reduction of one weight gradient: launched now (queue == nullptr) or appended to the caller's queue (its slabs
// must then stay untouched until dam_wgrad_queue_flush).
int reduce_submit(void* queue, const float* partial, float* dw, int nsplit, int nx, int TNB, int TKB, int TA, int TB, int n_real,
                  int k_real, int KH, int KW, int tap_groups, int tiles_k, hipStream_t st) {
    ReduceJob j;
    j.partial = partial; j.dw = dw; j.nsplit = nsplit; j.nx = nx; j.TNB = TNB; j.TKB = TKB; j.TA = TA; j.TB = TB;
    j.n_real = n_real; j.k_real = k_real; j.KH = KH; j.KW = KW; j.tap_groups = tap_groups; j.tiles_k = tiles_k;
    const int64_t per_split4 = (int64_t)nx * TNB * TKB * TA * TB * 64;
    if (nsplit >= 64) {
        j.sl = 32;
        j.blocks = (int)(cdiv(per_split4, 8) < 4096 ? cdiv(per_split4, 8) : 4096);
    } else if (per_split4 >= 8192 && !getenv("DAM_REDUCE_NO_SL1")) {
        // few big slabs (the thick stages): one element per thread, the splits added in order by that thread (4 chains, all
        // loads in flight) -- no LDS step at all; the 32 x 8 shape spent its time in one-load-then-barrier rounds
        j.sl = 1;
        j.blocks = (int)(cdiv(per_split4, 256) < 4096 ? cdiv(per_split4, 256) : 4096);
    } else {
        j.sl = 8;
        j.blocks = (int)(cdiv(per_split4, 32) < 2048 ? cdiv(per_split4, 32) : 2048);
    }
    if (!queue) {
        ReduceBatch one;
        one.job[0] = j; one.njobs = 1;
        return reduce_flush(one, st);
    }
    WgradQueue* q = static_cast<WgradQueue*>(queue);
    if (q->magic != WGRAD_QUEUE_MAGIC) return DAM_ERR_BAD_ARG;
    if (q->batch.njobs == REDUCE_MAX_JOBS) {
        const int rc = reduce_flush(q->batch, st);
        if (rc != DAM_OK) return rc;
    }
    q->batch.job[q->batch.njobs++] = j;
    return DAM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-streaming variant for the 3x3 / stride 1 / pad 1 layers with equal channel counts on both sides (every BasicBlock
// convolution except the strided ones): same partial-slab output and reduce kernel as above, different data movement.
//
//   * A workgroup (4 compute waves + 8 loader waves, one per CU) owns one (out-channel tile, in-channel tile) and a
//     strip of output rows of one image.  X rows and dY rows stream through two LDS rings of whole rows; row pitch P4
//     cells of 16 channels, cell e holds pixel column e-1, the cells outside the image stay zero.  With that pitch the
//     pixel index is linear inside a row for all three column taps: the X cell of tap b is (dY cell) + b - 1.
//   * Slot = 4 output rows, one per compute wave.  A wave's operand addresses for a row are 3*TKB + TNB registers
//     (row bases, scalar ring arithmetic) + immediates: the MFMA stream carries no address arithmetic at all
//     (a streaming MFMA wave owns the SIMD's vector issue port, see dam_conv_strip.hip).
//   * Loader waves: scalar plane arithmetic, buffer_load with per-lane column offsets computed once, ds_write with the
//     column mask in EXEC.  Two slots ahead in two register sets (a slot is ~4 us of MFMAs, about the HBM latency under load).
struct RowsGeo {
    int B, H, W, C;          // C channels per pixel in X and in dY
    int P4;                  // ring row pitch in cells: roundup4(W + 2)
    int rps, spi;            // output rows per strip (multiple of 4), strips per image
    int tiles_k;             // in-channel tiles
    int gpp;                 // 1 KB pieces per row plane
};
// Diagnostic build only (-DDAM_WGR_STAMPS, tools/wgr_stamps_probe.py; results are wrong): instead of its slab a workgroup leaves
// s_memtime stamps of compute wave 0 and loader wave 0, [role][32] of (tag << 56 | time): 1 start, 2 first rows staged,
// 5 end of a slot's MFMAs / commits, 7 behind the slot's barrier, 8 end.
#ifdef DAM_WGR_STAMPS
#define DAM_WSTAMP(tag)                                                                                                \
    do {                                                                                                              \
        if (lane == 0 && (wave == 0 || wave == 4) && stamp_n < 32)                                                    \
            stamp_v[stamp_n++] = ((unsigned long long)(tag) << 56) | ((unsigned long long)__builtin_readcyclecounter() & ((1ull << 56) - 1)); \
    } while (0)
#else
#define DAM_WSTAMP(tag) do { } while (0)
#endif
constexpr int RW_LOADERS = 8;                 // loader waves (waves 4..11)
constexpr int RW_THREADS = 256 + 64 * RW_LOADERS, RW_GUARD = 64, RW_TAIL = 1280;   // (tail: the step-split last slot prefetches up to 1 KB past a row)
// v % d for the two ring sizes (scalar multiply-shift, v < 30000)
template <int D> __device__ __forceinline__ int rw_mod(int v) { return D == 10 ? v - ((v * 52429) >> 19) * 10 : v - ((v * 43691) >> 18) * 6; }

// HV: 1 = a compute wave takes a whole output row per slot (4 rows per slot), 2 = half a row (2 rows per slot, for rows too
// wide for ten-row rings: the reference's native 216-frame spectrograms); STEPS = MFMA steps per wave and row (part)
// NL: loader waves, 8 or 4.  The hardware starts a workgroup's waves one after the other (the 12th wave of a 768-thread workgroup
// begins ~7 k clocks after the first when all CUs start at once); the narrow stages' few planes per slot do not need eight loaders,
// and at eight waves per workgroup two workgroups really are co-resident on a CU (107 registers x 16 waves).
template <int TNB, int TKB, int STEPS, int KP, int GPP, int HV, int NL>
__global__ __launch_bounds__(256 + 64 * NL) void wgrad_rows_kernel(const RowsGeo g, const float* __restrict__ X,
                                                                const float* __restrict__ dY, const float* __restrict__ in_scale,
                                                                const float* __restrict__ in_shift, int relu_in,
                                                                float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBLK = TNB * TKB * 9;
    constexpr int RPS = 4 / HV, RW_NRX = 2 * RPS + 2, RW_NRD = 2 * RPS;     // rows per slot; ring rows (in use + being written)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    // XCD-aware placement: workgroups go to the 8 XCDs round-robin in dispatch order (x fastest), so the nx workgroups that
    // stream the SAME rows (one per (out tile, in tile)) would sit on nx different L2s and every row would be fetched from the
    // fabric nx times (measured: 6 x the algorithmic bytes on the 96-channel stage).  Re-index so that they share an XCD:
    // position r inside XCD c -> (tile r % nx, strip (r / nx) * 8 + c); the strips beyond a multiple of 8 keep their place.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int nxg = gridDim.x, L = blockIdx.y * nxg + blockIdx.x, full = (gridDim.y >> 3) << 3;
        if (L < full * nxg) { const int c = L & 7, r = L >> 3; bx = r % nxg; by = (r / nxg) * 8 + c; }
    }
    const int tn = bx / g.tiles_k, tk = bx - tn * g.tiles_k;
    // strips: the image's rows in spi nearly equal parts (the feature maps are 2^k + 1 rows high: with strips of whole slots
    // 1025 rows on 32 strips took 9 slots of 4 rows per workgroup for 8.01 slots of work); a last slot of one row is split
    // over all four compute waves by MFMA steps (the waves' sums are added at the end whatever produced them)
    const int img = by / g.spi, sidx = by - img * g.spi;
    const int r_begin = (int)(((long long)sidx * g.H) / g.spi), r_end = (int)(((long long)(sidx + 1) * g.H) / g.spi);
    const int n_slots = (r_end - r_begin + RPS - 1) / RPS;
    const int n_slots2 = (n_slots + 1) & ~1;
    const int ROWB = g.P4 * 64;
    const int XPLANE = RW_NRX * ROWB, DPLANE = RW_NRD * ROWB;
    const int XBASE = RW_GUARD, DBASE = XBASE + TKB * XPLANE;
    const int lds_bytes = DBASE + TNB * DPLANE + RW_TAIL;

    // The cells the loaders never write must read as zero for the lifetime of the workgroup: per ring row the padding cells
    // (cell 0 = column -1, cells W + 1 .. P4 - 1), the guard in front of the first row and the tail behind the last.  Only
    // those are cleared (one pass; clearing the whole 150 KB image took the 12 waves 5.5 k clocks of a 27 k-clock prologue) and
    // they are disjoint from what a commit writes, so the clearing needs no barrier of its own.
#define DAM_RW_ZERO()                                                                                                      \
    do {                                                                                                                   \
        const int npad_ = g.P4 - g.W, nrow_ = TKB * RW_NRX + TNB * RW_NRD;                                                 \
        for (int e = tid; e < nrow_ * npad_ * 4; e += (256 + 64 * NL)) {                                                        \
            const int q_ = e & 3, c_ = (e >> 2) % npad_, r_ = (e >> 2) / npad_;                                            \
            const int cell_ = c_ == 0 ? 0 : g.W + c_;                                                                      \
            *reinterpret_cast<float4*>(smem + XBASE + r_ * ROWB + cell_ * 64 + q_ * 16) = make_float4(0.f, 0.f, 0.f, 0.f); \
        }                                                                                                                  \
        for (int e = tid * 16; e < RW_GUARD; e += (256 + 64 * NL) * 16)                                                         \
            *reinterpret_cast<float4*>(smem + e) = make_float4(0.f, 0.f, 0.f, 0.f);                                        \
        for (int e = tid * 16; e < RW_TAIL; e += (256 + 64 * NL) * 16)                                                          \
            *reinterpret_cast<float4*>(smem + lds_bytes - RW_TAIL + e) = make_float4(0.f, 0.f, 0.f, 0.f);                  \
    } while (0)

    v4f acc[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
#ifdef DAM_WGR_STAMPS
    unsigned long long stamp_v[32];
    int stamp_n = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) stamp_v[i] = 0;
#endif
    DAM_WSTAMP(1);

    if (wave >= 4) {
        // ================================ loader waves ================================
        const int cwl = wave - 4;
        int loffb[GPP];
        unsigned long long cmask[GPP];
#pragma unroll
        for (int gi = 0; gi < GPP; ++gi) {
            const int L = gi * 64 + lane, e = L >> 2, quad = L & 3, col = e - 1;
            const bool ok = gi < g.gpp && e < g.P4 && col >= 0 && col < g.W;
            const int colc = col < 0 ? 0 : (col >= g.W ? g.W - 1 : col);
            loffb[gi] = (colc * g.C + quad * 4) * 4;
            cmask[gi] = __ballot(ok);
        }
        const int img_bytes = g.H * g.W * g.C * 4;
        const __amdgpu_buffer_rsrc_t xrsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X) + (size_t)img * g.H * g.W * g.C, 0, img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t drsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY) + (size_t)img * g.H * g.W * g.C, 0, img_bytes, 0x00020000);
        const int lane16 = lane * 16;
        const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
        const float relu_lo = relu_in ? 0.f : -__builtin_inff();
        const v4f relu_lo4 = {relu_lo, relu_lo, relu_lo, relu_lo};
        // fused input affine on the X planes (the producer's BatchNorm + ReLU): a lane holds channel quad (lane & 3) of a cell
        const bool has_aff = in_scale != nullptr;
        v4f scq[TKB], shq[TKB];
#pragma unroll
        for (int kb = 0; kb < TKB; ++kb) {
            const int ch = (tk * TKB + kb) * 16 + (lane & 3) * 4;
            scq[kb] = has_aff ? *reinterpret_cast<const v4f*>(in_scale + ch) : (v4f){1.f, 1.f, 1.f, 1.f};
            shq[kb] = has_aff ? *reinterpret_cast<const v4f*>(in_shift + ch) : (v4f){0.f, 0.f, 0.f, 0.f};
        }
        v4f lv[KP][GPP];
        int dst[KP];               // scalar: LDS byte offset of the plane | 1 << 30 (row outside the image: zeros), -1 = none
        int aff[KP], aff2[KP];     // scalar: chunk index kb of an X plane that gets the input affine, -1 otherwise
        const int rowstride = g.W * g.C * 4;
        // TIMING EXPERIMENT ONLY (-DDAM_DIAG_DXHAT=1|2, tools/dxhat_ladder.py; results are wrong): what it would cost this kernel to
        // form its dY operand dc = a * (dy . mask) + b * c + k (BatchNorm backward, the bn_bwd_apply launch folded into the
        // loaders) itself: every dY plane is accompanied by a second plane from a THIRD tensor (read through the slab workspace
        // pointer at +128 MB: real HBM traffic, no aliasing with X or dY); 1 = the loads and one fma, 2 = the full arithmetic
        // with the mask recomputed from the second stream.
#ifdef DAM_DIAG_DXHAT
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(partial) + (size_t)(32u << 20) + (size_t)img * g.H * g.W * g.C, 0, img_bytes, 0x00020000);
        v4f lvc_a[KP][GPP], lvc_b[KP][GPP], lvc_c[(2 * TKB + NL - 1) / NL][GPP];
#define DAM_DXHAT_ON 1
#define DAM_DXHAT_REQ(LVC_, K_, SOFF_)                                                                                      \
    _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi)                                                                     \
        LVC_[K_][gi] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(crsrc, loffb[gi], SOFF_, 0))
#if DAM_DIAG_DXHAT == 1
#define DAM_DXHAT_COMBINE(LV_, LVC_, K_, GI_) __builtin_elementwise_fma(LVC_[K_][GI_], shq[0], LV_[K_][GI_])
#else
#define DAM_DXHAT_COMBINE(LV_, LVC_, K_, GI_)                                                                              \
    ([&]() {                                                                                                               \
        const v4f c_ = LVC_[K_][GI_], dy_ = LV_[K_][GI_];                                                    \
        const v4f m_ = __builtin_elementwise_fma(c_, scq[0], shq[0]);                                                      \
        v4f dz_;                                                                                                           \
        dz_.x = m_.x > 0.f ? dy_.x : 0.f; dz_.y = m_.y > 0.f ? dy_.y : 0.f;                                                \
        dz_.z = m_.z > 0.f ? dy_.z : 0.f; dz_.w = m_.w > 0.f ? dy_.w : 0.f;                                                \
        return __builtin_elementwise_fma(dz_, scq[0], __builtin_elementwise_fma(c_, shq[0], relu_lo4));                    \
    }())
#endif
#else
#define DAM_DXHAT_ON 0
#define DAM_DXHAT_REQ(LVC_, K_, SOFF_) do { } while (0)
#define DAM_DXHAT_COMBINE(LV_, LVC_, K_, GI_) (LV_[K_][GI_])
#endif
        // X rows xr0 .. xr0+nx-1 (in-channel chunks of this tile) then dY rows dr0 .. dr0+nd-1 (out-channel blocks), one
        // plane = one (row, 16 channels); loader wave cwl takes planes cwl, cwl+4, ...  Loads are unconditional (clamped).
#define DAM_RW_REQUEST(XR0_, NX_, DR0_, ND_, LV_, DST_, KP_, AFF_, LVC_)                                                                               \
    do {                                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < KP_; ++k) {                                                                  \
            const int pl_ = cwl + NL * k;                                                                          \
            const int nxp_ = (NX_) * TKB;                                                                                  \
            const bool isx_ = pl_ < nxp_;                                                                                  \
            const int q_ = isx_ ? pl_ : pl_ - nxp_;                                                                        \
            const int i_ = isx_ ? q_ / TKB : q_ / TNB, c_ = isx_ ? q_ - i_ * TKB : q_ - i_ * TNB;                          \
            const int row_ = (isx_ ? (XR0_) : (DR0_)) + i_;                                                                \
            const bool need_ = isx_ ? row_ <= r_end : (i_ < (ND_) && row_ < r_end);                                        \
            const bool inimg_ = need_ && row_ >= 0 && row_ < g.H;                                                          \
            const int rel_ = row_ - r_begin + 1;                               /* >= 0 for every needed row */             \
            const int xi_ = rw_mod<RW_NRX>(rel_);                                                                          \
            const int di_ = (row_ - r_begin) & (RW_NRD - 1);                                                               \
            const int ldsoff_ = isx_ ? XBASE + c_ * XPLANE + xi_ * ROWB : DBASE + c_ * DPLANE + di_ * ROWB;                \
            const int chan_ = isx_ ? (tk * TKB + c_) * 64 : (tn * TNB + c_) * 64;                                          \
            const int soff_ = (inimg_ ? row_ : 0) * rowstride + chan_;                                                     \
            if (isx_) {                                                                                                    \
                _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi)                                                         \
                    LV_[k][gi] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, loffb[gi], soff_, 0)); \
            } else {                                                                                                       \
                _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi)                                                         \
                    LV_[k][gi] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(drsrc, loffb[gi], soff_, 0)); \
                DAM_DXHAT_REQ(LVC_, k, soff_);                                                                             \
            }                                                                                                              \
            DST_[k] = need_ ? (ldsoff_ | (inimg_ ? 0 : 1 << 30)) : -1;                                                     \
            AFF_[k] = (isx_ && has_aff) ? c_ : (isx_ ? -1 : -2);                                                           \
        }                                                                                                                  \
    } while (0)
#define DAM_RW_WRITE(ADDR_, DATA_, GI_)                                                                                    \
    asm volatile("s_mov_b64 exec, %2\n\tds_write_b128 %0, %1 offset:%3\n\ts_mov_b64 exec, -1"                              \
                 : : "v"(ADDR_), "v"(DATA_), "s"(cmask[GI_]), "n"((GI_) * 1024) : "memory")
#define DAM_RW_COMMIT(LV_, DST_, KP_, AFF_, LVC_)                                                                              \
    do {                                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < KP_; ++k) {                                                                  \
            if (DST_[k] >= 0) {                                                                                            \
                const int va_ = lane16 + (DST_[k] & 0x3fffffff);                                                           \
                if (DST_[k] >> 30) {                                                                                       \
                    _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi) DAM_RW_WRITE(va_, zero4, gi);                       \
                } else if (AFF_[k] >= 0) {                                                                                 \
                    const v4f sc_ = (TKB > 1 && AFF_[k] > 0) ? scq[TKB - 1] : scq[0];                                      \
                    const v4f sh_ = (TKB > 1 && AFF_[k] > 0) ? shq[TKB - 1] : shq[0];                                      \
                    _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi) {                                                   \
                        v4f v_ = __builtin_elementwise_fma(LV_[k][gi], sc_, sh_);                                          \
                        v_ = __builtin_elementwise_max(v_, relu_lo4);       /* -inf without ReLU: no selects */             \
                        DAM_RW_WRITE(va_, v_, gi);                                                                         \
                    }                                                                                                      \
                } else if (DAM_DXHAT_ON && AFF_[k] == -2) {                                                                \
                    _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi) {                                                   \
                        v4f v_ = DAM_DXHAT_COMBINE(LV_, LVC_, k, gi);                                                          \
                        DAM_RW_WRITE(va_, v_, gi);                                                                         \
                    }                                                                                                      \
                } else {                                                                                                   \
                    _Pragma("unroll") for (int gi = 0; gi < GPP; ++gi) DAM_RW_WRITE(va_, LV_[k][gi], gi);                  \
                }                                                                                                          \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)
        // rows of slot 0: X rows r_begin-1 .. r_begin+RPS, dY rows r_begin .. r_begin+RPS-1: both rounds requested up front
        constexpr int KPB = (2 * TKB + NL - 1) / NL;
        v4f lvb[KPB][GPP], lv2[KP][GPP];
        int dstb[KPB], dst2[KP], affb[KPB];
        DAM_WSTAMP(9);                                                         // setup done
        DAM_RW_REQUEST(r_begin - 1, RPS, r_begin, RPS, lv, dst, KP, aff, lvc_a);
        DAM_RW_REQUEST(r_begin + RPS - 1, 2, r_begin, 0, lvb, dstb, KPB, affb, lvc_c);
        DAM_WSTAMP(10);                                                        // first rows requested
        DAM_RW_ZERO();
        DAM_WSTAMP(11);                                                        // padding cells cleared
        DAM_RW_COMMIT(lv, dst, KP, aff, lvc_a);
        DAM_RW_COMMIT(lvb, dstb, KPB, affb, lvc_c);
        DAM_WSTAMP(12);                                                        // first rows written (their loads have landed)
        // the compute waves start slot 0 HERE; the requests for slots 1 and 2 follow (in front of this barrier they held the
        // first MFMA back by 5-8 k clocks: with three slots of rows wanted by 256 CUs at once the request queue backs up)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // steady state, two slots ahead (HBM latency under this load is about one slot): slot s writes the rows slot s+1
        // adds (X rows r_begin+RPS(s+1)+1 .. +RPS, dY rows r_begin+RPS(s+1) .. +RPS-1; requested during slot s-1) and
        // requests those of slot s+3.  Two register sets alternate; slots come in pairs so that no load sits inside a
        // conditional.
#define DAM_RW_SLOT(T_, LV_, DST_, AFF_, LVC_) DAM_RW_REQUEST(r_begin + RPS * (T_) + 1, RPS, r_begin + RPS * (T_), RPS, LV_, DST_, KP, AFF_, LVC_)
        DAM_WSTAMP(2);
        DAM_RW_SLOT(1, lv, dst, aff, lvc_a);
        DAM_RW_SLOT(2, lv2, dst2, aff2, lvc_b);
        for (int s = 0; s < n_slots2; s += 2) {
            DAM_RW_COMMIT(lv, dst, KP, aff, lvc_a);
            DAM_RW_SLOT(s + 3, lv, dst, aff, lvc_a);
            DAM_WSTAMP(5);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            DAM_WSTAMP(7);
            DAM_RW_COMMIT(lv2, dst2, KP, aff2, lvc_b);
            DAM_RW_SLOT(s + 4, lv2, dst2, aff2, lvc_b);
            DAM_WSTAMP(5);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            DAM_WSTAMP(7);
        }
#undef DAM_RW_SLOT
#undef DAM_RW_REQUEST
#undef DAM_RW_WRITE
#undef DAM_RW_COMMIT
#undef DAM_DXHAT_ON
#undef DAM_DXHAT_REQ
#undef DAM_DXHAT_COMBINE
#ifndef DAM_DIAG_DXHAT
#undef lvc_a
#undef lvc_b
#undef lvc_c
#endif
    } else {
        // ================================ compute waves ================================
        const int cw = wave;
        const int lane_b = kq * 64 + j * 4;       // lane group kq owns cell 4t + kq of step t, lane j channel j of the cell
        DAM_RW_ZERO();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // rows of slot 0 are in LDS, padding cleared
        DAM_WSTAMP(2);
        // operand reads of step t + 1 against the MFMAs of step t: one read behind each of the first MFMAs (as a burst in front
        // of them -- build flag DAM_WGR_NO_INTERLEAVE, scheduling barriers only -- the launch is 1.3-1.6 us slower: the pipe drains
        // while seven reads issue)
#ifndef DAM_WGR_IL_STRIDE
#define DAM_WGR_IL_STRIDE (NL == 4 ? 2 : 1)   /* MFMAs between two operand reads: 2 measured -1.1 us on the four-loader 129x17 kernel, +0.5 us on the others */
#endif
#ifndef DAM_WGR_NO_INTERLEAVE
#define DAM_RW_SCHED_A() do { } while (0)
#define DAM_RW_SCHED_B()                                                                                                   \
    do {                                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                                 \
            __builtin_amdgcn_sched_group_barrier(0x008, DAM_WGR_IL_STRIDE, 0);                                             \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                             \
        }                                                                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 9 * TNB * TKB, 0);                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)
#else
#define DAM_RW_SCHED_A() __builtin_amdgcn_sched_barrier(0)
#define DAM_RW_SCHED_B() __builtin_amdgcn_sched_barrier(0)
#endif
        // one (row, first cell, step stride) assignment of this wave for the slot: F = 1 a whole row part (the steady state),
        // F = 4 / 2 every fourth / second MFMA step of the one / two rows of a strip's last slot
#define DAM_RW_LOAD(T_, BUF_)                                                                                              \
    do {                                                                                                                   \
        _Pragma("unroll") for (int nb = 0; nb < TNB; ++nb)                                                                 \
            av[BUF_][nb] = *reinterpret_cast<const float*>(smem + vd[nb] + (T_) * 256);                                    \
        _Pragma("unroll") for (int kb = 0; kb < TKB; ++kb)                                                                 \
            _Pragma("unroll") for (int a = 0; a < 3; ++a)                                                                  \
                _Pragma("unroll") for (int b = 0; b < 3; ++b)                                                              \
                    bv[BUF_][kb][a * 3 + b] = *reinterpret_cast<const float*>(smem + vx[kb][a] + ((T_) * 256 + b * 64));   \
    } while (0)
#define DAM_RW_ROW(F_, RW_, PART_, SUB_)                      /* steps SUB_, SUB_ + F_, ... of row RW_ of the slot */ \
    do {                                                                                                                   \
        int vx[TKB][3], vd[TNB];                                                                                           \
        _Pragma("unroll") for (int a = 0; a < 3; ++a) {                                                                    \
            const int xi = rw_mod<RW_NRX>(RPS * s + (RW_) + a);               /* row r - 1 + a relative to r_begin - 1 */  \
            _Pragma("unroll") for (int kb = 0; kb < TKB; ++kb) vx[kb][a] = lane_b + (XBASE - 64 + kb * XPLANE + xi * ROWB + (PART_)); \
        }                                                                                                                  \
        _Pragma("unroll") for (int nb = 0; nb < TNB; ++nb)                                                                 \
            vd[nb] = lane_b + (DBASE + nb * DPLANE + ((RPS * s + (RW_)) & (RW_NRD - 1)) * ROWB + (PART_));                 \
        float av[2][TNB], bv[2][TKB][9];                                                                                   \
        constexpr int NS_ = (STEPS + (F_) - 1) / (F_);                                                                     \
        DAM_RW_LOAD(0, 0);                                                                                                 \
        _Pragma("unroll") for (int t = 0; t < NS_; ++t) {                                                                  \
            if ((F_) > 1 && (SUB_) + (F_) * t >= STEPS) break;                /* (scalar; last step of the split form only) */ \
            if (t + 1 < NS_) DAM_RW_LOAD((F_) * (t + 1), (t + 1) & 1);                                                     \
            DAM_RW_SCHED_A();                                                                                              \
            _Pragma("unroll") for (int nb = 0; nb < TNB; ++nb)                                                             \
                _Pragma("unroll") for (int kb = 0; kb < TKB; ++kb)                                                         \
                    _Pragma("unroll") for (int tap = 0; tap < 9; ++tap) {                                                  \
                        const int idx = (nb * TKB + kb) * 9 + tap;                                                         \
                        acc[idx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t & 1][nb], bv[t & 1][kb][tap], acc[idx], 0, 0, 0); \
                    }                                                                                                      \
            DAM_RW_SCHED_B();                                                                                              \
        }                                                                                                                  \
    } while (0)
        for (int s = 0; s < n_slots2; ++s) {
            const int rows_here = r_end - (r_begin + RPS * s);                // rows of this slot (<= 0: none)
            // a strip's last row: every fourth step per wave.  Only for the one-block tile: with 18 accumulator blocks a second
            // copy of the MFMA stream -- unrolled, rolled, or the same copy with an early exit (which halves its speed) -- makes
            // the register allocator spill (168 registers at three waves per SIMD)
            if (HV == 1 && TNB * TKB == 1 && rows_here == 1) {
                DAM_RW_ROW(4, 0, cw * 256, cw);
            } else {
                const int rw = cw / HV, half = cw - rw * HV;                  // row of the slot, part of the row
                if (rw < rows_here) DAM_RW_ROW(1, rw, half * STEPS * 256, 0);
            }
            DAM_WSTAMP(5);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            DAM_WSTAMP(7);
        }
    }
#undef DAM_RW_ROW
#undef DAM_RW_LOAD
#undef DAM_RW_SCHED_A
#undef DAM_RW_SCHED_B

    // combine the 4 compute waves through LDS (sequential adds: fixed order), one slab per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NBLK; ++i) {
                float4* d4 = reinterpret_cast<float4*>(red + (i * 64 + lane) * 4);
                if (w == 0) {
                    *d4 = make_float4(acc[i].x, acc[i].y, acc[i].z, acc[i].w);
                } else {
                    float4 o = *d4;
                    o.x += acc[i].x; o.y += acc[i].y; o.z += acc[i].z; o.w += acc[i].w;
                    *d4 = o;
                }
            }
        }
        __syncthreads();
    }
    float4* out = reinterpret_cast<float4*>(partial) + ((size_t)by * gridDim.x + bx) * NBLK * 64;
#ifdef DAM_WGR_STAMPS
    DAM_WSTAMP(8);
    if (lane == 0 && (wave == 0 || wave == 4)) {
        unsigned long long* sp = reinterpret_cast<unsigned long long*>(out) + (wave >> 2) * 32;
        for (int i = 0; i < 32; ++i) sp[i] = stamp_v[i];
    }
#else
    for (int e = tid; e < NBLK * 64; e += (256 + 64 * NL)) out[e] = reinterpret_cast<const float4*>(red)[e];
#endif
}

#undef DAM_RW_ZERO

template <int TNB, int TKB, int STEPS, int KP, int GPP, int HV = 1, int NL = 8>
int launch_wgrad_rows(int B, int H, int W, int C, const float* X, const float* dY, const float* in_scale, const float* in_shift,
                      int relu_in, float* partial, int64_t ws_floats, float* dw, int n_real, int k_real, void* queue,
                      hipStream_t st) {
    constexpr int NBLK = TNB * TKB * 9;
    RowsGeo g;
    g.B = B; g.H = H; g.W = W; g.C = C;
    constexpr int RPS = 4 / HV, NRX = 2 * RPS + 2, NRD = 2 * RPS;
    g.P4 = ((W + 2 + 4 * HV - 1) / (4 * HV)) * (4 * HV);
    if (g.P4 != STEPS * 4 * HV) return DAM_ERR_UNSUPPORTED;
    g.gpp = (g.P4 * 64 + 1023) / 1024;
    if (g.gpp > GPP || (RPS * (TKB + TNB) + NL - 1) / NL > KP) return DAM_ERR_UNSUPPORTED;
    const int nblk = C / 16;
    if (nblk % TNB || nblk % TKB || H >= 8000) return DAM_ERR_UNSUPPORTED;
    const int tiles_n = nblk / TNB;
    g.tiles_k = nblk / TKB;
    const int nx = tiles_n * g.tiles_k;
    const size_t rowb = (size_t)g.P4 * 64;
    size_t lds = RW_GUARD + (size_t)TKB * NRX * rowb + (size_t)TNB * NRD * rowb + RW_TAIL;
    if (lds < (size_t)NBLK * 1024) lds = (size_t)NBLK * 1024;
    if (lds > 160 * 1024) return DAM_ERR_UNSUPPORTED;
    // Strips: as many workgroups as the chip holds at once and no more (a strip that has to wait for a CU doubles the
    // launch: 96 channels on 8 images gave 288 workgroups on 256 CUs, 63 us).  One workgroup per CU when whole strips per
    // image fill >= 80 % of the CUs that way; otherwise size for two co-resident workgroups per CU (the register file
    // holds two of these 8-wave workgroups, the narrow stages' rings leave the LDS for both): 432 workgroups, 50 us.
    // Measured both ways (DAM_WGR_PERCU=1|2): where one per CU already fills the chip, two cost 15 % (more slabs, shared pipe).
    auto strips = [&](int per_cu) { const int w = 256 * per_cu / (nx * B); return w > 0 ? w : 1; };
    int spi = strips(1);
    if (nx * B * spi < 205 && lds * 2 <= 160 * 1024) spi = strips(2);
    if (const char* e = getenv("DAM_WGR_PERCU")) spi = strips(atoi(e) == 2 && lds * 2 <= 160 * 1024 ? 2 : 1);      // diagnostic
    if (spi > (int)cdiv(H, RPS)) spi = (int)cdiv(H, RPS);
    if (spi > H) spi = H;
    for (;; --spi) {                              // (rows in spi nearly equal parts, see the kernel)
        g.spi = spi;
        g.rps = (int)cdiv(H, spi);
        if ((int64_t)B * g.spi * nx * NBLK * 256 <= ws_floats || spi == 1) break;
    }
    const int nsplit = B * g.spi;
    if ((int64_t)nsplit * nx * NBLK * 256 > ws_floats) return DAM_ERR_WORKSPACE;
    static PerDevice<bool> raised_pd; bool& raised = raised_pd();
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_rows_kernel<TNB, TKB, STEPS, KP, GPP, HV, NL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return DAM_ERR_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((wgrad_rows_kernel<TNB, TKB, STEPS, KP, GPP, HV, NL>), dim3(nx, nsplit), dim3((256 + 64 * NL)), lds, st, g, X, dY, in_scale,
                       in_shift, relu_in, partial);
    DAM_CHECK_LAUNCH();
    return reduce_submit(queue, partial, dw, nsplit, nx, TNB, TKB, 3, 3, n_real, k_real, 3, 3, 1, g.tiles_k, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-streaming variant for 3x3 / stride 2 / pad 1 (the first convolution of a down-sampling block,
// models/model_resnet.py:18-21 of the reference): the same workgroup shape, rings, slabs and reduce kernel as
// wgrad_rows_kernel, with the X rows kept as two column-parity planes so that all three column taps stay "dY cell + constant":
//
//   * X ring row = [odd plane | even plane], P4 cells of 16 channels each: odd cell e holds input column 2e-1 (cell 0 = the
//     left padding column), even cell e holds column 2e.  Output column ow reads tap 0 at odd cell ow, tap 1 at even cell ow
//     and tap 2 at odd cell ow+1; the dY ring row holds output column e in cell e (pitch P4 as well).
//   * Output row r reads X rows 2r-1, 2r, 2r+1.  A slot of RPS output rows brings in the 2*RPS rows 2r, 2r+1 of its output
//     rows (row 2*r_begin-1 comes with slot 0), so the X ring holds 2*RPS+1 rows in use + 2*RPS being written.
//   * one 16-channel chunk of X per workgroup (TKB = 1): the ring of two chunks does not fit beside the dY ring.
struct RowsS2Geo {
    int B, H, W, C;          // X
    int Ho, Wo, N;           // dY
    int rps, spi;            // output rows per strip (multiple of the slot's rows), strips per image
    int tiles_k;             // in-channel tiles (= chunks)
};
template <> __device__ __forceinline__ int rw_mod<18>(int v) { return v - ((v * 58255) >> 20) * 18; }     // v < 36000

// HV as in wgrad_rows_kernel; GPPX / GPPD: 1 KB pieces per X row plane (both parities) / dY row plane; KP planes per loader wave
template <int TNB, int STEPS, int KP, int GPPX, int GPPD, int HV>
__global__ __launch_bounds__(RW_THREADS) void wgrad_rows_s2_kernel(const RowsS2Geo g, const float* __restrict__ X,
                                                                   const float* __restrict__ dY, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBLK = TNB * 9;
    constexpr int RPS = 4 / HV, NRX = 4 * RPS + 2, NRD = 2 * RPS;
    constexpr int P4 = STEPS * 4 * HV;
    constexpr int XROWB = 2 * P4 * 64, DROWB = P4 * 64;
    constexpr int XPLANE = NRX * XROWB, DPLANE = NRD * DROWB;
    constexpr int XBASE = RW_GUARD, DBASE = XBASE + XPLANE;
    constexpr int lds_bytes = DBASE + TNB * DPLANE + RW_TAIL;
    constexpr int NXP = 2 * RPS;                      // X planes a slot brings in
    constexpr int GPP = GPPX > GPPD ? GPPX : GPPD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    // XCD-aware placement as in wgrad_rows_kernel: the nx workgroups of a strip share an L2
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int nxg = gridDim.x, L = blockIdx.y * nxg + blockIdx.x, full = (gridDim.y >> 3) << 3;
        if (L < full * nxg) { const int c = L & 7, r = L >> 3; bx = r % nxg; by = (r / nxg) * 8 + c; }
    }
    const int tn = bx / g.tiles_k, tk = bx - tn * g.tiles_k;
    const int img = by / g.spi, sidx = by - img * g.spi;             // strips and their last slot as in wgrad_rows_kernel
    const int r_begin = (int)(((long long)sidx * g.Ho) / g.spi), r_end = (int)(((long long)(sidx + 1) * g.Ho) / g.spi);
    const int n_slots = (r_end - r_begin + RPS - 1) / RPS;
    const int n_slots2 = (n_slots + 1) & ~1;

#define DAM_RW_ZERO()                                                                                                      \
    do {                                                                                                                   \
        for (int e = tid * 16; e < lds_bytes; e += RW_THREADS * 16)                                                        \
            *reinterpret_cast<float4*>(smem + e) = make_float4(0.f, 0.f, 0.f, 0.f);                                        \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                    \
    } while (0)

    v4f acc[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};

    if (wave >= 4) {
        // ================================ loader waves ================================
        // plane cwl + 8k of a slot: the first NXP are X rows, the rest dY (row, out-channel block) planes -- which kind a
        // (wave, k) pair carries never changes, so the per-lane column offsets and write masks are set up once per k
        const int cwl = wave - 4;
        int loffb[KP][GPP];
        unsigned long long cmask[KP][GPP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const bool isx = cwl + RW_LOADERS * k < NXP;
#pragma unroll
            for (int gi = 0; gi < GPP; ++gi) {
                const int L = gi * 64 + lane, e = L >> 2, quad = L & 3;
                const int col = isx ? (e < P4 ? 2 * e - 1 : 2 * (e - P4)) : e;
                const int lim = isx ? g.W : g.Wo, cells = isx ? 2 * P4 : P4;
                const bool ok = e < cells && col >= 0 && col < lim;
                const int colc = col < 0 ? 0 : (col >= lim ? lim - 1 : col);
                loffb[k][gi] = (colc * (isx ? g.C : g.N) + quad * 4) * 4;
                cmask[k][gi] = __ballot(ok);
            }
        }
        const int ximg_bytes = g.H * g.W * g.C * 4, dimg_bytes = g.Ho * g.Wo * g.N * 4;
        const __amdgpu_buffer_rsrc_t xrsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X) + (size_t)img * g.H * g.W * g.C, 0, ximg_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t drsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY) + (size_t)img * g.Ho * g.Wo * g.N, 0, dimg_bytes, 0x00020000);
        const int lane16 = lane * 16;
        const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
        const int xrowstride = g.W * g.C * 4, drowstride = g.Wo * g.N * 4;
        const int x_first = 2 * r_begin - 1, x_last = 2 * r_end - 1;            // X rows this strip reads
        v4f lv[KP][GPP], lv2[KP][GPP], lvb[GPPX];
        int dst[KP], dst2[KP], dstb;   // scalar: LDS byte offset of the plane | 1 << 30 (row outside the image: zeros), -1 = none
        // X rows XR0_ .. XR0_+NXP-1, then dY rows DR0_ .. DR0_+RPS-1 (TNB planes each).  Loads are unconditional (clamped).
#define DAM_RW2_REQUEST(XR0_, DR0_, LV_, DST_)                                                                             \
    do {                                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < KP; ++k) {                                                                   \
            const int pl_ = cwl + RW_LOADERS * k;                                                                          \
            const bool isx_ = pl_ < NXP;                                                                                   \
            const int q_ = isx_ ? pl_ : pl_ - NXP;                                                                         \
            const int i_ = isx_ ? q_ : q_ / TNB, c_ = isx_ ? 0 : q_ - i_ * TNB;                                            \
            const int row_ = (isx_ ? (XR0_) : (DR0_)) + i_;                                                                \
            const bool need_ = isx_ ? row_ <= x_last : (i_ < RPS && row_ < r_end);                                         \
            const bool inimg_ = need_ && row_ >= 0 && row_ < (isx_ ? g.H : g.Ho);                                          \
            const int xi_ = rw_mod<NRX>(isx_ ? row_ - x_first : 0);                                                        \
            const int di_ = (row_ - r_begin) & (NRD - 1);                                                                  \
            const int ldsoff_ = isx_ ? XBASE + xi_ * XROWB : DBASE + c_ * DPLANE + di_ * DROWB;                            \
            const int soff_ = (inimg_ ? row_ : 0) * (isx_ ? xrowstride : drowstride) + (isx_ ? tk * 64 : (tn * TNB + c_) * 64); \
            if (isx_) {                                                                                                    \
                _Pragma("unroll") for (int gi = 0; gi < GPPX; ++gi)                                                        \
                    LV_[k][gi] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, loffb[k][gi], soff_, 0)); \
            } else {                                                                                                       \
                _Pragma("unroll") for (int gi = 0; gi < GPPD; ++gi)                                                        \
                    LV_[k][gi] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(drsrc, loffb[k][gi], soff_, 0)); \
            }                                                                                                              \
            DST_[k] = need_ ? (ldsoff_ | (inimg_ ? 0 : 1 << 30) | (isx_ ? 1 << 29 : 0)) : -1;                              \
        }                                                                                                                  \
    } while (0)
#define DAM_RW2_WRITE(ADDR_, DATA_, MASK_, GI_)                                                                            \
    asm volatile("s_mov_b64 exec, %2\n\tds_write_b128 %0, %1 offset:%3\n\ts_mov_b64 exec, -1"                              \
                 : : "v"(ADDR_), "v"(DATA_), "s"(MASK_), "n"((GI_) * 1024) : "memory")
#define DAM_RW2_COMMIT(LV_, DST_)                                                                                          \
    do {                                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < KP; ++k) {                                                                   \
            if (DST_[k] >= 0) {                                                                                            \
                const int va_ = lane16 + (DST_[k] & 0x1fffffff);                                                           \
                const bool zero_ = (DST_[k] >> 30) & 1;                                                                    \
                if ((DST_[k] >> 29) & 1) {                                                                                 \
                    _Pragma("unroll") for (int gi = 0; gi < GPPX; ++gi) {                                                  \
                        const v4f v_ = zero_ ? zero4 : LV_[k][gi];                                                         \
                        DAM_RW2_WRITE(va_, v_, cmask[k][gi], gi);                                                          \
                    }                                                                                                      \
                } else {                                                                                                   \
                    _Pragma("unroll") for (int gi = 0; gi < GPPD; ++gi) {                                                  \
                        const v4f v_ = zero_ ? zero4 : LV_[k][gi];                                                         \
                        DAM_RW2_WRITE(va_, v_, cmask[k][gi], gi);                                                          \
                    }                                                                                                      \
                }                                                                                                          \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)
        // slot 0: X rows 2*r_begin .. +NXP-1 and dY rows r_begin .. +RPS-1 like every slot, plus X row 2*r_begin-1 (loader
        // wave 0, whose k = 0 plane is an X plane: its offsets and masks apply)
        DAM_RW2_REQUEST(2 * r_begin, r_begin, lv, dst);
        {
            const bool inimg = x_first >= 0;
            const int soff = (inimg ? x_first : 0) * xrowstride + tk * 64;
#pragma unroll
            for (int gi = 0; gi < GPPX; ++gi)
                lvb[gi] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, loffb[0][gi], soff, 0));
            dstb = cwl == 0 ? (XBASE | (inimg ? 0 : 1 << 30)) : -1;
        }
        DAM_RW_ZERO();
        DAM_RW2_COMMIT(lv, dst);
        if (dstb >= 0) {
            const int va = lane16 + (dstb & 0x1fffffff);
#pragma unroll
            for (int gi = 0; gi < GPPX; ++gi) {
                const v4f v = (dstb >> 30) & 1 ? zero4 : lvb[gi];
                DAM_RW2_WRITE(va, v, cmask[0][gi], gi);
            }
        }
        // steady state as in wgrad_rows_kernel: slot s writes the rows of slot s+1 and requests those of slot s+3
#define DAM_RW2_SLOT(T_, LV_, DST_) DAM_RW2_REQUEST(2 * (r_begin + RPS * (T_)), r_begin + RPS * (T_), LV_, DST_)
        DAM_RW2_SLOT(1, lv, dst);
        DAM_RW2_SLOT(2, lv2, dst2);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int s = 0; s < n_slots2; s += 2) {
            DAM_RW2_COMMIT(lv, dst);
            DAM_RW2_SLOT(s + 3, lv, dst);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            DAM_RW2_COMMIT(lv2, dst2);
            DAM_RW2_SLOT(s + 4, lv2, dst2);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#undef DAM_RW2_SLOT
#undef DAM_RW2_REQUEST
#undef DAM_RW2_WRITE
#undef DAM_RW2_COMMIT
    } else {
        // ================================ compute waves ================================
        const int cw = wave;
        const int lane_b = kq * 64 + j * 4;       // lane group kq owns cell 4t + kq of step t, lane j channel j of the cell
        DAM_RW_ZERO();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // rows of slot 0 are in LDS
#ifdef DAM_WGR_S2_INTERLEAVE                     /* as in wgrad_rows_kernel; here measured SLOWER (42.0 -> 43.6, 40.3 -> 42.2, 46.2 -> 47.0 us): off */
#define DAM_RW2_SCHED_A() do { } while (0)
#define DAM_RW2_SCHED_B()                                                                                                  \
    do {                                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                             \
        }                                                                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 9 * TNB, 0);                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)
#else
#define DAM_RW2_SCHED_A() __builtin_amdgcn_sched_barrier(0)
#define DAM_RW2_SCHED_B() __builtin_amdgcn_sched_barrier(0)
#endif
        // column taps: odd-plane cell ow, even-plane cell ow, odd-plane cell ow + 1
#define DAM_RW2_LOAD(T_, BUF_)                                                                                             \
    do {                                                                                                                   \
        _Pragma("unroll") for (int nb = 0; nb < TNB; ++nb)                                                                 \
            av[BUF_][nb] = *reinterpret_cast<const float*>(smem + vd[nb] + (T_) * 256);                                    \
        _Pragma("unroll") for (int a = 0; a < 3; ++a) {                                                                    \
            bv[BUF_][a * 3 + 0] = *reinterpret_cast<const float*>(smem + vx[a] + (T_) * 256);                              \
            bv[BUF_][a * 3 + 1] = *reinterpret_cast<const float*>(smem + vx[a] + ((T_) * 256 + P4 * 64));                  \
            bv[BUF_][a * 3 + 2] = *reinterpret_cast<const float*>(smem + vx[a] + ((T_) * 256 + 64));                       \
        }                                                                                                                  \
    } while (0)
#define DAM_RW2_ROW(F_, RW_, PART_, SUB_)                     /* as DAM_RW_ROW of wgrad_rows_kernel */                      \
    do {                                                                                                                   \
        int vx[3], vd[TNB];                                                                                                \
        _Pragma("unroll") for (int a = 0; a < 3; ++a)                         /* X row 2r - 1 + a, relative to 2*r_begin - 1 */ \
            vx[a] = lane_b + (XBASE + rw_mod<NRX>(2 * (RPS * s + (RW_)) + a) * XROWB + (PART_));                           \
        _Pragma("unroll") for (int nb = 0; nb < TNB; ++nb)                                                                 \
            vd[nb] = lane_b + (DBASE + nb * DPLANE + ((RPS * s + (RW_)) & (NRD - 1)) * DROWB + (PART_));                   \
        float av[2][TNB], bv[2][9];                                                                                        \
        constexpr int NS_ = (STEPS + (F_) - 1) / (F_);                                                                     \
        DAM_RW2_LOAD(0, 0);                                                                                                \
        _Pragma("unroll") for (int t = 0; t < NS_; ++t) {                                                                  \
            if ((F_) > 1 && (SUB_) + (F_) * t >= STEPS) break;                                                             \
            if (t + 1 < NS_) DAM_RW2_LOAD((F_) * (t + 1), (t + 1) & 1);                                                    \
            DAM_RW2_SCHED_A();                                                                                             \
            _Pragma("unroll") for (int nb = 0; nb < TNB; ++nb)                                                             \
                _Pragma("unroll") for (int tap = 0; tap < 9; ++tap)                                                        \
                    acc[nb * 9 + tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t & 1][nb], bv[t & 1][tap], acc[nb * 9 + tap], 0, 0, 0); \
            DAM_RW2_SCHED_B();                                                                                             \
        }                                                                                                                  \
    } while (0)
        for (int s = 0; s < n_slots2; ++s) {
            const int rows_here = r_end - (r_begin + RPS * s);                // rows of this slot (<= 0: none)
            if (HV == 1 && TNB == 1 && rows_here == 1) {
                DAM_RW2_ROW(4, 0, cw * 256, cw);
            } else {
                const int rw = cw / HV, half = cw - rw * HV;                  // row of the slot, part of the row
                if (rw < rows_here) DAM_RW2_ROW(1, rw, half * STEPS * 256, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
#undef DAM_RW2_ROW
#undef DAM_RW2_LOAD
#undef DAM_RW2_SCHED_A
#undef DAM_RW2_SCHED_B
#undef DAM_RW_ZERO

    // combine the 4 compute waves through LDS (sequential adds: fixed order), one slab per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NBLK; ++i) {
                float4* d4 = reinterpret_cast<float4*>(red + (i * 64 + lane) * 4);
                if (w == 0) {
                    *d4 = make_float4(acc[i].x, acc[i].y, acc[i].z, acc[i].w);
                } else {
                    float4 o = *d4;
                    o.x += acc[i].x; o.y += acc[i].y; o.z += acc[i].z; o.w += acc[i].w;
                    *d4 = o;
                }
            }
        }
        __syncthreads();
    }
    float4* out = reinterpret_cast<float4*>(partial) + ((size_t)by * gridDim.x + bx) * NBLK * 64;
    for (int e = tid; e < NBLK * 64; e += RW_THREADS) out[e] = reinterpret_cast<const float4*>(red)[e];
}

template <int TNB, int STEPS, int KP, int GPPX, int GPPD, int HV>
int launch_wgrad_rows_s2(int B, int H, int W, int C, int Ho, int Wo, int N, const float* X, const float* dY, float* partial,
                         int64_t ws_floats, float* dw, int n_real, int k_real, void* queue, hipStream_t st) {
    constexpr int NBLK = TNB * 9;
    constexpr int RPS = 4 / HV, NRX = 4 * RPS + 2, NRD = 2 * RPS, P4 = STEPS * 4 * HV;
    static_assert(NRX == 18 || NRX == 10, "ring sizes with a multiply-shift modulo");
    static_assert((2 * RPS + RPS * TNB + RW_LOADERS - 1) / RW_LOADERS <= KP, "planes per loader wave");
    static_assert(2 * P4 * 64 <= GPPX * 1024 && P4 * 64 <= GPPD * 1024, "pieces per plane");
    if (Ho != (H - 1) / 2 + 1 || Wo != (W - 1) / 2 + 1) return DAM_ERR_UNSUPPORTED;
    if (((Wo + 1 + 4 * HV - 1) / (4 * HV)) * (4 * HV) != P4) return DAM_ERR_UNSUPPORTED;
    const int nblk = N / 16;
    if (nblk % TNB || C % 16 || H >= 8000) return DAM_ERR_UNSUPPORTED;
    if ((int64_t)H * W * C * 4 >= (1ll << 31) || (int64_t)Ho * Wo * N * 4 >= (1ll << 31)) return DAM_ERR_UNSUPPORTED;
    RowsS2Geo g;
    g.B = B; g.H = H; g.W = W; g.C = C; g.Ho = Ho; g.Wo = Wo; g.N = N;
    const int tiles_n = nblk / TNB;
    g.tiles_k = C / 16;
    const int nx = tiles_n * g.tiles_k;
    size_t lds = RW_GUARD + (size_t)NRX * 2 * P4 * 64 + (size_t)TNB * NRD * P4 * 64 + RW_TAIL;
    if (lds < (size_t)NBLK * 1024) lds = (size_t)NBLK * 1024;
    if (lds > 160 * 1024) return DAM_ERR_UNSUPPORTED;
    // strips as in launch_wgrad_rows: fill the chip once
    auto strips = [&](int per_cu) { const int w = 256 * per_cu / (nx * B); return w > 0 ? w : 1; };
    int spi = strips(1);
    if (nx * B * spi < 205 && lds * 2 <= 160 * 1024) spi = strips(2);
    if (spi > (int)cdiv(Ho, RPS)) spi = (int)cdiv(Ho, RPS);
    if (spi > Ho) spi = Ho;
    for (;; --spi) {                              // (rows in spi nearly equal parts, see the kernel)
        g.spi = spi;
        g.rps = (int)cdiv(Ho, spi);
        if ((int64_t)B * g.spi * nx * NBLK * 256 <= ws_floats || spi == 1) break;
    }
    const int nsplit = B * g.spi;
    if ((int64_t)nsplit * nx * NBLK * 256 > ws_floats) return DAM_ERR_WORKSPACE;
    static PerDevice<bool> raised_pd; bool& raised = raised_pd();
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_rows_s2_kernel<TNB, STEPS, KP, GPPX, GPPD, HV>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return DAM_ERR_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((wgrad_rows_s2_kernel<TNB, STEPS, KP, GPPX, GPPD, HV>), dim3(nx, nsplit), dim3(RW_THREADS), lds, st, g, X, dY,
                       partial);
    DAM_CHECK_LAUNCH();
    return reduce_submit(queue, partial, dw, nsplit, nx, TNB, 1, 3, 3, n_real, k_real, 3, 3, 1, g.tiles_k, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Direct variant: nothing goes through LDS.  dW[n][k][a][b] = sum over output pixels of dY[p][n] * X[s*p + tap][k]; a
// wave walks whole output rows, lane = (pixel 4t + kq, channel j), both MFMA operands are buffer loads (dword per lane,
// 64-byte segments) straight from HBM / L2 -- every X element is wanted by at most KH*KW/s^2 taps and those re-reads are
// L2 hits.  Addresses are linear in t; lanes past the row end or on a padding column get an out-of-range offset and read
// 0; a tap row outside the image is skipped (scalar branch).  U steps of loads are in flight per wave.
//   1x1: the strided shortcut of a down-sampling block (models/model_resnet.py:18-21) -- a skinny memory-bound GEMM;
//   3x3: the strided first convolution of those blocks (a staged patch would hold both column parities of every row).
// Output: the same partial slabs + reduce kernel as the other variants.
// (bx, by) of a (gx, gy) grid: the launch's own block indices, or a job's share of a batched launch
template <int TNB, int TKB, int KH, int KW>
__device__ __forceinline__ void wgrad_direct_body(const float* __restrict__ X, const float* __restrict__ dY, int rows, int Ho, int Wo,
                                                  int H, int W, int C, int N, int s, int pad, int dil, int tiles_k,
                                                  float* __restrict__ partial, int bx, int by, int gx, int gy) {
    constexpr int NBLK = TNB * TKB * KH * KW, U = KH * KW == 1 ? 8 : 4;
    __shared__ float4 red[NBLK * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int tn = bx / tiles_k, tk = bx - tn * tiles_k;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY), 0, 0x7fffffff, 0x00020000);
    v4f acc[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int steps = (Wo + 3) >> 2;
    const int lane_x = (kq * s * C + tk * TKB * 16 + j) * 4, lane_d = (kq * N + tn * TNB * 16 + j) * 4;   // bytes
    const int step_x = 4 * s * C * 4, step_d = 4 * N * 4;
    for (int row = by * 4 + wave; row < rows; row += gy * 4) {        // row = (image, output row)
        const int img = row / Ho, oh = row - img * Ho;
        const int ds = (int)(((int64_t)row * Wo) * N * 4);                           // scalar byte offsets (< 2^31: host check)
        for (int t0 = 0; t0 < steps; t0 += U) {
            float av[U][TNB];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u;
                const int vd = (t < steps && 4 * t + kq < Wo) ? lane_d + t * step_d : 0x7fffffff;
#pragma unroll
                for (int nb = 0; nb < TNB; ++nb)
                    av[u][nb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dr, vd + nb * 64, ds, 0));
            }
#pragma unroll
            for (int a = 0; a < KH; ++a) {
                const int ih = oh * s + a * dil - pad;
                if (ih < 0 || ih >= H) continue;                                     // wave-uniform
                const int xs = (int)(((int64_t)(img * H + ih) * W) * C * 4);
                float bv[U][KW][TKB];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = t0 + u;
#pragma unroll
                    for (int b = 0; b < KW; ++b) {
                        const int iw = (4 * t + kq) * s + b * dil - pad;
                        const int vx = (t < steps && 4 * t + kq < Wo && iw >= 0 && iw < W) ? lane_x + t * step_x + (b * dil - pad) * C * 4
                                                                                         : 0x7fffffff;
#pragma unroll
                        for (int kb = 0; kb < TKB; ++kb)
                            bv[u][b][kb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, vx + kb * 64, xs, 0));
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int nb = 0; nb < TNB; ++nb)
#pragma unroll
                        for (int kb = 0; kb < TKB; ++kb)
#pragma unroll
                            for (int b = 0; b < KW; ++b) {
                                const int idx = ((nb * TKB + kb) * KH + a) * KW + b;
                                acc[idx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][nb], bv[u][b][kb], acc[idx], 0, 0, 0);
                            }
            }
        }
    }
    for (int w = 0; w < 4; ++w) {            // combine the 4 waves in fixed order
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NBLK; ++i) {
                float4 o = w == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : red[i * 64 + lane];
                o.x += acc[i].x; o.y += acc[i].y; o.z += acc[i].z; o.w += acc[i].w;
                red[i * 64 + lane] = o;
            }
        }
        __syncthreads();
    }
    float4* out = reinterpret_cast<float4*>(partial) + ((size_t)by * gx + bx) * NBLK * 64;
    for (int e = tid; e < NBLK * 64; e += 256) out[e] = red[e];
}

template <int TNB, int TKB, int KH, int KW>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(const float* __restrict__ X, const float* __restrict__ dY, int rows,
                                                           int Ho, int Wo, int H, int W, int C, int N, int s, int pad, int dil,
                                                           int tiles_k, float* __restrict__ partial) {
    wgrad_direct_body<TNB, TKB, KH, KW>(X, dY, rows, Ho, Wo, H, W, C, N, s, pad, dil, tiles_k, partial, blockIdx.x, blockIdx.y,
                                        gridDim.x, gridDim.y);
}

// Direct launches of ONE instantiation and DIFFERENT geometries in one launch (round 5): the 1x1 shortcut convolutions of the
// four 32 -> 256-channel down-sampling blocks (models/model_resnet.py:18-21) are 11-16 us launches of a memory- and latency-bound
// kernel, leaves of the backward pass; recorded in a batching queue they run as one flat grid at the flush (a workgroup finds its
// job by the prefix sums `wg0`), the jobs sharing the chip's workgroup slots (launch_pending_direct).
template <int TNB, int TKB, int KH, int KW>
__global__ __launch_bounds__(256) void wgrad_direct_batch_kernel(const DirectJobs jobs) {
    int ji = 0;
    while (ji + 1 < jobs.njobs && (int)blockIdx.x >= jobs.job[ji + 1].wg0) ++ji;
    const DirectJob& j = jobs.job[ji];
    const int li = (int)blockIdx.x - j.wg0;
    wgrad_direct_body<TNB, TKB, KH, KW>(j.X, j.dY, j.rows, j.Ho, j.Wo, j.H, j.W, j.C, j.N, j.s, j.pad, j.dil, j.tiles_k, j.partial,
                                        li % j.nx, li / j.nx, j.nx, j.nsplit);
}

template <int TNB, int TKB, int KH, int KW>
int launch_pending_direct(WgradQueue* q, hipStream_t st) {
    PendingDirect& p = q->pend_direct;
    const int n = p.jobs.njobs;
    p.jobs.njobs = 0;
    if (n <= 0) return DAM_OK;
    // Pixel splits: alone, a launch wants 3-4 workgroups per CU (`nsplit` as recorded); n jobs in one launch fill the chip
    // together, so each takes its share -- a quarter of the slabs to write and to reduce, four times the rows per wave (the
    // 129 x 17 shortcut had ONE 17-pixel row per wave: 40 us for the four jobs, 8.4 MB of slabs for a 32 KB gradient).  A batch of
    // one keeps the split of the plain launch (bitwise the same slabs).
    int total = 0;
    for (int i = 0; i < n; ++i) {
        DirectJob& j = p.jobs.job[i];
        if (n > 1) {
            const int share = (int)cdiv(1536, (int64_t)n * j.nx);
            if (share < j.nsplit) j.nsplit = share < 1 ? 1 : share;
        }
        j.wg0 = total;
        total += j.nx * j.nsplit;
    }
    DirectJobs jobs = p.jobs;
    jobs.njobs = n;
    hipLaunchKernelGGL((wgrad_direct_batch_kernel<TNB, TKB, KH, KW>), dim3(total), dim3(256), 0, st, jobs);
    DAM_CHECK_LAUNCH();
    for (int i = 0; i < n; ++i) {
        const DirectJob& j = jobs.job[i];
        const int rc = reduce_submit(q, j.partial, p.dw[i], j.nsplit, j.nx, TNB, TKB, KH, KW, p.n_real[i], p.k_real[i], KH, KW, 1,
                                     j.tiles_k, st);
        if (rc != DAM_OK) return rc;
    }
    return DAM_OK;
}

template <int TNB, int TKB, int KH, int KW>
int launch_wgrad_direct(int B, int H, int W, int C, int Ho, int Wo, int N, int s, int pad, int dil, const float* X,
                        const float* dY, float* partial, int64_t ws_floats, float* dw, int n_real, int k_real, void* queue,
                        hipStream_t st) {
    constexpr int NBLK = TNB * TKB * KH * KW;
    const int nblk = N / 16, nch = C / 16;
    if (nblk % TNB || nch % TKB) return DAM_ERR_UNSUPPORTED;
    if ((int64_t)B * H * W * C * 4 >= (1ll << 31) || (int64_t)B * Ho * Wo * N * 4 >= (1ll << 31)) return DAM_ERR_UNSUPPORTED;
    const int tiles_n = nblk / TNB, tiles_k = nch / TKB, nx = tiles_n * tiles_k;
    const int rows = B * Ho;
    int nsplit = (int)cdiv(KH * KW == 1 ? 1024 : 768, nx);      // 3-4 workgroups per CU: the kernel lives on loads in flight
    if (nsplit > (int)cdiv(rows, 4)) nsplit = (int)cdiv(rows, 4);
    while (nsplit > 1 && (int64_t)nsplit * nx * NBLK * 256 > ws_floats) --nsplit;
    if ((int64_t)nsplit * nx * NBLK * 256 > ws_floats) return DAM_ERR_WORKSPACE;
    WgradQueue* q = static_cast<WgradQueue*>(queue);
    if (q && q->magic != WGRAD_QUEUE_MAGIC) return DAM_ERR_BAD_ARG;
    static const bool no_batch = getenv("DAM_WG_DIRECT_NO_BATCH") != nullptr;      // A/B switch
    if (q && q->batching && !no_batch) {
        // record instead of launching: launches of this instantiation wait for each other until the queue is flushed (the caller
        // keeps X, dY and the slabs valid until then: the batching contract of include/dam_hip.h)
        PendingDirect& p = q->pend_direct;
        constexpr int sig = TNB | TKB << 4 | KH << 8 | KW << 12;
        if (p.jobs.njobs > 0 && (p.sig != sig || p.jobs.njobs == WD_MAX_JOBS)) {
            const int rc = p.launch(q, st);
            if (rc != DAM_OK) return rc;
        }
        const int i = p.jobs.njobs++;
        p.sig = sig; p.launch = &launch_pending_direct<TNB, TKB, KH, KW>;
        DirectJob& j = p.jobs.job[i];
        j.X = X; j.dY = dY; j.partial = partial; j.rows = rows; j.Ho = Ho; j.Wo = Wo; j.H = H; j.W = W; j.C = C; j.N = N; j.s = s;
        j.pad = pad; j.dil = dil; j.tiles_k = tiles_k; j.nx = nx; j.nsplit = nsplit; j.wg0 = 0;
        p.dw[i] = dw; p.n_real[i] = n_real; p.k_real[i] = k_real;
        return DAM_OK;
    }
    hipLaunchKernelGGL((wgrad_direct_kernel<TNB, TKB, KH, KW>), dim3(nx, nsplit), dim3(256), 0, st, X, dY, rows, Ho, Wo, H, W, C,
                       N, s, pad, dil, tiles_k, partial);
    DAM_CHECK_LAUNCH();
    return reduce_submit(queue, partial, dw, nsplit, nx, TNB, TKB, KH, KW, n_real, k_real, KH, KW, 1, tiles_k, st);
}

// One launch of the tile kernel for `njobs` weight gradients of geometry g (blockIdx.z = job), then their slab reductions.
template <int TNB, int TKB, int TA, int TB>
int launch_wgrad_jobs(WgradGeo g, const WgradJobs& jobs, int njobs, int64_t ws_floats, float* const* dw, const int* n_real,
                      const int* k_real, void* queue, hipStream_t st) {
    constexpr int NBLK = TNB * TKB * TA * TB;
    const int nx = g.tiles_n * g.tiles_k * g.tap_groups;
    size_t lds0 = (size_t)TKB * g.PR * g.PWT * 64 + (size_t)TNB * g.TMW * 64;
    // Loader-wave form (two LDS images, one 8-wave workgroup per CU) where two images fit.  MEASURED SLOWER than two plain
    // workgroups per CU (profiles/r05_wgrad_loader_waves_ab.txt: C2 5.28 -> 5.52 ms, C1 3.71 -> 3.86, C3 +8 us): one MFMA wave per
    // SIMD beside a VALU-heavy staging wave hides LDS operand latency worse than two MFMA waves taking turns.  OFF by default;
    // DAM_WG_LW=1 enables it for the big-tile instantiations, 2 for all (A/B).
    static const int lw_env = [] { const char* e = getenv("DAM_WG_LW"); return e ? atoi(e) : 0; }();
    // (default: the big-tile instantiations only -- 2x2 / 3x2 channel blocks hold 144+ accumulator registers, so at most two of
    //  their plain workgroups share a CU; the one-block tiles run up to five waves per SIMD and overlap by themselves)
    const bool lw_fits = lw_env != 0 && (TNB * TKB >= 4 || lw_env == 2) && 2 * lds0 <= 160 * 1024 && 2 * lds0 >= (size_t)NBLK * 1024;
    auto split_for = [&](int64_t slots, double fixed) {
        // Pixel split by MAKESPAN: every CU works through ceil(workgroups / slots) workgroups of ceil(total_tiles / nsplit) tiles
        // each (co-resident workgroups share the matrix pipe, so it is the count per CU that matters).  The first rule -- at least
        // 512 workgroups -- gave the 9x9 / 64 -> 128 layer of the scalar models 72 x 8 = 576 workgroups of 40 tiles: a third round
        // for a quarter of the chip (1.64 ms); 72 x 7 = 504 of 46 tiles is two rounds.  Batched jobs multiply the workgroups, not the
        // tiles of one: three jobs need a third of the splits -- and slabs.
        int ns_best = 1;
        double best = 1e300;
        // (up to 512 splits: a layer with ONE channel tile -- the scalar models' 4 -> 16 first convolution -- had 63 workgroups on
        // 256 CUs under the former cap of 64: 170 us for 0.15 GFLOP)
        const int hi = g.total_tiles < 512 ? g.total_tiles : 512;
        for (int ns = 1; ns <= hi; ++ns) {
            const int64_t wgs = (int64_t)nx * ns * njobs;
            if ((int64_t)nx * ns * NBLK * 256 > ws_floats && ns > 1) break;
            const double cost = (double)cdiv(wgs, slots) * ((double)cdiv(g.total_tiles, ns) + fixed);   // + fixed: per-workgroup part
            if (cost < best * 0.999) { best = cost; ns_best = ns; }
        }
        return ns_best;
    };
    int nsplit;
    bool lw = false;
    static const bool old_rule = getenv("DAM_WG_NSPLIT_OLD") != nullptr;
    if (old_rule) {                          // DAM_WG_NSPLIT_OLD: the first rule (A/B)
        nsplit = (int)cdiv(512, (int64_t)nx * njobs);
        if (nsplit > g.total_tiles) nsplit = g.total_tiles;
        if (nsplit < 1) nsplit = 1;
    } else {
        // slots: two workgroups per CU where the LDS holds two (the small-patch layers: their staging phases overlap each
        // other -- 256 workgroups of two tiles measured slower than 512 of one on the 33 x 5 stage), one otherwise
        if (lds0 < (size_t)NBLK * 1024) lds0 = (size_t)NBLK * 1024;
        static const int slots_forced = [] { const char* e = getenv("DAM_WG_SLOTS"); return e ? atoi(e) : 0; }();      // A/B knob
        const int64_t slots = slots_forced ? slots_forced : (lds0 * 2 <= 160 * 1024 ? 512 : 256);
        nsplit = split_for(slots, 1.0);
        if (lw_fits) {
            // the loader-wave form is one workgroup per CU and pays its staging only for the FIRST tile: worth it when a
            // workgroup has several tiles to stream (lw_env 2 forces it)
            const int ns_lw = split_for(256, 1.0);
            if (cdiv(g.total_tiles, ns_lw) >= 2 || lw_env == 2) { lw = true; nsplit = ns_lw; }
        }
    }
    while (nsplit > 1 && (int64_t)nsplit * nx * NBLK * 256 > ws_floats) --nsplit;
    if ((int64_t)nsplit * nx * NBLK * 256 > ws_floats) return DAM_ERR_WORKSPACE;
    g.nsplit = nsplit;
    size_t lds = (size_t)TKB * g.PR * g.PWT * 64 + (size_t)TNB * g.TMW * 64;
    if (lw) lds *= 2;
    if (lds < (size_t)NBLK * 1024) lds = (size_t)NBLK * 1024;
    if (lds > 160 * 1024) return DAM_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        static PerDevice<bool> raised_pd[2];
        bool& raised = raised_pd[lw ? 1 : 0]();
        if (!raised) {
            const void* fn = lw ? reinterpret_cast<const void*>(&wgrad_kernel<TNB, TKB, TA, TB, true>)
                                : reinterpret_cast<const void*>(&wgrad_kernel<TNB, TKB, TA, TB, false>);
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DAM_ERR_LAUNCH;
            raised = true;
        }
    }
    if (lw)
        hipLaunchKernelGGL((wgrad_kernel<TNB, TKB, TA, TB, true>), dim3(nx, nsplit, njobs), dim3(512), lds, st, g, jobs);
    else
        hipLaunchKernelGGL((wgrad_kernel<TNB, TKB, TA, TB, false>), dim3(nx, nsplit, njobs), dim3(256), lds, st, g, jobs);
    DAM_CHECK_LAUNCH();
    for (int i = 0; i < njobs; ++i) {
        const int rc = reduce_submit(queue, jobs.partial[i], dw[i], g.nsplit, nx, TNB, TKB, TA, TB, n_real[i], k_real[i], g.KH, g.KW,
                                     g.tap_groups, g.tiles_k, st);
        if (rc != DAM_OK) return rc;
    }
    return DAM_OK;
}

// The recorded tile launches of a queue, as one launch (and nothing pending afterwards, whatever the outcome).
template <int TNB, int TKB, int TA, int TB>
int launch_pending_tile(WgradQueue* q, hipStream_t st) {
    PendingTile& p = q->pend;
    const int n = p.njobs;
    p.njobs = 0;
    if (n <= 0) return DAM_OK;
    int64_t ws = p.ws_floats[0];
    for (int i = 1; i < n; ++i) ws = p.ws_floats[i] < ws ? p.ws_floats[i] : ws;
    return launch_wgrad_jobs<TNB, TKB, TA, TB>(p.g, p.jobs, n, ws, p.dw, p.n_real, p.k_real, q, st);
}

template <int TNB, int TKB, int TA, int TB>
int launch_wgrad(WgradGeo& g, const float* X, const float* dY, const float* sc, const float* sh, float* partial,
                 int64_t ws_floats, float* dw, int n_real, int k_real, void* queue, hipStream_t st) {
    g.tiles_n = (int)cdiv(g.nblk, TNB);
    g.tiles_k = (int)cdiv(g.nchunks, TKB);
    g.tap_groups = (int)cdiv(g.KH, TA);
    g.nsplit = 0;
    const int relu_in = g.relu_in;
    g.relu_in = 0;                           // travels per job, not in the shared geometry
    WgradQueue* q = static_cast<WgradQueue*>(queue);
    if (q && q->magic != WGRAD_QUEUE_MAGIC) return DAM_ERR_BAD_ARG;
    if (q && q->batching) {
        // record instead of launching: jobs of the same instantiation and geometry wait for each other until the queue is
        // flushed (or another geometry arrives); the caller keeps X, dY, the affine and the slabs valid until then
        PendingTile& p = q->pend;
        constexpr int sig = TNB | TKB << 4 | TA << 8 | TB << 12;
        if (p.njobs > 0 && (p.sig != sig || memcmp(&p.g, &g, sizeof(WgradGeo)) != 0 || p.njobs == WG_MAX_JOBS)) {
            const int rc = p.launch(q, st);
            if (rc != DAM_OK) return rc;
        }
        const int i = p.njobs++;
        p.sig = sig; p.g = g; p.launch = &launch_pending_tile<TNB, TKB, TA, TB>;
        p.jobs.X[i] = X; p.jobs.dY[i] = dY; p.jobs.sc[i] = sc; p.jobs.sh[i] = sh; p.jobs.partial[i] = partial;
        p.jobs.relu_in[i] = relu_in;
        p.dw[i] = dw; p.ws_floats[i] = ws_floats; p.n_real[i] = n_real; p.k_real[i] = k_real;
        return DAM_OK;
    }
    WgradJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    jobs.X[0] = X; jobs.dY[0] = dY; jobs.sc[0] = sc; jobs.sh[0] = sh; jobs.partial[0] = partial; jobs.relu_in[0] = relu_in;
    return launch_wgrad_jobs<TNB, TKB, TA, TB>(g, jobs, 1, ws_floats, &dw, &n_real, &k_real, queue, st);
}

}  // namespace
}  // namespace dam

extern "C" int64_t dam_wgrad_queue_bytes(void) { return (int64_t)sizeof(dam::WgradQueue); }

extern "C" int dam_wgrad_queue_init(void* queue) {
    if (!queue) return DAM_ERR_BAD_ARG;
    dam::WgradQueue* q = static_cast<dam::WgradQueue*>(queue);
    const int batching = q->magic == dam::WGRAD_QUEUE_MAGIC ? q->batching : 0;      // re-initialising keeps the mode
    memset(q, 0, sizeof(*q));
    q->magic = dam::WGRAD_QUEUE_MAGIC;
    q->batching = batching;
    return DAM_OK;
}

extern "C" int dam_wgrad_queue_set_batching(void* queue, int on) {
    dam::WgradQueue* q = static_cast<dam::WgradQueue*>(queue);
    if (!q || q->magic != dam::WGRAD_QUEUE_MAGIC) return DAM_ERR_BAD_ARG;
    if (q->pend.njobs > 0 || q->pend_direct.jobs.njobs > 0) return DAM_ERR_BAD_ARG;          // flush first
    q->batching = on ? 1 : 0;
    return DAM_OK;
}

extern "C" int dam_wgrad_queue_pending(const void* queue) {
    const dam::WgradQueue* q = static_cast<const dam::WgradQueue*>(queue);
    return q && q->magic == dam::WGRAD_QUEUE_MAGIC ? q->batch.njobs + q->pend.njobs + q->pend_direct.jobs.njobs : -1;
}

extern "C" int dam_wgrad_queue_flush(void* queue, void* stream) {
    dam::WgradQueue* q = static_cast<dam::WgradQueue*>(queue);
    if (!q || q->magic != dam::WGRAD_QUEUE_MAGIC) return DAM_ERR_BAD_ARG;
    if (q->pend.njobs > 0) {
        const int rc = q->pend.launch(q, (hipStream_t)stream);
        if (rc != DAM_OK) return rc;
    }
    if (q->pend_direct.jobs.njobs > 0) {
        const int rc = q->pend_direct.launch(q, (hipStream_t)stream);
        if (rc != DAM_OK) return rc;
    }
    return dam::reduce_flush(q->batch, (hipStream_t)stream);
}

extern "C" int64_t dam_conv2d_wgrad_workspace_floats(int n_out, int c_in, int kh, int kw) {
    if (n_out <= 0 || c_in <= 0 || kh <= 0 || kw <= 0) return 0;
    // enough for ~512 workgroup slabs of the largest tile, and at least 4 splits of the whole gradient
    const int64_t whole = dam::cdiv(n_out, 32) * dam::cdiv(c_in, 32) * 4 * 256 * (int64_t)kh * kw;
    const int64_t a = 512LL * 36 * 256 + 4 * whole;
    return a;
}

extern "C" int dam_conv2d_wgrad_f32(const float* x, int B, int H, int W, int C, int in_nchw, const float* in_scale,
                                    const float* in_shift, int relu_in, const float* dy, int Ho, int Wo, int n_chan,
                                    int n_out, int kh, int kw, int stride, int pad, int dil, float* dw, int c_real,
                                    float* workspace, int64_t workspace_floats, void* reduce_queue, void* stream) {
    using namespace dam;
    if (!x || !dy || !dw || !workspace || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return DAM_ERR_BAD_ARG;
    if (c_real < 0 || c_real > C) return DAM_ERR_BAD_ARG;
    const int k_real = c_real ? c_real : C;         // dw rows kept: the real input channels (the stem's zero-padded planes drop out)
    if (n_chan % 16 || n_out > n_chan || (stride != 1 && stride != 2)) return DAM_ERR_UNSUPPORTED;
    if (in_nchw ? C > 16 : C % 16) return DAM_ERR_UNSUPPORTED;
    if (in_scale && !in_shift) return DAM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1 && !in_nchw && Ho == H && Wo == W && C == n_chan) {
        // row-streaming kernel for the shapes of the ResNet stages; anything else takes the tile kernel below
        int rc = DAM_ERR_UNSUPPORTED;
#define DAM_WGR(...) launch_wgrad_rows<__VA_ARGS__>(B, H, W, C, x, dy, in_scale, in_shift, relu_in, workspace, workspace_floats, dw, n_out, k_real, reduce_queue, st)
        // <TN, TK, steps, planes per loader wave, pieces per plane, row parts>: the shapes of the ResNet stages at 130 frames
        // (3 s clips) and at the reference's native 216 frames
        // (the last parameter, loader waves: four measured 53.7 -> 48.5 us on the 129x17 stage, where two workgroups then share a
        // CU; no difference on the three wide stages -- 56.6 / 57.5 / 59.5 against 57.0 / 57.0 / 59.2 us -- and 31.8 -> 32.7 on 65x9)
        if (C == 16) { rc = DAM_WGR(1, 1, 33, 1, 9); if (rc == DAM_ERR_UNSUPPORTED) rc = DAM_WGR(1, 1, 28, 1, 14, 2); }
        else if (W > 80) rc = DAM_WGR(2, 1, 14, 1, 7, 2);
        else if (W > 48) { rc = DAM_WGR(2, 1, 17, 2, 5); if (rc == DAM_ERR_UNSUPPORTED) rc = DAM_WGR(2, 1, 14, 2, 4); }
        else if (W > 20) rc = DAM_WGR(2, 1, 9, 2, 3);
        else if (W > 12) rc = getenv("DAM_WGR_NL8") ? DAM_WGR(2, 1, 5, 2, 2) : DAM_WGR(2, 1, 5, 3, 2, 1, 4);   // 17-pixel rows (129x17 stage), four loader waves
        else if (W > 6) rc = DAM_WGR(2, 1, 3, 2, 1);        // 9-pixel rows (65x9 stage): 40.8 -> 32.1 us; narrower: tile kernel
#undef DAM_WGR
        if (rc != DAM_ERR_UNSUPPORTED) return rc;
    }
#define DAM_WGD(TN_, TK_, KH_, KW_) \
    launch_wgrad_direct<TN_, TK_, KH_, KW_>(B, H, W, C, Ho, Wo, n_chan, stride, pad, dil, x, dy, workspace, workspace_floats, dw, n_out, k_real, reduce_queue, st)
    if (kh == 1 && kw == 1 && pad == 0 && !in_nchw && !in_scale) {
        int rc = DAM_WGD(2, 2, 1, 1);
        if (rc == DAM_ERR_UNSUPPORTED) rc = DAM_WGD(2, 1, 1, 1);
        if (rc == DAM_ERR_UNSUPPORTED) rc = DAM_WGD(1, 1, 1, 1);
        if (rc != DAM_ERR_UNSUPPORTED) return rc;
    }
    // strided 3x3 (the first convolution of a down-sampling block): measured 61/62/58 us against 80/78/67 us for the tile
    // kernel; for stride-1 narrow rows (17 pixels, 96 channels) it is slower (117 vs 60 us: nine L2 reads per element)
    if (kh == 3 && kw == 3 && !in_nchw && !in_scale && stride == 2 && pad == 1 && dil == 1 && !getenv("DAM_WGR_S2_DIRECT")) {
        // <TN, steps, planes per loader wave, pieces per X plane, per dY plane, row parts>: the down-sampling convolutions of
        // the ResNet stages at 130 frames (65-, 33- and 17-pixel output rows) and at the reference's native 216 (54, 27)
#define DAM_WGR2(...) launch_wgrad_rows_s2<__VA_ARGS__>(B, H, W, C, Ho, Wo, n_chan, x, dy, workspace, workspace_floats, dw, n_out, k_real, reduce_queue, st)
        int rc = DAM_ERR_UNSUPPORTED;
        if (Wo > 56) rc = DAM_WGR2(2, 9, 1, 9, 5, 2);
        else if (Wo > 34) rc = DAM_WGR2(2, 7, 1, 7, 4, 2);
        else if (Wo > 28) rc = DAM_WGR2(2, 9, 2, 5, 3, 1);
        else if (Wo > 18) rc = DAM_WGR2(2, 7, 2, 4, 2, 1);
        else if (Wo > 14) rc = DAM_WGR2(2, 5, 2, 3, 2, 1);
        // narrower rows stay on the tile kernel: slots of 3 / 2 MFMA steps are all barrier (measured 38.5 vs 38.9 us on the
        // 65x9 stage, 31.6 vs 29.5 us on 33x5)
#undef DAM_WGR2
        if (rc != DAM_ERR_UNSUPPORTED) return rc;
    }
    if (kh == 3 && kw == 3 && !in_nchw && !in_scale && Wo >= 16 && stride == 2) {
        int rc = DAM_WGD(2, 1, 3, 3);
        if (rc == DAM_ERR_UNSUPPORTED) rc = DAM_WGD(1, 1, 3, 3);
        if (rc != DAM_ERR_UNSUPPORTED) return rc;
    }
#undef DAM_WGD
    WgradGeo g;
    memset(&g, 0, sizeof(g));               // (compared bytewise when launches of one geometry are batched)
    g.B = B; g.H = H; g.W = W; g.C = C; g.Ho = Ho; g.Wo = Wo; g.N = n_chan; g.s = stride; g.KH = kh; g.KW = kw;
    g.off_h = -pad; g.step_h = dil; g.off_w = -pad; g.step_w = dil;
    g.r0 = -pad; g.c0 = -pad;
    g.PWin = (Wo - 1) * stride + (kw - 1) * dil + 1;
    g.PWs = (int)cdiv(g.PWin, stride);
    g.PWT = g.PWs * stride;
    const int ta_rows = (kh == 3 && kw == 3) ? 3 : 1;               // TA of the instantiations below
    auto set_tile = [&](int tmw_max) {
        // pixels per tile: the image in equal parts of at most tmw_max pixels, rounded up to the 16 a step of the four waves takes
        // (165 pixels of the 33 x 5 stage in a 256-pixel tile were 16 MFMA steps per wave for 10.3 of work: 176 -> 11)
        const int64_t npix_ = (int64_t)Ho * Wo, parts_ = cdiv(npix_, tmw_max);
        const int tmw = getenv("DAM_WG_FIXED_TILE") ? tmw_max : (int)(cdiv(cdiv(npix_, parts_), 16) * 16);
        int rows_out = (tmw + Wo - 2) / Wo + 1;
        if (rows_out > Ho) rows_out = Ho;
        g.TMW = tmw;
        g.PR = (rows_out - 1) * stride + (ta_rows - 1) * dil + 1;        // rows one tap group (TA kernel rows) touches
        g.tiles_m = (int)cdiv((int64_t)Ho * Wo, tmw);
        g.total_tiles = g.tiles_m * B;
    };
    auto lds_bytes = [&](int tnb, int tkb) { return (size_t)tkb * g.PR * g.PWT * 64 + (size_t)tnb * g.TMW * 64; };
    g.nchunks = in_nchw ? 1 : C / 16;
    g.nblk = n_chan / 16;
    g.in_nchw = in_nchw; g.relu_in = relu_in;
    // tile choice: 2x2 channel blocks when both sides have them and LDS allows two workgroups per CU.  48 OUTPUT channels (three blocks:
    // the scalar models' 5x5 / 32 -> 48 layer) take a three-block tile on that side instead of two tiles of two with the fourth block
    // empty (a quarter of the MFMAs on zeros): 167 -> 115 us; DAM_WG_NO_T3 keeps 2x2 (A/B).  The same on the INPUT side (7x7 / 48 -> 64
    // as a 1x3 tile) measured 476 us against 463: 22 LDS reads per 21 MFMAs; a 2x3 tile needs 273 VGPRs.  Not instantiated.
    static const bool no_t3 = getenv("DAM_WG_NO_T3") != nullptr;
    int tnb = 2, tkb = 2;
    if (!no_t3 && kh == 5 && kw == 5 && g.nblk == 3 && g.nchunks % 2 == 0) { tnb = 3; tkb = 2; }
    bool small = g.nchunks == 1 || g.nblk == 1;
    set_tile(256);
    if (!small && lds_bytes(tnb, tkb) > 72 * 1024) {
        set_tile(128);
        if (lds_bytes(tnb, tkb) > 72 * 1024) {
            if (tnb == 2 && tkb == 2) small = true;
            else { tnb = tkb = 2; set_tile(256); if (lds_bytes(2, 2) > 72 * 1024) { set_tile(128); if (lds_bytes(2, 2) > 72 * 1024) small = true; } }
        }
    }
    if (small) {
        set_tile(256);
        if (lds_bytes(1, 1) > 72 * 1024) set_tile(128);
    }
#define DAM_WG(TN_, TK_, TA_, TB_) \
    return launch_wgrad<TN_, TK_, TA_, TB_>(g, x, dy, in_scale, in_shift, workspace, workspace_floats, dw, n_out, k_real, reduce_queue, st)
    if (kw == 3 && kh == 3) { if (small) DAM_WG(1, 1, 3, 3); else DAM_WG(2, 2, 3, 3); }
    if (kw == 1 && kh == 1) { if (small) DAM_WG(1, 1, 1, 1); else DAM_WG(2, 2, 1, 1); }
    if (kw == 5) { if (small) DAM_WG(1, 1, 1, 5); else if (tnb == 3) DAM_WG(3, 2, 1, 5); else DAM_WG(2, 2, 1, 5); }
    if (kw == 7) { if (small) DAM_WG(1, 1, 1, 7); else DAM_WG(2, 2, 1, 7); }
    if (kw == 9) { if (small) DAM_WG(1, 1, 1, 9); else DAM_WG(2, 2, 1, 9); }
#undef DAM_WG
    return DAM_ERR_UNSUPPORTED;
}
