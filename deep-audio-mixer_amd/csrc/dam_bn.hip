// dam_bn.hip -- BatchNorm2d (training and eval mode) forward / backward on NHWC float32, HBM-bound kernels.
//
// Replaces nn.BatchNorm2d + F.relu (+ residual add) of models/model_resnet.py:12-27,65,97 (eps 1e-5,
// momentum 0.1) and models/model_scalar_1s.py:174-186 / model_scalar_2s.py:32-44 (eps 1e-3, momentum 0.9),
// including the running-statistics update and the autograd backward reached from model_trainer.py:36.
//
//   * statistics: every workgroup streams a contiguous pixel range with 16-byte loads (a thread owns 4
//     channels), keeps SHIFTED sums (x - K, K = first value seen) so that mean^2 >> var cannot cancel,
//     converts to (n, mean, M2) and the partials are merged with Chan's formula (in double in the finalize
//     kernel): deterministic, no atomics, matches a two-pass CPU BatchNorm;
//   * finalize also updates running_mean / running_var (unbiased) / num_batches_tracked exactly like
//     torch, and emits scale = gamma*invstd, shift = beta - mean*scale for the fused consumers
//     (bn_apply below, or the convolution kernels' load prologue);
//   * apply: y = relu?(x*scale + shift (+ r  or  + r*rscale + rshift))  -- the residual variant covers
//     both BasicBlock shortcuts (identity, or the shortcut conv's own BatchNorm folded in);
//   * backward: one reduction pass (sum dz, sum dz*xhat; dz = dy * (y > 0)) and one apply pass
//     dx = c1*dz + c2*x + c3 with per-channel constants.
#include <cstdlib>
#include "dam_common.h"
#include "dam_bn_fin.h"

namespace dam {
namespace {

constexpr int BN_MAX_PARTS = 1024;
static_assert(BN_MAX_PARTS == BN_BWD_RECORDS_MAX, "records a data-gradient epilogue may leave");

struct BnLaunch { int threads, q, r, parts; int64_t ppb; };

inline BnLaunch bn_plan(int64_t P, int C, int max_parts = BN_MAX_PARTS) {
    BnLaunch l;
    l.q = C / 4;
    l.r = 256 / l.q;
    if (l.r < 1) l.r = 1;
    l.threads = l.q * l.r;
    int64_t parts = cdiv(P, (int64_t)l.r * 16);          // ~16 pixels (two batches of 8 loads) per thread
    if (parts > max_parts) {
        // a consumer that merges the records itself wants few of them (fa_max_parts): the workgroups grow instead, so that the
        // pass still has >= 2048 waves to cover the memory latency (170 workgroups of 4 waves ran a 34 MB pair pass 8 us slower)
        parts = max_parts;
        while (l.threads < 1024 && parts * l.threads < 131072 && (size_t)(l.r * 2) * C * 3 * sizeof(float) <= 48 * 1024) {
            l.r *= 2;
            l.threads *= 2;
        }
    }
    if (parts < 1) parts = 1;
    l.ppb = cdiv(P, parts);
    l.parts = (int)cdiv(P, l.ppb);
    return l;
}

// gridDim.y == 2: a second tensor of the same shape (x2 -> partial2) in the same launch (dam_bn_stats_pair_f32).
__global__ void bn_stats_partial_kernel(const float* __restrict__ x, int64_t P, int C, int Q, int R, int64_t ppb,
                                        float* __restrict__ partial /* [parts][C][3] */, const BnFinArgs fin,
                                        const float* __restrict__ x2, float* __restrict__ partial2) {
    extern __shared__ __attribute__((aligned(16))) float sm[];    // [R][C][3]
    if (blockIdx.y) { x = x2; partial = partial2; }
    const int cq = threadIdx.x % Q, pr = threadIdx.x / Q;
    const int64_t lo = blockIdx.x * ppb, hi = (lo + ppb < P) ? lo + ppb : P;
    float k[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    int n = 0;
    constexpr int U = 8;      // loads in flight per thread: the kernel is a pure stream, latency must be covered by MLP
    for (int64_t p = lo + pr; p < hi; p += (int64_t)R * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = p + (int64_t)u * R;
            v[u] = q < hi ? *reinterpret_cast<const float4*>(x + q * C + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (n == 0) { k[0] = v[0].x; k[1] = v[0].y; k[2] = v[0].z; k[3] = v[0].w; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + (int64_t)u * R < hi) {
                const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = e[i] - k[i];
                    s1[i] += d;
                    s2[i] = fmaf(d, d, s2[i]);
                }
                ++n;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* o = sm + ((size_t)pr * C + cq * 4 + i) * 3;
        const float md = n ? s1[i] / n : 0.f;
        o[0] = (float)n;
        o[1] = k[i] + md;
        o[2] = n ? fmaxf(s2[i] - s1[i] * md, 0.f) : 0.f;
    }
    __syncthreads();
    // tree merge over the R pixel rows (Chan), all threads active; fixed order -> deterministic
    int span = 1;
    while (span < R) span <<= 1;
    for (int stride = span >> 1; stride >= 1; stride >>= 1) {
        if (pr < stride && pr + stride < R) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* a = sm + ((size_t)pr * C + cq * 4 + i) * 3;
                const float* b = sm + ((size_t)(pr + stride) * C + cq * 4 + i) * 3;
                const float na = a[0], nb = b[0];
                if (nb != 0.f) {
                    const float nn = na + nb, d = b[1] - a[1];
                    a[1] += d * (nb / nn);
                    a[2] += b[2] + d * d * (na * nb / nn);
                    a[0] = nn;
                }
            }
        }
        __syncthreads();
    }
    if (pr == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cq * 4 + i;
            const float* a = sm + (size_t)c * 3;
            float* out = partial + ((size_t)blockIdx.x * C + c) * 3;
            store_sc1(out, a[0]); store_sc1(out + 1, a[1]); store_sc1(out + 2, a[2]);
        }
    }
    if (fin.counter) {      // the last workgroup to arrive merges all records (dam_bn_fin.h): no finalize launch
        __shared__ unsigned ticket;
        if (block_arrive_last(fin.counter, gridDim.x, &ticket))
            bn_stats_finalize_block(partial, (int)gridDim.x, C, fin, reinterpret_cast<double*>(sm), threadIdx.x, blockDim.x);
    }
}

// One wave per channel: lanes merge a strided subset of the partials (Chan), then a shuffle tree merges the lanes.
__global__ __launch_bounds__(64) void bn_stats_finalize_kernel(const float* __restrict__ partial, int parts, int C,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                         long long* __restrict__ num_batches, float momentum, float eps,
                                         float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                         float* __restrict__ scale, float* __restrict__ shift, const BnFinArgs second,
                                         const float* __restrict__ partial2) {
    if (blockIdx.y) {       // second BatchNorm of a pair launch
        partial = partial2; gamma = second.gamma; beta = second.beta; running_mean = second.running_mean;
        running_var = second.running_var; num_batches = second.num_batches; momentum = second.momentum; eps = second.eps;
        save_mean = second.save_mean; save_invstd = second.save_invstd; scale = second.scale; shift = second.shift;
    }
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c == 0 && lane == 0 && num_batches) *num_batches += 1;
    // the channel's parameters and running statistics are requested together with the records: read behind the reduction
    // (lane 0 only) they were a second memory round trip in a kernel that is nothing but round trips
    const float gam_c = gamma[c], bet_c = beta[c];
    const float rm_c = running_mean ? running_mean[c] : 0.f, rv_c = running_mean ? running_var[c] : 0.f;
    // Every lane requests ALL its records (<= 16: parts <= 1024) before it touches the first: the records come from other
    // XCDs' workgroups, each load is a memory-side round trip, and a load-merge-load chain made this 5 us kernel cost 5-7 us.
    // Merging is two plain wave reductions instead of a chain of Chan updates (no divide per record):
    //   N = sum n_i,  mu = sum n_i mean_i / N,  M2 = sum (m2_i + n_i (mean_i - mu)^2).
    constexpr int U = 16;
    float rn[U], rmn[U], rq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int p = lane + 64 * u;
        const float* o = partial + ((size_t)(p < parts ? p : 0) * C + c) * 3;
        rn[u] = o[0]; rmn[u] = o[1]; rq[u] = o[2];
        if (p >= parts) rn[u] = 0.f;
    }
    double na = 0, sa = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) { na += (double)rn[u]; sa += (double)rn[u] * (double)rmn[u]; }
    for (int p = lane + 64 * U; p < parts; p += 64) {        // more than 1024 records: not produced by this library
        const float* o = partial + ((size_t)p * C + c) * 3;
        na += (double)o[0]; sa += (double)o[0] * (double)o[1];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { na += __shfl_xor(na, off); sa += __shfl_xor(sa, off); }
    const double ma = sa / na;
    double qa = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) { const double d = (double)rmn[u] - ma; qa += (double)rq[u] * (rn[u] != 0.f ? 1.0 : 0.0) + (double)rn[u] * d * d; }
    for (int p = lane + 64 * U; p < parts; p += 64) {
        const float* o = partial + ((size_t)p * C + c) * 3;
        const double d = (double)o[1] - ma;
        qa += (double)o[2] + (double)o[0] * d * d;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) qa += __shfl_xor(qa, off);
    if (lane != 0) return;
    const double var = qa / na;
    const float mean = (float)ma;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    const float sc = gam_c * invstd;
    scale[c] = sc;
    shift[c] = bet_c - mean * sc;
    if (running_mean) {
        const double unbiased = na > 1 ? qa / (na - 1) : var;
        running_mean[c] = (1.f - momentum) * rm_c + momentum * mean;
        running_var[c] = (1.f - momentum) * rv_c + momentum * (float)unbiased;
    }
}

__global__ void bn_eval_affine_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float eps, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                      float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(running_var[c] + eps);
    const float sc = gamma[c] * invstd;
    save_mean[c] = running_mean[c];
    save_invstd[c] = invstd;
    scale[c] = sc;
    shift[c] = beta[c] - running_mean[c] * sc;
}

__global__ void bn_apply_kernel(const float* __restrict__ x, int64_t nquads, int Q, const float* __restrict__ scale,
                                const float* __restrict__ shift, const float* __restrict__ res,
                                const float* __restrict__ rscale, const float* __restrict__ rshift, int relu,
                                float* __restrict__ y, unsigned char* __restrict__ sign_bits) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nquads; e += (int64_t)gridDim.x * blockDim.x) {
        const int cq = (int)(e % Q);
        const float4 v = reinterpret_cast<const float4*>(x)[e];
        const float4 sc = reinterpret_cast<const float4*>(scale)[cq], sh = reinterpret_cast<const float4*>(shift)[cq];
        float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
        if (res) {
            float4 r = reinterpret_cast<const float4*>(res)[e];
            if (rscale) {
                const float4 a = reinterpret_cast<const float4*>(rscale)[cq], b = reinterpret_cast<const float4*>(rshift)[cq];
                r = make_float4(fmaf(r.x, a.x, b.x), fmaf(r.y, a.y, b.y), fmaf(r.z, a.z, b.z), fmaf(r.w, a.w, b.w));
            }
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        reinterpret_cast<float4*>(y)[e] = o;
        // one byte per channel quad: bit i = (y[4e + i] > 0) -- the ReLU mask the backward passes want, 1/16 of the bytes of y
        if (sign_bits) sign_bits[e] = (unsigned char)((o.x > 0.f) | ((o.y > 0.f) << 1) | ((o.z > 0.f) << 2) | ((o.w > 0.f) << 3));
    }
}

struct BnBwdFin {
    double count;
    const float* gamma; const float* mean; const float* invstd;
    int training;
    float* dgamma; float* dbeta; float* coef;      // coef [3][C]
    unsigned* counter;                             // null: separate finalize launch
};

// dgamma / dbeta and the three per-channel constants of the apply pass from the partial sums [parts][C][2].
// Thread layout as bn_stats_finalize_block: channel = t % W, slice = t / W; scratch >= nthreads * 2 doubles.
__device__ __forceinline__ void bn_bwd_finalize_block(const float* partial, int parts, int C, const BnBwdFin& f,
                                                      double* scratch, int tid, int nthreads) {
    const int W = C < nthreads ? C : nthreads, S = nthreads / W;
    for (int c0 = 0; c0 < C; c0 += nthreads) {
        const int c = c0 + tid % W, sl = tid / W;
        double s1 = 0, s2 = 0;
        if (c < C && sl < S) {
            constexpr int U = 8;            // eight records in flight per thread (see bn_stats_finalize_block)
            for (int p0 = sl; p0 < parts; p0 += S * U) {
                float ra[U], rb[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = p0 + u * S;
                    const float* o = partial + ((size_t)(p < parts ? p : sl) * C + c) * 2;
                    ra[u] = load_sc1(o); rb[u] = load_sc1(o + 1);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (p0 + u * S < parts) { s1 += (double)ra[u]; s2 += (double)rb[u]; }
            }
        }
        scratch[tid * 2] = s1; scratch[tid * 2 + 1] = s2;
        __syncthreads();
        if (sl == 0 && c < C) {
            for (int s = 1; s < S; ++s) { s1 += scratch[(size_t)(tid + s * W) * 2]; s2 += scratch[(size_t)(tid + s * W) * 2 + 1]; }
            f.dbeta[c] = (float)s1;
            f.dgamma[c] = (float)s2;
            const double g = (double)f.gamma[c] * f.invstd[c];
            double c2 = 0, c3 = 0;
            if (f.training) {
                c2 = -g * f.invstd[c] * s2 / f.count;
                c3 = -g * s1 / f.count - c2 * f.mean[c];
            }
            f.coef[c] = (float)g; f.coef[C + c] = (float)c2; f.coef[2 * C + c] = (float)c3;
        }
        __syncthreads();
    }
}

// partial[blk][c] = (sum dz, sum dz*xhat)
__device__ __forceinline__ float4 sign_quad(unsigned b) {       // sign byte -> (1 or 0) x 4
    return make_float4((float)(b & 1u), (float)((b >> 1) & 1u), (float)((b >> 2) & 1u), (float)((b >> 3) & 1u));
}

// MASK: 3 = from the sign bytes written by bn_apply (one byte per channel quad: y_mask then points at bytes);
// MASK: 0 none, 1 from y_mask (saved output), 2 recomputed as fma(x, mscale, mshift) > 0 -- the forward's own expression, so
// the bits agree and the saved activation is not read at all
template <int MASK>
__global__ void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ y_mask,
                                      const float* __restrict__ x, int64_t P, int C, int Q, int R, int64_t ppb,
                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                      const float* __restrict__ mscale, const float* __restrict__ mshift,
                                      float* __restrict__ partial /* [parts][C][2] */, const BnBwdFin fin) {
    extern __shared__ __attribute__((aligned(16))) float sm[];    // [R][C][2]
    const int cq = threadIdx.x % Q, pr = threadIdx.x / Q;
    const int64_t lo = blockIdx.x * ppb, hi = (lo + ppb < P) ? lo + ppb : P;
    const float4 mu = reinterpret_cast<const float4*>(mean)[cq], is = reinterpret_cast<const float4*>(invstd)[cq];
    float4 msc = make_float4(0.f, 0.f, 0.f, 0.f), msh = msc;
    if (MASK == 2) { msc = reinterpret_cast<const float4*>(mscale)[cq]; msh = reinterpret_cast<const float4*>(mshift)[cq]; }
    float a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    constexpr int U = 4;      // 3 streams x 4 loads in flight per thread
    for (int64_t p = lo + pr; p < hi; p += (int64_t)R * U) {
        float4 gv[U], mv[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = p + (int64_t)u * R;
            const bool ok = q < hi;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            gv[u] = ok ? *reinterpret_cast<const float4*>(dy + q * C + cq * 4) : z;
            xv[u] = ok ? *reinterpret_cast<const float4*>(x + q * C + cq * 4) : z;
            mv[u] = (ok && MASK == 1) ? *reinterpret_cast<const float4*>(y_mask + q * C + cq * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
            if (MASK == 3) mv[u] = sign_quad(ok ? reinterpret_cast<const unsigned char*>(y_mask)[q * Q + cq] : 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 g = gv[u];
            const float4 v = xv[u];
            float4 m = mv[u];
            if (MASK == 2) m = make_float4(fmaf(v.x, msc.x, msh.x), fmaf(v.y, msc.y, msh.y), fmaf(v.z, msc.z, msh.z), fmaf(v.w, msc.w, msh.w));
            g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f; g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
            a[0] += g.x; a[1] += g.y; a[2] += g.z; a[3] += g.w;
            b[0] = fmaf(g.x, (v.x - mu.x) * is.x, b[0]); b[1] = fmaf(g.y, (v.y - mu.y) * is.y, b[1]);
            b[2] = fmaf(g.z, (v.z - mu.z) * is.z, b[2]); b[3] = fmaf(g.w, (v.w - mu.w) * is.w, b[3]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* o = sm + ((size_t)pr * C + cq * 4 + i) * 2;
        o[0] = a[i]; o[1] = b[i];
    }
    __syncthreads();
    int span = 1;
    while (span < R) span <<= 1;
    for (int stride = span >> 1; stride >= 1; stride >>= 1) {
        if (pr < stride && pr + stride < R) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* a2 = sm + ((size_t)pr * C + cq * 4 + i) * 2;
                const float* b2 = sm + ((size_t)(pr + stride) * C + cq * 4 + i) * 2;
                a2[0] += b2[0]; a2[1] += b2[1];
            }
        }
        __syncthreads();
    }
    if (pr == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cq * 4 + i;
            store_sc1(partial + ((size_t)blockIdx.x * C + c) * 2, sm[(size_t)c * 2]);
            store_sc1(partial + ((size_t)blockIdx.x * C + c) * 2 + 1, sm[(size_t)c * 2 + 1]);
        }
    }
    if (fin.counter) {      // the last workgroup to arrive finalizes (dam_bn_fin.h)
        __shared__ unsigned ticket;
        if (block_arrive_last(fin.counter, gridDim.x, &ticket))
            bn_bwd_finalize_block(partial, (int)gridDim.x, C, fin, reinterpret_cast<double*>(sm), threadIdx.x, blockDim.x);
    }
}

__device__ __forceinline__ double wave_sum64_f64(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// One wave per channel.
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int parts, int C, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, int training, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ coef /* [3][C] */) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const float gam_c = gamma[c], inv_c = invstd[c], mean_c = mean[c];      // (requested with the records, see bn_stats_finalize_kernel)
    double s1 = 0, s2 = 0;
    {   // all loads of the lane in flight before the first add (see bn_stats_finalize_kernel)
        constexpr int U = 16;
        float ra[U], rb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = lane + 64 * u;
            const float* o = partial + ((size_t)(p < parts ? p : 0) * C + c) * 2;
            ra[u] = p < parts ? o[0] : 0.f; rb[u] = p < parts ? o[1] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { s1 += (double)ra[u]; s2 += (double)rb[u]; }
        for (int p = lane + 64 * U; p < parts; p += 64) { s1 += partial[((size_t)p * C + c) * 2]; s2 += partial[((size_t)p * C + c) * 2 + 1]; }
    }
    s1 = wave_sum64_f64(s1);
    s2 = wave_sum64_f64(s2);
    if (lane != 0) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    const double g = (double)gam_c * inv_c;
    double c2 = 0, c3 = 0;
    if (training) {
        c2 = -g * inv_c * s2 / count;
        c3 = -g * s1 / count - c2 * mean_c;
    }
    coef[c] = (float)g; coef[C + c] = (float)c2; coef[2 * C + c] = (float)c3;
}

template <int MASK>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y_mask,
                                    const float* __restrict__ x, int64_t nquads, int Q, int C,
                                    const float* __restrict__ coef, const float* __restrict__ mscale,
                                    const float* __restrict__ mshift, float* __restrict__ dx) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nquads; e += (int64_t)gridDim.x * blockDim.x) {
        const int cq = (int)(e % Q);
        float4 g = reinterpret_cast<const float4*>(dy)[e];
        const float4 v = reinterpret_cast<const float4*>(x)[e];
        if (MASK != 0) {
            float4 m;
            if (MASK == 1) {
                m = reinterpret_cast<const float4*>(y_mask)[e];
            } else if (MASK == 3) {
                m = sign_quad(reinterpret_cast<const unsigned char*>(y_mask)[e]);
            } else {
                const float4 sc = reinterpret_cast<const float4*>(mscale)[cq], sh = reinterpret_cast<const float4*>(mshift)[cq];
                m = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
            }
            g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f; g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
        }
        const float4 c1 = reinterpret_cast<const float4*>(coef)[cq], c2 = reinterpret_cast<const float4*>(coef + C)[cq],
                     c3 = reinterpret_cast<const float4*>(coef + 2 * C)[cq];
        float4 o;
        o.x = fmaf(c1.x, g.x, fmaf(c2.x, v.x, c3.x)); o.y = fmaf(c1.y, g.y, fmaf(c2.y, v.y, c3.y));
        o.z = fmaf(c1.z, g.z, fmaf(c2.z, v.z, c3.z)); o.w = fmaf(c1.w, g.w, fmaf(c2.w, v.w, c3.w));
        reinterpret_cast<float4*>(dx)[e] = o;
    }
}

// ---- two BatchNorms that share dy and the ReLU mask (a residual block's bn2 and its shortcut BatchNorm, both fed by the
// block's output gradient): one partial / finalize / apply launch for both -- dy and the mask are read once per pass
// instead of twice, three launches instead of six.  Same sums in the same order as the single form: bitwise the same result.
template <bool BITS>        // y_mask points at the sign bytes of bn_apply instead of the float output
__global__ void bn_bwd_partial_pair_kernel(const float* __restrict__ dy, const float* __restrict__ y_mask,
                                           const float* __restrict__ xa, const float* __restrict__ xb, int64_t P, int C, int Q,
                                           int R, int64_t ppb, const float* __restrict__ mean_a, const float* __restrict__ invstd_a,
                                           const float* __restrict__ mean_b, const float* __restrict__ invstd_b,
                                           float* __restrict__ partial /* [parts][C][3]: sum dz, sum dz*xhat_a, sum dz*xhat_b */) {
    extern __shared__ __attribute__((aligned(16))) float sm[];    // [R][C][3]
    const int cq = threadIdx.x % Q, pr = threadIdx.x / Q;
    const int64_t lo = blockIdx.x * ppb, hi = (lo + ppb < P) ? lo + ppb : P;
    const float4 mua = reinterpret_cast<const float4*>(mean_a)[cq], isa = reinterpret_cast<const float4*>(invstd_a)[cq];
    const float4 mub = reinterpret_cast<const float4*>(mean_b)[cq], isb = reinterpret_cast<const float4*>(invstd_b)[cq];
    float a[4] = {0, 0, 0, 0}, ba[4] = {0, 0, 0, 0}, bb[4] = {0, 0, 0, 0};
    constexpr int U = 4;
    for (int64_t p = lo + pr; p < hi; p += (int64_t)R * U) {
        float4 gv[U], mv[U], va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = p + (int64_t)u * R;
            const bool ok = q < hi;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            gv[u] = ok ? *reinterpret_cast<const float4*>(dy + q * C + cq * 4) : z;
            va[u] = ok ? *reinterpret_cast<const float4*>(xa + q * C + cq * 4) : z;
            vb[u] = ok ? *reinterpret_cast<const float4*>(xb + q * C + cq * 4) : z;
            if (BITS) mv[u] = sign_quad(ok ? reinterpret_cast<const unsigned char*>(y_mask)[q * Q + cq] : 0);
            else mv[u] = ok ? *reinterpret_cast<const float4*>(y_mask + q * C + cq * 4) : z;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 g = gv[u];
            const float4 m = mv[u], v = va[u], w = vb[u];
            g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f; g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
            a[0] += g.x; a[1] += g.y; a[2] += g.z; a[3] += g.w;
            ba[0] = fmaf(g.x, (v.x - mua.x) * isa.x, ba[0]); ba[1] = fmaf(g.y, (v.y - mua.y) * isa.y, ba[1]);
            ba[2] = fmaf(g.z, (v.z - mua.z) * isa.z, ba[2]); ba[3] = fmaf(g.w, (v.w - mua.w) * isa.w, ba[3]);
            bb[0] = fmaf(g.x, (w.x - mub.x) * isb.x, bb[0]); bb[1] = fmaf(g.y, (w.y - mub.y) * isb.y, bb[1]);
            bb[2] = fmaf(g.z, (w.z - mub.z) * isb.z, bb[2]); bb[3] = fmaf(g.w, (w.w - mub.w) * isb.w, bb[3]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* o = sm + ((size_t)pr * C + cq * 4 + i) * 3;
        o[0] = a[i]; o[1] = ba[i]; o[2] = bb[i];
    }
    __syncthreads();
    int span = 1;
    while (span < R) span <<= 1;
    for (int stride = span >> 1; stride >= 1; stride >>= 1) {
        if (pr < stride && pr + stride < R) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* a2 = sm + ((size_t)pr * C + cq * 4 + i) * 3;
                const float* b2 = sm + ((size_t)(pr + stride) * C + cq * 4 + i) * 3;
                a2[0] += b2[0]; a2[1] += b2[1]; a2[2] += b2[2];
            }
        }
        __syncthreads();
    }
    if (pr == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cq * 4 + i;
            float* o = partial + ((size_t)blockIdx.x * C + c) * 3;
            o[0] = sm[(size_t)c * 3]; o[1] = sm[(size_t)c * 3 + 1]; o[2] = sm[(size_t)c * 3 + 2];
        }
    }
}

// One wave per channel; coef [2][3][C].
__global__ __launch_bounds__(64) void bn_bwd_finalize_pair_kernel(const float* __restrict__ partial, int parts, int C, double count,
                                                                  const float* __restrict__ gamma_a, const float* __restrict__ mean_a,
                                                                  const float* __restrict__ invstd_a, const float* __restrict__ gamma_b,
                                                                  const float* __restrict__ mean_b, const float* __restrict__ invstd_b,
                                                                  int training, float* __restrict__ dgamma_a, float* __restrict__ dbeta_a,
                                                                  float* __restrict__ dgamma_b, float* __restrict__ dbeta_b,
                                                                  float* __restrict__ coef) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const float gam_a = gamma_a[c], inv_a = invstd_a[c], mu_a = mean_a[c];   // (requested with the records)
    const float gam_b = gamma_b[c], inv_b = invstd_b[c], mu_b = mean_b[c];
    double s1 = 0, s2 = 0, s3 = 0;
    {   // all loads of the lane in flight before the first add (see bn_stats_finalize_kernel)
        constexpr int U = 16;
        float ra[U], rb[U], rc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = lane + 64 * u;
            const float* o = partial + ((size_t)(p < parts ? p : 0) * C + c) * 3;
            ra[u] = p < parts ? o[0] : 0.f; rb[u] = p < parts ? o[1] : 0.f; rc[u] = p < parts ? o[2] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { s1 += (double)ra[u]; s2 += (double)rb[u]; s3 += (double)rc[u]; }
        for (int p = lane + 64 * U; p < parts; p += 64) {
            const float* o = partial + ((size_t)p * C + c) * 3;
            s1 += o[0]; s2 += o[1]; s3 += o[2];
        }
    }
    s1 = wave_sum64_f64(s1);
    s2 = wave_sum64_f64(s2);
    s3 = wave_sum64_f64(s3);
    if (lane != 0) return;
    dbeta_a[c] = (float)s1; dbeta_b[c] = (float)s1;
    dgamma_a[c] = (float)s2; dgamma_b[c] = (float)s3;
    const double ga = (double)gam_a * inv_a, gb = (double)gam_b * inv_b;
    double a2 = 0, a3 = 0, b2 = 0, b3 = 0;
    if (training) {
        a2 = -ga * inv_a * s2 / count; a3 = -ga * s1 / count - a2 * mu_a;
        b2 = -gb * inv_b * s3 / count; b3 = -gb * s1 / count - b2 * mu_b;
    }
    coef[c] = (float)ga; coef[C + c] = (float)a2; coef[2 * C + c] = (float)a3;
    coef[3 * C + c] = (float)gb; coef[4 * C + c] = (float)b2; coef[5 * C + c] = (float)b3;
}

template <bool BITS>
__global__ void bn_bwd_apply_pair_kernel(const float* __restrict__ dy, const float* __restrict__ y_mask,
                                         const float* __restrict__ xa, const float* __restrict__ xb, int64_t nquads, int Q, int C,
                                         const float* __restrict__ coef, float* __restrict__ dxa, float* __restrict__ dxb) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nquads; e += (int64_t)gridDim.x * blockDim.x) {
        const int cq = (int)(e % Q);
        float4 g = reinterpret_cast<const float4*>(dy)[e];
        const float4 m = BITS ? sign_quad(reinterpret_cast<const unsigned char*>(y_mask)[e]) : reinterpret_cast<const float4*>(y_mask)[e];
        const float4 v = reinterpret_cast<const float4*>(xa)[e], w = reinterpret_cast<const float4*>(xb)[e];
        g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f; g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
        const float4 a1 = reinterpret_cast<const float4*>(coef)[cq], a2 = reinterpret_cast<const float4*>(coef + C)[cq],
                     a3 = reinterpret_cast<const float4*>(coef + 2 * C)[cq];
        const float4 b1 = reinterpret_cast<const float4*>(coef + 3 * C)[cq], b2 = reinterpret_cast<const float4*>(coef + 4 * C)[cq],
                     b3 = reinterpret_cast<const float4*>(coef + 5 * C)[cq];
        float4 o, r;
        o.x = fmaf(a1.x, g.x, fmaf(a2.x, v.x, a3.x)); o.y = fmaf(a1.y, g.y, fmaf(a2.y, v.y, a3.y));
        o.z = fmaf(a1.z, g.z, fmaf(a2.z, v.z, a3.z)); o.w = fmaf(a1.w, g.w, fmaf(a2.w, v.w, a3.w));
        r.x = fmaf(b1.x, g.x, fmaf(b2.x, w.x, b3.x)); r.y = fmaf(b1.y, g.y, fmaf(b2.y, w.y, b3.y));
        r.z = fmaf(b1.z, g.z, fmaf(b2.z, w.z, b3.z)); r.w = fmaf(b1.w, g.w, fmaf(b2.w, w.w, b3.w));
        reinterpret_cast<float4*>(dxa)[e] = o;
        reinterpret_cast<float4*>(dxb)[e] = r;
    }
}

// out[c] = sum over pixels of x[p][c] (conv bias gradient); reuses the bwd partial layout with one column.
__global__ void channel_sum_partial_kernel(const float* __restrict__ x, int64_t P, int C, int Q, int R, int64_t ppb,
                                           float* __restrict__ partial) {
    extern __shared__ float sm[];
    const int cq = threadIdx.x % Q, pr = threadIdx.x / Q;
    const int64_t lo = blockIdx.x * ppb, hi = (lo + ppb < P) ? lo + ppb : P;
    float a[4] = {0, 0, 0, 0};
    constexpr int U = 8;
    for (int64_t p = lo + pr; p < hi; p += (int64_t)R * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = p + (int64_t)u * R;
            v[u] = q < hi ? *reinterpret_cast<const float4*>(x + q * C + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { a[0] += v[u].x; a[1] += v[u].y; a[2] += v[u].z; a[3] += v[u].w; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[(size_t)pr * C + cq * 4 + i] = a[i];
    __syncthreads();
    int span = 1;
    while (span < R) span <<= 1;
    for (int stride = span >> 1; stride >= 1; stride >>= 1) {
        if (pr < stride && pr + stride < R) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sm[(size_t)pr * C + cq * 4 + i] += sm[(size_t)(pr + stride) * C + cq * 4 + i];
        }
        __syncthreads();
    }
    if (pr == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) partial[(size_t)blockIdx.x * C + cq * 4 + i] = sm[cq * 4 + i];
    }
}
__global__ __launch_bounds__(64) void channel_sum_finalize_kernel(const float* __restrict__ partial, int parts, int C, int n_real,
                                            float* __restrict__ out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0;
    for (int p = lane; p < parts; p += 64) s += partial[(size_t)p * C + c];
    s = wave_sum64_f64(s);
    if (lane == 0) out[c] = (float)s;
}

// ================================================================================================================================
// Finalize INSIDE the elementwise consumer (round 4).  A finalize launch is 4.6-5 us whatever it does and a ResNet18 step had fifty of
// them; round 3's attempt to merge the records in every workgroup of the 4096-workgroup apply launches lost to the table traffic
// (every workgroup re-read every channel's records: ~6 M cache-line requests per launch).  What makes it pay:
//   * the grid is what the chip holds (<= 512 workgroups), each workgroup walks a contiguous pixel range, and
//   * a workgroup only merges the channels it applies: workgroup (slice, range) owns CS = 16 or 32 channels, so its table is
//     parts x CS records (<= ~100 KB, usually 20-60), requested in one burst behind the workgroup's first data loads;
//   * the merge is one pass in double around a pivot (record 0's mean): N = sum n, S = sum n (mean - m0),
//     Q = sum (M2 + n (mean - m0)^2); mean = m0 + S / N, M2 = Q - S^2 / N -- the shifted-data form of the pooled variance, exact to
//     double rounding (no Chan chain, no second pass over the records);
//   * range 0 of every slice writes what later kernels read (save_mean, save_invstd, scale, shift, the running statistics;
//     dgamma / dbeta in the backward form).
// The records' producers ran in earlier launches (the kernel boundary is the synchronisation), so plain loads are fine.
// ================================================================================================================================
#define DAM_Z4 make_float4(0.f, 0.f, 0.f, 0.f)
constexpr int FA_THREADS = 256;
constexpr int FA_U = 4;                 // pixel pieces per thread, stream and register set (two sets: the next batch's loads are
                                        // requested before the current batch is computed and stored)

struct FaPlan { int cs, nslices, nranges, q, ppi; int64_t ppr; };

// CS: the whole row for thin layers (one 128-byte line per pixel at 32 channels), 16-channel slices above that (64-byte pieces:
// the wide layers' tensors are small and L2 resident; what matters there is the table a workgroup has to merge)
inline int fa_cs(int C) { return C <= 32 ? C : 16; }
inline FaPlan fa_plan(int64_t P, int C) {
    FaPlan f;
    f.cs = fa_cs(C);
    f.nslices = C / f.cs;
    f.q = f.cs / 4;
    f.ppi = FA_THREADS / f.q;
    static const int wgs = [] { const char* e = getenv("DAM_BN_FA_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
    int64_t want = P / ((int64_t)f.ppi * FA_U * 2);       // >= two batches of loads per thread
    const int64_t cap = wgs / f.nslices > 0 ? wgs / f.nslices : 1;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    f.ppr = cdiv(P, want);
    f.nranges = (int)cdiv(P, f.ppr);
    return f;
}
// a slice table beyond this costs a workgroup more to read than the finalize launch it replaces (measured: 110 KB tables made the
// 129 x 17 stage's fused launch 5 us slower than finalize + apply, 74 KB ones broke even)
constexpr int FA_TABLE_BYTES_MAX = 80 * 1024;
inline bool fa_table_ok(int C, int parts, int rec_floats) { return (int64_t)parts * fa_cs(C) * rec_floats * 4 <= FA_TABLE_BYTES_MAX; }
// The FORWARD form pays more per record (12 bytes, a divide-free but longer merge, sqrt) and its launches on the full-resolution
// stages measured 2-3 us SLOWER than finalize + apply (48 KB tables x 512 workgroups = 24 MB of table reads in front of a 35 us
// stream), the small stages' about equal: it is taken only for tables up to DAM_BN_FUSED_FWD_KB (default 24) KB per workgroup --
// the backward forms win 1.5-2.5 us per launch on every stage and are taken up to FA_TABLE_BYTES_MAX.
inline bool fa_fwd_table_ok(int C, int parts) {
    static const int kb = [] { const char* e = getenv("DAM_BN_FUSED_FWD_KB"); const int v = e ? atoi(e) : -1; return v >= 0 ? v : 24; }();
    return fa_table_ok(C, parts, 3) && (int64_t)parts * fa_cs(C) * 12 <= (int64_t)kb * 1024;
}

// Sums NV per-record values over the slice's records: thread (c = tid % CS, i = tid / CS) takes records i, i + TPC, ...; every
// record of a thread is requested before the first is used (<= 16 per thread and round: one round trip for tables up to 256
// records of 16 channels); tid < CS ends up with the channel's totals in acc[].  `term(rec, v)`: one record (NR floats) -> NV addends.
template <int NR, int NV, typename F>
__device__ __forceinline__ void fa_slice_sums(const float* __restrict__ partial, int parts, int C, int ch, int CS, double* red,
                                              double (&acc)[NV], F term) {
    const int tid = threadIdx.x, TPC = FA_THREADS / CS, i0 = tid / CS;
    constexpr int RU = 16;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0;
    for (int p0 = i0; p0 < parts; p0 += TPC * RU) {
        float rec[RU][NR];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int p = p0 + u * TPC;
            const float* o = partial + ((size_t)(p < parts ? p : i0) * C + ch) * NR;
#pragma unroll
            for (int k = 0; k < NR; ++k) rec[u][k] = o[k];
        }
#pragma unroll
        for (int u = 0; u < RU; ++u)
            if (p0 + u * TPC < parts) {
                double v[NV];
                term(rec[u], v);
#pragma unroll
                for (int k = 0; k < NV; ++k) acc[k] += v[k];
            }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) red[tid * NV + k] = acc[k];
    __syncthreads();
    if (tid < CS)
        for (int i = 1; i < TPC; ++i)
#pragma unroll
            for (int k = 0; k < NV; ++k) acc[k] += red[(tid + i * CS) * NV + k];
}

// Two register sets alternate: LOAD(set, p0) requests a batch, EMIT(set, p0) computes and stores it.
#define DAM_FA_STREAM(LOAD_, EMIT_, STEP_)                                                                                     \
    for (int64_t p0 = lo + tp;;) {                                                                                            \
        if (p0 + (STEP_) < hi) { LOAD_(1, p0 + (STEP_)); }                                                                    \
        EMIT_(0, p0);                                                                                                         \
        p0 += (STEP_);                                                                                                        \
        if (p0 >= hi) break;                                                                                                  \
        if (p0 + (STEP_) < hi) { LOAD_(0, p0 + (STEP_)); }                                                                    \
        EMIT_(1, p0);                                                                                                         \
        p0 += (STEP_);                                                                                                        \
        if (p0 >= hi) break;                                                                                                  \
    }

__global__ __launch_bounds__(FA_THREADS) void bn_fin_apply_kernel(const float* __restrict__ partial, int parts, int C, int CS,
                                                                   const BnFinArgs fin, const float* __restrict__ x, int64_t P,
                                                                   int64_t ppr, const float* __restrict__ res,
                                                                   const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                                   int relu, float* __restrict__ y, unsigned char* __restrict__ sign_bits) {
    __shared__ double red[FA_THREADS * 3];
    __shared__ __attribute__((aligned(16))) float tab[2 * 32];
    const int tid = threadIdx.x, slice = blockIdx.y, q = CS / 4, tq = tid % q, tp = tid / q, ppi = FA_THREADS / q;
    const int Q = C / 4, cq = slice * q + tq;
    const int64_t lo = blockIdx.x * ppr, hi = (lo + ppr < P) ? lo + ppr : P;
    const int64_t step = (int64_t)ppi * FA_U;
    float4 xv[2][FA_U], rv[2][FA_U];
#define DAM_FA_LOAD(S_, P0_)                                                                                                  \
    _Pragma("unroll") for (int u = 0; u < FA_U; ++u) {                                                                        \
        const int64_t p = (P0_) + (int64_t)u * ppi;                                                                           \
        xv[S_][u] = p < hi ? reinterpret_cast<const float4*>(x)[p * Q + cq] : DAM_Z4;                                         \
        if (res) rv[S_][u] = p < hi ? reinterpret_cast<const float4*>(res)[p * Q + cq] : DAM_Z4;                              \
    }
    // the first pieces are requested before the records: the table's round trip hides behind them
    DAM_FA_LOAD(0, lo + tp)
    float4 ra = make_float4(1.f, 1.f, 1.f, 1.f), rb = DAM_Z4;
    if (rscale) { ra = reinterpret_cast<const float4*>(rscale)[cq]; rb = reinterpret_cast<const float4*>(rshift)[cq]; }
    {
        const int cl = tid % CS, ch = slice * CS + cl;
        const float m0 = partial[(size_t)ch * 3 + 1];
        const float gam = fin.gamma[ch], bet = fin.beta[ch];
        const bool writer = blockIdx.x == 0 && tid < CS;
        const float rm = (writer && fin.running_mean) ? fin.running_mean[ch] : 0.f, rvv = (writer && fin.running_mean) ? fin.running_var[ch] : 0.f;
        double acc[3];
        fa_slice_sums<3, 3>(partial, parts, C, ch, CS, red, acc, [m0](const float (&r)[3], double (&v)[3]) {
            const double n = (double)r[0], d = (double)r[1] - (double)m0;
            v[0] = n; v[1] = n * d; v[2] = n != 0.0 ? (double)r[2] + n * d * d : 0.0;
        });
        if (tid < CS) {
            const double na = acc[0], ds = acc[1] / na, qa = acc[2] - acc[1] * ds;
            const double var = qa / na;
            const float mean = (float)((double)m0 + ds);
            const float invstd = (float)(1.0 / sqrt((var > 0 ? var : 0.0) + (double)fin.eps));
            const float sc = gam * invstd, sh = bet - mean * sc;
            tab[cl] = sc; tab[CS + cl] = sh;
            if (writer) {
                fin.save_mean[ch] = mean; fin.save_invstd[ch] = invstd; fin.scale[ch] = sc; fin.shift[ch] = sh;
                if (fin.running_mean) {
                    const double unbiased = na > 1 ? (qa > 0 ? qa : 0.0) / (na - 1) : var;
                    fin.running_mean[ch] = (1.f - fin.momentum) * rm + fin.momentum * mean;
                    fin.running_var[ch] = (1.f - fin.momentum) * rvv + fin.momentum * (float)unbiased;
                }
                if (ch == 0 && fin.num_batches) *fin.num_batches += 1;
            }
        }
        __syncthreads();
    }
    const float4 sc = *reinterpret_cast<const float4*>(tab + tq * 4), sh = *reinterpret_cast<const float4*>(tab + CS + tq * 4);
#define DAM_FA_EMIT(S_, P0_)                                                                                                  \
    _Pragma("unroll") for (int u = 0; u < FA_U; ++u) {                                                                        \
        const int64_t p = (P0_) + (int64_t)u * ppi;                                                                           \
        if (p < hi) {                                                                                                         \
            const float4 v = xv[S_][u];                                                                                       \
            float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)); \
            if (res) {                                                                                                        \
                float4 r = rv[S_][u];                                                                                         \
                if (rscale) r = make_float4(fmaf(r.x, ra.x, rb.x), fmaf(r.y, ra.y, rb.y), fmaf(r.z, ra.z, rb.z), fmaf(r.w, ra.w, rb.w)); \
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;                                                               \
            }                                                                                                                 \
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }         \
            reinterpret_cast<float4*>(y)[p * Q + cq] = o;                                                                     \
            if (sign_bits)                                                                                                    \
                sign_bits[p * Q + cq] = (unsigned char)((o.x > 0.f) | ((o.y > 0.f) << 1) | ((o.z > 0.f) << 2) | ((o.w > 0.f) << 3)); \
        }                                                                                                                     \
    }
    DAM_FA_STREAM(DAM_FA_LOAD, DAM_FA_EMIT, step)
#undef DAM_FA_LOAD
#undef DAM_FA_EMIT
}

// Backward: records [parts][C][2] = (sum dz, sum dz * xhat) -> dgamma / dbeta (range 0 writes them) and dx = c1 dz + c2 x + c3.
template <int MASK>
__global__ __launch_bounds__(FA_THREADS) void bn_bwd_fin_apply_kernel(const float* __restrict__ partial, int parts, int C, int CS,
                                                                       double count, const float* __restrict__ gamma,
                                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                       int training, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                       const float* __restrict__ dy, const float* __restrict__ y_mask,
                                                                       const float* __restrict__ x, int64_t P, int64_t ppr,
                                                                       const float* __restrict__ mscale, const float* __restrict__ mshift,
                                                                       float* __restrict__ dx) {
    __shared__ double red[FA_THREADS * 2];
    __shared__ __attribute__((aligned(16))) float tab[3 * 32];
    const int tid = threadIdx.x, slice = blockIdx.y, q = CS / 4, tq = tid % q, tp = tid / q, ppi = FA_THREADS / q;
    const int Q = C / 4, cq = slice * q + tq;
    const int64_t lo = blockIdx.x * ppr, hi = (lo + ppr < P) ? lo + ppr : P;
    constexpr int UB = 3;               // three streams: three pieces each per register set
    const int64_t step = (int64_t)ppi * UB;
    float4 gv[2][UB], xv[2][UB], mv[2][UB];
#define DAM_FA_LOAD(S_, P0_)                                                                                                  \
    _Pragma("unroll") for (int u = 0; u < UB; ++u) {                                                                          \
        const int64_t p = (P0_) + (int64_t)u * ppi;                                                                           \
        const bool ok = p < hi;                                                                                               \
        gv[S_][u] = ok ? reinterpret_cast<const float4*>(dy)[p * Q + cq] : DAM_Z4;                                            \
        xv[S_][u] = ok ? reinterpret_cast<const float4*>(x)[p * Q + cq] : DAM_Z4;                                             \
        if (MASK == 1) mv[S_][u] = ok ? reinterpret_cast<const float4*>(y_mask)[p * Q + cq] : DAM_Z4;                         \
        if (MASK == 3) mv[S_][u] = sign_quad(ok ? reinterpret_cast<const unsigned char*>(y_mask)[p * Q + cq] : 0);            \
    }
    DAM_FA_LOAD(0, lo + tp)
    float4 msc = DAM_Z4, msh = DAM_Z4;
    if (MASK == 2) { msc = reinterpret_cast<const float4*>(mscale)[cq]; msh = reinterpret_cast<const float4*>(mshift)[cq]; }
    {
        const int cl = tid % CS, ch = slice * CS + cl;
        const float gam = gamma[ch], inv = invstd[ch], mu = mean[ch];
        double acc[2];
        fa_slice_sums<2, 2>(partial, parts, C, ch, CS, red, acc, [](const float (&r)[2], double (&v)[2]) { v[0] = (double)r[0]; v[1] = (double)r[1]; });
        if (tid < CS) {
            const double s1 = acc[0], s2 = acc[1];
            const double g = (double)gam * inv;
            double c2 = 0, c3 = 0;
            if (training) { c2 = -g * inv * s2 / count; c3 = -g * s1 / count - c2 * mu; }
            tab[cl] = (float)g; tab[CS + cl] = (float)c2; tab[2 * CS + cl] = (float)c3;
            if (blockIdx.x == 0) { dbeta[ch] = (float)s1; dgamma[ch] = (float)s2; }
        }
        __syncthreads();
    }
    const float4 c1 = *reinterpret_cast<const float4*>(tab + tq * 4), c2 = *reinterpret_cast<const float4*>(tab + CS + tq * 4),
                 c3 = *reinterpret_cast<const float4*>(tab + 2 * CS + tq * 4);
#define DAM_FA_EMIT(S_, P0_)                                                                                                  \
    _Pragma("unroll") for (int u = 0; u < UB; ++u) {                                                                          \
        const int64_t p = (P0_) + (int64_t)u * ppi;                                                                           \
        if (p < hi) {                                                                                                         \
            float4 g = gv[S_][u];                                                                                             \
            const float4 v = xv[S_][u];                                                                                       \
            if (MASK != 0) {                                                                                                  \
                float4 m = mv[S_][u];                                                                                         \
                if (MASK == 2) m = make_float4(fmaf(v.x, msc.x, msh.x), fmaf(v.y, msc.y, msh.y), fmaf(v.z, msc.z, msh.z), fmaf(v.w, msc.w, msh.w)); \
                g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f; g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f; \
            }                                                                                                                 \
            float4 o;                                                                                                         \
            o.x = fmaf(c1.x, g.x, fmaf(c2.x, v.x, c3.x)); o.y = fmaf(c1.y, g.y, fmaf(c2.y, v.y, c3.y));                       \
            o.z = fmaf(c1.z, g.z, fmaf(c2.z, v.z, c3.z)); o.w = fmaf(c1.w, g.w, fmaf(c2.w, v.w, c3.w));                       \
            reinterpret_cast<float4*>(dx)[p * Q + cq] = o;                                                                    \
        }                                                                                                                     \
    }
    DAM_FA_STREAM(DAM_FA_LOAD, DAM_FA_EMIT, step)
#undef DAM_FA_LOAD
#undef DAM_FA_EMIT
}

// The pair form (a residual block's bn2 and its shortcut BatchNorm): records [parts][C][3] = (sum dz, sum dz xhat_a, sum dz xhat_b).
template <bool BITS>
__global__ __launch_bounds__(FA_THREADS) void bn_bwd_fin_apply_pair_kernel(const float* __restrict__ partial, int parts, int C, int CS,
        double count, const float* __restrict__ gamma_a, const float* __restrict__ mean_a, const float* __restrict__ invstd_a,
        const float* __restrict__ gamma_b, const float* __restrict__ mean_b, const float* __restrict__ invstd_b, int training,
        float* __restrict__ dgamma_a, float* __restrict__ dbeta_a, float* __restrict__ dgamma_b, float* __restrict__ dbeta_b,
        const float* __restrict__ dy, const float* __restrict__ y_mask, const float* __restrict__ xa, const float* __restrict__ xb,
        int64_t P, int64_t ppr, float* __restrict__ dxa, float* __restrict__ dxb) {
    __shared__ double red[FA_THREADS * 3];
    __shared__ __attribute__((aligned(16))) float tab[6 * 32];
    const int tid = threadIdx.x, slice = blockIdx.y, q = CS / 4, tq = tid % q, tp = tid / q, ppi = FA_THREADS / q;
    const int Q = C / 4, cq = slice * q + tq;
    const int64_t lo = blockIdx.x * ppr, hi = (lo + ppr < P) ? lo + ppr : P;
    constexpr int UP = 2;               // four streams: two pieces each per register set
    const int64_t step = (int64_t)ppi * UP;
    float4 gv[2][UP], va[2][UP], vb[2][UP], mv[2][UP];
#define DAM_FA_LOAD(S_, P0_)                                                                                                  \
    _Pragma("unroll") for (int u = 0; u < UP; ++u) {                                                                          \
        const int64_t p = (P0_) + (int64_t)u * ppi;                                                                           \
        const bool ok = p < hi;                                                                                               \
        gv[S_][u] = ok ? reinterpret_cast<const float4*>(dy)[p * Q + cq] : DAM_Z4;                                            \
        va[S_][u] = ok ? reinterpret_cast<const float4*>(xa)[p * Q + cq] : DAM_Z4;                                            \
        vb[S_][u] = ok ? reinterpret_cast<const float4*>(xb)[p * Q + cq] : DAM_Z4;                                            \
        if (BITS) mv[S_][u] = sign_quad(ok ? reinterpret_cast<const unsigned char*>(y_mask)[p * Q + cq] : 0);                 \
        else mv[S_][u] = ok ? reinterpret_cast<const float4*>(y_mask)[p * Q + cq] : DAM_Z4;                                   \
    }
    DAM_FA_LOAD(0, lo + tp)
    {
        const int cl = tid % CS, ch = slice * CS + cl;
        const float ga_ = gamma_a[ch], ia = invstd_a[ch], ma = mean_a[ch], gb_ = gamma_b[ch], ib = invstd_b[ch], mb = mean_b[ch];
        double acc[3];
        fa_slice_sums<3, 3>(partial, parts, C, ch, CS, red, acc,
                            [](const float (&r)[3], double (&v)[3]) { v[0] = (double)r[0]; v[1] = (double)r[1]; v[2] = (double)r[2]; });
        if (tid < CS) {
            const double s1 = acc[0], s2 = acc[1], s3 = acc[2];
            const double ga = (double)ga_ * ia, gb = (double)gb_ * ib;
            double a2 = 0, a3 = 0, b2 = 0, b3 = 0;
            if (training) {
                a2 = -ga * ia * s2 / count; a3 = -ga * s1 / count - a2 * ma;
                b2 = -gb * ib * s3 / count; b3 = -gb * s1 / count - b2 * mb;
            }
            tab[cl] = (float)ga; tab[CS + cl] = (float)a2; tab[2 * CS + cl] = (float)a3;
            tab[3 * CS + cl] = (float)gb; tab[4 * CS + cl] = (float)b2; tab[5 * CS + cl] = (float)b3;
            if (blockIdx.x == 0) {
                dbeta_a[ch] = (float)s1; dbeta_b[ch] = (float)s1; dgamma_a[ch] = (float)s2; dgamma_b[ch] = (float)s3;
            }
        }
        __syncthreads();
    }
    float4 k[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) k[i] = *reinterpret_cast<const float4*>(tab + i * CS + tq * 4);
#define DAM_FA_EMIT(S_, P0_)                                                                                                  \
    _Pragma("unroll") for (int u = 0; u < UP; ++u) {                                                                          \
        const int64_t p = (P0_) + (int64_t)u * ppi;                                                                           \
        if (p < hi) {                                                                                                         \
            float4 g = gv[S_][u];                                                                                             \
            const float4 m = mv[S_][u], v = va[S_][u], w = vb[S_][u];                                                         \
            g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f; g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f; \
            float4 o, r;                                                                                                      \
            o.x = fmaf(k[0].x, g.x, fmaf(k[1].x, v.x, k[2].x)); o.y = fmaf(k[0].y, g.y, fmaf(k[1].y, v.y, k[2].y));           \
            o.z = fmaf(k[0].z, g.z, fmaf(k[1].z, v.z, k[2].z)); o.w = fmaf(k[0].w, g.w, fmaf(k[1].w, v.w, k[2].w));           \
            r.x = fmaf(k[3].x, g.x, fmaf(k[4].x, w.x, k[5].x)); r.y = fmaf(k[3].y, g.y, fmaf(k[4].y, w.y, k[5].y));           \
            r.z = fmaf(k[3].z, g.z, fmaf(k[4].z, w.z, k[5].z)); r.w = fmaf(k[3].w, g.w, fmaf(k[4].w, w.w, k[5].w));           \
            reinterpret_cast<float4*>(dxa)[p * Q + cq] = o;                                                                   \
            reinterpret_cast<float4*>(dxb)[p * Q + cq] = r;                                                                   \
        }                                                                                                                     \
    }
    DAM_FA_STREAM(DAM_FA_LOAD, DAM_FA_EMIT, step)
#undef DAM_FA_LOAD
#undef DAM_FA_EMIT
}
#undef DAM_FA_STREAM
#undef DAM_Z4
inline int elt_blocks(int64_t n) {
    int64_t b = cdiv(n, 256);
    return (int)(b < 4096 ? (b < 1 ? 1 : b) : 4096);
}

// DAM_BN_FUSED_FIN=0 keeps the separate finalize launches (A/B switch, read once)
inline bool fa_enabled() {
    static const bool on = [] { const char* e = getenv("DAM_BN_FUSED_FIN"); return !(e && e[0] == '0'); }();
    return on;
}

// records a partial pass may leave for a fused consumer: a workgroup's slice table stays <= 64 KB
inline int fa_max_parts(int C, int rec_floats) {
    static const int kb = [] { const char* e = getenv("DAM_BN_FA_PARTS_KB"); const int v = e ? atoi(e) : 0; return v >= 16 && v <= 80 ? v : 64; }();   // A/B knob
    int m = kb * 1024 / (fa_cs(C) * rec_floats * 4);
    if (m > BN_MAX_PARTS) m = BN_MAX_PARTS;
    return m < 64 ? 64 : m;
}

}  // namespace
}  // namespace dam

using namespace dam;

extern "C" int64_t dam_bn_workspace_floats(int C) { return (int64_t)BN_RECORDS_MAX * C * 3; }

// First half of dam_bn_stats_f32 on its own: the partial records [*parts_host][C][3], sized for a fused consumer
// (dam_bn_finalize_apply_f32) -- or for dam_bn_finalize_f32.
extern "C" int dam_bn_stats_partial_f32(const float* x, int64_t n_pixels, int C, float* workspace, int* parts_host, void* stream) {
    if (!x || !workspace || !parts_host || n_pixels <= 0) return DAM_ERR_BAD_ARG;
    if (C % 16 || C > 1024) return DAM_ERR_UNSUPPORTED;
    const BnLaunch l = bn_plan(n_pixels, C, fa_max_parts(C, 3));
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(l.parts), dim3(l.threads), (size_t)l.r * C * 3 * sizeof(float), (hipStream_t)stream,
                       x, n_pixels, C, l.q, l.r, l.ppb, workspace, BnFinArgs{}, (const float*)nullptr, (float*)nullptr);
    DAM_CHECK_LAUNCH();
    *parts_host = l.parts;
    return DAM_OK;
}

// Finalize + apply in one launch: merges `parts` records [parts][C][3] (a convolution epilogue's, dam_bn_stats_partial_f32's)
// exactly as dam_bn_finalize_f32 does -- fin's outputs and running statistics are written -- and applies
// y = relu?(x * scale + shift [+ res [* res_scale + res_shift]]) [+ sign bytes] as dam_bn_apply_f32 does.
extern "C" int dam_bn_finalize_apply_f32(const float* partial, int parts, int C, const dam_bn_fin* fin, const float* x,
                                         int64_t n_pixels, const float* res, const float* res_scale, const float* res_shift,
                                         int relu, float* y, uint8_t* sign_bits, void* stream) {
    if (!partial || parts <= 0 || !fin || !x || !y || n_pixels <= 0) return DAM_ERR_BAD_ARG;
    if (!fin->gamma || !fin->beta || !fin->save_mean || !fin->save_invstd || !fin->scale || !fin->shift) return DAM_ERR_BAD_ARG;
    if (res_scale && (!res || !res_shift)) return DAM_ERR_BAD_ARG;
    if (C % 16 || C > 1024) return DAM_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (!fa_enabled() || !fa_fwd_table_ok(C, parts)) {
        int rc = dam_bn_finalize_f32(partial, parts, C, fin->gamma, fin->beta, fin->running_mean, fin->running_var,
                                     fin->num_batches_tracked, fin->momentum, fin->eps, fin->save_mean, fin->save_invstd, fin->scale,
                                     fin->shift, stream);
        if (rc != DAM_OK) return rc;
        return dam_bn_apply_f32(x, n_pixels, C, fin->scale, fin->shift, res, res_scale, res_shift, relu, y, sign_bits, stream);
    }
    const FaPlan f = fa_plan(n_pixels, C);
    const BnFinArgs a{fin->gamma, fin->beta, fin->running_mean, fin->running_var, (long long*)fin->num_batches_tracked, fin->momentum,
                      fin->eps, fin->save_mean, fin->save_invstd, fin->scale, fin->shift, nullptr};
    hipLaunchKernelGGL(bn_fin_apply_kernel, dim3(f.nranges, f.nslices), dim3(FA_THREADS), 0, st, partial, parts, C, f.cs, a, x,
                       n_pixels, f.ppr, res, res_scale, res_shift, relu, y, sign_bits);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_bn_stats_f32(const float* x, int64_t n_pixels, int C, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                float momentum, float eps, float* save_mean, float* save_invstd, float* scale,
                                float* shift, float* workspace, uint32_t* counter, void* stream) {
    if (!x || !gamma || !beta || !save_mean || !save_invstd || !scale || !shift || !workspace || n_pixels <= 0)
        return DAM_ERR_BAD_ARG;
    if (C % 16 || C > 1024) return DAM_ERR_UNSUPPORTED;
    const BnLaunch l = bn_plan(n_pixels, C);
    hipStream_t st = (hipStream_t)stream;
    const BnFinArgs fin{gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, momentum, eps,
                        save_mean, save_invstd, scale, shift, counter};
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(l.parts), dim3(l.threads), (size_t)l.r * C * 3 * sizeof(float), st,
                       x, n_pixels, C, l.q, l.r, l.ppb, workspace, fin, (const float*)nullptr, (float*)nullptr);
    DAM_CHECK_LAUNCH();
    if (!counter) {
        hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(64), 0, st, workspace, l.parts, C,
                           gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, momentum, eps,
                           save_mean, save_invstd, scale, shift, BnFinArgs{}, (const float*)nullptr);
        DAM_CHECK_LAUNCH();
    }
    return DAM_OK;
}

// Training-mode statistics of TWO tensors of one shape (a block's conv1 output and its shortcut convolution's output: two
// independent BatchNorms that become ready together) in one partial + one finalize launch.
extern "C" int dam_bn_stats_pair_f32(const float* x_a, const float* x_b, int64_t n_pixels, int C, const dam_bn_fin* a,
                                     const dam_bn_fin* b, float* workspace, void* stream) {
    if (!x_a || !x_b || !a || !b || !workspace || n_pixels <= 0) return DAM_ERR_BAD_ARG;
    if (!a->gamma || !a->beta || !a->save_mean || !a->save_invstd || !a->scale || !a->shift || !b->gamma || !b->beta ||
        !b->save_mean || !b->save_invstd || !b->scale || !b->shift)
        return DAM_ERR_BAD_ARG;
    if (C % 16 || C > 1024) return DAM_ERR_UNSUPPORTED;
    const BnLaunch l = bn_plan(n_pixels, C);
    hipStream_t st = (hipStream_t)stream;
    float* ws_b = workspace + (size_t)BN_RECORDS_MAX * C * 3;
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(l.parts, 2), dim3(l.threads), (size_t)l.r * C * 3 * sizeof(float), st, x_a,
                       n_pixels, C, l.q, l.r, l.ppb, workspace, BnFinArgs{}, x_b, ws_b);
    DAM_CHECK_LAUNCH();
    const BnFinArgs fb{b->gamma, b->beta, b->running_mean, b->running_var, (long long*)b->num_batches_tracked, b->momentum,
                       b->eps, b->save_mean, b->save_invstd, b->scale, b->shift, nullptr};
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C, 2), dim3(64), 0, st, workspace, l.parts, C, a->gamma, a->beta,
                       a->running_mean, a->running_var, (long long*)a->num_batches_tracked, a->momentum, a->eps, a->save_mean,
                       a->save_invstd, a->scale, a->shift, fb, (const float*)ws_b);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_bn_finalize_f32(const float* partial, int parts, int C, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                   float momentum, float eps, float* save_mean, float* save_invstd, float* scale,
                                   float* shift, void* stream) {
    if (!partial || parts <= 0 || C <= 0 || !gamma || !beta || !save_mean || !save_invstd || !scale || !shift)
        return DAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, partial, parts, C, gamma, beta,
                       running_mean, running_var, (long long*)num_batches_tracked, momentum, eps, save_mean, save_invstd,
                       scale, shift, BnFinArgs{}, (const float*)nullptr);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_bn_finalize_pair_f32(const float* partial_a, const float* partial_b, int parts, int C, const dam_bn_fin* a,
                                        const dam_bn_fin* b, void* stream) {
    if (!partial_a || !partial_b || parts <= 0 || C <= 0 || !a || !b) return DAM_ERR_BAD_ARG;
    if (!a->gamma || !a->beta || !a->save_mean || !a->save_invstd || !a->scale || !a->shift || !b->gamma || !b->beta ||
        !b->save_mean || !b->save_invstd || !b->scale || !b->shift)
        return DAM_ERR_BAD_ARG;
    const BnFinArgs fb{b->gamma, b->beta, b->running_mean, b->running_var, (long long*)b->num_batches_tracked, b->momentum,
                       b->eps, b->save_mean, b->save_invstd, b->scale, b->shift, nullptr};
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C, 2), dim3(64), 0, (hipStream_t)stream, partial_a, parts, C, a->gamma, a->beta,
                       a->running_mean, a->running_var, (long long*)a->num_batches_tracked, a->momentum, a->eps, a->save_mean,
                       a->save_invstd, a->scale, a->shift, fb, partial_b);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_bn_eval_affine_f32(int C, const float* gamma, const float* beta, const float* running_mean,
                                      const float* running_var, float eps, float* save_mean, float* save_invstd,
                                      float* scale, float* shift, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !save_mean || !save_invstd || !scale || !shift || C <= 0)
        return DAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, C, gamma, beta,
                       running_mean, running_var, eps, save_mean, save_invstd, scale, shift);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_bn_apply_f32(const float* x, int64_t n_pixels, int C, const float* scale, const float* shift,
                                const float* res, const float* res_scale, const float* res_shift, int relu, float* y,
                                uint8_t* sign_bits, void* stream) {
    if (!x || !scale || !shift || !y || n_pixels <= 0) return DAM_ERR_BAD_ARG;
    if (C % 16) return DAM_ERR_UNSUPPORTED;
    if (res_scale && (!res || !res_shift)) return DAM_ERR_BAD_ARG;
    const int64_t nq = n_pixels * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(elt_blocks(nq)), dim3(256), 0, (hipStream_t)stream, x, nq, C / 4, scale, shift,
                       res, res_scale, res_shift, relu, y, sign_bits);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_bn_backward_f32(const float* dy, const float* y_mask, const float* x, int64_t n_pixels, int C,
                                   const float* gamma, const float* save_mean, const float* save_invstd, int training,
                                   const float* mask_scale, const float* mask_shift, const uint8_t* mask_bits, float* dx,
                                   float* dgamma, float* dbeta, float* workspace, int partials_given, uint32_t* counter,
                                   void* stream) {
    if (!dy || !x || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace || n_pixels <= 0)
        return DAM_ERR_BAD_ARG;
    if ((mask_scale != nullptr) != (mask_shift != nullptr) || (y_mask && mask_scale) || (mask_bits && (y_mask || mask_scale)))
        return DAM_ERR_BAD_ARG;
    if (mask_bits) y_mask = reinterpret_cast<const float*>(mask_bits);       // MASK == 3 reads it as bytes
    if (C % 16 || C > 1024) return DAM_ERR_UNSUPPORTED;
    const bool fused = fa_enabled() && !counter;
    BnLaunch l = bn_plan(n_pixels, C, fused ? fa_max_parts(C, 2) : BN_MAX_PARTS);
    if (partials_given < 0 || partials_given > BN_MAX_PARTS) return DAM_ERR_BAD_ARG;
    if (partials_given) { l.parts = partials_given; counter = nullptr; }       // records from a data-gradient epilogue
    const bool fuse_now = fused && fa_table_ok(C, l.parts, 2);
    hipStream_t st = (hipStream_t)stream;
    float* coef = workspace + (size_t)BN_MAX_PARTS * C * 2;    // workspace holds [parts][C][2] then [3][C]
    const int mask = mask_bits ? 3 : (y_mask ? 1 : (mask_scale ? 2 : 0));
    const BnBwdFin fin{(double)n_pixels, gamma, save_mean, save_invstd, training, dgamma, dbeta, coef, counter};
#define DAM_BN_PARTIAL(M_)                                                                                                   \
    hipLaunchKernelGGL(bn_bwd_partial_kernel<M_>, dim3(l.parts), dim3(l.threads), (size_t)l.r * C * 2 * sizeof(float), st, dy, \
                       y_mask, x, n_pixels, C, l.q, l.r, l.ppb, save_mean, save_invstd, mask_scale, mask_shift, workspace, fin)
    if (partials_given) { }
    else if (mask == 1) DAM_BN_PARTIAL(1); else if (mask == 2) DAM_BN_PARTIAL(2); else if (mask == 3) DAM_BN_PARTIAL(3); else DAM_BN_PARTIAL(0);
#undef DAM_BN_PARTIAL
    DAM_CHECK_LAUNCH();
    if (fuse_now) {     // finalize inside the apply launch (bn_bwd_fin_apply_kernel)
        const FaPlan f = fa_plan(n_pixels, C);
#define DAM_BN_FA(M_)                                                                                                        \
    hipLaunchKernelGGL(bn_bwd_fin_apply_kernel<M_>, dim3(f.nranges, f.nslices), dim3(FA_THREADS), 0, st, workspace, l.parts, C, \
                       f.cs, (double)n_pixels, gamma, save_mean, save_invstd, training, dgamma, dbeta, dy, y_mask, x, n_pixels, \
                       f.ppr, mask_scale, mask_shift, dx)
        if (mask == 1) DAM_BN_FA(1); else if (mask == 2) DAM_BN_FA(2); else if (mask == 3) DAM_BN_FA(3); else DAM_BN_FA(0);
#undef DAM_BN_FA
        DAM_CHECK_LAUNCH();
        return DAM_OK;
    }
    if (!counter) {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, st, workspace, l.parts, C,
                           (double)n_pixels, gamma, save_mean, save_invstd, training, dgamma, dbeta, coef);
        DAM_CHECK_LAUNCH();
    }
    const int64_t nq = n_pixels * (C / 4);
#define DAM_BN_APPLY(M_)                                                                                                     \
    hipLaunchKernelGGL(bn_bwd_apply_kernel<M_>, dim3(elt_blocks(nq)), dim3(256), 0, st, dy, y_mask, x, nq, C / 4, C, coef,   \
                       mask_scale, mask_shift, dx)
    if (mask == 1) DAM_BN_APPLY(1); else if (mask == 2) DAM_BN_APPLY(2); else if (mask == 3) DAM_BN_APPLY(3); else DAM_BN_APPLY(0);
#undef DAM_BN_APPLY
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int64_t dam_bn_pair_workspace_floats(int C) { return (int64_t)BN_MAX_PARTS * C * 3 + 6 * (int64_t)C; }

extern "C" int dam_bn_backward_pair_f32(const float* dy, const float* y_mask, const uint8_t* mask_bits, int64_t n_pixels, int C,
                                        int training,
                                        const float* x_a, const float* gamma_a, const float* mean_a, const float* invstd_a,
                                        float* dx_a, float* dgamma_a, float* dbeta_a,
                                        const float* x_b, const float* gamma_b, const float* mean_b, const float* invstd_b,
                                        float* dx_b, float* dgamma_b, float* dbeta_b, float* workspace, void* stream) {
    if ((y_mask != nullptr) == (mask_bits != nullptr)) return DAM_ERR_BAD_ARG;      // exactly one form of the mask
    const bool bits = mask_bits != nullptr;
    if (bits) y_mask = reinterpret_cast<const float*>(mask_bits);
    if (!dy || !y_mask || !x_a || !gamma_a || !mean_a || !invstd_a || !dx_a || !dgamma_a || !dbeta_a || !x_b || !gamma_b ||
        !mean_b || !invstd_b || !dx_b || !dgamma_b || !dbeta_b || !workspace || n_pixels <= 0)
        return DAM_ERR_BAD_ARG;
    if (C % 16 || C > 1024) return DAM_ERR_UNSUPPORTED;
    const bool fused = fa_enabled();
    const BnLaunch l = bn_plan(n_pixels, C, fused ? fa_max_parts(C, 3) : BN_MAX_PARTS);
    if ((size_t)l.r * C * 3 * sizeof(float) > 64 * 1024) return DAM_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    float* coef = workspace + (size_t)BN_MAX_PARTS * C * 3;
    if (bits)
        hipLaunchKernelGGL(bn_bwd_partial_pair_kernel<true>, dim3(l.parts), dim3(l.threads), (size_t)l.r * C * 3 * sizeof(float), st,
                           dy, y_mask, x_a, x_b, n_pixels, C, l.q, l.r, l.ppb, mean_a, invstd_a, mean_b, invstd_b, workspace);
    else
        hipLaunchKernelGGL(bn_bwd_partial_pair_kernel<false>, dim3(l.parts), dim3(l.threads), (size_t)l.r * C * 3 * sizeof(float), st,
                           dy, y_mask, x_a, x_b, n_pixels, C, l.q, l.r, l.ppb, mean_a, invstd_a, mean_b, invstd_b, workspace);
    DAM_CHECK_LAUNCH();
    if (fused) {
        const FaPlan f = fa_plan(n_pixels, C);
        if (bits)
            hipLaunchKernelGGL(bn_bwd_fin_apply_pair_kernel<true>, dim3(f.nranges, f.nslices), dim3(FA_THREADS), 0, st, workspace, l.parts,
                               C, f.cs, (double)n_pixels, gamma_a, mean_a, invstd_a, gamma_b, mean_b, invstd_b, training, dgamma_a,
                               dbeta_a, dgamma_b, dbeta_b, dy, y_mask, x_a, x_b, n_pixels, f.ppr, dx_a, dx_b);
        else
            hipLaunchKernelGGL(bn_bwd_fin_apply_pair_kernel<false>, dim3(f.nranges, f.nslices), dim3(FA_THREADS), 0, st, workspace, l.parts,
                               C, f.cs, (double)n_pixels, gamma_a, mean_a, invstd_a, gamma_b, mean_b, invstd_b, training, dgamma_a,
                               dbeta_a, dgamma_b, dbeta_b, dy, y_mask, x_a, x_b, n_pixels, f.ppr, dx_a, dx_b);
        DAM_CHECK_LAUNCH();
        return DAM_OK;
    }
    hipLaunchKernelGGL(bn_bwd_finalize_pair_kernel, dim3(C), dim3(64), 0, st, workspace, l.parts, C, (double)n_pixels, gamma_a,
                       mean_a, invstd_a, gamma_b, mean_b, invstd_b, training, dgamma_a, dbeta_a, dgamma_b, dbeta_b, coef);
    DAM_CHECK_LAUNCH();
    const int64_t nq = n_pixels * (C / 4);
    if (bits)
        hipLaunchKernelGGL(bn_bwd_apply_pair_kernel<true>, dim3(elt_blocks(nq)), dim3(256), 0, st, dy, y_mask, x_a, x_b, nq, C / 4, C,
                           coef, dx_a, dx_b);
    else
        hipLaunchKernelGGL(bn_bwd_apply_pair_kernel<false>, dim3(elt_blocks(nq)), dim3(256), 0, st, dy, y_mask, x_a, x_b, nq, C / 4, C,
                           coef, dx_a, dx_b);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_channel_sum_f32(const float* x, int64_t n_pixels, int C, int n_real, float* out, float* workspace,
                                   void* stream) {
    if (!x || !out || !workspace || n_pixels <= 0 || n_real <= 0 || n_real > C) return DAM_ERR_BAD_ARG;
    if (C % 16 || C > 1024) return DAM_ERR_UNSUPPORTED;
    const BnLaunch l = bn_plan(n_pixels, C);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(l.parts), dim3(l.threads), (size_t)l.r * C * sizeof(float), st, x,
                       n_pixels, C, l.q, l.r, l.ppb, workspace);
    DAM_CHECK_LAUNCH();
    hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(n_real), dim3(64), 0, st, workspace, l.parts, C,
                       n_real, out);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}
