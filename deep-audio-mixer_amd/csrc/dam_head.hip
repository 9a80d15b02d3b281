// dam_head.hip -- per-stem gain heads, gain-weighted spectrogram sum, fused MSE.
//
// Replaces models/model_resnet.py:75-85,108-126 (identical code in models/model_scalar_1s.py:222-232,244-273
// and models/model_scalar_2s.py:79-132):  for every stem s
//     h_s = relu(conv1x1_s(trunk) + cb_s)          [B, P]      (P = flattened_dim, NCHW flatten == pixel order)
//     g_s = fc_s(h_s) + fcb_s                       [B, 1]
//     masked = sum_s g_s * x[:, s]                  [B, F, T]
// and the MSELoss the trainer applies to `masked` (model_trainer.py:35), forward and backward.
// All S heads are evaluated together: the trunk is read once (one wave per pixel, 16-byte loads),
// instead of S separate 1x1 convolutions + S Linear layers + S+1 element-wise launches.
#include "dam_common.h"

namespace dam {
namespace {

constexpr int MAX_STEMS = 16;

// h[b][s][p] = relu(cb[s] + sum_c trunk[b][p][c] * cw[s][c]);  one lane GROUP per pixel: 64 lanes for >= 132 channels (the ResNet's
// 256), 32 / 16 lanes for <= 128 / <= 64 (the scalar models' 128: a whole wave per pixel left half its lanes idle and did six
// shuffle steps per stem where five serve -- 37.7 us for 41 MB).
template <int GS>
__global__ __launch_bounds__(256) void head_conv_kernel(const float* __restrict__ trunk, int P, int C, int S,
                                                        const float* __restrict__ cw, const float* __restrict__ cb,
                                                        float* __restrict__ h) {
    extern __shared__ float w_s[];   // [S][C]
    for (int e = threadIdx.x; e < S * C; e += blockDim.x) w_s[e] = cw[e];
    __syncthreads();
    constexpr int G = 64 / GS;                               // pixels per wave and round
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    const int gl = lane % GS, gi = lane / GS;
    const int Q = C / 4;
    for (int p0 = (blockIdx.x * 4 + wave) * G; p0 < P; p0 += gridDim.x * 4 * G) {
        const int p = p0 + gi;
        const bool live = p < P;
        float acc[MAX_STEMS];
#pragma unroll
        for (int s = 0; s < MAX_STEMS; ++s) acc[s] = 0.f;
        const float4* row = reinterpret_cast<const float4*>(trunk + ((size_t)b * P + (live ? p : 0)) * C);
        for (int q = gl; q < Q; q += GS) {
            const float4 v = live ? row[q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int s = 0; s < MAX_STEMS; ++s) {
                if (s < S) {
                    const float4 w = reinterpret_cast<const float4*>(w_s + s * C)[q];
                    acc[s] = fmaf(v.x, w.x, fmaf(v.y, w.y, fmaf(v.z, w.z, fmaf(v.w, w.w, acc[s]))));
                }
            }
        }
#pragma unroll
        for (int s = 0; s < MAX_STEMS; ++s) {
            if (s < S) {
                float t = wave_sum16(acc[s]);
                if (GS >= 32) t += __shfl_xor(t, 16);
                if (GS >= 64) t += __shfl_xor(t, 32);
                if (gl == 0 && live) h[((size_t)b * S + s) * P + p] = fmaxf(t + cb[s], 0.f);
            }
        }
    }
}

__device__ __forceinline__ float block_sum(float v, float* red) {   // blockDim.x == 256
    v = wave_sum64(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// g[b][s] = fcb[s] + sum_p fcw[s][p] * h[b][s][p];  grid (S, B)
__global__ __launch_bounds__(256) void head_fc_kernel(const float* __restrict__ h, int P, int S,
                                                      const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                      float* __restrict__ g) {
    __shared__ float red[4];
    const int s = blockIdx.x, b = blockIdx.y;
    const float* hr = h + ((size_t)b * S + s) * P;
    const float* wr = fcw + (size_t)s * P;
    // eight independent chains, sixteen loads in flight per thread: the scalar models' 20049-pixel rows took 78 dependent
    // round trips per thread (29.5 us for 16 workgroups)
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int p = threadIdx.x;
    for (; p + 7 * 256 < P; p += 8 * 256) {
        float hv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { hv[u] = hr[p + u * 256]; wv[u] = wr[p + u * 256]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) a8[u] = fmaf(hv[u], wv[u], a8[u]);
    }
    for (; p < P; p += 256) a8[0] = fmaf(hr[p], wr[p], a8[0]);
    float a = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    a = block_sum(a, red);
    if (threadIdx.x == 0) g[b * S + s] = a + fcb[s];
}

// e[b][s][p] = dg[b][s] * fcw[s][p] * (h > 0);  dtrunk[b][p][:] = sum_s e * cw[s][:]
template <int GS>      // lanes per pixel, as in head_conv_kernel
__global__ __launch_bounds__(256) void head_bwd_pixel_kernel(const float* __restrict__ dg, const float* __restrict__ h,
                                                             int P, int C, int S, const float* __restrict__ cw,
                                                             const float* __restrict__ fcw, float* __restrict__ e_out,
                                                             float* __restrict__ dtrunk, int B, float* __restrict__ dfcw,
                                                             float* __restrict__ dfcb) {
    extern __shared__ float w_s[];   // [S][C]
    if (blockIdx.y == (unsigned)B) {
        // the extra row of workgroups: the Linear layer's gradients (dfcw[s][p] = sum_b dg[b][s] * h[b][s][p], dfcb[s] = sum_b
        // dg[b][s]) -- the same inputs, a launch of their own cost the step ~5 us for 1320 sums
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S * P; i += gridDim.x * blockDim.x) {
            const int s = i / P, p = i - s * P;
            float a = 0.f;
            for (int bb = 0; bb < B; ++bb) a = fmaf(dg[bb * S + s], h[((size_t)bb * S + s) * P + p], a);
            dfcw[i] = a;
            if (i < S) {
                float t = 0.f;
                for (int bb = 0; bb < B; ++bb) t += dg[bb * S + i];
                dfcb[i] = t;
            }
        }
        return;
    }
    for (int e = threadIdx.x; e < S * C; e += blockDim.x) w_s[e] = cw[e];
    __syncthreads();
    constexpr int G = 64 / GS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    const int gl = lane % GS, gi = lane / GS;
    const int Q = C / 4;
    for (int p0 = (blockIdx.x * 4 + wave) * G; p0 < P; p0 += gridDim.x * 4 * G) {
        const int p = p0 + gi;
        if (p >= P) continue;                                        // (no barrier in the loop)
        float ev[MAX_STEMS];
#pragma unroll
        for (int s = 0; s < MAX_STEMS; ++s) {
            ev[s] = 0.f;
            if (s < S) {
                const size_t i = ((size_t)b * S + s) * P + p;
                ev[s] = h[i] > 0.f ? dg[b * S + s] * fcw[(size_t)s * P + p] : 0.f;
                if (gl == 0) e_out[i] = ev[s];
            }
        }
        float4* row = reinterpret_cast<float4*>(dtrunk + ((size_t)b * P + p) * C);
        for (int q = gl; q < Q; q += GS) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int s = 0; s < MAX_STEMS; ++s) {
                if (s < S) {
                    const float4 w = reinterpret_cast<const float4*>(w_s + s * C)[q];
                    o.x = fmaf(ev[s], w.x, o.x); o.y = fmaf(ev[s], w.y, o.y);
                    o.z = fmaf(ev[s], w.z, o.z); o.w = fmaf(ev[s], w.w, o.w);
                }
            }
            row[q] = o;
        }
    }
}

// partial[blk][s][c] = sum over the block's (b,p) range of e[b][s][p] * trunk[b][p][c];  partial_b[blk][s] = sum of e
// (the conv-bias gradient: the same e values, already in registers)
__global__ void head_bwd_cw_partial_kernel(const float* __restrict__ e, const float* __restrict__ trunk, int B, int P,
                                           int C, int S, int Q, int R, int64_t ppb, float* __restrict__ partial,
                                           float* __restrict__ partial_b) {
    extern __shared__ float sm[];    // [R][S][C]
    const int cq = threadIdx.x % Q, pr = threadIdx.x / Q;
    const int64_t BP = (int64_t)B * P;
    const int64_t lo = blockIdx.x * ppb, hi = (lo + ppb < BP) ? lo + ppb : BP;
    float acc[MAX_STEMS][4], esum[MAX_STEMS];
#pragma unroll
    for (int s = 0; s < MAX_STEMS; ++s) { acc[s][0] = acc[s][1] = acc[s][2] = acc[s][3] = 0.f; esum[s] = 0.f; }
    for (int64_t bp = lo + pr; bp < hi; bp += R) {
        const int b = (int)(bp / P), p = (int)(bp - (int64_t)b * P);
        const float4 v = *reinterpret_cast<const float4*>(trunk + bp * C + cq * 4);
#pragma unroll
        for (int s = 0; s < MAX_STEMS; ++s) {
            if (s < S) {
                const float ev = e[((size_t)b * S + s) * P + p];
                esum[s] += ev;
                acc[s][0] = fmaf(ev, v.x, acc[s][0]); acc[s][1] = fmaf(ev, v.y, acc[s][1]);
                acc[s][2] = fmaf(ev, v.z, acc[s][2]); acc[s][3] = fmaf(ev, v.w, acc[s][3]);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < MAX_STEMS; ++s)
        if (s < S)
#pragma unroll
            for (int i = 0; i < 4; ++i) sm[((size_t)pr * S + s) * C + cq * 4 + i] = acc[s][i];
    __syncthreads();
    if (pr == 0) {
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = 0.f;
                for (int r = 0; r < R; ++r) a += sm[((size_t)r * S + s) * C + cq * 4 + i];
                partial[((size_t)blockIdx.x * S + s) * C + cq * 4 + i] = a;
            }
    }
    __syncthreads();                                   // second use of sm: [R][S] sums of e (rows pr, column quad 0 only)
    if (cq == 0)
        for (int s = 0; s < S; ++s) sm[pr * S + s] = esum[s];
    __syncthreads();
    if (threadIdx.x < S) {
        float a = 0.f;
        for (int r = 0; r < R; ++r) a += sm[r * S + threadIdx.x];
        partial_b[(size_t)blockIdx.x * S + threadIdx.x] = a;
    }
}
// dcw[i] = sum over the blocks' records (i < S*C), dcb[s] likewise (i = S*C + s).  Block = 8 outputs x 32 record lanes: a lane
// adds every 32nd record in double, the 32 subtotals are combined by a fixed tree (deterministic).  (One thread per
// output walking all ~1000 records took 134-258 us on the scalar models.)
__global__ __launch_bounds__(256) void head_bwd_cw_finalize_kernel(const float* __restrict__ partial,
                                                                   const float* __restrict__ partial_b, int parts, int C, int S,
                                                                   float* __restrict__ dcw, float* __restrict__ dcb) {
    __shared__ double sub[32][8];
    const int el = threadIdx.x & 7, ys = threadIdx.x >> 3;
    const int i = blockIdx.x * 8 + el, n = S * C;
    double a = 0;
    if (i < n) {
        for (int p = ys; p < parts; p += 32) a += (double)partial[(size_t)p * n + i];
    } else if (i < n + S) {
        for (int p = ys; p < parts; p += 32) a += (double)partial_b[(size_t)p * S + (i - n)];
    }
    sub[ys][el] = a;
    __syncthreads();
#pragma unroll
    for (int stride = 16; stride >= 1; stride >>= 1) {
        if (ys < stride) sub[ys][el] += sub[ys + stride][el];
        __syncthreads();
    }
    if (ys == 0) {
        if (i < n) dcw[i] = (float)sub[0][el];
        else if (i < n + S) dcb[i - n] = (float)sub[0][el];
    }
}

// masked[b][i] = sum_s g[b][s] * x[b][s][i]
__global__ void masksum_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, int S, int64_t FT,
                                   float* __restrict__ masked) {
    const int b = blockIdx.y;
    float gv[MAX_STEMS];
#pragma unroll
    for (int s = 0; s < MAX_STEMS; ++s) gv[s] = s < S ? g[b * S + s] : 0.f;
    const float* xb = x + (size_t)b * S * FT;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < FT; i += (int64_t)gridDim.x * blockDim.x) {
        float a = 0.f;
#pragma unroll
        for (int s = 0; s < MAX_STEMS; ++s)
            if (s < S) a = fmaf(gv[s], xb[(size_t)s * FT + i], a);
        masked[(size_t)b * FT + i] = a;
    }
}

// partial[b][blk][s] = sum_i d[b][i] * x[b][s][i]  where d = dmasked (MODE 0) or (masked - gt) (MODE 1, also sum d^2)
// SC: the stem count at compile time (2, 4, 8: the models of the reference) or 0 = run time (<= MAX_STEMS).  With SC the thread
// requests the S + 1 loads of TWO elements of its stride walk before it touches the first (the kernel is a pure stream of four or
// five dependent round trips per thread otherwise: 16 us for 43 MB); the generic form predicates 16 loads per element and is
// slower that way.  Even plane lengths take 8-byte loads (two neighbouring elements per lane).
template <int MODE, int SC>
__global__ __launch_bounds__(256) void masksum_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                  const float* __restrict__ d_or_gt, int S_rt, int64_t FT,
                                                                  float* __restrict__ masked_out,
                                                                  float* __restrict__ partial /* [B][nblk][S+1] */) {
    constexpr int SM = SC ? SC : MAX_STEMS, U = SC ? 2 : 1;
    const int S = SC ? SC : S_rt;
    const int b = blockIdx.y;
    float gv[SM], acc[SM + 1];
#pragma unroll
    for (int s = 0; s < SM; ++s) { gv[s] = (MODE == 1 && s < S) ? g[b * S + s] : 0.f; acc[s] = 0.f; }
    acc[SM] = 0.f;
    const float* xb = x + (size_t)b * S * FT;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    if (SC && (FT & 1) == 0) {
        // even plane length (every plane then starts 8-byte aligned): 8-byte loads, two elements per lane and request
        const int64_t F2 = FT >> 1;
        const float2* tb = reinterpret_cast<const float2*>(d_or_gt + (size_t)b * FT);
        for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i0 < F2; i0 += step * U) {
            float2 xv[U][SM], tv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * step;
                const bool ok = i < F2;
#pragma unroll
                for (int s = 0; s < SM; ++s)
                    xv[u][s] = ok ? reinterpret_cast<const float2*>(xb + (size_t)s * FT)[i] : make_float2(0.f, 0.f);
                tv[u] = ok ? tb[i] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * step;
                if (i < F2) {
                    float2 d;
                    if (MODE == 1) {
                        float2 m = make_float2(0.f, 0.f);
#pragma unroll
                        for (int s = 0; s < SM; ++s) { m.x = fmaf(gv[s], xv[u][s].x, m.x); m.y = fmaf(gv[s], xv[u][s].y, m.y); }
                        if (masked_out) reinterpret_cast<float2*>(masked_out + (size_t)b * FT)[i] = m;
                        d = make_float2(m.x - tv[u].x, m.y - tv[u].y);
                        acc[SM] = fmaf(d.x, d.x, acc[SM]);
                        acc[SM] = fmaf(d.y, d.y, acc[SM]);
                    } else {
                        d = tv[u];
                    }
#pragma unroll
                    for (int s = 0; s < SM; ++s) { acc[s] = fmaf(d.x, xv[u][s].x, acc[s]); acc[s] = fmaf(d.y, xv[u][s].y, acc[s]); }
                }
            }
        }
    } else {
        for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i0 < FT; i0 += step * U) {
            float xv[U][SM], tv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * step;
                const bool ok = i < FT;
#pragma unroll
                for (int s = 0; s < SM; ++s) xv[u][s] = (ok && s < S) ? xb[(size_t)s * FT + i] : 0.f;
                tv[u] = ok ? d_or_gt[(size_t)b * FT + i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * step;
                if (i < FT) {
                    float d;
                    if (MODE == 1) {
                        float m = 0.f;
#pragma unroll
                        for (int s = 0; s < SM; ++s) m = fmaf(gv[s], xv[u][s], m);
                        if (masked_out) masked_out[(size_t)b * FT + i] = m;
                        d = m - tv[u];
                        acc[SM] = fmaf(d, d, acc[SM]);
                    } else {
                        d = tv[u];
                    }
#pragma unroll
                    for (int s = 0; s < SM; ++s) acc[s] = fmaf(d, xv[u][s], acc[s]);
                }
            }
        }
    }
    // all S + 1 sums in ONE pass: wave sums by shuffles, one barrier, thread s adds the four waves' values (nine block_sum calls
    // in a row were eighteen barriers)
    __shared__ float redw[4][MAX_STEMS + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k <= SM; ++k) {
        const float t = wave_sum64(acc[k]);
        if (lane == 0) redw[wave][k == SM ? MAX_STEMS : k] = t;
    }
    __syncthreads();
    float* out = partial + ((size_t)b * gridDim.x + blockIdx.x) * (S + 1);
    if ((int)threadIdx.x <= S) {
        const int k = (int)threadIdx.x == S ? MAX_STEMS : (int)threadIdx.x;
        out[threadIdx.x] = redw[0][k] + redw[1][k] + redw[2][k] + redw[3][k];
    }
}
// dg[b][s] = scale * sum_blk partial;  loss = sum of squared error / count   (MODE 1).  One wave per (b, s), last block: loss.
__global__ __launch_bounds__(64) void masksum_bwd_finalize_kernel(const float* __restrict__ partial, int B, int nblk, int S,
                                                                  double scale, double inv_count, float* __restrict__ dg,
                                                                  float* __restrict__ loss) {
    const int i = blockIdx.x, lane = threadIdx.x;
    double a = 0;
    if (i < B * S) {
        const int b = i / S, s = i - b * S;
        for (int k = lane; k < nblk; k += 64) a += partial[((size_t)b * nblk + k) * (S + 1) + s];
    } else {
        for (int k = lane; k < B * nblk; k += 64) a += partial[(size_t)k * (S + 1) + S];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off);
    if (lane != 0) return;
    if (i < B * S) dg[i] = (float)(a * scale);
    else if (loss) *loss = (float)(a * inv_count);
}

}  // namespace
}  // namespace dam

using namespace dam;

extern "C" int dam_heads_fwd_f32(const float* trunk, int B, int P, int C, int S, const float* conv_w, const float* conv_b,
                                 const float* fc_w, const float* fc_b, float* h, float* gains, void* stream) {
    if (!trunk || !conv_w || !conv_b || !fc_w || !fc_b || !h || !gains || B <= 0 || P <= 0) return DAM_ERR_BAD_ARG;
    if (C % 4 || S < 1 || S > MAX_STEMS || B > 65535) return DAM_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int Q = C / 4, gs = Q <= 16 ? 16 : Q <= 32 ? 32 : 64, ppw = 4 * (64 / gs);        // pixels per workgroup and round
    const int gx = (int)(cdiv(P, ppw) < 1024 ? cdiv(P, ppw) : 1024);
    if (gs == 16) hipLaunchKernelGGL(head_conv_kernel<16>, dim3(gx, B), dim3(256), (size_t)S * C * sizeof(float), st, trunk, P, C, S, conv_w, conv_b, h);
    else if (gs == 32) hipLaunchKernelGGL(head_conv_kernel<32>, dim3(gx, B), dim3(256), (size_t)S * C * sizeof(float), st, trunk, P, C, S, conv_w, conv_b, h);
    else hipLaunchKernelGGL(head_conv_kernel<64>, dim3(gx, B), dim3(256), (size_t)S * C * sizeof(float), st, trunk, P, C, S, conv_w, conv_b, h);
    DAM_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_fc_kernel, dim3(S, B), dim3(256), 0, st, h, P, S, fc_w, fc_b, gains);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int64_t dam_heads_bwd_workspace_floats(int B, int P, int C, int S) {
    return (int64_t)B * S * P + (int64_t)1024 * S * C + (int64_t)1024 * S;     // e, weight-grad records, bias-grad records
}

extern "C" int dam_heads_bwd_f32(const float* dgains, const float* h, const float* trunk, int B, int P, int C, int S,
                                 const float* conv_w, const float* fc_w, float* dtrunk, float* dconv_w, float* dconv_b,
                                 float* dfc_w, float* dfc_b, float* workspace, void* stream) {
    if (!dgains || !h || !trunk || !conv_w || !fc_w || !dtrunk || !dconv_w || !dconv_b || !dfc_w || !dfc_b || !workspace)
        return DAM_ERR_BAD_ARG;
    if (C % 16 || S < 1 || S > MAX_STEMS || B > 65534 || C > 1024) return DAM_ERR_UNSUPPORTED;   // (grid y = B + 1)
    hipStream_t st = (hipStream_t)stream;
    float* e = workspace;
    float* partial = workspace + (size_t)B * S * P;
    float* partial_b = partial + (size_t)1024 * S * C;
    const int Qp = C / 4, gs = Qp <= 16 ? 16 : Qp <= 32 ? 32 : 64, ppw = 4 * (64 / gs);
    const int gx = (int)(cdiv(P, ppw) < 1024 ? cdiv(P, ppw) : 1024);
#define DAM_HEAD_PIXEL(GS_)                                                                                                     \
    hipLaunchKernelGGL(head_bwd_pixel_kernel<GS_>, dim3(gx, B + 1), dim3(256), (size_t)S * C * sizeof(float), st, dgains, h, P, C, S, \
                       conv_w, fc_w, e, dtrunk, B, dfc_w, dfc_b)        /* (row B of the grid: the Linear layer's gradients) */
    if (gs == 16) DAM_HEAD_PIXEL(16); else if (gs == 32) DAM_HEAD_PIXEL(32); else DAM_HEAD_PIXEL(64);
#undef DAM_HEAD_PIXEL
    DAM_CHECK_LAUNCH();
    const int Q = C / 4;
    int R = 256 / Q; if (R < 1) R = 1;
    while ((size_t)R * S * C * sizeof(float) > 60 * 1024 && R > 1) R >>= 1;
    const int64_t BP = (int64_t)B * P;
    // ~2 pixels per thread row: the loop is a chain of dependent small loads (8 pixels per row on the 1320-pixel trunk of the
    // 130-frame ResNet: 42 workgroups, 21 us), so more, shorter workgroups are faster even though the finalize reads more records
    int64_t parts = cdiv(BP, (int64_t)R * 2); if (parts > 1024) parts = 1024;
    const int64_t ppb = cdiv(BP, parts);
    parts = cdiv(BP, ppb);
    hipLaunchKernelGGL(head_bwd_cw_partial_kernel, dim3((unsigned)parts), dim3(Q * R), (size_t)R * S * C * sizeof(float), st, e,
                       trunk, B, P, C, S, Q, R, ppb, partial, partial_b);
    DAM_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_bwd_cw_finalize_kernel, dim3((unsigned)cdiv((int64_t)S * C + S, 8)), dim3(256), 0, st, partial, partial_b,
                       (int)parts, C, S, dconv_w, dconv_b);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_masksum_fwd_f32(const float* x, const float* gains, int B, int S, int64_t FT, float* masked, void* stream) {
    if (!x || !gains || !masked || B <= 0 || FT <= 0) return DAM_ERR_BAD_ARG;
    if (S < 1 || S > MAX_STEMS || B > 65535) return DAM_ERR_UNSUPPORTED;
    const int gx = (int)(cdiv(FT, 256) < 512 ? cdiv(FT, 256) : 512);
    hipLaunchKernelGGL(masksum_fwd_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, gains, S, FT, masked);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int64_t dam_masksum_workspace_floats(int B, int S) { return (int64_t)B * 256 * (S + 1); }

extern "C" int dam_masksum_bwd_f32(const float* dmasked, const float* x, int B, int S, int64_t FT, float* dgains,
                                   float* workspace, void* stream) {
    if (!dmasked || !x || !dgains || !workspace || B <= 0 || FT <= 0) return DAM_ERR_BAD_ARG;
    if (S < 1 || S > MAX_STEMS || B > 65535) return DAM_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int gx = (int)(cdiv(FT, 1024) < 256 ? cdiv(FT, 1024) : 256);
#define DAM_MASKSUM_GO(MODE_, G_, D_, M_)                                                                                       \
    do {                                                                                                                        \
        if (S == 8) hipLaunchKernelGGL((masksum_bwd_partial_kernel<MODE_, 8>), dim3(gx, B), dim3(256), 0, st, x, G_, D_, S, FT, M_, workspace);      \
        else if (S == 4) hipLaunchKernelGGL((masksum_bwd_partial_kernel<MODE_, 4>), dim3(gx, B), dim3(256), 0, st, x, G_, D_, S, FT, M_, workspace); \
        else if (S == 2) hipLaunchKernelGGL((masksum_bwd_partial_kernel<MODE_, 2>), dim3(gx, B), dim3(256), 0, st, x, G_, D_, S, FT, M_, workspace); \
        else hipLaunchKernelGGL((masksum_bwd_partial_kernel<MODE_, 0>), dim3(gx, B), dim3(256), 0, st, x, G_, D_, S, FT, M_, workspace);             \
    } while (0)
    DAM_MASKSUM_GO(0, (const float*)nullptr, dmasked, (float*)nullptr);
    DAM_CHECK_LAUNCH();
    hipLaunchKernelGGL(masksum_bwd_finalize_kernel, dim3(B * S), dim3(64), 0, st, workspace, B, gx, S, 1.0, 0.0,
                       dgains, (float*)nullptr);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_masksum_mse_f32(const float* x, const float* gains, const float* gt, int B, int S, int64_t FT,
                                   float* masked, float* loss, float* dgains, float* workspace, void* stream) {
    if (!x || !gains || !gt || !loss || !dgains || !workspace || B <= 0 || FT <= 0) return DAM_ERR_BAD_ARG;
    if (S < 1 || S > MAX_STEMS || B > 65535) return DAM_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int gx = (int)(cdiv(FT, 1024) < 256 ? cdiv(FT, 1024) : 256);
    DAM_MASKSUM_GO(1, gains, gt, masked);
    DAM_CHECK_LAUNCH();
    const double count = (double)B * (double)FT;
    hipLaunchKernelGGL(masksum_bwd_finalize_kernel, dim3(B * S + 1), dim3(64), 0, st, workspace, B, gx, S,
                       2.0 / count, 1.0 / count, dgains, loss);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}
