// dam_stft.hip -- STFT -> |.| -> dB front-end for gfx950 (MI355X).
//
// Replaces data/dataset.py:132-162 (compute_features), :181-183 (_stereo_to_mono),
// :164-168 (_augment_audio) of the reference, for every track of a batch in one launch.
//
// Mapping (wave64, no port of a warp-32 design):
//   * one workgroup = 4 waves = one tile of TF=16 consecutive frames of one track;
//   * one WAVE owns one frame at a time: the 2048 real samples are packed as 1024
//     complex points z[n] = x[2n] + i x[2n+1]; lane j keeps the 16 points z[j + 64*n1]
//     in registers, read with 16 coalesced 16-byte loads straight from the interleaved
//     stereo PCM (one load = L0 R0 L1 R1 = one complex point after the channel mean).
//     The 50 % frame overlap is served by L2 (hop = n_fft/2 -> every sample is wanted by
//     two frames of the same workgroup), so HBM sees each sample once;
//   * 1024-point complex FFT = radix-16 (registers) x radix-16 (registers) x radix-4
//     with three exchanges through a private 8.5 KB LDS scratch per wave (padded rows,
//     conflict-free ds_read_b64); window, all twiddles of the three stages and of the
//     real-FFT split are lane-invariant, so they live in registers for the whole kernel;
//   * split post-pass produces bins k and 1024-k from Z[k], Z[1024-k], then
//     20*log10(max(|X|, amin)) goes into an XOR-swizzled [1025][16] LDS tile;
//   * the tile is written out with T contiguous (64-byte row segments) in the
//     reference layout out[track][bin][frame]; optional per-frame max-abs normalise.
#include "dam_common.h"

namespace dam {
namespace {

constexpr int NFFT = 2048;
constexpr int NCPX = NFFT / 2;       // complex points per frame
constexpr int NBINS = NFFT / 2 + 1;  // 1025
constexpr int TF = 16;               // frames per workgroup
constexpr int STFT_WAVES = 4;
constexpr int ROW = 68;              // float2 per scratch row (64 + 4 pad: 8-dword bank shift per row)
constexpr int SCRATCH = 16 * ROW;    // float2 per wave (>= 1024)

__device__ __forceinline__ void radix4(float2& a0, float2& a1, float2& a2, float2& a3) {
    float2 s0 = cadd(a0, a2), s1 = csub(a0, a2), s2 = cadd(a1, a3), s3 = csub(a1, a3);
    a0 = cadd(s0, s2);
    a2 = csub(s0, s2);
    a1 = make_float2(s1.x + s3.y, s1.y - s3.x);   // s1 + (-i) s3
    a3 = make_float2(s1.x - s3.y, s1.y + s3.x);   // s1 + (+i) s3
}

// In-register 16-point forward DFT.  Input v[n]; output X[k] is left in v[4*(k&3) + (k>>2)].
__device__ __forceinline__ void fft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f;   // cos/sin(pi/8)
    constexpr float C2 = 0.70710678118654752f;                             // cos(pi/4)
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) radix4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    // v[4*k1 + n2] *= W16^(n2*k1),  W16^m = (cos(2 pi m/16), -sin(2 pi m/16))
    v[5] = cmul(v[5], make_float2(C1, -S1));     // m = 1
    v[6] = cmul(v[6], make_float2(C2, -C2));     // m = 2
    v[7] = cmul(v[7], make_float2(S1, -C1));     // m = 3
    v[9] = cmul(v[9], make_float2(C2, -C2));     // m = 2
    v[10] = make_float2(v[10].y, -v[10].x);      // m = 4 : * (-i)
    v[11] = cmul(v[11], make_float2(-C2, -C2));  // m = 6
    v[13] = cmul(v[13], make_float2(S1, -C1));   // m = 3
    v[14] = cmul(v[14], make_float2(-C2, -C2));  // m = 6
    v[15] = cmul(v[15], make_float2(-C1, S1));   // m = 9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) radix4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}
__device__ __forceinline__ constexpr int fft16_pos(int k) { return 4 * (k & 3) + (k >> 2); }

// Channel layouts: interleaved (PLANAR = false: sample p of channel c at trk[CH*p + c], what soundfile / a WAV decoder
// hands over) and planar (PLANAR = true: trk[c*cs + p], the [channels, n] arrays inference_utils.py works on).
template <typename PCM, int CH, bool PLANAR>
__device__ __forceinline__ float mono_at(const PCM* __restrict__ trk, int64_t cs, int64_t p) {
    if (CH == 1) return (float)trk[p];
    if (PLANAR) return (float)((trk[p] + trk[cs + p]) * (PCM)0.5);
    return (float)((trk[2 * p] + trk[2 * p + 1]) * (PCM)0.5);
}

typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));

// Loads complex point (x[p], x[p+1]) of the reflect-padded mono signal; interior = no mirroring.
template <typename PCM, int CH, bool PLANAR>
__device__ __forceinline__ float2 load_pair_interior(const PCM* __restrict__ trk, int64_t cs, int64_t p) {
    if constexpr (CH == 2 && sizeof(PCM) == 4 && !PLANAR) {
        f32x4_u q = *reinterpret_cast<const f32x4_u*>(trk + 2 * p);
        return make_float2((q.x + q.y) * 0.5f, (q.z + q.w) * 0.5f);
    } else if constexpr (CH == 2 && sizeof(PCM) == 4 && PLANAR) {
        f32x2_u a = *reinterpret_cast<const f32x2_u*>(trk + p), b = *reinterpret_cast<const f32x2_u*>(trk + cs + p);
        return make_float2((a.x + b.x) * 0.5f, (a.y + b.y) * 0.5f);
    } else if constexpr (CH == 1 && sizeof(PCM) == 4) {
        f32x2_u q = *reinterpret_cast<const f32x2_u*>(trk + p);
        return make_float2(q.x, q.y);
    } else {
        return make_float2(mono_at<PCM, CH, PLANAR>(trk, cs, p), mono_at<PCM, CH, PLANAR>(trk, cs, p + 1));
    }
}
__device__ __forceinline__ int64_t reflect(int64_t p, int64_t n) {
    p = p < 0 ? -p : p;
    return p >= n ? 2 * (n - 1) - p : p;
}

// 20*log10(max(m, amin)); the floor is passed in (computed in double on the host) so that
// silence maps to exactly 20*log10(amin) = -100 dB as in the reference.
__device__ __forceinline__ float to_db(float m, float amin, float floor_db) {
    return m <= amin ? floor_db : 20.0f * log10f(m);
}

template <typename PCM, int CH, bool PLANAR>
__global__ __launch_bounds__(STFT_WAVES* WAVE) void stft_logmag_kernel(
    const PCM* __restrict__ pcm, int64_t n_samples, int64_t outer_stride, int n_inner, int64_t inner_stride, int64_t cs,
    const float* __restrict__ window,
    const float2* __restrict__ tw /* W_2048^k */, const float* __restrict__ gain, int hop, int n_frames,
    float amin, float floor_db, int normalize, float* __restrict__ out, float* __restrict__ out_tail, int n_tail) {
    __shared__ __attribute__((aligned(16))) float2 scratch_all[STFT_WAVES * SCRATCH];
    __shared__ float tile[NBINS * TF];
    __shared__ float colmax[TF * TF];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t track = blockIdx.y;
    const int t0 = blockIdx.x * TF;
    const PCM* trk = pcm + (track / n_inner) * outer_stride + (track % n_inner) * inner_stride;
    float2* S = scratch_all + wave * SCRATCH;
    const float g = gain ? gain[track] : 1.0f;

    // lane-invariant tables -> registers
    float2 win[16], tw1[16], tw2[16], tw3[8];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int n = lane + 64 * n1;
        win[n1] = make_float2(window[2 * n] * g, window[2 * n + 1] * g);
        tw1[n1] = tw[(2 * lane * n1) & (NFFT - 1)];          // W_1024^(lane*k1)
        tw2[n1] = tw[(32 * (lane & 3) * n1) & (NFFT - 1)];   // W_64^(b*c)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) tw3[i] = tw[lane + 64 * i];  // W_2048^k

    // float32 input: the raw samples of the wave's NEXT frame are requested before the FFT of the current one and only
    // converted (channel mean) when that frame's turn comes, so the HBM latency of a frame hides under the previous FFT.
    constexpr bool PREFETCH = sizeof(PCM) == 4;
    typedef float raw_t __attribute__((ext_vector_type(2 * CH), aligned(4)));
    raw_t rawn[PREFETCH ? 16 : 1];
    auto frame_interior = [&](int t) {
        const int64_t p0 = (int64_t)t * hop - NFFT / 2;
        return t < n_frames && p0 >= 0 && p0 + NFFT <= n_samples;
    };
    auto issue_frame = [&](int t) {
        if (PREFETCH && frame_interior(t)) {
            const int64_t p0 = (int64_t)t * hop - NFFT / 2;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int64_t p = p0 + 2 * (lane + 64 * n1);
                if constexpr (PLANAR && CH == 2) {
                    const f32x2_u a = *reinterpret_cast<const f32x2_u*>(trk + p), b = *reinterpret_cast<const f32x2_u*>(trk + cs + p);
                    rawn[n1] = (raw_t){a.x, b.x, a.y, b.y};
                } else {
                    rawn[n1] = *reinterpret_cast<const raw_t*>(trk + (int64_t)CH * p);
                }
            }
        }
    };
    issue_frame(t0 + wave);

    for (int q = 0; q < TF / STFT_WAVES; ++q) {
        const int tl = q * STFT_WAVES + wave;
        const int t = t0 + tl;
        if (t >= n_frames) break;   // wave-uniform; no workgroup barrier inside this loop
        const int64_t p0 = (int64_t)t * hop - NFFT / 2;
        float2 v[16];
        if (PREFETCH && frame_interior(t)) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                if constexpr (CH == 2) v[n1] = make_float2((rawn[n1].x + rawn[n1].y) * 0.5f, (rawn[n1].z + rawn[n1].w) * 0.5f);
                else v[n1] = make_float2(rawn[n1].x, rawn[n1].y);
            }
        } else if (p0 >= 0 && p0 + NFFT <= n_samples) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) v[n1] = load_pair_interior<PCM, CH, PLANAR>(trk, cs, p0 + 2 * (lane + 64 * n1));
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int64_t p = p0 + 2 * (lane + 64 * n1);
                v[n1] = make_float2(mono_at<PCM, CH, PLANAR>(trk, cs, reflect(p, n_samples)),
                                    mono_at<PCM, CH, PLANAR>(trk, cs, reflect(p + 1, n_samples)));
            }
        }
        if (q + 1 < TF / STFT_WAVES) issue_frame(t + STFT_WAVES);     // in flight during this frame's FFT
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = make_float2(v[n1].x * win[n1].x, v[n1].y * win[n1].y);

        // stage 1: DFT16 over n1 (n = 64 n1 + lane), twiddle W_1024^(lane k1), scatter A[k1][lane]
        fft16(v);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            float2 a = v[fft16_pos(k1)];
            if (k1) a = cmul(a, tw1[k1]);
            S[k1 * ROW + lane] = a;
        }
        wave_lds_sync();
        // stage 2: lane = (k1, b); DFT16 over a (n2 = 4a + b), twiddle W_64^(b c)
        {
            const int k1 = lane >> 2, b = lane & 3;
#pragma unroll
            for (int a = 0; a < 16; ++a) v[a] = S[k1 * ROW + 4 * a + b];
            fft16(v);
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float2 x = v[fft16_pos(c)];
                if (c) x = cmul(x, tw2[c]);
                S[k1 * ROW + 4 * c + b] = x;
            }
        }
        wave_lds_sync();
        // stage 3: lane handles (k1 = lane&15, c = (lane>>4) + 4i); radix-4 over b; Z[k1 + 16c + 256d]
        {
            const int k1 = lane & 15, cq = lane >> 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = cq + 4 * i;
                const float4 lo = *reinterpret_cast<const float4*>(&S[k1 * ROW + 4 * c]);
                const float4 hi = *reinterpret_cast<const float4*>(&S[k1 * ROW + 4 * c + 2]);
                v[4 * i + 0] = make_float2(lo.x, lo.y);
                v[4 * i + 1] = make_float2(lo.z, lo.w);
                v[4 * i + 2] = make_float2(hi.x, hi.y);
                v[4 * i + 3] = make_float2(hi.z, hi.w);
                radix4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
            }
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int d = 0; d < 4; ++d) S[lane + 64 * i + 256 * d] = v[4 * i + d];
        }
        wave_lds_sync();
        // real-FFT split: bins k and 1024-k from Z[k], Z[1024-k]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = lane + 64 * i;
            const float2 zk = S[k], zn = S[(NCPX - k) & (NCPX - 1)];
            const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
            const float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
            const float2 tt = cmul(tw3[i], o);
            const float2 xa = cadd(e, tt), xb = csub(e, tt);
            const float ma = sqrtf(xa.x * xa.x + xa.y * xa.y), mb = sqrtf(xb.x * xb.x + xb.y * xb.y);
            const int fa = k, fb = NCPX - k;
            tile[fa * TF + (tl ^ (fa & 15))] = to_db(ma, amin, floor_db);
            tile[fb * TF + (tl ^ (fb & 15))] = to_db(mb, amin, floor_db);
        }
        if (lane == 0) {
            const float2 z = S[NCPX / 2];
            const float m = sqrtf(z.x * z.x + z.y * z.y);
            tile[(NCPX / 2) * TF + (tl ^ ((NCPX / 2) & 15))] = to_db(m, amin, floor_db);
        }
        wave_lds_sync();
    }
    __syncthreads();

    // write-out: thread -> (frame tl = tid&15, bin f = tid>>4 + 16*it); rows of 16 frames = 64 B
    const int tl = threadIdx.x & 15, f0 = threadIdx.x >> 4;
    const bool col_ok = (t0 + tl) < n_frames;
    float scale = 1.0f;
    if (normalize) {
        float m = 0.0f;
        if (col_ok)
            for (int f = f0; f < NBINS; f += 16) m = fmaxf(m, fabsf(tile[f * TF + (tl ^ (f & 15))]));
        colmax[f0 * TF + tl] = m;
        __syncthreads();
        m = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, colmax[r * TF + tl]);
        scale = m;
    }
    if (col_ok) {
        // the last n_tail tracks of every outer group go to out_tail (dataset item = (stems, mix), data/dataset.py:207-210)
        const int64_t og = track / n_inner;
        const int ig = (int)(track % n_inner), n_main = n_inner - n_tail;
        float* o = (ig < n_main ? out + ((og * n_main + ig) * NBINS) * (int64_t)n_frames
                                : out_tail + ((og * n_tail + (ig - n_main)) * NBINS) * (int64_t)n_frames) + t0 + tl;
        for (int f = f0; f < NBINS; f += 16) {
            float vdb = tile[f * TF + (tl ^ (f & 15))];
            if (normalize && scale >= 1.17549435e-38f) vdb = vdb / scale;
            o[(int64_t)f * n_frames] = vdb;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Any power-of-two window (data/dataset.py:132-133 exposes window_size / hop_length as parameters; every caller in the
// reference leaves them at 2048 / 1024, which is what the kernel above is built for).  One workgroup = one frame:
// M = n_fft/2 complex points z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1] in LDS, log2(M) autosort (Stockham) radix-2 passes,
// the same real-FFT split and dB epilogue, bins written straight out.  Not tuned: correctness path.
template <typename PCM, int CH, bool PLANAR>
__global__ __launch_bounds__(256) void stft_generic_kernel(
    const PCM* __restrict__ pcm, int64_t n_samples, int64_t outer_stride, int n_inner, int64_t inner_stride, int64_t cs,
    const float* __restrict__ window, const float2* __restrict__ tw /* W_nfft^k */, const float* __restrict__ gain, int n_fft,
    int hop, int n_frames, float amin, float floor_db, int normalize, float* __restrict__ out, float* __restrict__ out_tail,
    int n_tail) {
    extern __shared__ __attribute__((aligned(16))) float2 buf[];      // [2][M]
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int M = n_fft >> 1, nbins = M + 1;
    const int64_t track = blockIdx.y;
    const int t = blockIdx.x;
    const PCM* trk = pcm + (track / n_inner) * outer_stride + (track % n_inner) * inner_stride;
    const float g = gain ? gain[track] : 1.0f;
    const int64_t p0 = (int64_t)t * hop - M;
    for (int n = tid; n < M; n += 256) {
        const int64_t a = reflect(p0 + 2 * n, n_samples), b = reflect(p0 + 2 * n + 1, n_samples);
        buf[n] = make_float2(mono_at<PCM, CH, PLANAR>(trk, cs, a) * (window[2 * n] * g),
                             mono_at<PCM, CH, PLANAR>(trk, cs, b) * (window[2 * n + 1] * g));
    }
    __syncthreads();
    float2* x = buf;
    float2* y = buf + M;
    int sshift = 0;                                   // s = 1 << sshift
    for (int n = M; n > 1; n >>= 1, ++sshift) {
        const int m = n >> 1, s = 1 << sshift;
        for (int e = tid; e < (M >> 1); e += 256) {
            const int p = e >> sshift, q = e & (s - 1);
            const float2 a = x[q + s * p], b = x[q + s * (p + m)];
            const float2 w = tw[(2 * p * s) & (n_fft - 1)];                 // W_n^p = W_nfft^(2 p s)
            y[q + s * (2 * p)] = cadd(a, b);
            y[q + s * (2 * p + 1)] = cmul(csub(a, b), w);
        }
        __syncthreads();
        float2* tmp = x; x = y; y = tmp;
    }
    // real-FFT split + dB; the result of bin f goes to y[] as a float (reusing the other buffer)
    float* db = reinterpret_cast<float*>(y);
    float mx = 0.f;
    for (int k = tid; k <= M; k += 256) {
        const float2 zk = x[k & (M - 1)], zn = x[(M - k) & (M - 1)];
        const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        const float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        const float2 wk = k < n_fft ? tw[k & (n_fft - 1)] : make_float2(1.f, 0.f);
        const float2 xa = cadd(e, cmul(wk, o));
        const float v = to_db(sqrtf(xa.x * xa.x + xa.y * xa.y), amin, floor_db);
        db[k] = v;
        mx = fmaxf(mx, fabsf(v));
    }
    float scale = 1.0f;
    if (normalize) {
        red[tid] = mx;
        __syncthreads();
        for (int st = 128; st >= 1; st >>= 1) {
            if (tid < st) red[tid] = fmaxf(red[tid], red[tid + st]);
            __syncthreads();
        }
        scale = red[0];
    }
    const int64_t og = track / n_inner;
    const int ig = (int)(track % n_inner), n_main = n_inner - n_tail;
    float* o = (ig < n_main ? out + ((og * n_main + ig) * (int64_t)nbins) * n_frames
                            : out_tail + ((og * n_tail + (ig - n_main)) * (int64_t)nbins) * n_frames) + t;
    for (int k = tid; k <= M; k += 256) {
        float v = db[k];
        if (normalize && scale >= 1.17549435e-38f) v = v / scale;
        o[(int64_t)k * n_frames] = v;
    }
}

}  // namespace
}  // namespace dam

extern "C" int64_t dam_stft_twiddle_count(int n_fft) { return n_fft > 0 ? n_fft : 0; }

extern "C" int dam_stft_fill_twiddles_host(int n_fft, float* table_host) {
    if (n_fft <= 0 || !table_host) return DAM_ERR_BAD_ARG;
    const double w = -2.0 * 3.14159265358979323846264338327950288 / (double)n_fft;
    for (int k = 0; k < n_fft; ++k) {
        table_host[2 * k] = (float)cos(w * k);
        table_host[2 * k + 1] = (float)sin(w * k);
    }
    return DAM_OK;
}

extern "C" int dam_stft_logmag_strided_f32(const void* pcm, int pcm_dtype, int64_t n_outer, int64_t outer_stride,
                                           int64_t n_inner, int64_t inner_stride, int64_t n_samples, int channels,
                                           int64_t sample_stride, int64_t channel_stride, const float* window,
                                           const float* twiddles, const float* gain, int n_fft, int hop, float amin,
                                           int normalize, float* out, float* out_tail, int n_tail, void* stream) {
    using namespace dam;
    if (!pcm || !window || !twiddles || !out || n_outer <= 0 || n_inner <= 0 || hop <= 0) return DAM_ERR_BAD_ARG;
    if (n_tail < 0 || n_tail >= n_inner || (n_tail > 0 && !out_tail)) return DAM_ERR_BAD_ARG;
    if (n_samples <= n_fft / 2) return DAM_ERR_BAD_ARG;   // reflect padding needs N > n_fft/2 (torch.stft raises too)
    if (channels != 1 && channels != 2) return DAM_ERR_UNSUPPORTED;
    const bool fast = n_fft == NFFT && !(hop & 1);         // the tuned 2048-point kernel; else any power of two 64..4096
    if (!fast && (n_fft < 64 || n_fft > 4096 || (n_fft & (n_fft - 1)))) return DAM_ERR_UNSUPPORTED;
    if (pcm_dtype != DAM_PCM_F32 && pcm_dtype != DAM_PCM_F64) return DAM_ERR_UNSUPPORTED;
    const int64_t n_tracks = n_outer * n_inner;
    if (n_tracks > 65535 || n_inner > 0x7fffffff) return DAM_ERR_UNSUPPORTED;
    bool planar;
    if (channels == 1) {
        if (sample_stride != 1) return DAM_ERR_UNSUPPORTED;
        planar = false;
    } else if (sample_stride == channels && channel_stride == 1) {
        planar = false;
    } else if (sample_stride == 1 && channel_stride >= n_samples) {
        planar = true;
    } else {
        return DAM_ERR_UNSUPPORTED;
    }
    const int n_frames = (int)(1 + n_samples / hop);
    dim3 grid((unsigned)cdiv(n_frames, TF), (unsigned)n_tracks), block(STFT_WAVES * WAVE);
    hipStream_t s = (hipStream_t)stream;
    const float2* tw = reinterpret_cast<const float2*>(twiddles);
    const float floor_db = (float)(20.0 * log10((double)amin));
    if (!fast) {
        if (n_frames > 0x7fffffff / 2) return DAM_ERR_UNSUPPORTED;
        const dim3 ggrid((unsigned)n_frames, (unsigned)n_tracks);
        const size_t lds = (size_t)n_fft * sizeof(float2);        // two buffers of n_fft/2 complex points
#define DAM_STFT_GENERIC(T, C, P)                                                                                     \
    hipLaunchKernelGGL((stft_generic_kernel<T, C, P>), ggrid, dim3(256), lds, s, (const T*)pcm, n_samples, outer_stride, \
                       (int)n_inner, inner_stride, channel_stride, window, tw, gain, n_fft, hop, n_frames, amin, floor_db, \
                       normalize, out, out_tail, n_tail)
        if (pcm_dtype == DAM_PCM_F32) {
            if (channels == 1) DAM_STFT_GENERIC(float, 1, false);
            else if (planar) DAM_STFT_GENERIC(float, 2, true);
            else DAM_STFT_GENERIC(float, 2, false);
        } else {
            if (channels == 1) DAM_STFT_GENERIC(double, 1, false);
            else if (planar) DAM_STFT_GENERIC(double, 2, true);
            else DAM_STFT_GENERIC(double, 2, false);
        }
#undef DAM_STFT_GENERIC
        DAM_CHECK_LAUNCH();
        return DAM_OK;
    }
#define DAM_STFT_LAUNCH(T, C, P)                                                                              \
    hipLaunchKernelGGL((stft_logmag_kernel<T, C, P>), grid, block, 0, s, (const T*)pcm, n_samples, outer_stride, \
                       (int)n_inner, inner_stride, channel_stride, window, tw, gain, hop, n_frames, amin, floor_db, \
                       normalize, out, out_tail, n_tail)
    if (pcm_dtype == DAM_PCM_F32) {
        if (channels == 1) DAM_STFT_LAUNCH(float, 1, false);
        else if (planar) DAM_STFT_LAUNCH(float, 2, true);
        else DAM_STFT_LAUNCH(float, 2, false);
    } else {
        if (channels == 1) DAM_STFT_LAUNCH(double, 1, false);
        else if (planar) DAM_STFT_LAUNCH(double, 2, true);
        else DAM_STFT_LAUNCH(double, 2, false);
    }
#undef DAM_STFT_LAUNCH
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_stft_logmag_f32(const void* pcm, int pcm_dtype, int64_t n_tracks, int64_t n_samples, int channels,
                                   int64_t pcm_track_stride, const float* window, const float* twiddles,
                                   const float* gain, int n_fft, int hop, float amin, int normalize, float* out,
                                   void* stream) {
    return dam_stft_logmag_strided_f32(pcm, pcm_dtype, n_tracks, pcm_track_stride, 1, 0, n_samples, channels, channels, 1,
                                       window, twiddles, gain, n_fft, hop, amin, normalize, out, nullptr, 0, stream);
}
