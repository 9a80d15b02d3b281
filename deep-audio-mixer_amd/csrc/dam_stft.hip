// dam_stft.hip -- STFT -> |.| -> dB front-end for gfx950 (MI355X).
//
// Replaces data/dataset.py:132-162 (compute_features), :181-183 (_stereo_to_mono),
// :164-168 (_augment_audio) of the reference, for every track of a batch in one launch.
//
// Mapping (wave64, no port of a warp-32 design): one WAVE owns one frame -- the 2048 real samples packed as 1024 complex
// points, 16 per lane, read with coalesced 16-byte loads straight from the interleaved stereo PCM (one load = L0 R0 L1 R1
// = one complex point after the channel mean); the 50 % frame overlap is served by L2, HBM sees each sample once.  A
// workgroup = 8 waves = 8 consecutive frames of one track, whose dB values are transposed through LDS into the reference
// layout out[track][bin][frame].  Details at stft2048_kernel below; other window sizes: stft_generic_kernel.
#include "dam_common.h"

namespace dam {
namespace {

constexpr int NFFT = 2048;
constexpr int NCPX = NFFT / 2;       // complex points per frame
constexpr int NBINS = NFFT / 2 + 1;  // 1025
constexpr int ROW = 68;              // floats per row of a wave's exchange plane (64 + 4 pad: conflict-free column reads)

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
// Integer PCM (DAM_PCM_S16 / DAM_PCM_S32: the samples as the WAV file holds them, data/dataset.py:192-196 reads them through
// soundfile, which divides by 2^(bits-1)): every sample is converted to float exactly as that division rounds it, the
// power-of-two scale 2^-(bits-1) rides on the window * gain product (exact), so the result is bit for bit what the float32
// kernel computes from host-converted samples -- without the host conversion and with half the bytes over PCIe for 16 bit.
template <typename PCM> struct pcm_traits { static constexpr bool integer = false; static constexpr float scale = 1.0f; };
template <> struct pcm_traits<int16_t> { static constexpr bool integer = true; static constexpr float scale = 1.0f / 32768.0f; };
template <> struct pcm_traits<int32_t> { static constexpr bool integer = true; static constexpr float scale = 1.0f / 2147483648.0f; };

template <typename PCM>
__device__ __forceinline__ float mean2(PCM a, PCM b) {
    if constexpr (pcm_traits<PCM>::integer) return ((float)a + (float)b) * 0.5f;
    else return (float)((a + b) * (PCM)0.5);
}

template <typename PCM, int CH, bool PLANAR>
__device__ __forceinline__ float mono_at(const PCM* __restrict__ trk, int64_t cs, int64_t p) {
    if (CH == 1) return (float)trk[p];
    if (PLANAR) return mean2<PCM>(trk[p], trk[cs + p]);
    return mean2<PCM>(trk[2 * p], trk[2 * p + 1]);
}

typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// Loads complex point (x[p], x[p+1]) of the reflect-padded mono signal; interior = no mirroring.
template <typename PCM, int CH, bool PLANAR>
__device__ __forceinline__ float2 load_pair_interior(const PCM* __restrict__ trk, int64_t cs, int64_t p) {
    constexpr bool F32 = sizeof(PCM) == 4 && !pcm_traits<PCM>::integer;
    if constexpr (CH == 2 && F32 && !PLANAR) {
        f32x4_u q = *reinterpret_cast<const f32x4_u*>(trk + 2 * p);
        return make_float2((q.x + q.y) * 0.5f, (q.z + q.w) * 0.5f);
    } else if constexpr (CH == 2 && F32 && PLANAR) {
        f32x2_u a = *reinterpret_cast<const f32x2_u*>(trk + p), b = *reinterpret_cast<const f32x2_u*>(trk + cs + p);
        return make_float2((a.x + b.x) * 0.5f, (a.y + b.y) * 0.5f);
    } else if constexpr (CH == 1 && F32) {
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
// The same from the POWER p = |X|^2: 20*log10(sqrt(p)) = 10*log10(p) -- no square root (a correctly rounded sqrtf is a dozen
// instructions, seventeen times per lane and frame); v_log_f32 is good to 1 ulp of log2(p), i.e. < 1e-5 dB over the whole range.
__device__ __forceinline__ float power_to_db(float p, float amin2, float floor_db) {
    return p <= amin2 ? floor_db : 3.01029995663981195f * __log2f(p);
}

// ---------------------------------------------------------------------------------------------------------------------
// The 2048-point kernel, built for OCCUPANCY.  (Round 1's version kept every lane-invariant table and a prefetched frame in
// registers: 256 VGPR + 68 AGPR and 101 KB of LDS per 4-wave workgroup = one wave per SIMD, nothing to hide an LDS or HBM
// round trip behind: 77 + 27 us for the 72 tracks of a C3 batch.  This one: 127 VGPR, 76 KB, 16 waves per CU, 59 us in one
// launch, of which 26 us are the [bin][frame] write-out in 32-byte pieces -- profiles/README.md.)
//   * radix-16 x radix-16 x radix-4 over the 1024 complex points z[n] = x[2n] + i x[2n+1], lane j of a wave holds
//     z[j + 64 n1]; same twiddles and the same window * gain product as before;
//   * the lane-invariant tables are NOT in the register file: stage twiddles come from a 8.5 KB LDS copy shared by the
//     workgroup, window and split twiddles from L1 (coalesced 8-byte loads); no register prefetch of the next frame --
//     four waves per SIMD hide the latency instead;
//   * the three in-wave exchanges go through ONE float plane per wave (real parts, then imaginary parts: 4.3 KB instead of
//     8.7 KB), the real-FFT split fetches the partner Z[1024-k] through the same plane and every lane turns its own 16 points
//     Z[lane + 64 i + 256 d] into 16 bins (+ Nyquist on lane 0);
//   * a workgroup = 8 waves = 8 consecutive frames of one track; the [8][1025] dB tile is written out as 16-byte pieces
//     (4 frames of one bin); persistent: the workgroup walks tiles blockIdx.x, + gridDim.x, ... so the tables load once.
//   LDS 76 KB per workgroup -> 2 workgroups = 16 waves per CU; <= 128 VGPR.
constexpr int TF2 = 8;                 // frames (= waves) per workgroup tile
constexpr int TROW = NBINS + 3;        // tile row pitch in floats (1028: rows stay 16-byte aligned)
constexpr int PLANE = 16 * ROW;        // floats of one exchange plane (1088 >= 1024)

// indirect: 0 = `pcm` is the batch; 1 = DAM_PCM_INDIRECT, a device word holding its address; 2 = DAM_PCM_ROTATE, a device table
// {address of an int64 step counter, n, offset, addr[0..n)}: the batch is addr[(counter + offset) % n] (include/dam_hip.h)
template <typename PCM>
__device__ __forceinline__ const PCM* resolve_pcm(const PCM* pcm, int indirect) {
    if (indirect == 2) {
        const long long* t = reinterpret_cast<const long long*>(pcm);
        const long long c = *reinterpret_cast<const long long*>(t[0]) + t[2];
        return reinterpret_cast<const PCM*>(t[3 + (c % t[1])]);
    }
    if (indirect) return *reinterpret_cast<const PCM* const*>(pcm);
    return pcm;
}

template <typename PCM, int CH, bool PLANAR>
__global__ __launch_bounds__(TF2* WAVE, 4) void stft2048_kernel(
    const PCM* __restrict__ pcm, int64_t n_samples, int64_t outer_stride, int n_inner, int64_t inner_stride, int64_t cs,
    const float* __restrict__ window, const float2* __restrict__ tw /* W_2048^k */, const float* __restrict__ gain, int hop,
    int n_frames, int tiles_per_track, int n_tiles, float amin, float floor_db, int normalize, float* __restrict__ out,
    float* __restrict__ out_tail, int n_tail, int indirect) {
    // DAM_PCM_INDIRECT: `pcm` is a device word that holds the batch's address (a captured launch then follows whichever
    // resident batch the word points at: no copy into a fixed input buffer); DAM_PCM_ROTATE: a table of batches walked by a
    // device-side step counter
    pcm = resolve_pcm(pcm, indirect);
    __shared__ __attribute__((aligned(16))) float2 tw1s[16 * 64];      // [k1][lane]  W_1024^(lane*k1)
    __shared__ __attribute__((aligned(16))) float2 tw2s[16 * 4];       // [c][b]      W_64^(b*c)
    __shared__ __attribute__((aligned(16))) float planes[TF2 * PLANE];
    __shared__ __attribute__((aligned(16))) float tile[TF2 * TROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 16 * 64; e += TF2 * WAVE) tw1s[e] = tw[(2 * (e & 63) * (e >> 6)) & (NFFT - 1)];
    if (tid < 64) tw2s[tid] = tw[(32 * (tid & 3) * (tid >> 2)) & (NFFT - 1)];
    __syncthreads();
    float* P = planes + wave * PLANE;
    const int n_main = n_inner - n_tail;
    const float amin2 = amin * amin;

    // Tile walk, XCD-aware: workgroup b runs on XCD b % 8 (round-robin dispatch) and every XCD has its own L2.  Each XCD takes
    // one CONTIGUOUS eighth of the tiles and its workgroups walk it side by side, so the tiles that share an output cache line
    // (four consecutive 8-frame tiles of a track: 32-byte pieces of one 128-byte line) and the PCM of overlapping frames meet in
    // ONE L2 at about the same time: partial lines are merged there instead of going out as masked writes, the second read of
    // every sample hits.  (With the plain stride walk, neighbours sat on different XCDs: WRITE_SIZE 1.7 x the output bytes.)
    const int n_xcd = (gridDim.x & 7) == 0 ? 8 : 1;
    const int per_xcd = (n_tiles + n_xcd - 1) / n_xcd, wg_per_xcd = gridDim.x / n_xcd;
    const int tile_lo = (blockIdx.x % n_xcd) * per_xcd;
    for (int v = blockIdx.x / n_xcd; v < per_xcd && tile_lo + v < n_tiles; v += wg_per_xcd) {
        const int tile_i = tile_lo + v;
        const int64_t track = tile_i / tiles_per_track;
        const int t0 = (tile_i - (int)track * tiles_per_track) * TF2;
        const int t = t0 + wave;
        if (t < n_frames) {             // wave-uniform
            const PCM* trk = pcm + (track / n_inner) * outer_stride + (track % n_inner) * inner_stride;
            const float g = (gain ? gain[track] : 1.0f) * pcm_traits<PCM>::scale;     // power-of-two scale: exact
            const int64_t p0 = (int64_t)t * hop - NFFT / 2;
            float2 v[16];
            const bool interior = p0 >= 0 && p0 + NFFT <= n_samples;
            // window (and, for interleaved float32 PCM, the samples) through buffer loads: scalar base + ONE lane offset
            // register + immediates, instead of sixteen 64-bit address pairs each (the register file is what limits occupancy)
            const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(window), 0, NFFT * 4, 0x00020000);
            constexpr bool BUF = sizeof(PCM) <= 4 && !PLANAR;           // float32 / int32 / int16, interleaved or mono
            constexpr int PB = (int)sizeof(PCM) * 2 * CH;               // bytes of one complex point (two frames)
            // two halves of eight points: samples and window of a half are requested together, the compiler must not hoist the
            // second half's loads over the first half's arithmetic (it would spill)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (interior) {
                    if constexpr (BUF) {
                        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
                            const_cast<PCM*>(trk + (int64_t)CH * p0), 0, NFFT * CH * (int)sizeof(PCM), 0x00020000);
#pragma unroll
                        for (int n1 = 8 * half; n1 < 8 * half + 8; ++n1) {
                            if constexpr (PB == 16) {               // 4-byte samples, stereo: L0 R0 L1 R1
                                const auto raw = __builtin_amdgcn_raw_buffer_load_b128(prs, lane * 16, n1 * 1024, 0);
                                if constexpr (pcm_traits<PCM>::integer) {
                                    const i32x4 q = __builtin_bit_cast(i32x4, raw);
                                    v[n1] = make_float2(mean2<int32_t>(q.x, q.y), mean2<int32_t>(q.z, q.w));
                                } else {
                                    const f32x4 q = __builtin_bit_cast(f32x4, raw);
                                    v[n1] = make_float2((q.x + q.y) * 0.5f, (q.z + q.w) * 0.5f);
                                }
                            } else if constexpr (PB == 8 && sizeof(PCM) == 4) {      // 4-byte samples, mono
                                const auto raw = __builtin_amdgcn_raw_buffer_load_b64(prs, lane * 8, n1 * 512, 0);
                                if constexpr (pcm_traits<PCM>::integer) {
                                    const i32x2 q = __builtin_bit_cast(i32x2, raw);
                                    v[n1] = make_float2((float)q.x, (float)q.y);
                                } else {
                                    const f32x2 q = __builtin_bit_cast(f32x2, raw);
                                    v[n1] = make_float2(q.x, q.y);
                                }
                            } else if constexpr (PB == 8) {         // int16 stereo: one 8-byte load = L0 R0 L1 R1
                                const i32x2 q = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(prs, lane * 8, n1 * 512, 0));
                                v[n1] = make_float2(mean2<int32_t>((int)(short)(q.x & 0xffff), q.x >> 16),
                                                    mean2<int32_t>((int)(short)(q.y & 0xffff), q.y >> 16));
                            } else {                                // int16 mono: one 4-byte load = two samples
                                const int q = __builtin_amdgcn_raw_buffer_load_b32(prs, lane * 4, n1 * 256, 0);
                                v[n1] = make_float2((float)(int)(short)(q & 0xffff), (float)(q >> 16));
                            }
                        }
                    } else {
#pragma unroll
                        for (int n1 = 8 * half; n1 < 8 * half + 8; ++n1)
                            v[n1] = load_pair_interior<PCM, CH, PLANAR>(trk, cs, p0 + 2 * (lane + 64 * n1));
                    }
                } else {
#pragma unroll
                    for (int n1 = 8 * half; n1 < 8 * half + 8; ++n1) {
                        const int64_t p = p0 + 2 * (lane + 64 * n1);
                        v[n1] = make_float2(mono_at<PCM, CH, PLANAR>(trk, cs, reflect(p, n_samples)),
                                            mono_at<PCM, CH, PLANAR>(trk, cs, reflect(p + 1, n_samples)));
                    }
                }
#pragma unroll
                for (int n1 = 8 * half; n1 < 8 * half + 8; ++n1) {
                    const f32x2 w = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(wrs, lane * 8, n1 * 512, 0));
                    v[n1] = make_float2(v[n1].x * (w.x * g), v[n1].y * (w.y * g));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // stage 1: DFT16 over n1 (n = 64 n1 + lane), twiddle W_1024^(lane k1), exchange A[k1][lane] -> lane (k1, b)
            fft16(v);
            float re[16];
            {
                float2 a[16];
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4) {          // four twiddles in flight at a time (register pressure)
#pragma unroll
                    for (int k1 = k0; k1 < k0 + 4; ++k1) {
                        a[k1] = v[fft16_pos(k1)];
                        if (k1) a[k1] = cmul(a[k1], tw1s[k1 * 64 + lane]);
                        P[k1 * ROW + lane] = a[k1].x;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                wave_lds_order();
#pragma unroll
                for (int q = 0; q < 16; ++q) re[q] = P[(lane >> 2) * ROW + 4 * q + (lane & 3)];
                wave_lds_order();
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) P[k1 * ROW + lane] = a[k1].y;
                wave_lds_order();
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = make_float2(re[q], P[(lane >> 2) * ROW + 4 * q + (lane & 3)]);
                wave_lds_order();
            }
            // stage 2: lane = (k1, b); DFT16 over a (n2 = 4a + b), twiddle W_64^(b c); exchange -> lane (k1 = lane&15, cq)
            fft16(v);
            {
                const int k1 = lane >> 2, b = lane & 3;
                float2 x[16];
#pragma unroll
                for (int c0 = 0; c0 < 16; c0 += 4) {
#pragma unroll
                    for (int c = c0; c < c0 + 4; ++c) {
                        x[c] = v[fft16_pos(c)];
                        if (c) x[c] = cmul(x[c], tw2s[c * 4 + b]);
                        P[k1 * ROW + 4 * c + b] = x[c].x;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                wave_lds_order();
                float4 r4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) r4[i] = *reinterpret_cast<const float4*>(&P[(lane & 15) * ROW + 4 * ((lane >> 4) + 4 * i)]);
                wave_lds_order();
#pragma unroll
                for (int c = 0; c < 16; ++c) P[k1 * ROW + 4 * c + b] = x[c].y;
                wave_lds_order();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 i4 = *reinterpret_cast<const float4*>(&P[(lane & 15) * ROW + 4 * ((lane >> 4) + 4 * i)]);
                    v[4 * i + 0] = make_float2(r4[i].x, i4.x);
                    v[4 * i + 1] = make_float2(r4[i].y, i4.y);
                    v[4 * i + 2] = make_float2(r4[i].z, i4.z);
                    v[4 * i + 3] = make_float2(r4[i].w, i4.w);
                }
                wave_lds_order();
            }
            // stage 3: radix-4 over b: v[4i + d] = Z[lane + 64 i + 256 d]
#pragma unroll
            for (int i = 0; i < 4; ++i) radix4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
            // real-FFT split: bin k of every point this lane holds needs Z[(1024 - k) & 1023], fetched through the plane
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int d = 0; d < 4; ++d) P[lane + 64 * i + 256 * d] = v[4 * i + d].x;
            wave_lds_order();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int d = 0; d < 4; ++d) re[4 * i + d] = P[(NCPX - (lane + 64 * i + 256 * d)) & (NCPX - 1)];
            wave_lds_order();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int d = 0; d < 4; ++d) P[lane + 64 * i + 256 * d] = v[4 * i + d].y;
            wave_lds_order();
            float db[16], nyq = 0.f, mx = 0.f;
            const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(tw), 0, NFFT * 8, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_barrier(0);            // four bins at a time: twiddle + partner loads stay short-lived
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int k = lane + 64 * i + 256 * d;
                    const float2 zk = v[4 * i + d], zn = make_float2(re[4 * i + d], P[(NCPX - k) & (NCPX - 1)]);
                    const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
                    const float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
                    const f32x2 wk = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(trs, lane * 8, (64 * i + 256 * d) * 8, 0));
                    const float2 tt = cmul(make_float2(wk.x, wk.y), o);
                    const float2 xa = cadd(e, tt);
#ifdef DAM_STFT_DIAG_NO_LOG
                    db[4 * i + d] = xa.x * xa.x + xa.y * xa.y;
#else
                    db[4 * i + d] = power_to_db(xa.x * xa.x + xa.y * xa.y, amin2, floor_db);
#endif
                    mx = fmaxf(mx, fabsf(db[4 * i + d]));
                    if (i == 0 && d == 0 && lane == 0) {          // k = 0: Nyquist bin X[1024] = e - tt
                        const float2 xb = csub(e, tt);
                        nyq = power_to_db(xb.x * xb.x + xb.y * xb.y, amin2, floor_db);
                    }
                }
            }
            wave_lds_order();                                           // the plane is free again for the next frame
            if (normalize) {        // librosa.util.normalize(features): each frame divided by its max-abs over the bins
                mx = fmaxf(mx, fabsf(__shfl(nyq, 0)));
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
                if (mx >= 1.17549435e-38f) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) db[q] = db[q] / mx;
                    nyq = nyq / mx;
                }
            }
            float* trow = tile + wave * TROW;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int d = 0; d < 4; ++d) trow[lane + 64 * i + 256 * d] = db[4 * i + d];
            if (lane == 0) trow[NCPX] = nyq;
        }
        __syncthreads();
#ifndef DAM_STFT_DIAG_NO_WRITEOUT
        {   // write-out: (bin, half) -> 4 consecutive frames = one 16-byte piece of out[track][bin][t0 + 4 half ...]
            const int64_t og = track / n_inner;
            const int ig = (int)(track % n_inner);
            float* obase = (ig < n_main ? out + ((og * n_main + ig) * NBINS) * (int64_t)n_frames
                                        : out_tail + ((og * n_tail + (ig - n_main)) * NBINS) * (int64_t)n_frames) + t0;
            for (int e = tid; e < NBINS * 2; e += TF2 * WAVE) {
                const int f = e >> 1, h = e & 1;
                const int left = n_frames - (t0 + 4 * h);                  // frames of this piece inside the track
                if (left <= 0) continue;
                f32x4_u q;
                q.x = tile[(4 * h + 0) * TROW + f]; q.y = tile[(4 * h + 1) * TROW + f];
                q.z = tile[(4 * h + 2) * TROW + f]; q.w = tile[(4 * h + 3) * TROW + f];
                float* o = obase + (int64_t)f * n_frames + 4 * h;
                if (left >= 4) {
                    *reinterpret_cast<f32x4_u*>(o) = q;
                } else {
                    o[0] = q.x;
                    if (left > 1) o[1] = q.y;
                    if (left > 2) o[2] = q.z;
                }
            }
        }
#else
        if (tile_i == 0x7fffffff) out[tid] = tile[tid];
#endif
        __syncthreads();
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
    int n_tail, int indirect) {
    pcm = resolve_pcm(pcm, indirect);
    extern __shared__ __attribute__((aligned(16))) float2 buf[];      // [2][M]
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int M = n_fft >> 1, nbins = M + 1;
    const int64_t track = blockIdx.y;
    const int t = blockIdx.x;
    const PCM* trk = pcm + (track / n_inner) * outer_stride + (track % n_inner) * inner_stride;
    const float g = (gain ? gain[track] : 1.0f) * pcm_traits<PCM>::scale;
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
    bool fast = n_fft == NFFT && !(hop & 1);               // the tuned 2048-point kernel; else any power of two 64..4096
    // 16-bit mono tracks at an odd sample stride start on odd 2-byte boundaries: the tuned kernel's 4-byte point loads would
    // be misaligned, the generic kernel reads sample by sample
    if ((pcm_dtype & ~(DAM_PCM_INDIRECT | DAM_PCM_ROTATE)) == DAM_PCM_S16 && channels == 1 && ((outer_stride | inner_stride) & 1)) fast = false;
    if (!fast && (n_fft < 64 || n_fft > 16384 || (n_fft & (n_fft - 1)))) return DAM_ERR_UNSUPPORTED;    // (two LDS buffers of n_fft/2 points: 128 KB at 16384)
    if ((pcm_dtype & DAM_PCM_INDIRECT) && (pcm_dtype & DAM_PCM_ROTATE)) return DAM_ERR_BAD_ARG;
    const int indirect = (pcm_dtype & DAM_PCM_ROTATE) ? 2 : (pcm_dtype & DAM_PCM_INDIRECT) ? 1 : 0;
    pcm_dtype &= ~(DAM_PCM_INDIRECT | DAM_PCM_ROTATE);
    if (pcm_dtype != DAM_PCM_F32 && pcm_dtype != DAM_PCM_F64 && pcm_dtype != DAM_PCM_S16 && pcm_dtype != DAM_PCM_S32)
        return DAM_ERR_UNSUPPORTED;
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
    const bool integer = pcm_dtype == DAM_PCM_S16 || pcm_dtype == DAM_PCM_S32;
    if (integer && planar) return DAM_ERR_UNSUPPORTED;     // integer PCM is what a WAV decoder hands over: interleaved
    // the 2-byte kernels fetch a complex point (two frames) with one aligned 4- / 8-byte load
    if (pcm_dtype == DAM_PCM_S16 && !indirect && ((uintptr_t)pcm & 3)) return DAM_ERR_BAD_ARG;
    const int n_frames = (int)(1 + n_samples / hop);
    dim3 grid, block;
    hipStream_t s = (hipStream_t)stream;
    const float2* tw = reinterpret_cast<const float2*>(twiddles);
    const float floor_db = (float)(20.0 * log10((double)amin));
    if (!fast) {
        if (n_frames > 0x7fffffff / 2) return DAM_ERR_UNSUPPORTED;
        const dim3 ggrid((unsigned)n_frames, (unsigned)n_tracks);
        const size_t lds = (size_t)n_fft * sizeof(float2);        // two buffers of n_fft/2 complex points
#define DAM_STFT_GENERIC(T, C, P)                                                                                     \
    do {                                                                                                              \
        if (lds > 48 * 1024) {                                     /* 8192 / 16384-point windows: raise the kernel's LDS limit once */ \
            static PerDevice<bool> raised_pd; bool& raised = raised_pd();\
            if (!raised) {                                                                                            \
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_generic_kernel<T, C, P>),                 \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024) != hipSuccess)        \
                    return DAM_ERR_LAUNCH;                                                                            \
                raised = true;                                                                                        \
            }                                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((stft_generic_kernel<T, C, P>), ggrid, dim3(256), lds, s, (const T*)pcm, n_samples, outer_stride, \
                           (int)n_inner, inner_stride, channel_stride, window, tw, gain, n_fft, hop, n_frames, amin, floor_db, \
                           normalize, out, out_tail, n_tail, indirect);                                               \
    } while (0)
        if (pcm_dtype == DAM_PCM_F32) {
            if (channels == 1) DAM_STFT_GENERIC(float, 1, false);
            else if (planar) DAM_STFT_GENERIC(float, 2, true);
            else DAM_STFT_GENERIC(float, 2, false);
        } else if (pcm_dtype == DAM_PCM_S16) {
            if (channels == 1) DAM_STFT_GENERIC(int16_t, 1, false);
            else DAM_STFT_GENERIC(int16_t, 2, false);
        } else if (pcm_dtype == DAM_PCM_S32) {
            if (channels == 1) DAM_STFT_GENERIC(int32_t, 1, false);
            else DAM_STFT_GENERIC(int32_t, 2, false);
        } else {
            if (channels == 1) DAM_STFT_GENERIC(double, 1, false);
            else if (planar) DAM_STFT_GENERIC(double, 2, true);
            else DAM_STFT_GENERIC(double, 2, false);
        }
#undef DAM_STFT_GENERIC
        DAM_CHECK_LAUNCH();
        return DAM_OK;
    }
    const int tiles_per_track = (int)cdiv(n_frames, TF2);
    const int64_t n_tiles64 = (int64_t)tiles_per_track * n_tracks;
    if (n_tiles64 > 0x7fffffff) return DAM_ERR_UNSUPPORTED;
    const int n_tiles = (int)n_tiles64;
    grid = dim3((unsigned)(n_tiles < 512 ? n_tiles : 512));           // persistent: two workgroups per CU
    block = dim3(TF2 * WAVE);
#define DAM_STFT_LAUNCH(T, C, P)                                                                              \
    hipLaunchKernelGGL((stft2048_kernel<T, C, P>), grid, block, 0, s, (const T*)pcm, n_samples, outer_stride, \
                       (int)n_inner, inner_stride, channel_stride, window, tw, gain, hop, n_frames, tiles_per_track, \
                       n_tiles, amin, floor_db, normalize, out, out_tail, n_tail, indirect)
    if (pcm_dtype == DAM_PCM_F32) {
        if (channels == 1) DAM_STFT_LAUNCH(float, 1, false);
        else if (planar) DAM_STFT_LAUNCH(float, 2, true);
        else DAM_STFT_LAUNCH(float, 2, false);
    } else if (pcm_dtype == DAM_PCM_S16) {
        if (channels == 1) DAM_STFT_LAUNCH(int16_t, 1, false);
        else DAM_STFT_LAUNCH(int16_t, 2, false);
    } else if (pcm_dtype == DAM_PCM_S32) {
        if (channels == 1) DAM_STFT_LAUNCH(int32_t, 1, false);
        else DAM_STFT_LAUNCH(int32_t, 2, false);
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
