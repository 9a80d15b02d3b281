// dam_loudness.hip -- ITU-R BS.1770 K-weighting + 400 ms block energies.
//
// Replaces what the reference gets from pyloudnorm (third-party, not vendored; `pyln.Meter(sr).integrated_loudness`):
// data/dataset.py:115-130 (compute_mean_loudness), evaluation.py:39-46,59-66 (per-stem and mix loudness),
// models/baselines/mean_loudness_model.py:10-20.  The device part is everything that touches samples: the two-biquad
// K-weighting filter (scipy.signal.lfilter, transposed direct form II, float64) and the mean square of every gating
// block.  The gating itself (two thresholds over a few thousand block energies) stays on the host (loudness.py).
//
// An IIR filter is a linear recurrence; it is made parallel exactly, not by warm-up: the signal is cut into chunks,
//   pass 1  every chunk is filtered from a zero state            -> its zero-state end state            (parallel)
//   scan    s_in[c+1] = T s_in[c] + s_end0[c], T = (4x4 state transition)^L obtained by running the zero-input filter
//           from the four unit states                           -> the true state entering every chunk (two-level scan,
//           one workgroup per channel)
//   pass 2  every chunk is filtered again from its true entry state; y^2 is stored                       (parallel)
//   blocks  z[ch][j] = sum y^2 over [lo_j, hi_j) / (hi_j - lo_j nominal length)                          (parallel)
// float64 throughout (the reference filters float64 arrays); HBM-bound.
#include "dam_common.h"

namespace dam {
namespace {

struct KwCoef { double b[2][3], a[2][2]; };      // stage 0 = high shelf, stage 1 = high pass; a0 == 1

// one sample through both stages (transposed direct form II, the same recurrence as scipy's lfilter)
__device__ __forceinline__ double kw_step(const KwCoef& k, double x, double (&s)[4]) {
    const double y0 = k.b[0][0] * x + s[0];
    s[0] = k.b[0][1] * x - k.a[0][0] * y0 + s[1];
    s[1] = k.b[0][2] * x - k.a[0][1] * y0;
    const double y1 = k.b[1][0] * y0 + s[2];
    s[2] = k.b[1][1] * y0 - k.a[1][0] * y1 + s[3];
    s[3] = k.b[1][2] * y0 - k.a[1][1] * y1;
    return y1;
}

template <typename T>
__device__ __forceinline__ double kw_load(const T* x, int64_t n, int64_t sample_stride) { return (double)x[n * sample_stride]; }

// pass 1 (PASS == 1): end state of chunk c filtered from zero;  pass 2: y^2 from the true entry state
template <typename T, int PASS>
__global__ __launch_bounds__(64) void kw_chunk_kernel(const T* __restrict__ x, int64_t n_samples, int64_t sample_stride,
                                                      int64_t channel_stride, KwCoef k, int L, int64_t n_chunks,
                                                      double* __restrict__ state /* [ch][n_chunks][4] */,
                                                      double* __restrict__ ysq /* [ch][n_samples] */) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int ch = blockIdx.y;
    if (c >= n_chunks) return;
    const T* xc = x + ch * channel_stride;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    double* st = state + ((int64_t)ch * n_chunks + c) * 4;
    if (PASS == 2) { s[0] = st[0]; s[1] = st[1]; s[2] = st[2]; s[3] = st[3]; }
    const int64_t lo = c * L, hi = lo + L < n_samples ? lo + L : n_samples;
    for (int64_t n = lo; n < hi; ++n) {
        const double y = kw_step(k, kw_load(xc, n, sample_stride), s);
        if (PASS == 2) ysq[(int64_t)ch * n_samples + n] = y * y;
    }
    if (PASS == 1) { st[0] = s[0]; st[1] = s[1]; st[2] = s[2]; st[3] = s[3]; }
}

// state[c] := true entry state of chunk c (in place: on input state[c] is the zero-state END state of chunk c).
// One workgroup per channel, two levels: thread t owns K consecutive chunks; it first propagates a zero entry state
// through them, thread 0 chains the 256 results with T^K, then every thread replays its chunks from its true entry state.
__global__ __launch_bounds__(256) void kw_scan_kernel(KwCoef k, int L, int64_t n_chunks, double* __restrict__ state) {
    __shared__ double Tm[4][4];          // Tm[r][q]: component r of the state after L zero-input samples from unit state q
    __shared__ double TK[4][4];          // the same after K chunks
    __shared__ double seg[256][4];       // per thread: end state of its K chunks from a zero entry, then its true entry state
    const int ch = blockIdx.x, t = threadIdx.x;
    const int64_t K = (n_chunks + 255) / 256;
    if (t < 4) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        s[t] = 1.0;
        for (int n = 0; n < L; ++n) kw_step(k, 0.0, s);
        for (int r = 0; r < 4; ++r) Tm[r][t] = s[r];
    }
    __syncthreads();
    if (t < 4) {
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        v[t] = 1.0;
        for (int64_t i = 0; i < K; ++i) {
            double w[4];
            for (int r = 0; r < 4; ++r) w[r] = Tm[r][0] * v[0] + Tm[r][1] * v[1] + Tm[r][2] * v[2] + Tm[r][3] * v[3];
            for (int r = 0; r < 4; ++r) v[r] = w[r];
        }
        for (int r = 0; r < 4; ++r) TK[r][t] = v[r];
    }
    double* st = state + (int64_t)ch * n_chunks * 4;
    const int64_t c0 = t * K, c1 = c0 + K < n_chunks ? c0 + K : n_chunks;
    double cur[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t c = c0; c < c1; ++c) {
        double nxt[4];
        for (int r = 0; r < 4; ++r)
            nxt[r] = st[c * 4 + r] + (Tm[r][0] * cur[0] + Tm[r][1] * cur[1] + Tm[r][2] * cur[2] + Tm[r][3] * cur[3]);
        for (int r = 0; r < 4; ++r) cur[r] = nxt[r];
    }
    // threads whose range is short or empty: pad with zero-input chunks so that every segment spans exactly K chunks
    for (int64_t c = (c1 > c0 ? c1 : c0); c < c0 + K; ++c) {
        double nxt[4];
        for (int r = 0; r < 4; ++r) nxt[r] = Tm[r][0] * cur[0] + Tm[r][1] * cur[1] + Tm[r][2] * cur[2] + Tm[r][3] * cur[3];
        for (int r = 0; r < 4; ++r) cur[r] = nxt[r];
    }
    for (int r = 0; r < 4; ++r) seg[t][r] = cur[r];
    __syncthreads();
    if (t == 0) {
        double e[4] = {0.0, 0.0, 0.0, 0.0};          // entry state of segment 0
        for (int i = 0; i < 256; ++i) {
            double nxt[4];
            for (int r = 0; r < 4; ++r)
                nxt[r] = seg[i][r] + (TK[r][0] * e[0] + TK[r][1] * e[1] + TK[r][2] * e[2] + TK[r][3] * e[3]);
            for (int r = 0; r < 4; ++r) { seg[i][r] = e[r]; e[r] = nxt[r]; }
        }
    }
    __syncthreads();
    for (int r = 0; r < 4; ++r) cur[r] = seg[t][r];
    for (int64_t c = c0; c < c1; ++c) {
        double end0[4], nxt[4];
        for (int r = 0; r < 4; ++r) end0[r] = st[c * 4 + r];
        for (int r = 0; r < 4; ++r) nxt[r] = end0[r] + (Tm[r][0] * cur[0] + Tm[r][1] * cur[1] + Tm[r][2] * cur[2] + Tm[r][3] * cur[3]);
        for (int r = 0; r < 4; ++r) { st[c * 4 + r] = cur[r]; cur[r] = nxt[r]; }
    }
}

// z[ch][j] = sum_{n in [lo_j, hi_j)} ysq[ch][n] * inv_len      (hi_j may exceed n_samples: clipped, like a numpy slice)
__global__ __launch_bounds__(256) void kw_block_energy_kernel(const double* __restrict__ ysq, int64_t n_samples,
                                                              const int64_t* __restrict__ lo, const int64_t* __restrict__ hi,
                                                              int n_blocks, double inv_len, double* __restrict__ z) {
    __shared__ double red[256];
    const int j = blockIdx.x, ch = blockIdx.y;
    const int64_t a = lo[j], b = hi[j] < n_samples ? hi[j] : n_samples;
    const double* y = ysq + (int64_t)ch * n_samples;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t n = a + threadIdx.x;
    for (; n + 768 < b; n += 1024) { s0 += y[n]; s1 += y[n + 256]; s2 += y[n + 512]; s3 += y[n + 768]; }
    for (; n < b; n += 256) s0 += y[n];
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) z[(int64_t)ch * n_blocks + j] = red[0] * inv_len;
}

constexpr int KW_CHUNK = 1024;

}  // namespace
}  // namespace dam

// RBJ biquads exactly as pyloudnorm 0.1.x builds its "K-weighting" (iirfilter.py): high shelf G = 4 dB, Q = 1/sqrt(2),
// fc = 1500 Hz; high pass Q = 0.5, fc = 38 Hz; both normalised by a0.  coef12 = {b0,b1,b2,a0(=1),a1,a2} x 2 stages.
extern "C" int dam_loudness_kweight_coeffs(double rate, double* coef12) {
    if (!coef12 || !(rate > 0.0)) return DAM_ERR_BAD_ARG;
    const double pi = 3.14159265358979323846;
    {
        const double G = 4.0, Q = 1.0 / sqrt(2.0), fc = 1500.0;
        const double A = pow(10.0, G / 40.0), w0 = 2.0 * pi * (fc / rate), alpha = sin(w0) / (2.0 * Q);
        const double b0 = A * ((A + 1) + (A - 1) * cos(w0) + 2 * sqrt(A) * alpha);
        const double b1 = -2 * A * ((A - 1) + (A + 1) * cos(w0));
        const double b2 = A * ((A + 1) + (A - 1) * cos(w0) - 2 * sqrt(A) * alpha);
        const double a0 = (A + 1) - (A - 1) * cos(w0) + 2 * sqrt(A) * alpha;
        const double a1 = 2 * ((A - 1) - (A + 1) * cos(w0));
        const double a2 = (A + 1) - (A - 1) * cos(w0) - 2 * sqrt(A) * alpha;
        coef12[0] = b0 / a0; coef12[1] = b1 / a0; coef12[2] = b2 / a0; coef12[3] = 1.0; coef12[4] = a1 / a0; coef12[5] = a2 / a0;
    }
    {
        const double Q = 0.5, fc = 38.0;
        const double w0 = 2.0 * pi * (fc / rate), alpha = sin(w0) / (2.0 * Q);
        const double b0 = (1 + cos(w0)) / 2, b1 = -(1 + cos(w0)), b2 = (1 + cos(w0)) / 2;
        const double a0 = 1 + alpha, a1 = -2 * cos(w0), a2 = 1 - alpha;
        coef12[6] = b0 / a0; coef12[7] = b1 / a0; coef12[8] = b2 / a0; coef12[9] = 1.0; coef12[10] = a1 / a0; coef12[11] = a2 / a0;
    }
    return DAM_OK;
}

extern "C" int64_t dam_loudness_workspace_bytes(int64_t n_samples, int channels) {
    if (n_samples <= 0 || channels <= 0) return 0;
    const int64_t n_chunks = dam::cdiv(n_samples, dam::KW_CHUNK);
    return (int64_t)channels * (n_chunks * 4 + n_samples) * (int64_t)sizeof(double);
}

extern "C" int dam_loudness_block_energy(const void* x, int x_is_f64, int64_t n_samples, int channels, int64_t sample_stride,
                                         int64_t channel_stride, const double* coef12_host, const int64_t* blk_lo,
                                         const int64_t* blk_hi, int n_blocks, double block_len, double* z, void* workspace,
                                         void* stream) {
    using namespace dam;
    if (!x || !coef12_host || !blk_lo || !blk_hi || !z || !workspace) return DAM_ERR_BAD_ARG;
    if (n_samples <= 0 || channels <= 0 || channels > 65535 || n_blocks <= 0 || !(block_len > 0.0)) return DAM_ERR_BAD_ARG;
    KwCoef k;
    for (int s = 0; s < 2; ++s) {
        if (coef12_host[s * 6 + 3] != 1.0) return DAM_ERR_BAD_ARG;        // normalised sections only
        for (int i = 0; i < 3; ++i) k.b[s][i] = coef12_host[s * 6 + i];
        k.a[s][0] = coef12_host[s * 6 + 4]; k.a[s][1] = coef12_host[s * 6 + 5];
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_chunks = cdiv(n_samples, KW_CHUNK);
    double* state = reinterpret_cast<double*>(workspace);
    double* ysq = state + (int64_t)channels * n_chunks * 4;
    const dim3 grid((unsigned)cdiv(n_chunks, 64), (unsigned)channels);
    if (x_is_f64) {
        hipLaunchKernelGGL((kw_chunk_kernel<double, 1>), grid, dim3(64), 0, st, reinterpret_cast<const double*>(x), n_samples,
                           sample_stride, channel_stride, k, KW_CHUNK, n_chunks, state, ysq);
    } else {
        hipLaunchKernelGGL((kw_chunk_kernel<float, 1>), grid, dim3(64), 0, st, reinterpret_cast<const float*>(x), n_samples,
                           sample_stride, channel_stride, k, KW_CHUNK, n_chunks, state, ysq);
    }
    DAM_CHECK_LAUNCH();
    hipLaunchKernelGGL(kw_scan_kernel, dim3(channels), dim3(256), 0, st, k, KW_CHUNK, n_chunks, state);
    DAM_CHECK_LAUNCH();
    if (x_is_f64) {
        hipLaunchKernelGGL((kw_chunk_kernel<double, 2>), grid, dim3(64), 0, st, reinterpret_cast<const double*>(x), n_samples,
                           sample_stride, channel_stride, k, KW_CHUNK, n_chunks, state, ysq);
    } else {
        hipLaunchKernelGGL((kw_chunk_kernel<float, 2>), grid, dim3(64), 0, st, reinterpret_cast<const float*>(x), n_samples,
                           sample_stride, channel_stride, k, KW_CHUNK, n_chunks, state, ysq);
    }
    DAM_CHECK_LAUNCH();
    hipLaunchKernelGGL(kw_block_energy_kernel, dim3(n_blocks, channels), dim3(256), 0, st, ysq, n_samples, blk_lo, blk_hi,
                       n_blocks, 1.0 / block_len, z);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}
