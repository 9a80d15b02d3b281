// dam_mix.hip -- the full-song inference tail on the device (BASELINE config C5).
//
//   dam_gains_smooth            inference_utils.py:125-130 (10 ** (0.5 * g) per chunk and stem) and :136-141
//                               (scipy.signal.savgol_filter(raw, window, 2), default mode='interp')
//   dam_gain_ramp_apply         inference_utils.py:12-41 (interpolate_mask: piecewise-constant stretch of the gain sequence
//                               to sample resolution) fused with :143 (loaded_tracks[track] * mask)
//   dam_mixdown_peak_normalize  the callers' next step (inference.ipynb cells 9/11, evaluation.py:59-66): sum of the mixed
//                               stems and librosa.util.normalize(track_sum, axis=1), fused with the gain ramp
//
// The 8-byte-per-sample mask is never materialised: sample n of every channel is scaled by gains[min(n / seg, n_gains-1)],
// seg = int(n_samples / n_gains).  The grid is laid out over the GAIN SEGMENTS (blockIdx.y = segment), so no thread ever
// divides a sample index: a workgroup streams a contiguous piece of one segment with 16-byte loads and its gain is a
// scalar.  HBM-bound, one pass over the song.
#include "dam_common.h"

namespace dam {
namespace {

template <typename T> struct vec16;
template <> struct vec16<float> { typedef f32x4_u type; static constexpr int N = 4; };
template <> struct vec16<double> { typedef f64x2_u type; static constexpr int N = 2; };

// sample range of gain segment k: [k*seg, (k+1)*seg), the last one runs to the end of the row (interpolate_mask :37-39)
__device__ __forceinline__ void seg_range(int k, int n_gains, int64_t seg, int64_t n_samples, int64_t& lo, int64_t& hi) {
    lo = (int64_t)k * seg;
    hi = k == n_gains - 1 ? n_samples : lo + seg;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void gain_ramp_apply_kernel(const TI* __restrict__ audio, const double* __restrict__ gains,
                                                              int64_t rows, int64_t rows_per_gain, int64_t n_samples,
                                                              int n_gains, int64_t seg, TO* __restrict__ out) {
    constexpr int V = vec16<TI>::N;
    for (int k = blockIdx.y; k < n_gains; k += gridDim.y) {
        int64_t lo, hi;
        seg_range(k, n_gains, seg, n_samples, lo, hi);
        for (int64_t row = blockIdx.z; row < rows; row += gridDim.z) {
            const TO g = (TO)gains[(row / rows_per_gain) * n_gains + k];
            const TI* a = audio + row * n_samples;
            TO* o = out + row * n_samples;
            for (int64_t n = lo + ((int64_t)blockIdx.x * 256 + threadIdx.x) * V; n < hi; n += (int64_t)gridDim.x * 256 * V) {
                if (n + V <= hi) {
                    const typename vec16<TI>::type x = *reinterpret_cast<const typename vec16<TI>::type*>(a + n);
#pragma unroll
                    for (int j = 0; j < V; ++j) o[n + j] = (TO)x[j] * g;
                } else {
                    for (int64_t m = n; m < hi; ++m) o[m] = (TO)a[m] * g;
                }
            }
        }
    }
}

// mix[r][n] = sum_s audio[s][r][n] * gains[s][segment of n], plus per-(row, block) max-abs partials.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void mixdown_kernel(const TI* __restrict__ audio, const double* __restrict__ gains, int S,
                                                      int64_t n_samples, int n_gains, int64_t seg, TO* __restrict__ mix,
                                                      TO* __restrict__ peak_partial) {
    constexpr int V = vec16<TI>::N;
    __shared__ TO red[256];
    const int row = blockIdx.z, rows = gridDim.z;
    TO m = 0;
    for (int k = blockIdx.y; k < n_gains; k += gridDim.y) {
        int64_t lo, hi;
        seg_range(k, n_gains, seg, n_samples, lo, hi);
        for (int64_t n = lo + ((int64_t)blockIdx.x * 256 + threadIdx.x) * V; n < hi; n += (int64_t)gridDim.x * 256 * V) {
            if (n + V <= hi) {
                TO acc[V];
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] = 0;
                for (int s = 0; s < S; ++s) {
                    const TO g = (TO)gains[(int64_t)s * n_gains + k];
                    const typename vec16<TI>::type x =
                        *reinterpret_cast<const typename vec16<TI>::type*>(audio + ((int64_t)s * rows + row) * n_samples + n);
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] += (TO)x[j] * g;
                }
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    mix[(int64_t)row * n_samples + n + j] = acc[j];
                    const TO b = acc[j] < 0 ? -acc[j] : acc[j];
                    m = b > m ? b : m;
                }
            } else {
                for (int64_t p = n; p < hi; ++p) {
                    TO a = 0;
                    for (int s = 0; s < S; ++s)
                        a += (TO)audio[((int64_t)s * rows + row) * n_samples + p] * (TO)gains[(int64_t)s * n_gains + k];
                    mix[(int64_t)row * n_samples + p] = a;
                    const TO b = a < 0 ? -a : a;
                    m = b > m ? b : m;
                }
            }
        }
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] = red[threadIdx.x + st] > red[threadIdx.x] ? red[threadIdx.x + st] : red[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) peak_partial[((int64_t)row * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0];
}

// mix[r][:] /= max_n |mix[r][n]|   (librosa.util.normalize(x, axis=1): rows whose peak is below `tiny` are left alone)
template <typename T>
__global__ __launch_bounds__(256) void peak_normalize_kernel(T* __restrict__ mix, int64_t n_samples, const T* __restrict__ peak_partial,
                                                             int nblk, T tiny) {
    constexpr int V = vec16<T>::N;
    __shared__ T red[256];
    const int row = blockIdx.y;
    T m = 0;
    for (int i = threadIdx.x; i < nblk; i += 256) { const T v = peak_partial[(int64_t)row * nblk + i]; m = v > m ? v : m; }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] = red[threadIdx.x + st] > red[threadIdx.x] ? red[threadIdx.x + st] : red[threadIdx.x];
        __syncthreads();
    }
    const T peak = red[0];
    if (peak < tiny) return;
    T* r = mix + (int64_t)row * n_samples;
    for (int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V; n < n_samples; n += (int64_t)gridDim.x * 256 * V) {
        if (n + V <= n_samples) {
            typename vec16<T>::type x = *reinterpret_cast<typename vec16<T>::type*>(r + n);
#pragma unroll
            for (int j = 0; j < V; ++j) x[j] = x[j] / peak;
            *reinterpret_cast<typename vec16<T>::type*>(r + n) = x;
        } else {
            for (int64_t p = n; p < n_samples; ++p) r[p] /= peak;
        }
    }
}

// One workgroup per stem.  amp[c] = 10^(0.5 * raw_db[c][s]) in float64 (numpy's np.power(10.0, 0.5 * x)), then the
// Savitzky-Golay smoothing of that sequence exactly as scipy.signal.savgol_filter(amp, window, polyorder) computes it in
// its default mode 'interp': output i is the least-squares polynomial of degree `polyorder` through the `window` samples
// starting at a = clamp(i - h, 0, n - window) (h = window / 2), evaluated at i -- for interior points that is the
// classic symmetric filter, for the first / last h points the polynomial fitted to the first / last window (scipy's
// _fit_edge).  The fit is done on the orthogonal polynomials of the symmetric integer grid u = -h..h (Stieltjes
// three-term recurrence p_{k+1} = u p_k - beta_k p_{k-1}, beta_k = N_k / N_{k-1}, N_k = sum_u p_k(u)^2): no linear
// system, float64 throughout, agrees with scipy's lstsq coefficients to ~1e-14 relative.
constexpr int SG_MAX_ORDER = 5;

__global__ __launch_bounds__(256) void gains_smooth_kernel(const float* __restrict__ raw_db, int n, int S, int window,
                                                           int order, double* __restrict__ amp, double* __restrict__ smooth,
                                                           float* __restrict__ smooth_f32) {
    extern __shared__ double y[];
    const int s = blockIdx.x;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const double a = pow(10.0, 0.5 * (double)raw_db[(int64_t)c * S + s]);
        y[c] = a;
        amp[(int64_t)s * n + c] = a;
    }
    __syncthreads();
    const int h = window / 2;
    // norms and recurrence coefficients (every thread computes the same few numbers: window * order^2 operations)
    double beta[SG_MAX_ORDER + 1], norm[SG_MAX_ORDER + 1];
    for (int k = 0; k <= order; ++k) {
        double acc = 0.0;
        for (int u = -h; u <= h; ++u) {
            double pm = 0.0, p = 1.0;
            for (int j = 0; j < k; ++j) { const double pn = (double)u * p - (j ? beta[j] : 0.0) * pm; pm = p; p = pn; }
            acc += p * p;
        }
        norm[k] = acc;
        beta[k] = k ? norm[k] / norm[k - 1] : 0.0;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int a = i - h;
        a = a < 0 ? 0 : (a > n - window ? n - window : a);
        const double v = (double)(i - a - h);
        double proj[SG_MAX_ORDER + 1];
        for (int k = 0; k <= order; ++k) proj[k] = 0.0;
        for (int t = 0; t < window; ++t) {
            const double u = (double)(t - h), yt = y[a + t];
            double pm = 0.0, p = 1.0;
            for (int k = 0; k <= order; ++k) {
                proj[k] += p * yt;
                const double pn = u * p - (k ? beta[k] : 0.0) * pm;
                pm = p; p = pn;
            }
        }
        double r = 0.0, pm = 0.0, p = 1.0;
        for (int k = 0; k <= order; ++k) {
            r += p * proj[k] / norm[k];
            const double pn = v * p - (k ? beta[k] : 0.0) * pm;
            pm = p; p = pn;
        }
        smooth[(int64_t)s * n + i] = r;
        if (smooth_f32) smooth_f32[(int64_t)s * n + i] = (float)r;
    }
}

struct MixGrid { unsigned bx, by; };
static MixGrid mix_grid(int64_t seg_max, int n_gains, int64_t rows, int vec, int64_t max_xy) {
    MixGrid g;
    g.by = (unsigned)(n_gains < 1024 ? n_gains : 1024);
    int64_t bx = cdiv(4096, (int64_t)g.by * rows);                  // aim at >= 4096 workgroups ...
    const int64_t cap = cdiv(seg_max, 256 * (int64_t)vec);          // ... but no more than one pass per workgroup
    if (bx > cap) bx = cap;
    if (max_xy > 0 && bx * g.by > max_xy) bx = max_xy / g.by;
    g.bx = (unsigned)(bx < 1 ? 1 : bx);
    return g;
}

}  // namespace
}  // namespace dam

extern "C" int dam_gains_smooth(const float* raw_db, int n_chunks, int n_stems, int window, int polyorder, double* amp,
                                double* smooth, float* smooth_f32, void* stream) {
    using namespace dam;
    if (!raw_db || !amp || !smooth || n_chunks <= 0 || n_stems <= 0) return DAM_ERR_BAD_ARG;
    // scipy's own argument checks (savgol_coeffs / savgol_filter mode='interp'): odd window > polyorder, window <= n
    if (window <= 0 || window % 2 == 0 || polyorder < 0 || polyorder >= window || window > n_chunks) return DAM_ERR_BAD_ARG;
    if (polyorder > SG_MAX_ORDER || n_chunks > 8192 || n_stems > 65535) return DAM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gains_smooth_kernel, dim3((unsigned)n_stems), dim3(256), (size_t)n_chunks * sizeof(double),
                       (hipStream_t)stream, raw_db, n_chunks, n_stems, window, polyorder, amp, smooth, smooth_f32);
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int dam_gain_ramp_apply(const void* audio, int audio_is_f64, const double* gains, int64_t rows,
                                   int64_t rows_per_gain, int64_t n_samples, int n_gains, void* out, int out_is_f64,
                                   void* stream) {
    using namespace dam;
    if (!audio || !gains || !out || rows <= 0 || rows_per_gain <= 0 || n_samples <= 0 || n_gains <= 0 || n_gains > n_samples)
        return DAM_ERR_BAD_ARG;
    const int64_t seg = n_samples / n_gains;
    const int64_t seg_max = n_samples - (int64_t)(n_gains - 1) * seg;
    const unsigned bz = (unsigned)(rows < 64 ? rows : 64);
    const MixGrid g = mix_grid(seg_max, n_gains, bz, audio_is_f64 ? 2 : 4, 0);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(g.bx, g.by, bz), block(256);
#define DAM_RAMP(TI, TO)                                                                                             \
    hipLaunchKernelGGL((gain_ramp_apply_kernel<TI, TO>), grid, block, 0, st, (const TI*)audio, gains, rows, rows_per_gain, \
                       n_samples, n_gains, seg, (TO*)out)
    if (audio_is_f64) { if (out_is_f64) DAM_RAMP(double, double); else DAM_RAMP(double, float); }
    else { if (out_is_f64) DAM_RAMP(float, double); else DAM_RAMP(float, float); }
#undef DAM_RAMP
    DAM_CHECK_LAUNCH();
    return DAM_OK;
}

extern "C" int64_t dam_mixdown_workspace_elems(int64_t rows) { return rows > 0 ? rows * 4096 : 0; }

extern "C" int dam_mixdown_peak_normalize(const void* audio, int audio_is_f64, const double* gains, int n_stems,
                                          int64_t rows, int64_t n_samples, int n_gains, int normalize, void* mix,
                                          int mix_is_f64, void* workspace, void* stream) {
    using namespace dam;
    if (!audio || !gains || !mix || !workspace || n_stems <= 0 || rows <= 0 || n_samples <= 0 || n_gains <= 0 || n_gains > n_samples)
        return DAM_ERR_BAD_ARG;
    if (rows > 65535) return DAM_ERR_UNSUPPORTED;
    const int64_t seg = n_samples / n_gains;
    const int64_t seg_max = n_samples - (int64_t)(n_gains - 1) * seg;
    const MixGrid g = mix_grid(seg_max, n_gains, rows, audio_is_f64 ? 2 : 4, 4096);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(g.bx, g.by, (unsigned)rows), block(256);
    const int nblk = (int)(g.bx * g.by);
    int64_t nx = cdiv(n_samples, 256 * 4 * 4);
    if (nx > 2048) nx = 2048;
    const dim3 ngrid((unsigned)nx, (unsigned)rows);
#define DAM_MIX(TI, TO)                                                                                           \
    hipLaunchKernelGGL((mixdown_kernel<TI, TO>), grid, block, 0, st, (const TI*)audio, gains, n_stems, n_samples, n_gains, \
                       seg, (TO*)mix, (TO*)workspace)
    if (audio_is_f64) { if (mix_is_f64) DAM_MIX(double, double); else DAM_MIX(double, float); }
    else { if (mix_is_f64) DAM_MIX(float, double); else DAM_MIX(float, float); }
#undef DAM_MIX
    DAM_CHECK_LAUNCH();
    if (normalize) {
        if (mix_is_f64)
            hipLaunchKernelGGL(peak_normalize_kernel<double>, ngrid, block, 0, st, (double*)mix, n_samples,
                               (const double*)workspace, nblk, 2.2250738585072014e-308);
        else
            hipLaunchKernelGGL(peak_normalize_kernel<float>, ngrid, block, 0, st, (float*)mix, n_samples,
                               (const float*)workspace, nblk, 1.17549435e-38f);
        DAM_CHECK_LAUNCH();
    }
    return DAM_OK;
}
