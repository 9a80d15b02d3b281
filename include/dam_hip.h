/*
 * dam_hip.h -- C ABI of libdam_hip.so, the MI355X (gfx950) implementation of the
 * deep-audio-mixer hot path.
 *
 * The reference (apelykh/deep-audio-mixer) has no FFI: its hot path is reached through
 * torch.stft / torch.nn on one device.  Each entry point below names the reference
 * code (file:line, relative to the reference root) whose arithmetic it replaces; the
 * Python mirror of the reference interface (deep-audio-mixer_amd/) binds these through
 * ctypes, see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in
 *     _host; the library allocates nothing and keeps no state;
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*), never
 *     synchronises, and is hipGraph-capturable;
 *   - returns DAM_OK (0) or a negative dam_status; never throws across the ABI;
 *   - activations are NHWC float32 ("pixels x channels", channel count a multiple of 16
 *     unless stated); tensors the reference API exposes (features, masked, gains) are in
 *     the reference's own layout ([B,S,F,T] / [B,F,T] / [B,S]).
 */
#ifndef DAM_HIP_H
#define DAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dam_status {
    DAM_OK = 0,
    DAM_ERR_BAD_ARG = -1,        /* null pointer, non-positive size, N <= n_fft/2 ... */
    DAM_ERR_UNSUPPORTED = -2,    /* shape/dtype outside what the gfx950 kernels implement */
    DAM_ERR_LAUNCH = -3,         /* hipGetLastError() != hipSuccess after the launch */
    DAM_ERR_WORKSPACE = -4       /* caller-provided workspace too small */
} dam_status;

typedef enum dam_pcm_dtype { DAM_PCM_F32 = 0, DAM_PCM_F64 = 1 } dam_pcm_dtype;

/* Library / build identification ("gfx950"). */
const char* dam_arch(void);
int dam_abi_version(void);

/* ---------------------------------------------------------------------------------
 * Feature front-end.  Replaces data/dataset.py:132-162 (compute_features: torch.stft ->
 * abs -> amplitude_to_DB), :181-183 (_stereo_to_mono), :164-168 (_augment_audio) and the
 * stacking at :207-210, for a whole batch of tracks in one launch.
 * --------------------------------------------------------------------------------- */

/* Host helper: number of float2 entries of the twiddle table for n_fft (n_fft entries:
 * W_{n_fft}^k = exp(-2 pi i k / n_fft)), and a routine that fills it (computed in double). */
int64_t dam_stft_twiddle_count(int n_fft);
int dam_stft_fill_twiddles_host(int n_fft, float* table_host /* [2*count] re,im */);

/* out[track][f][t] = 20*log10(max(|STFT(mean_c pcm[track][:, c] * gain[track])|, amin))
 *   pcm      : [n_tracks][n_samples][channels], dtype per pcm_dtype, interleaved channels;
 *              track k starts at pcm + k*pcm_track_stride elements
 *   window   : [n_fft] float32 (the caller uploads torch.hann_window(n_fft))
 *   twiddles : table from dam_stft_fill_twiddles_host
 *   gain     : optional [n_tracks] float32 (nullptr = 1), the augmentation draw
 *   normalize: 0, or 1 = divide every frame (column) by its max-abs over bins
 *              (librosa.util.normalize semantics of the call disabled at data/dataset.py:159-160)
 *   out      : [n_tracks][n_fft/2+1][T] float32, T = 1 + n_samples/hop
 * center=True / reflect padding / onesided / unnormalised, as torch.stft's defaults.
 * Supported: n_fft == 2048, hop even and >= 1, channels 1 or 2, n_samples > n_fft/2. */
int dam_stft_logmag_f32(const void* pcm, int pcm_dtype, int64_t n_tracks, int64_t n_samples, int channels,
                        int64_t pcm_track_stride, const float* window, const float* twiddles,
                        const float* gain, int n_fft, int hop, float amin, int normalize,
                        float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DAM_HIP_H */
