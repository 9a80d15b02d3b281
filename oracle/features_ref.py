"""Oracle: feature front-end (test infrastructure, see oracle/__init__.py).

Restates, in numpy, what the reference computes through ``torch.stft`` and
``torchaudio.functional.amplitude_to_DB``:

  data/dataset.py:181-183   _stereo_to_mono     mean over the channel axis
  data/dataset.py:164-168   _augment_audio      audio * U(0.6, 1.4)
  data/dataset.py:132-162   compute_features    STFT -> |.| -> 20*log10(max(., 1e-5))
  data/dataset.py:207-210   stacking            stems stacked in tracklist order, mix = target

The STFT conventions are torch.stft's defaults as the reference calls it
(``center=True``, ``pad_mode='reflect'``, ``onesided=True``, ``normalized=False``)
with the *float32* periodic Hann window ``torch.hann_window(n_fft)``.
"""
import numpy as np

TRACKLIST = ['bass', 'drums', 'vocals', 'other', 'mix']   # data/dataset.py:42


def hann_window_f32(n_fft: int) -> np.ndarray:
    """torch.hann_window(n_fft) (periodic, float32) -- data/dataset.py:148.

    The table is taken from torch itself so that it is bit-identical to what the
    reference feeds to torch.stft; the closed form is 0.5 - 0.5*cos(2*pi*n/n_fft).
    """
    import torch
    return torch.hann_window(n_fft).numpy().copy()


def stereo_to_mono(audio: np.ndarray) -> np.ndarray:
    """data/dataset.py:181-183: np.mean(audio, axis=1) on [n, channels]."""
    return np.mean(audio, axis=1)


def augment_audio(audio: np.ndarray, gain: float) -> np.ndarray:
    """data/dataset.py:164-168 with the random draw made explicit."""
    return gain * audio


def num_frames(n: int, hop: int) -> int:
    """torch.stft with center=True: T = 1 + floor(N / hop)."""
    return 1 + n // hop


def reflect_pad(x: np.ndarray, p: int) -> np.ndarray:
    """torch.stft center=True, pad_mode='reflect' (no edge repeat)."""
    if x.shape[-1] <= p:
        raise ValueError('reflect padding needs N > n_fft/2')
    return np.concatenate([x[p:0:-1], x, x[-2:-p - 2:-1]])


def stft_mag(x: np.ndarray, n_fft: int = 2048, hop: int = 1024, dtype=np.float64) -> np.ndarray:
    """|STFT| of a mono signal, shape [n_fft/2+1, T] -- data/dataset.py:145-151."""
    x = np.asarray(x, dtype=dtype)
    w = hann_window_f32(n_fft).astype(dtype)
    xp = reflect_pad(x, n_fft // 2)
    t = num_frames(x.shape[0], hop)
    idx = np.arange(n_fft)[None, :] + hop * np.arange(t)[:, None]
    frames = xp[idx] * w[None, :]
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)     # [T, F]
    return np.abs(spec).T.astype(dtype)


def amplitude_to_db(mag: np.ndarray, multiplier: float = 20.0, amin: float = 1e-5,
                    db_multiplier: float = 0.0) -> np.ndarray:
    """torchaudio.functional.amplitude_to_DB as called at data/dataset.py:152-155
    (no top_db): multiplier*log10(clamp(x, min=amin)) - multiplier*db_multiplier.
    Third-party formula, torchaudio absent from the image: parity unpinned.
    The logarithm is evaluated in float64 and rounded to mag's dtype (numpy's float32 log10 is
    not correctly rounded at the 1e-5 floor; torch's is, giving exactly -100.0 dB for silence)."""
    m = np.maximum(mag, np.asarray(amin, dtype=mag.dtype)).astype(np.float64)
    return (multiplier * np.log10(m) - multiplier * db_multiplier).astype(mag.dtype)


def normalize_columns(feat: np.ndarray) -> np.ndarray:
    """librosa.util.normalize(features) (norm=inf, axis=0) -- the call that is
    commented out at data/dataset.py:159-160.  Each frame (column) is
    divided by its max-abs; columns whose max-abs is below tiny are left alone."""
    m = np.max(np.abs(feat), axis=0, keepdims=True)
    tiny = np.finfo(feat.dtype).tiny
    return np.where(m < tiny, feat, feat / np.where(m < tiny, 1.0, m))


def compute_features(audio: np.ndarray, window_size: int = 2048, hop_length: int = 1024,
                     dtype=np.float64, normalize: bool = False) -> np.ndarray:
    """data/dataset.py:132-162 for a mono signal -> [window_size/2+1, T] dB."""
    f = amplitude_to_db(stft_mag(audio, window_size, hop_length, dtype)).astype(dtype)
    return normalize_columns(f) if normalize else f


def clip_features(pcm: np.ndarray, window_size: int = 2048, hop_length: int = 1024,
                  dtype=np.float64, gains=None, normalize: bool = False):
    """One dataset item from in-memory audio -- data/dataset.py:185-210.

    pcm: [S+1, n, channels] (stems in tracklist order, mix last).
    Returns (train_features [S, F, T], gt_features [F, T]).
    """
    feats = []
    for k in range(pcm.shape[0]):
        a = stereo_to_mono(pcm[k].astype(np.float64)) if pcm.ndim == 3 else pcm[k].astype(np.float64)
        if gains is not None:
            a = augment_audio(a, gains[k])
        feats.append(compute_features(a.astype(dtype), window_size, hop_length, dtype, normalize))
    return np.stack(feats[:-1]), feats[-1]


def augment_gain_ref(seed: int, item: int, track: int, lo: float = 0.6, hi: float = 1.4) -> np.float32:
    """The product's reproducible stand-in for ``np.random.uniform(0.6, 1.4)`` at data/dataset.py:164-168 (the reference
    draws from numpy's global state; there is nothing to match bit for bit): u = top 24 bits of
    splitmix64(seed * K + item * 4096 + track) / 2^24, gain = lo + (hi - lo) * u, all in float32."""
    m = (1 << 64) - 1
    z = (seed * 0xD1342543DE82EF95 + item * 4096 + track) & m
    z = (z + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    z ^= z >> 31
    r = z >> 32
    u = np.float32(r >> 8) * np.float32(1.0 / 16777216.0)
    return np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u
