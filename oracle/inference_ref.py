"""Oracle: full-song inference tail on numpy (test infrastructure).

  inference_utils.py:12-41     interpolate_mask
  inference_utils.py:105-145   mix_song_smooth
  data/dataset_utils.py:46-50  scalar_dB_to_amplitude  (10**(0.5*x), sic)

mix_song_smooth is not runnable at the reference HEAD (it hands
[C, n] slices to torch.stft).  The restatement below uses the documented fix --
features are computed on the channel mean, gains are applied to the original
multichannel audio -- and is checked piecewise against the golden vectors for the
pieces that do run at HEAD.
"""
import numpy as np
from scipy.signal import savgol_filter

from . import features_ref


def scalar_db_to_amplitude(x):
    return np.power(10.0, 0.5 * x)


def interpolate_mask(spec_mask, tgt_len):
    spec_mask = np.asarray(spec_mask, dtype=np.float64)
    n = len(spec_mask)
    assert n <= tgt_len
    out = np.zeros(tgt_len)
    seg = int(tgt_len / n)
    for i in range(n - 1):
        out[i * seg:(i + 1) * seg] = spec_mask[i]
    if n > 1:
        out[(n - 1) * seg:] = spec_mask[-1]
    return out


def savgol_window(num_chunks):
    """inference_utils.py:136-139: int(num_chunks/4) made odd."""
    w = int(num_chunks / 4)
    return w if w % 2 else w + 1


def mix_song_smooth(model_fn, loaded_tracks, stems, chunk_length=1, sr=44100,
                    window_size=2048, hop_length=1024):
    """model_fn(features [1,S,F,T] float32 ndarray) -> gains [S] (raw model outputs)."""
    chunk = chunk_length * sr
    n = loaded_tracks[stems[0]].shape[1]
    num_chunks = int(n / chunk)
    raw = {t: [] for t in stems}
    for c in range(1, num_chunks):
        lo, hi = (c - 1) * chunk, c * chunk
        feats = [features_ref.compute_features(loaded_tracks[t][:, lo:hi].mean(axis=0),
                                               window_size, hop_length) for t in stems]
        g = model_fn(np.stack(feats)[None].astype(np.float32))
        for t, gv in zip(stems, g):
            raw[t].append(float(scalar_db_to_amplitude(np.float64(gv))))
    smooth, mixed = {}, {}
    for t in stems:
        sm = savgol_filter(raw[t], savgol_window(num_chunks), 2)
        smooth[t] = list(sm)
        mixed[t] = loaded_tracks[t] * interpolate_mask(sm, loaded_tracks[t].shape[1])
    return mixed, raw, smooth
