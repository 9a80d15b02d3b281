"""Full-song inference -- drop-in for the reference's inference_utils.py (interpolate_mask :12-41,
mix_song_smooth :105-145) on the GPU.

mix_song_smooth at the reference HEAD cannot run (it hands [channels, n] slices to torch.stft); this
module implements the intended semantics: features of the channel MEAN, gains applied to the original
multichannel audio.  All chunks of the song go through the front-end in ONE launch.  The model is applied as the
reference applies it -- whatever ``model.training`` is, never toggled here: in eval mode all chunks
run as one batch; in training mode BatchNorm uses per-call batch statistics, so chunks run one by one (batch of 1)
exactly as in the reference loop.  Gain smoothing (Savitzky-Golay, 59 numbers per stem) stays on the host with
scipy as in the reference; the sample-rate gain ramp and the multiply are one HIP kernel (dam_gain_ramp_apply).
"""
import numpy as np
import torch
from scipy.signal import savgol_filter

from . import features, ops
from .data.dataset_utils import scalar_dB_to_amplitude

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')      # inference_utils.py:9


def interpolate_mask(spec_mask: np.array, tgt_len: int) -> np.array:
    """inference_utils.py:12-41 (host version, kept for API parity; mix_song_smooth uses the fused kernel)."""
    assert len(spec_mask) <= tgt_len, "Target mask should be longer than the initial one"
    sample_mask = np.zeros(tgt_len)
    interp_coef = int(tgt_len / len(spec_mask))
    final_i = -1
    for chunk_i in range(0, len(spec_mask) - 1):
        i_from, i_to = chunk_i * interp_coef, (chunk_i + 1) * interp_coef
        sample_mask[i_from:i_to] = spec_mask[chunk_i]
        final_i = i_to
    if final_i > -1:
        sample_mask[final_i:] = spec_mask[-1]
    return sample_mask


def _savgol_window(num_chunks):
    w = int(num_chunks / 4)          # inference_utils.py:136-139
    return w if w % 2 else w + 1


def predict_chunk_gains(model, pcm, n_stems, n_chunks, chunk_samples, window_size=2048, hop_length=1024):
    """pcm: CUDA [n_stems, channels, n] -> raw model outputs [n_chunks-1, n_stems] for chunks 0..n_chunks-2
    (the reference loop ``range(1, num_chunks)`` processes exactly those, inference_utils.py:111-113)."""
    n_proc = n_chunks - 1
    ch = pcm.shape[1]
    seg = pcm[:, :, :n_proc * chunk_samples].reshape(n_stems, ch, n_proc, chunk_samples)
    tracks = seg.permute(2, 0, 3, 1).reshape(n_proc * n_stems, chunk_samples, ch).contiguous()   # interleaved channels
    feats = features.stft_logmag(tracks, window_size, hop_length)
    feats = feats.view(n_proc, n_stems, feats.shape[1], feats.shape[2])
    with torch.no_grad():
        if model.training:
            gains = [torch.cat(model(feats[i:i + 1])[1], 1) for i in range(n_proc)]
            return torch.cat(gains, 0)
        return torch.cat(model(feats)[1], 1)


def mix_song_smooth(dataset, model, loaded_tracks: dict, chunk_length=1, sr=44100):
    """Returns (mixed_tracks {track: ndarray[channels, n]}, raw_gains {track: [float]}, smooth_gains {track: list})."""
    stems = [t for t in dataset.get_tracklist() if t != 'mix']
    chunk_samples = chunk_length * sr
    n = len(loaded_tracks[stems[0]][0])
    num_chunks = int(n / chunk_samples)
    dev = next(model.parameters()).device
    audio = {t: np.ascontiguousarray(loaded_tracks[t]) for t in stems}
    pcm = torch.stack([torch.from_numpy(audio[t]) for t in stems]).to(dev)          # [S, channels, n]
    g = predict_chunk_gains(model, pcm, len(stems), num_chunks, chunk_samples).double().cpu().numpy()
    raw_gains = {t: [float(scalar_dB_to_amplitude(v)) for v in g[:, i]] for i, t in enumerate(stems)}
    smooth_gains = {t: [] for t in stems}
    mixed_tracks = {}
    for i, t in enumerate(stems):
        smoothed = savgol_filter(raw_gains[t], _savgol_window(num_chunks), 2)
        smooth_gains[t].extend(smoothed)
        gains_dev = torch.from_numpy(np.ascontiguousarray(smoothed)).to(device=dev, dtype=pcm.dtype)
        mixed_tracks[t] = ops.gain_ramp_apply(pcm[i], gains_dev).cpu().numpy()
    return mixed_tracks, raw_gains, smooth_gains


def mix_song_to_master(dataset, model, loaded_tracks: dict, chunk_length=1, sr=44100, normalize=True):
    """mix_song_smooth followed by what every caller of the reference does next (inference.ipynb cells 9/11,
    evaluation.py:59-66): ``track_sum = np.sum(list(mixed_tracks.values()), axis=0)`` and, if ``normalize``,
    ``librosa.util.normalize(track_sum, axis=1)`` -- fused into one pass over the song on the GPU (the per-stem mixed
    tracks are never materialised).  Returns (mix ndarray[channels, n], raw_gains, smooth_gains)."""
    stems = [t for t in dataset.get_tracklist() if t != 'mix']
    chunk_samples = chunk_length * sr
    num_chunks = int(len(loaded_tracks[stems[0]][0]) / chunk_samples)
    dev = next(model.parameters()).device
    pcm = torch.stack([torch.from_numpy(np.ascontiguousarray(loaded_tracks[t])) for t in stems]).to(dev)
    g = predict_chunk_gains(model, pcm, len(stems), num_chunks, chunk_samples).double().cpu().numpy()
    raw_gains = {t: [float(scalar_dB_to_amplitude(v)) for v in g[:, i]] for i, t in enumerate(stems)}
    smooth = np.stack([savgol_filter(raw_gains[t], _savgol_window(num_chunks), 2) for t in stems])
    gains_dev = torch.from_numpy(np.ascontiguousarray(smooth)).to(device=dev, dtype=pcm.dtype)
    mix = ops.mixdown_peak_normalize(pcm, gains_dev, normalize=normalize).cpu().numpy()
    return mix, raw_gains, {t: list(smooth[i]) for i, t in enumerate(stems)}
