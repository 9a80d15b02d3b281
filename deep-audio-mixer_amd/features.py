"""GPU feature front-end: batched STFT -> |.| -> dB through ``dam_stft_logmag_f32``.

Mirrors the arithmetic of the reference's ``MultitrackAudioDataset.compute_features``
(data/dataset.py:132-162) with ``_stereo_to_mono`` (:181-183) and ``_augment_audio``
(:164-168) fused in, for all tracks of a batch in one launch.
"""
import ctypes

import numpy as np
import torch

from . import _lib

AMIN = 1e-5      # data/dataset.py:154
_tables = {}


def _get_tables(device, n_fft):
    key = (device.type, device.index, n_fft)
    if key not in _tables:
        L = _lib.lib()
        n = L.dam_stft_twiddle_count(n_fft)
        tw = np.empty(2 * n, dtype=np.float32)
        _lib.check(L.dam_stft_fill_twiddles_host(n_fft, tw.ctypes.data_as(ctypes.c_void_p)), 'dam_stft_fill_twiddles_host')
        # the window is torch's own float32 periodic Hann table, computed on the CPU exactly as the
        # reference does (torch.hann_window(window_size), data/dataset.py:148) and uploaded
        win = torch.hann_window(n_fft, dtype=torch.float32)
        _tables[key] = (win.to(device), torch.from_numpy(tw).to(device))
    return _tables[key]


def num_frames(n_samples, hop):
    return 1 + n_samples // hop


def stft_logmag(pcm, n_fft=2048, hop=1024, gain=None, normalize=False, out=None):
    """pcm: CUDA tensor [n_tracks, n_samples, channels] or [n_tracks, n_samples], float32 or
    float64, channels in {1, 2} interleaved.  Returns float32 [n_tracks, n_fft/2+1, T] in dB."""
    _lib.require_cuda(pcm, gain, out)
    if pcm.dim() == 2:
        pcm = pcm.unsqueeze(-1)
    if pcm.dim() != 3:
        raise ValueError('pcm must be [tracks, samples(, channels)]')
    if pcm.dtype not in (torch.float32, torch.float64):
        raise TypeError('pcm must be float32 or float64')
    pcm = pcm.contiguous()
    n_tracks, n, ch = pcm.shape
    t = num_frames(n, hop)
    win, tw = _get_tables(pcm.device, n_fft)
    if out is None:
        out = torch.empty((n_tracks, n_fft // 2 + 1, t), dtype=torch.float32, device=pcm.device)
    elif out.shape != (n_tracks, n_fft // 2 + 1, t) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError('bad out tensor')
    if gain is not None:
        gain = gain.to(device=pcm.device, dtype=torch.float32).contiguous()
        if gain.numel() != n_tracks:
            raise ValueError('gain must have one entry per track')
    st = _lib.lib().dam_stft_logmag_f32(_lib.ptr(pcm), 0 if pcm.dtype == torch.float32 else 1, n_tracks, n, ch,
                                        n * ch, _lib.ptr(win), _lib.ptr(tw), _lib.ptr(gain), n_fft, hop,
                                        AMIN, 1 if normalize else 0, _lib.ptr(out), _lib.stream())
    _lib.check(st, 'dam_stft_logmag_f32')
    return out
