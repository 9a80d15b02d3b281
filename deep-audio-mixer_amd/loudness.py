"""BS.1770 integrated loudness on the device -- the part of the third-party ``pyloudnorm`` package the reference uses
(``pyln.Meter(sr).integrated_loudness(data)`` at data/dataset.py:118-126, evaluation.py:33,40,64,
models/baselines/mean_loudness_model.py:8,16 and ``pyln.normalize.loudness`` at evaluation.py:65,
mean_loudness_model.py:17), with the same names, argument meaning and error behaviour:

    meter = Meter(44100)                       # K-weighting, 400 ms blocks
    lufs = meter.integrated_loudness(data)     # data: [samples] or [samples, channels], numpy or torch
    out = normalize_loudness(data, lufs, -20.0)

The samples are filtered and block-averaged by ``dam_loudness_block_energy`` (HIP, float64; csrc/dam_loudness.hip); the
gating over the few thousand block energies is the host logic below, a restatement of pyloudnorm 0.1.x ``meter.py``.
There is no CPU path: without the HIP library this raises.
"""
import ctypes
import warnings

import numpy as np
import torch

from . import _lib

_CHANNEL_GAINS = (1.0, 1.0, 1.0, 1.41, 1.41)      # L, R, C, Ls, Rs (pyloudnorm meter.py: G)


def _as_device_2d(data):
    """[samples] or [samples, channels] (numpy or torch) -> CUDA tensor [samples, channels] of float32/float64."""
    if isinstance(data, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(data))
    elif torch.is_tensor(data):
        t = data
    else:
        raise ValueError('Data must be of type numpy.ndarray or torch.Tensor.')
    if t.dtype not in (torch.float32, torch.float64):
        raise ValueError('Data must be floating point.')
    if t.dim() == 1:
        t = t.reshape(-1, 1)
    elif t.dim() != 2:
        raise ValueError('Audio must be [samples] or [samples, channels].')
    if t.shape[1] > 5:
        raise ValueError('Audio must have five channels or less.')
    if not t.is_cuda:
        t = t.cuda()
    return t


class Meter:
    """Same constructor surface as ``pyloudnorm.Meter`` for what the reference uses: ``Meter(rate)``."""

    def __init__(self, rate, filter_class='K-weighting', block_size=0.400):
        if filter_class != 'K-weighting':
            raise ValueError('only the K-weighting filter class is provided')
        self.rate = rate
        self.block_size = block_size
        coef = (ctypes.c_double * 12)()
        _lib.check(_lib.lib().dam_loudness_kweight_coeffs(float(rate), coef), 'dam_loudness_kweight_coeffs')
        self._coef = coef
        self.coefficients = np.array(list(coef)).reshape(2, 6)          # [stage][b0 b1 b2 a0 a1 a2]

    # ---- device part: block mean squares z[channel][block]
    def block_energies(self, data):
        x = _as_device_2d(data)
        n, ch = x.shape
        T_g, step = self.block_size, 0.25                                # 75 % overlap
        if n < T_g * self.rate:
            raise ValueError('Audio must have length greater than the block size.')
        T = n / self.rate
        num_blocks = int(np.round(((T - T_g) / (T_g * step))) + 1)
        j = np.arange(0, num_blocks)
        # meter.py: l = int(T_g * (j * step) * rate), u = int(T_g * (j * step + 1) * rate) -- the same float64 operations
        # in the same order, element-wise (int() and astype both truncate)
        lo = (T_g * (j * step) * self.rate).astype(np.int64)
        hi = (T_g * (j * step + 1) * self.rate).astype(np.int64)
        dev = x.device
        lo_d, hi_d = torch.from_numpy(lo).to(dev), torch.from_numpy(hi).to(dev)
        z = torch.empty((ch, num_blocks), dtype=torch.float64, device=dev)
        L = _lib.lib()
        ws = torch.empty(L.dam_loudness_workspace_bytes(n, ch), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.dam_loudness_block_energy(_lib.ptr(x), 1 if x.dtype == torch.float64 else 0, n, ch, x.stride(0),
                                                   x.stride(1), self._coef, _lib.ptr(lo_d), _lib.ptr(hi_d), num_blocks,
                                                   float(T_g * self.rate), _lib.ptr(z), _lib.ptr(ws), _lib.stream()),
                       'dam_loudness_block_energy')
        return z.cpu().numpy()

    # ---- host part: two-stage gating (pyloudnorm meter.py integrated_loudness)
    def integrated_loudness(self, data):
        z = self.block_energies(data)
        return gated_loudness(z)


def gated_loudness(z):
    """LUFS from block mean squares z[channel][block]: absolute gate -70, relative gate -10 LU (BS.1770-4)."""
    num_channels, num_blocks = z.shape
    G = np.array(_CHANNEL_GAINS[:num_channels])
    Gamma_a = -70.0
    with np.errstate(divide='ignore', invalid='ignore'):
        l = -0.691 + 10.0 * np.log10(np.sum(G[:, None] * z, axis=0))
        J_g = [j for j, l_j in enumerate(l) if l_j >= Gamma_a]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', category=RuntimeWarning)
            z_avg_gated = [np.mean([z[i, j] for j in J_g]) for i in range(num_channels)]
        Gamma_r = -0.691 + 10.0 * np.log10(np.sum([G[i] * z_avg_gated[i] for i in range(num_channels)])) - 10.0
        J_g = [j for j, l_j in enumerate(l) if (l_j > Gamma_r and l_j > Gamma_a)]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', category=RuntimeWarning)
            z_avg_gated = np.nan_to_num(np.array([np.mean([z[i, j] for j in J_g]) for i in range(num_channels)]))
        LUFS = -0.691 + 10.0 * np.log10(np.sum([G[i] * z_avg_gated[i] for i in range(num_channels)]))
    return float(LUFS)


def normalize_loudness(data, input_loudness, target_loudness):
    """``pyloudnorm.normalize.loudness``: constant gain so that a signal measured at input_loudness reads target_loudness."""
    delta_loudness = target_loudness - input_loudness
    gain = np.power(10.0, delta_loudness / 20.0)
    output = gain * data
    peak = float(output.abs().max()) if torch.is_tensor(output) else float(np.max(np.abs(output)))
    if peak >= 1.0:
        warnings.warn('Possible clipped samples in output.')
    return output
