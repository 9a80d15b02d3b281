"""ORACLE (test infrastructure only -- never imported by the product): CPU restatement of the BS.1770 meter the
reference calls through the third-party package ``pyloudnorm`` (imported at data/dataset.py:5, evaluation.py:7,
models/baselines/mean_loudness_model.py:1; not vendored in /root/reference and not installed here; the reference pins
no version -- this follows pyloudnorm 0.1.x: ``iirfilter.py`` (RBJ biquads), ``meter.py`` (integrated_loudness),
``normalize.py`` (loudness)).

Parity status: UNPINNED against pyloudnorm itself (absent, and the reference holds no loudness fixture).  Anchored
instead on the standard's known answer (tests/test_loudness_cpu.py): a 0 dBFS 997 Hz sine in one front channel reads
-3.01 LKFS (ITU-R BS.1770-4, Annex 1); this biquad design reads -3.05 (its RBJ high pass has a 0.995 pass-band gain where
the standard's table has b = [1, -2, 1]) -- pyloudnorm's own test-suite asserts -3.0523 for its sine fixture.  Level
linearity, channel summation and silence gating are checked exactly.

numpy/scipy float64; scipy.signal.lfilter is what pyloudnorm applies.
"""
import warnings

import numpy as np
import scipy.signal


def kweight_coefficients(rate):
    """[stage][b0 b1 b2 a0 a1 a2], stage 0 = high shelf (G 4 dB, Q 1/sqrt2, 1500 Hz), 1 = high pass (Q 0.5, 38 Hz)."""
    out = []
    for G, Q, fc, kind in ((4.0, 1.0 / np.sqrt(2), 1500.0, 'high_shelf'), (0.0, 0.5, 38.0, 'high_pass')):
        A = 10 ** (G / 40.0)
        w0 = 2.0 * np.pi * (fc / rate)
        alpha = np.sin(w0) / (2.0 * Q)
        if kind == 'high_shelf':
            b0 = A * ((A + 1) + (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha)
            b1 = -2 * A * ((A - 1) + (A + 1) * np.cos(w0))
            b2 = A * ((A + 1) + (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha)
            a0 = (A + 1) - (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha
            a1 = 2 * ((A - 1) - (A + 1) * np.cos(w0))
            a2 = (A + 1) - (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha
        else:
            b0 = (1 + np.cos(w0)) / 2
            b1 = -(1 + np.cos(w0))
            b2 = (1 + np.cos(w0)) / 2
            a0 = 1 + alpha
            a1 = -2 * np.cos(w0)
            a2 = 1 - alpha
        out.append([b0 / a0, b1 / a0, b2 / a0, 1.0, a1 / a0, a2 / a0])
    return np.array(out)


def block_energies(data, rate, block_size=0.400):
    """z[channel][block] as pyloudnorm meter.py computes them (same int() truncations)."""
    x = np.asarray(data, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    n, ch = x.shape
    if ch > 5:
        raise ValueError('Audio must have five channels or less.')
    if n < block_size * rate:
        raise ValueError('Audio must have length greater than the block size.')
    y = x.copy()
    for c in kweight_coefficients(rate):
        for i in range(ch):
            y[:, i] = scipy.signal.lfilter(c[:3], c[3:], y[:, i])
    T_g, step = block_size, 0.25
    T = n / rate
    num_blocks = int(np.round(((T - T_g) / (T_g * step))) + 1)
    z = np.zeros((ch, num_blocks))
    for i in range(ch):
        for j in range(num_blocks):
            l = int(T_g * (j * step) * rate)
            u = int(T_g * (j * step + 1) * rate)
            z[i, j] = (1.0 / (T_g * rate)) * np.sum(np.square(y[l:u, i]))
    return z


def integrated_loudness(data, rate, block_size=0.400):
    z = block_energies(data, rate, block_size)
    ch, nb = z.shape
    G = [1.0, 1.0, 1.0, 1.41, 1.41]
    Gamma_a = -70.0
    with np.errstate(divide='ignore', invalid='ignore'), warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        l = [-0.691 + 10.0 * np.log10(np.sum([G[i] * z[i, j] for i in range(ch)])) for j in range(nb)]
        J_g = [j for j, l_j in enumerate(l) if l_j >= Gamma_a]
        z_avg = [np.mean([z[i, j] for j in J_g]) for i in range(ch)]
        Gamma_r = -0.691 + 10.0 * np.log10(np.sum([G[i] * z_avg[i] for i in range(ch)])) - 10.0
        J_g = [j for j, l_j in enumerate(l) if (l_j > Gamma_r and l_j > Gamma_a)]
        z_avg = np.nan_to_num(np.array([np.mean([z[i, j] for j in J_g]) for i in range(ch)]))
        return float(-0.691 + 10.0 * np.log10(np.sum([G[i] * z_avg[i] for i in range(ch)])))


def normalize_loudness(data, input_loudness, target_loudness):
    return np.power(10.0, (target_loudness - input_loudness) / 20.0) * data
