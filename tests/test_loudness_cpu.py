"""CPU: the BS.1770 oracle against the standard's known answers (pyloudnorm itself is absent: see oracle/loudness_ref.py),
and the library's host-side coefficient entry point against the oracle's."""
import ctypes

import numpy as np
import pytest

from oracle import loudness_ref as ref


def sine(freq, dbfs, seconds, rate=48000, channels=1):
    t = np.arange(int(seconds * rate)) / rate
    x = 10 ** (dbfs / 20.0) * np.sin(2 * np.pi * freq * t)
    return np.stack([x] * channels, axis=1)


def test_bs1770_known_answers():
    # ITU-R BS.1770-4 Annex 1: 0 dBFS 997 Hz sine in a front channel -> -3.01 LKFS.  pyloudnorm's RBJ high pass has a
    # pass-band gain of 0.995 (the standard's table has b = [1, -2, 1]), so this design reads 0.04 LU low: -3.05.
    base = ref.integrated_loudness(sine(997, 0.0, 5.0), 48000)
    assert abs(base - (-3.01)) < 0.05 and abs(base - (-3.0517)) < 1e-3
    assert abs(ref.integrated_loudness(sine(997, -23.0, 5.0), 48000) - (base - 23.0)) < 1e-9      # level linearity
    assert abs(ref.integrated_loudness(sine(997, 0.0, 5.0, channels=2), 48000) - (base + 10 * np.log10(2.0))) < 1e-9
    # 44.1 kHz (the reference's rate): the coefficients are re-derived for the rate, same reading to 1e-3
    assert abs(ref.integrated_loudness(sine(997, -20.0, 4.0, rate=44100), 44100) - (base - 20.0)) < 2e-3


def test_gating_ignores_silence_and_rejects_short_input():
    x = sine(997, -20.0, 6.0)
    padded = np.concatenate([np.zeros((48000 * 6, 1)), x, np.zeros((48000 * 6, 1))])
    # ungated the reading would drop by 10*log10(3) = 4.8 LU; gated only the partly filled edge blocks pull it down
    assert abs(ref.integrated_loudness(padded, 48000) - ref.integrated_loudness(x, 48000)) < 0.3
    with pytest.raises(ValueError):
        ref.integrated_loudness(np.zeros(100), 48000)
    assert ref.integrated_loudness(np.zeros(48000), 48000) == -np.inf


def test_library_coefficients_match_oracle(dam_lib):
    lib = dam_lib            # host-only entry point: no GPU needed
    for rate in (44100.0, 48000.0, 16000.0):
        c = (ctypes.c_double * 12)()
        assert lib.dam_loudness_kweight_coeffs(rate, c) == 0
        np.testing.assert_allclose(np.array(list(c)).reshape(2, 6), ref.kweight_coefficients(rate), rtol=1e-14, atol=0)
    assert lib.dam_loudness_kweight_coeffs(0.0, (ctypes.c_double * 12)()) != 0
