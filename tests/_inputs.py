"""Seeded synthetic inputs shared by the golden generator (oracle/gen_golden.py carries the
same recipes) and the tests.  numpy only."""
import numpy as np


def make_audio(kind, n, seed):
    if kind == 'noise':
        return 0.1 * np.random.default_rng(seed).standard_normal(n)
    if kind == 'sweep':
        t = np.arange(n) / 44100.0
        return 0.5 * np.sin(2 * np.pi * (50.0 + 4000.0 * t) * t)
    if kind == 'impulse':
        x = np.zeros(n)
        x[n // 3] = 1.0
        return x
    if kind == 'silence':
        return np.zeros(n)
    raise ValueError(kind)


def model_input(b, s, f, t, seed):
    rng = np.random.default_rng(seed)
    x = (-20.0 + 15.0 * rng.standard_normal((b, s, f, t))).astype(np.float32)
    g = np.linspace(0.5, 1.5, s, dtype=np.float32)
    gt = (x * g[None, :, None, None]).sum(1) + rng.standard_normal((b, f, t)).astype(np.float32)
    return x, gt


def synthetic_clips(n_clips, n_stems, n_samples, channels=2, seed=1234, dtype=np.float32):
    """stems 0.1*N(0,1), mix = sum_s linspace(0.5,1.5,S)[s]*stem_s; [clips, S+1, n, ch]."""
    rng = np.random.default_rng(seed)
    stems = (0.1 * rng.standard_normal((n_clips, n_stems, n_samples, channels))).astype(dtype)
    g = np.linspace(0.5, 1.5, n_stems).astype(dtype)
    mix = (stems * g[None, :, None, None]).sum(1, keepdims=True)
    return np.concatenate([stems, mix], axis=1)


def feature_error(got, want_db):
    """SURVEY section 7 'hard parts': dB features are compared in the LINEAR domain relative to the
    frame peak (spectral nulls amplify rounding in dB) plus a loose absolute dB bound away from nulls."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want_db, dtype=np.float64)
    lin_g, lin_w = 10.0 ** (got / 20.0), 10.0 ** (want / 20.0)
    peak = lin_w.max(axis=0, keepdims=True)
    rel_lin = np.abs(lin_g - lin_w) / np.maximum(peak, 1e-30)
    strong = lin_w > 1e-3 * peak          # bins within 60 dB of the frame peak
    db_err = np.abs(got - want)[strong].max() if strong.any() else 0.0
    return rel_lin.max(), db_err
