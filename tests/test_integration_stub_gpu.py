"""GPU: the two binding stubs shown in INTEGRATION.md, executed as written (plain ctypes on libdam_hip.so, no package
code), against the oracle: the front-end replacing MultitrackAudioDataset.compute_features (data/dataset.py:145-155) and the
BS.1770 block energies replacing pyloudnorm's filter + block loop (data/dataset.py:126)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import features_ref, loudness_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib(dam_lib):
    l = ctypes.CDLL(os.path.join(ROOT, 'deep-audio-mixer_amd', 'libdam_hip.so'))
    l.dam_stft_logmag_f32.restype = ctypes.c_int
    l.dam_stft_logmag_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    l.dam_loudness_kweight_coeffs.argtypes = [ctypes.c_double, ctypes.c_void_p]
    l.dam_loudness_workspace_bytes.restype = ctypes.c_int64
    l.dam_loudness_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
    l.dam_loudness_block_energy.restype = ctypes.c_int
    l.dam_loudness_block_energy.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                            ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return l


def test_compute_features_stub(lib):
    tw = np.empty(2 * 2048, np.float32)
    lib.dam_stft_fill_twiddles_host(2048, tw.ctypes.data_as(ctypes.c_void_p))
    tw_d, win_d = torch.from_numpy(tw).cuda(), torch.hann_window(2048).cuda()

    def compute_features(audio, window_size=2048, hop_length=1024):
        pcm = torch.from_numpy(audio).cuda()
        n, ch = pcm.shape[0], (pcm.shape[1] if pcm.dim() == 2 else 1)
        out = torch.empty((window_size // 2 + 1, 1 + n // hop_length), device='cuda')
        rc = lib.dam_stft_logmag_f32(pcm.data_ptr(), int(pcm.dtype == torch.float64), 1, n, ch, n * ch,
                                     win_d.data_ptr(), tw_d.data_ptr(), None, window_size, hop_length, 1e-5, 0,
                                     out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc:
            raise RuntimeError('dam_stft_logmag_f32: %d' % rc)
        return out

    audio = (0.1 * np.random.default_rng(3).standard_normal(44100)).astype(np.float32)
    got = compute_features(audio).cpu().numpy()
    want = features_ref.compute_features(audio, 2048, 1024, np.float32)
    assert got.shape == want.shape == (1025, 44)
    lin = lambda db: 10.0 ** (np.asarray(db, np.float64) / 20.0)
    assert np.max(np.abs(lin(got) - lin(want)) / lin(want).max(axis=0, keepdims=True)) < 2e-6


def test_loudness_stub(lib):
    rate = 44100
    track = (0.1 * np.random.default_rng(5).standard_normal((rate * 3, 2))).astype(np.float32)
    coef = (ctypes.c_double * 12)()
    lib.dam_loudness_kweight_coeffs(ctypes.c_double(float(rate)), coef)
    x = torch.from_numpy(track).cuda()
    n, ch = x.shape
    nb = int(round((n / rate - 0.4) / 0.1)) + 1
    j = np.arange(nb)
    lo = (0.4 * (j * 0.25) * rate).astype(np.int64)
    hi = (0.4 * (j * 0.25 + 1) * rate).astype(np.int64)
    lo_d, hi_d = torch.from_numpy(lo).cuda(), torch.from_numpy(hi).cuda()
    z = torch.empty((ch, nb), dtype=torch.float64, device='cuda')
    ws = torch.empty(lib.dam_loudness_workspace_bytes(n, ch), dtype=torch.uint8, device='cuda')
    rc = lib.dam_loudness_block_energy(x.data_ptr(), int(x.dtype == torch.float64), n, ch, x.stride(0), x.stride(1), coef,
                                       lo_d.data_ptr(), hi_d.data_ptr(), nb, ctypes.c_double(0.4 * rate), z.data_ptr(),
                                       ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    np.testing.assert_allclose(z.cpu().numpy(), loudness_ref.block_energies(track.astype(np.float64), rate), rtol=1e-9)
