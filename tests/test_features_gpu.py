"""GPU: dam_stft_logmag_f32 (through the C ABI) against the oracle and the reference's golden vectors.

Tolerance (hard parts): the f32 pipeline is compared in the linear domain relative
to the frame peak (<= 2e-6) and in dB (<= 2e-3 dB) on bins within 60 dB of the frame peak."""
import json
import os

import numpy as np
import pytest
import torch

from _inputs import feature_error, make_audio, synthetic_clips
from oracle import features_ref

pytestmark = pytest.mark.gpu
REL_LIN, ABS_DB = 2e-6, 2e-3


@pytest.fixture(scope='module')
def feats(dam_lib):
    from deep_audio_mixer_amd import features
    return features


def _check(got, want):
    rel, db = feature_error(got, want)
    assert rel <= REL_LIN, rel
    assert db <= ABS_DB, db


def test_golden_cases(feats, golden_dir):
    data = np.load(os.path.join(golden_dir, 'features.npz'))
    meta = json.load(open(os.path.join(golden_dir, 'features.json')))
    for c in meta['cases']:
        if c['dtype'] != 'f64':
            continue
        a = make_audio(c['kind'], c['n'], c['seed'])
        for tdt in (torch.float64, torch.float32):
            n_fft = c.get('n_fft', 2048)        # 2048 / even hop: the tuned kernel; anything else: the generic power-of-two one
            out = feats.stft_logmag(torch.from_numpy(a).to(tdt).cuda()[None], n_fft=n_fft, hop=c['hop'])[0].cpu().numpy()
            assert list(out.shape) == c['shape']
            if c['kind'] == 'silence':
                assert np.all(out == -100.0)
                continue
            if c['key'] + '_full' in data:
                _check(out, data[c['key'] + '_full'])
            want = features_ref.compute_features(a, n_fft, c['hop'])
            _check(out, want)
            samp = data[c['key'] + '_sample']        # straight dB comparison away from spectral nulls
            strong = samp > samp.max() - 60.0
            np.testing.assert_allclose(out[::37, ::5][strong], samp[strong], rtol=0, atol=ABS_DB)


@pytest.mark.parametrize('channels,dtype', [(2, np.float32), (1, np.float32), (2, np.float64), (1, np.float64)])
def test_batched_tracks_stereo_gain(feats, channels, dtype):
    rng = np.random.default_rng(3)
    n_tracks, n = 5, 44100 + 7          # ragged length: not a multiple of hop
    pcm = (0.1 * rng.standard_normal((n_tracks, n, channels))).astype(dtype)
    gains = rng.uniform(0.6, 1.4, n_tracks)
    out = feats.stft_logmag(torch.from_numpy(pcm).cuda(), hop=512,
                            gain=torch.from_numpy(gains).cuda()).cpu().numpy()
    for k in range(n_tracks):
        mono = features_ref.stereo_to_mono(pcm[k].astype(np.float64))
        want = features_ref.compute_features(features_ref.augment_audio(mono, gains[k]), 2048, 512)
        _check(out[k], want)


def test_normalize_and_edges(feats):
    a = make_audio('noise', 3000, 5)         # shortest legal-ish input: N > n_fft/2, every frame mirrored
    out = feats.stft_logmag(torch.from_numpy(a).cuda()[None], hop=1024, normalize=True)[0].cpu().numpy()
    want = features_ref.compute_features(a, 2048, 1024, normalize=True)
    assert out.shape == want.shape == (1025, 3)
    np.testing.assert_allclose(out, want, rtol=0, atol=5e-4)
    with pytest.raises(RuntimeError, match='DAM_ERR_BAD_ARG'):
        feats.stft_logmag(torch.zeros(1, 1024, device='cuda'), hop=1024)      # N <= n_fft/2: torch.stft raises too
    with pytest.raises(RuntimeError, match='DAM_ERR_UNSUPPORTED'):
        feats.stft_logmag(torch.zeros(1, 40960, device='cuda'), n_fft=1000, hop=256)      # not a power of two
    with pytest.raises(RuntimeError, match='DAM_ERR_UNSUPPORTED'):
        feats.stft_logmag(torch.zeros(1, 40960, device='cuda'), n_fft=32768, hop=256)      # above 16384: two LDS buffers do not fit
    with pytest.raises(RuntimeError, match='GPU only'):
        feats.stft_logmag(torch.zeros(1, 4096), hop=1024)


def test_full_size_properties(feats):
    """BASELINE config C3 size: 8 clips x 9 tracks x 3 s stereo.  Size-independent properties:
    linearity in dB (x2 amplitude = +6.0206 dB away from the floor) and batch independence."""
    pcm = torch.from_numpy(synthetic_clips(8, 8, 132300).reshape(72, 132300, 2)).cuda()
    out = feats.stft_logmag(pcm, hop=1024)
    assert out.shape == (72, 1025, 130)
    out2 = feats.stft_logmag(pcm * 2, hop=1024)
    d = (out2 - out)[out > -60]
    assert torch.all((d - 6.0206).abs() < 2e-3)
    single = feats.stft_logmag(pcm[17:18], hop=1024)
    assert torch.equal(single[0], out[17])
    want = features_ref.compute_features(pcm[40].double().mean(1).cpu().numpy(), 2048, 1024)
    _check(out[40].cpu().numpy(), want)


@pytest.mark.parametrize('n_fft,hop,channels,dtype', [(1024, 256, 2, np.float32), (4096, 1024, 1, np.float64), (512, 100, 2, np.float64),
                                                      (2048, 441, 2, np.float32), (64, 16, 1, np.float32), (8192, 2048, 2, np.float32),
                                                      (16384, 4096, 1, np.float32)])
def test_other_window_sizes(feats, n_fft, hop, channels, dtype):
    """compute_features(audio, window_size, hop_length) for windows other than 2048 and odd hops (generic kernel): stereo mean,
    gain, normalisation and the (stems, mix) output split behave as in the tuned kernel."""
    rng = np.random.default_rng(n_fft + hop)
    n = 20000 + 13
    pcm = (0.1 * rng.standard_normal((4, n, channels))).astype(dtype)
    gains = rng.uniform(0.6, 1.4, 4)
    out = feats.stft_logmag(torch.from_numpy(pcm).cuda(), n_fft=n_fft, hop=hop, gain=torch.from_numpy(gains).cuda()).cpu().numpy()
    assert out.shape == (4, n_fft // 2 + 1, 1 + n // hop)
    for k in (0, 3):
        mono = features_ref.stereo_to_mono(pcm[k].astype(np.float64))
        _check(out[k], features_ref.compute_features(features_ref.augment_audio(mono, gains[k]), n_fft, hop))
    nrm = feats.stft_logmag(torch.from_numpy(pcm[:1]).cuda(), n_fft=n_fft, hop=hop, normalize=True)[0].cpu().numpy()
    want = features_ref.compute_features(features_ref.stereo_to_mono(pcm[0].astype(np.float64)), n_fft, hop, normalize=True)
    np.testing.assert_allclose(nrm, want, rtol=0, atol=5e-4)
    x, gt = feats.stft_logmag_clips(torch.from_numpy(pcm.reshape(2, 2, n, channels)).cuda(), n_fft=n_fft, hop=hop)
    plain = feats.stft_logmag(torch.from_numpy(pcm).cuda(), n_fft=n_fft, hop=hop)
    assert torch.equal(x[:, 0], plain[[0, 2]]) and torch.equal(gt, plain[[1, 3]])
