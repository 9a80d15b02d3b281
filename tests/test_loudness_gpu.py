"""GPU: BS.1770 meter (HIP K-weighting filter + block energies through the C ABI, host gating) against the CPU oracle.
float64 recurrences evaluated in a different association order (chunked state propagation vs one sequential lfilter):
tolerance 1e-9 relative on block energies, 1e-8 LU on the loudness."""
import numpy as np
import pytest
import torch

from oracle import loudness_ref as ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def loud(dam_lib):
    from deep_audio_mixer_amd import loudness
    return loudness


def music_like(n, ch, seed, rate=44100):
    r = np.random.RandomState(seed)
    t = np.arange(n) / rate
    env = 0.5 + 0.5 * np.sin(2 * np.pi * 0.3 * t + r.rand())
    x = np.stack([env * (0.2 * np.sin(2 * np.pi * (110 * (i + 1)) * t) + 0.05 * r.randn(n)) for i in range(ch)], axis=1)
    x[n // 3: n // 3 + rate] = 0.0            # a second of digital silence: gated blocks
    return x


@pytest.mark.parametrize('n,ch,rate,dtype', [(44100 * 7 + 123, 2, 44100, np.float64), (48000 * 3, 1, 48000, np.float32),
                                            (17640, 1, 44100, np.float64), (44100 * 4 + 1, 5, 44100, np.float32),
                                            (1025 * 9 + 16000, 2, 16000, np.float64)])
def test_block_energies_and_loudness_match_oracle(loud, n, ch, rate, dtype):
    x = music_like(n, ch, n % 97, rate).astype(dtype)
    m = loud.Meter(rate)
    z = m.block_energies(x if ch > 1 else x[:, 0])
    want = ref.block_energies(x.astype(np.float64), rate)
    assert z.shape == want.shape
    np.testing.assert_allclose(z, want, rtol=1e-9, atol=1e-18)
    got = m.integrated_loudness(torch.from_numpy(x).cuda())
    assert abs(got - ref.integrated_loudness(x.astype(np.float64), rate)) < 1e-8


def test_strided_views_known_answer_and_errors(loud):
    rate = 48000
    t = np.arange(rate * 5) / rate
    tone = np.sin(2 * np.pi * 997 * t)
    m = loud.Meter(rate)
    assert abs(m.integrated_loudness(tone) - (-3.01)) < 0.05                     # BS.1770-4 Annex 1 (see the CPU test)
    # [channels, samples] storage passed as its transpose, as every reference call site does (tracks[name].T)
    planar = torch.from_numpy(np.stack([tone, 0.5 * tone])).cuda()
    assert abs(m.integrated_loudness(planar.T) - ref.integrated_loudness(planar.T.cpu().numpy(), rate)) < 1e-8
    with pytest.raises(ValueError):
        m.integrated_loudness(np.zeros(100))
    with pytest.raises(ValueError):
        m.integrated_loudness(np.zeros((rate, 6)))
    assert m.integrated_loudness(np.zeros(rate)) == -np.inf
    out = loud.normalize_loudness(planar.T, -10.0, -16.0)
    np.testing.assert_allclose(out.cpu().numpy(), ref.normalize_loudness(planar.T.cpu().numpy(), -10.0, -16.0), rtol=1e-15)


def test_full_song_length_property(loud):
    """4-minute stereo stem: gain linearity (+6.0206 dB in, +6.0206 LU out) and agreement with the oracle."""
    rate, n = 44100, 44100 * 240
    x = music_like(n, 2, 5, rate).astype(np.float32)
    m = loud.Meter(rate)
    xd = torch.from_numpy(x).cuda()
    a = m.integrated_loudness(xd)
    b = m.integrated_loudness(xd * 2.0)
    assert abs((b - a) - 20 * np.log10(2.0)) < 1e-6
    assert abs(a - ref.integrated_loudness(x.astype(np.float64), rate)) < 1e-8


def test_mean_loudness_baseline(loud):
    from deep_audio_mixer_amd.models.baselines.mean_loudness_model import MeanLoudnessModel
    rate = 44100
    tracks = {k: torch.from_numpy(music_like(rate * 3, 2, i, rate).T.copy()).cuda() for i, k in enumerate(('bass', 'drums', 'vocals', 'other'))}
    target = {'bass': -25.0, 'drums': -20.0, 'vocals': -18.0, 'other': -22.0}
    out = MeanLoudnessModel(target, rate).forward(tracks)
    m = loud.Meter(rate)
    for k in target:
        assert out[k].shape == tracks[k].shape
        assert abs(m.integrated_loudness(out[k].T) - target[k]) < 1e-6


def test_evaluator_and_dataset_mean_loudness(loud):
    """evaluation.py:39-53 profile/error and data/dataset.py:115-130 mean loudness against the oracle."""
    from deep_audio_mixer_amd.evaluation import LoudnessEvaluator
    from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
    rate = 44100
    keys = ('bass', 'drums', 'vocals', 'other')
    a = {k: music_like(rate * 3, 2, 10 + i, rate).T.copy() for i, k in enumerate(keys)}
    b = {k: a[k] * g for k, g in zip(keys, (1.0, 0.5, 2.0, 1.0))}
    ev = LoudnessEvaluator(rate, keys)
    pa, _ = ev._sum_and_evaluate_tracks({k: torch.from_numpy(v).cuda() for k, v in a.items()}, None)
    pb, err = ev._sum_and_evaluate_tracks({k: torch.from_numpy(v).cuda() for k, v in b.items()}, pa)
    want_a = np.array([ref.integrated_loudness(a[k].T, rate) for k in keys]); want_a -= want_a.mean()
    want_b = np.array([ref.integrated_loudness(b[k].T, rate) for k in keys]); want_b -= want_b.mean()
    np.testing.assert_allclose(list(pa.values()), want_a, atol=1e-8)
    assert abs(err - np.mean(np.abs(want_a - want_b))) < 1e-8
    mixed = ev.sum_tracks_to_target({k: torch.from_numpy(v).cuda() for k, v in a.items()}, -20.0)
    assert abs(ev.meter.integrated_loudness(mixed) - (-20.0)) < 1e-6
    songs = {'s%d' % i: {k: music_like(rate * 2, 2, 30 + 4 * i + q, rate) for q, k in enumerate(keys + ('mix',))} for i in range(2)}
    ds = MultitrackAudioDataset.from_arrays(songs, tracklist=list(keys) + ['mix'], chunk_length=1, sr=rate)
    got = ds.compute_mean_loudness()
    for k in keys:
        assert abs(got[k] - np.mean([ref.integrated_loudness(songs[s][k], rate) for s in songs])) < 1e-8
