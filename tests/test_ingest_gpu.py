"""GPU: the ingest side of the path -- page-locked staging pipes, the batched WAV -> front-end
loader, the device-side augmentation draw."""
import os
import wave

import numpy as np
import pytest
import torch

from oracle import features_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dam(dam_lib):
    import deep_audio_mixer_amd as pkg
    return pkg


def test_pinned_pipe_roundtrip(dam):
    from deep_audio_mixer_amd.staging import PinnedPipe
    pipe = PinnedPipe('cuda', piece_bytes=1 << 20)
    rng = np.random.default_rng(0)
    for shape, dt in (((3, 777777), np.float32), ((2, 100001), np.float64), ((5,), np.float32), ((1 << 18,), np.float32)):
        a = rng.standard_normal(shape).astype(dt)
        d = torch.empty(shape, dtype=torch.from_numpy(a).dtype, device='cuda')
        pipe.upload(d, a)
        assert np.array_equal(d.cpu().numpy(), a)
        d.mul_(2)
        back = pipe.download(d)
        assert back.dtype == dt and np.array_equal(back, 2 * a)
        out = np.empty_like(a)
        assert pipe.download(d, out=out) is out and np.array_equal(out, 2 * a)


def test_batch_stager_cycles_through_host_dataset(dam):
    from deep_audio_mixer_amd.staging import BatchStager
    host = torch.arange(10 * 6, dtype=torch.float32).reshape(10, 2, 3)
    pinned = torch.empty(host.shape, dtype=host.dtype, pin_memory=True)
    pinned.copy_(host)
    st = BatchStager(pinned, 4, 'cuda')           # 2 whole batches, then wraps around
    sink = torch.empty((4, 2, 3), device='cuda')
    for k in range(7):
        sink.copy_(st.next())
        want = host[(k % 2) * 4:(k % 2) * 4 + 4]
        assert torch.equal(sink.cpu(), want), k
    with pytest.raises(ValueError):
        BatchStager(host, 4, 'cuda')              # pageable memory is refused


def test_augment_gains_reproducible_and_uniform(dam):
    from deep_audio_mixer_amd import features
    g = features.augment_gains(321, 5, first_item=100, n_items=64).cpu().numpy()
    assert g.shape == (64, 5) and g.dtype == np.float32 and g.min() >= 0.6 and g.max() < 1.4
    for it, k in ((0, 0), (7, 4), (63, 2)):
        assert g[it, k] == features_ref.augment_gain_ref(321, 100 + it, k)
    # keyed by the GLOBAL item index: any batch composition gives the same draw for the same item
    sel = [163, 100, 131]
    g2 = features.augment_gains(321, 5, items=sel).cpu().numpy()
    assert np.array_equal(g2, g[[63, 0, 31]])
    assert not np.array_equal(features.augment_gains(322, 5, first_item=100, n_items=64).cpu().numpy(), g)
    big = features.augment_gains(1, 9, first_item=0, n_items=20000).cpu().numpy().ravel()
    assert abs(big.mean() - 1.0) < 2e-3 and abs(big.std() - 0.8 / np.sqrt(12)) < 2e-3
    hist = np.histogram(big, bins=8, range=(0.6, 1.4))[0] / big.size
    assert np.abs(hist - 0.125).max() < 5e-3


def _write_song(root, name, n, sr, rng, width=2):
    song = root / name / (name + '_STEMS_JOINED')
    song.mkdir(parents=True)
    for fn in ('%s_STEM_BASS.wav', '%s_STEM_DRUMS.wav', '%s_STEM_VOCALS.wav', '%s_STEM_OTHER.wav', '../%s_MIX.wav'):
        x = (rng.uniform(-0.5, 0.5, (n, 2)) * 32767).astype('<i2')
        with wave.open(str(song / (fn % name)), 'wb') as w:
            w.setnchannels(2), w.setsampwidth(width), w.setframerate(sr)
            w.writeframes(x.tobytes())


def test_iter_batches_equals_items(dam, tmp_path):
    """The batched ingest path (decode threads -> page-locked staging -> copy stream -> one launch per batch) yields what
    DataLoader over __getitem__ yields, ragged last batch and augmentation included."""
    from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
    sr = 8000
    rng = np.random.default_rng(5)
    _write_song(tmp_path, 'A', sr * 4 + 100, sr, rng)
    _write_song(tmp_path, 'B', sr * 3 + 5, sr, rng)
    d = MultitrackAudioDataset(str(tmp_path), chunk_length=1, sr=sr, seed=11)
    assert len(d) == 7
    items = [d[i] for i in range(len(d))]
    got = list(d.iter_batches(3, workers=4))
    assert [b[0].shape[0] for b in got] == [3, 3, 1]
    x = torch.cat([b[0] for b in got])
    gt = torch.cat([b[1] for b in got])
    for i, (xi, gi) in enumerate(items):
        assert torch.equal(x[i], xi) and torch.equal(gt[i], gi), i           # 16-bit PCM: float32 staging is lossless
    assert len(list(d.iter_batches(3, drop_last=True))) == 2
    sub = list(d.iter_batches(2, indices=[6, 1]))
    assert torch.equal(sub[0][0][0], items[6][0]) and torch.equal(sub[0][0][1], items[1][0])
    # augmentation: per-item device draws, identical through both paths and across runs
    a = MultitrackAudioDataset(str(tmp_path), chunk_length=1, sr=sr, seed=11, augment_data=True)
    xa = torch.cat([b[0] for b in a.iter_batches(4)])
    for i in (0, 5):
        assert torch.equal(xa[i], a[i][0])
    off = (xa[2] - items[2][0]).flatten(1)
    want = [20 * np.log10(features_ref.augment_gain_ref(11, 2, k)) for k in range(4)]
    assert torch.allclose(off.median(dim=1).values.cpu(), torch.tensor(want, dtype=torch.float32), atol=2e-3)
