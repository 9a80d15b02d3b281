"""CPU: host-side logic of the drop-in API (index arithmetic, WAV reading, inference helpers, state_dict
compatibility) and the data-parallel pieces over gloo with world_size 2."""
import json
import os
import socket
import wave

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import deep_audio_mixer_amd  # noqa: F401
from deep_audio_mixer_amd import distributed as ddist
from deep_audio_mixer_amd import inference_utils
from deep_audio_mixer_amd.data import dataset_utils
from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
from oracle import models_ref


def test_interpolate_mask_and_db(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'inference.json')))
    for c in g['interpolate_mask']:
        np.testing.assert_array_equal(inference_utils.interpolate_mask(np.array(c['mask']), c['n']), np.array(c['out']))
    np.testing.assert_allclose([dataset_utils.scalar_dB_to_amplitude(v) for v in g['db']], g['db_to_amplitude'], rtol=1e-15)
    assert inference_utils._savgol_window(60) == 15 and inference_utils._savgol_window(64) == 17


def _songs(durations_s, sr=8000, tracks=('bass', 'drums', 'vocals', 'other', 'mix')):
    rng = np.random.default_rng(0)
    return {'song%d' % i: {t: rng.standard_normal((int(d * sr), 2)) for t in tracks} for i, d in enumerate(durations_s)}


def test_dataset_index_arithmetic():
    """data/dataset.py:56-75,97-113: len = sum floor(dur/chunk); chunks of a song are consecutive."""
    d = MultitrackAudioDataset.from_arrays(_songs([7.3, 2.0, 4.9]), chunk_length=2, sr=8000, seed=3, device='cpu')
    assert d.get_tracklist() == ['bass', 'drums', 'vocals', 'other', 'mix']
    assert d.get_num_songs() == 3 and sorted(d.get_song_durations()) == [2, 4, 6]     # int seconds, trimmed (:70-73)
    assert all(isinstance(v, int) for v in d.get_song_durations())
    per_song = [int(x / 2) for x in d.get_song_durations()]
    assert len(d) == sum(per_song) == 6
    seen = [d._calculate_song_index(i) for i in range(len(d))]
    want = [(s, c) for s, n in enumerate(per_song) for c in range(n)]
    assert seen == want
    assert d._calculate_song_index(len(d) + 5)[0] == 2          # past the end: stays on the last song (reference loop)
    # the bisection against the reference's linear walk (data/dataset.py:97-113), songs shorter than a chunk included
    d2 = MultitrackAudioDataset.from_arrays(_songs([1.0, 6.5, 0.5, 0.7, 4.2, 1.9, 9.0, 1.1]), chunk_length=2, sr=8000, seed=5, device='cpu')

    def walk(chunk_i):
        song_i, n = 0, int(d2.song_durations[0] / 2)
        while chunk_i >= n and song_i < len(d2.songlist) - 1:
            chunk_i -= n
            song_i += 1
            n = int(d2.song_durations[song_i] / 2)
        return song_i, chunk_i
    assert [d2._calculate_song_index(i) for i in range(len(d2) + 3)] == [walk(i) for i in range(len(d2) + 3)]
    np.testing.assert_array_equal(d._stereo_to_mono(np.array([[1.0, 3.0], [2.0, -2.0]])), [2.0, 0.0])


def test_wav_partial_read(tmp_path):
    sr, n = 8000, 5000
    x = (np.random.default_rng(1).uniform(-1, 1, (n, 2)) * 32767).astype('<i2')
    song = tmp_path / 'A' / 'A_STEMS_JOINED'
    song.mkdir(parents=True)
    for name in ('A_STEM_BASS.wav', 'A_STEM_DRUMS.wav', 'A_STEM_VOCALS.wav', 'A_STEM_OTHER.wav', '../A_MIX.wav'):
        with wave.open(str(song / name), 'wb') as w:
            w.setnchannels(2), w.setsampwidth(2), w.setframerate(sr)
            w.writeframes(x.tobytes())
    a, got_sr = dataset_utils.read_wav(str(tmp_path / 'A' / 'A_MIX.wav'), 1000, 3000)
    assert got_sr == sr and a.shape == (2000, 2)
    np.testing.assert_array_equal(a, x[1000:3000] / 32768.0)          # soundfile's int16 normalisation
    d = MultitrackAudioDataset(str(tmp_path), chunk_length=1, sr=sr, device='cpu')
    assert len(d) == 0 and d.get_num_songs() == 1                     # 0.625 s < one chunk
    tracks = dataset_utils.load_tracks(str(tmp_path), 'A', sr=sr)
    assert tracks['drums'].shape == (2, n) and tracks['drums'].dtype == np.float32     # librosa.load(mono=False)
    # a rate mismatch is resampled on load, as librosa.load(path, sr=sr) does (data/dataset_utils.py:53-83): 8 kHz files asked
    # for at 44.1 kHz -- length ceil(n * 44100 / 8000), and a tone keeps its frequency and level (the resampler itself is scipy's
    # polyphase FIR: parity with librosa's soxr / resampy is unpinned, stated in the warning)
    t = np.arange(n) / sr
    tone = (0.5 * np.sin(2 * np.pi * 440.0 * t)[:, None] * np.ones((1, 2)) * 32767).astype('<i2')
    with wave.open(str(tmp_path / 'A' / 'A_MIX.wav'), 'wb') as w:
        w.setnchannels(2), w.setsampwidth(2), w.setframerate(sr)
        w.writeframes(tone.tobytes())
    with pytest.warns(RuntimeWarning, match='resampl'):
        up = dataset_utils.load_tracks(str(tmp_path), 'A', tracklist=('mix',))['mix']
    assert up.shape == (2, -(-n * 44100 // sr)) and up.dtype == np.float32
    mid = up[0, 4410:-4410].astype(np.float64)
    want = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(up.shape[1]) / 44100.0)[4410:-4410]
    assert np.abs(mid - want).max() < 2e-3


def test_wav_native_reads_from_many_threads(tmp_path):
    """read_wav_native(out=...) is the ingest path: positioned reads (os.preadv) on a cached descriptor straight into the
    caller's buffer.  Eight threads read overlapping chunks of two files at once, every chunk must be the file's own samples
    (a shared file position or a descriptor closed under a reader would show here); a rewritten file is read anew."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(11)
    data = {}
    for name, ch in (('a.wav', 2), ('b.wav', 1)):
        x = rng.integers(-30000, 30000, (50000, ch), dtype=np.int16)
        with wave.open(str(tmp_path / name), 'wb') as w:
            w.setnchannels(ch), w.setsampwidth(2), w.setframerate(44100)
            w.writeframes(x.tobytes())
        data[name] = x

    def one(i):
        name = 'a.wav' if i % 2 else 'b.wav'
        x = data[name]
        lo = (i * 977) % 40000
        out = np.empty((7001, x.shape[1]), dtype=np.int16)
        got, rate = dataset_utils.read_wav_native(str(tmp_path / name), lo, lo + 7001, out=out)
        return rate == 44100 and got is out and np.array_equal(out, x[lo:lo + 7001])

    with ThreadPoolExecutor(8) as pool:
        assert all(pool.map(one, range(200)))
    # the same path with other contents (new size): the header and descriptor caches key on (path, mtime, size)
    y = rng.integers(-100, 100, (1234, 2), dtype=np.int16)
    with wave.open(str(tmp_path / 'a.wav'), 'wb') as w:
        w.setnchannels(2), w.setsampwidth(2), w.setframerate(22050)
        w.writeframes(y.tobytes())
    out = np.empty((1000, 2), dtype=np.int16)
    got, rate = dataset_utils.read_wav_native(str(tmp_path / 'a.wav'), 100, 1100, out=out)
    assert rate == 22050 and np.array_equal(out, y[100:1100])
    with pytest.raises(ValueError):                     # a chunk past the end of the data: short read, reported
        dataset_utils.read_wav_native(str(tmp_path / 'a.wav'), 1000, 1300, out=np.empty((300, 2), dtype=np.int16))


def test_wav_extensible_float_and_musdb_layout(tmp_path):
    """WAVE_FORMAT_EXTENSIBLE (24-bit) and IEEE-float files, which the stdlib wave module rejects, decode like
    soundfile; MUSDB18-HQ layout loader; split_songlist."""
    import struct
    rng = np.random.default_rng(2)
    n, ch = 700, 2
    v = rng.integers(-2 ** 23, 2 ** 23, (n, ch))
    raw = b''.join(int(s).to_bytes(3, 'little', signed=True) for s in v.flatten())
    guid_tail = b'\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71'
    fmt = struct.pack('<HHIIHH', 0xFFFE, ch, 44100, 44100 * ch * 3, ch * 3, 24) + struct.pack('<HHI', 22, 24, 3) + \
        struct.pack('<H', 1) + guid_tail
    body = b'WAVE' + b'fmt ' + struct.pack('<I', len(fmt)) + fmt + b'LIST' + struct.pack('<I', 3) + b'abc\x00' + \
        b'data' + struct.pack('<I', len(raw)) + raw
    song = tmp_path / 'M'
    song.mkdir()
    for name in ('bass', 'drums', 'vocals', 'other', 'mixture'):
        (song / (name + '.wav')).write_bytes(b'RIFF' + struct.pack('<I', len(body)) + body)
    a, sr = dataset_utils.read_wav(str(song / 'bass.wav'), 10, 20)
    assert sr == 44100 and np.array_equal(a, v[10:20] / 8388608.0)
    a32, _ = dataset_utils.read_wav(str(song / 'bass.wav'), dtype=np.float32)
    assert a32.dtype == np.float32 and np.array_equal(a32.astype(np.float64), v / 8388608.0)    # 24-bit PCM is exact in f32
    tracks = dataset_utils.load_tracks_musdb18(str(tmp_path), 'M')
    assert sorted(tracks) == ['bass', 'drums', 'mix', 'other', 'vocals'] and tracks['mix'].shape == (2, n)
    f = rng.standard_normal((100, 1)).astype('<f4')
    fmt = struct.pack('<HHIIHH', 3, 1, 16000, 16000 * 4, 4, 32)
    body = b'WAVE' + b'fmt ' + struct.pack('<I', len(fmt)) + fmt + b'data' + struct.pack('<I', f.nbytes) + f.tobytes()
    (tmp_path / 'f.wav').write_bytes(b'RIFF' + struct.pack('<I', len(body)) + body)
    a, sr = dataset_utils.read_wav(str(tmp_path / 'f.wav'))
    assert sr == 16000 and np.array_equal(a, f.astype(np.float64))
    with pytest.raises(ValueError):
        (tmp_path / 'bad.wav').write_bytes(b'RIFFxxxxWAVEjunk')
        dataset_utils.read_wav(str(tmp_path / 'bad.wav'))
    np.random.seed(0)
    tr, va, te = dataset_utils.split_songlist(['s%d' % i for i in range(10)], (0.5, 0.25, 0.25), summary=False)
    assert len(tr) == 5 and len(va) == 2 and len(te) == 3 and sorted(tr + va + te) == ['s%d' % i for i in range(10)]


def test_reference_checkpoint_roundtrip():
    """A reference-keyed state_dict (built by the oracle, whose keys are pinned to the reference's by the golden
    test) loads into the product model and comes back unchanged, per-stem head keys included."""
    from deep_audio_mixer_amd.models.model_scalar_1s import MixingModelScalar1s
    ref = models_ref.closed_form_fill(models_ref.RefMixingModelScalar1s())
    m = MixingModelScalar1s()
    m.load_state_dict(ref.state_dict())
    out = m.state_dict()
    assert list(out.keys()) == list(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert torch.equal(out[k], v), k
    with pytest.raises(RuntimeError, match='GPU only'):
        m(torch.zeros(1, 4, 1025, 87))


def test_shard_indices():
    parts = [ddist.shard_indices(11, r, 3) for r in range(3)]
    assert parts == [[0, 3, 6], [1, 4, 7], [2, 5, 8]]
    assert ddist.shard_indices(11, 1, 3, drop_last=False) == [1, 4, 7, 10]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, w, _ = ddist.init_process_group('gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                         # deliberately different replicas before the broadcast
    model = models_ref.RefMixingModelScalar1s(n_stems=2, input_shape=(64, 48))
    ddist.broadcast_module(model)
    for mod in model.modules():
        if hasattr(mod, 'dropout_p'):
            mod.dropout_p = -1
    sampler = ddist.DistributedChunkSampler(4, rank, world)
    g = torch.Generator().manual_seed(5)
    xs, gts = torch.randn(4, 2, 64, 48, generator=g), torch.randn(4, 64, 48, generator=g)
    idx = list(sampler)
    masked, _ = model(xs[idx])
    torch.nn.functional.mse_loss(masked, gts[idx]).backward()
    bucket = ddist.GradBucket(model.parameters())
    flat = bucket.all_reduce_mean().clone()
    gains = ddist.all_gather_gains(torch.full((2, 3), float(rank)) + torch.arange(2.0)[:, None] * 10)
    torch.save({'flat': flat, 'idx': idx, 'w0': next(model.parameters()).detach().clone(), 'gains': gains},
               os.path.join(out_dir, 'r%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_data_parallel_gloo_world2(tmp_path):
    """2 ranks over gloo: broadcast makes replicas identical, the flat bucket holds the MEAN of the per-replica
    gradients (local BatchNorm statistics), every rank ends with the same gradients."""
    port = _free_port()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'r0.pt'), torch.load(tmp_path / 'r1.pt')
    assert r0['idx'] == [0, 2] and r1['idx'] == [1, 3]
    assert torch.equal(r0['w0'], r1['w0']) and torch.equal(r0['flat'], r1['flat'])
    # single-process restatement: same replica, the two micro-batches separately, gradients averaged
    torch.manual_seed(100)
    model = models_ref.RefMixingModelScalar1s(n_stems=2, input_shape=(64, 48))
    for mod in model.modules():
        if hasattr(mod, 'dropout_p'):
            mod.dropout_p = -1
    g = torch.Generator().manual_seed(5)
    xs, gts = torch.randn(4, 2, 64, 48, generator=g), torch.randn(4, 64, 48, generator=g)
    flats = []
    for idx in ([0, 2], [1, 3]):
        model.zero_grad()
        masked, _ = model(xs[idx])
        torch.nn.functional.mse_loss(masked, gts[idx]).backward()
        flats.append(torch.cat([p.grad.flatten() for p in model.parameters()]))
    want = (flats[0] + flats[1]) / 2
    assert torch.allclose(r0['flat'], want, rtol=1e-5, atol=1e-7)
    # gathered gains come back in chunk order (rank-strided)
    assert r0['gains'][:, 0].tolist() == [0.0, 1.0, 10.0, 11.0]


class _CutNet(torch.nn.Module):
    """Oracle RefMixingModelScalar1s with the activation at a bucket boundary exposed (what the product models' `tap` does)."""

    def __init__(self):
        super().__init__()
        self.m = models_ref.RefMixingModelScalar1s(n_stems=2, input_shape=(64, 48))
        for mod in self.m.modules():
            if hasattr(mod, 'dropout_p'):
                mod.dropout_p = -1

    def forward(self, x, tap):
        y = x
        for i in range(1, 6):
            y = getattr(self.m, 'conv_b%d' % i)(y)
            if i == 3:
                tap.append(y)
        return self.m._run_heads(x, y)

    def late(self):
        return [p for n, p in self.m.named_parameters() if not n.startswith(('conv_b1.', 'conv_b2.', 'conv_b3.'))]

    def early(self):
        return [p for n, p in self.m.named_parameters() if n.startswith(('conv_b1.', 'conv_b2.', 'conv_b3.'))]


def _staged_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    ddist.init_process_group('gloo')
    torch.manual_seed(7)
    net = _CutNet()
    g = torch.Generator().manual_seed(5)
    xs, gts = torch.randn(4, 2, 64, 48, generator=g), torch.randn(4, 64, 48, generator=g)
    idx = ddist.shard_indices(4, rank, world)
    tap = []
    masked, _ = net(xs[idx], tap)
    loss = torch.nn.functional.mse_loss(masked, gts[idx])
    # the step engine's order: late bucket first and on the wire while the early layers' backward runs
    late_b, early_b = ddist.GradBucket(net.late()), ddist.GradBucket(net.early())
    grads, dmid = ddist.backward_late(loss, net.late(), tap[0])
    assert all(p.grad is None for p in net.early())          # nothing in front of the boundary has been touched
    late_b.fill(grads)
    w1 = late_b.start_all_reduce()
    ddist.backward_early(tap[0], dmid, net.early())
    early_b.fill()
    w0 = early_b.start_all_reduce()
    late_b.finish(w1), early_b.finish(w0)
    torch.save({'late': late_b.flat.clone(), 'early': early_b.flat.clone()}, os.path.join(out_dir, 's%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_staged_backward_two_buckets_gloo_world2(tmp_path):
    """engine.TrainStep's multi-rank schedule on the CPU: backward cut at a bucket boundary (autograd.grad down to the
    boundary, backward from it), each bucket all-reduced asynchronously as soon as it exists == one plain backward per
    replica with the gradients averaged."""
    port = _free_port()
    mp.spawn(_staged_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 's0.pt'), torch.load(tmp_path / 's1.pt')
    assert torch.equal(r0['late'], r1['late']) and torch.equal(r0['early'], r1['early'])
    torch.manual_seed(7)
    net = _CutNet()
    g = torch.Generator().manual_seed(5)
    xs, gts = torch.randn(4, 2, 64, 48, generator=g), torch.randn(4, 64, 48, generator=g)
    acc = {'late': 0, 'early': 0}
    for idx in ([0, 2], [1, 3]):
        net.zero_grad()
        masked, _ = net(xs[idx], [])
        torch.nn.functional.mse_loss(masked, gts[idx]).backward()
        acc['late'] = acc['late'] + torch.cat([p.grad.flatten() for p in net.late()]) / 2
        acc['early'] = acc['early'] + torch.cat([p.grad.flatten() for p in net.early()]) / 2
    # (conv biases in front of a training-mode BatchNorm have an exactly-zero true gradient: rounding noise only)
    for k in ('late', 'early'):
        assert torch.allclose(r0[k], acc[k], rtol=1e-4, atol=1e-5 * acc[k].abs().max().item()), k


def test_bench_refuses_to_run_fewer_ranks():
    """`python bench.py --gpus N` with no WORLD_SIZE must start N ranks itself or fail -- never report a 1-rank line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    import bench
    if (bench.visible_gpu_count() or 0) >= 2:        # counted from sysfs: this process never initialises a GPU runtime
        pytest.skip('multi-GPU host: the spawn path would run for real')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'refusing' in r.stderr and '{' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=dict(env, WORLD_SIZE='1'),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE' in r.stderr


def test_bench_counts_gpus_from_sysfs_without_a_gpu_runtime(tmp_path, monkeypatch):
    """bench.visible_gpu_count(): KFD topology nodes with SIMDs whose render node is accessible, narrowed by
    HIP_VISIBLE_DEVICES -- no torch.cuda / HIP call in the parent of an N-rank run."""
    import builtins
    import glob as globmod
    import bench
    nodes = tmp_path / 'nodes'
    props = {0: 'cpu_cores_count 64\nsimd_count 0\ndrm_render_minor -1\n',
             1: 'cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor 128\n',
             2: 'cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor 129\n',
             3: 'cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor 130\n'}
    for i, txt in props.items():
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / 'properties').write_text(txt)
    real_glob, real_exists, real_access, real_open = globmod.glob, os.path.exists, os.access, builtins.open
    monkeypatch.setattr(globmod, 'glob', lambda pat: real_glob(str(nodes / '*' / 'properties')) if 'kfd' in pat else real_glob(pat))
    visible = {'/dev/dri/renderD128', '/dev/dri/renderD129'}            # the third GPU belongs to another container
    monkeypatch.setattr(os.path, 'exists', lambda q: q in visible or (not str(q).startswith('/dev/dri') and real_exists(q)))
    monkeypatch.setattr(os, 'access', lambda q, m: q in visible or (not str(q).startswith('/dev/dri') and real_access(q, m)))
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count() == 2
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '1')
    assert bench.visible_gpu_count() == 1


def _write_wav_song(root, name, n, sr, rng):
    song = root / name / (name + '_STEMS_JOINED')
    song.mkdir(parents=True)
    out = {}
    for track, fn in (('bass', '%s_STEM_BASS.wav'), ('drums', '%s_STEM_DRUMS.wav'), ('vocals', '%s_STEM_VOCALS.wav'),
                      ('other', '%s_STEM_OTHER.wav'), ('mix', '../%s_MIX.wav')):
        x = (rng.uniform(-0.5, 0.5, (n, 2)) * 32767).astype('<i2')
        with wave.open(str(song / (fn % name)), 'wb') as w:
            w.setnchannels(2), w.setsampwidth(2), w.setframerate(sr)
            w.writeframes(x.tobytes())
        out[track] = x
    return out


def test_dataloader_workers_return_host_pcm_and_touch_no_gpu(tmp_path):
    """training.ipynb cell 6 as written -- DataLoader(d_train, batch_size, shuffle=False, num_workers=6, pin_memory=True,
    ...) -- over data/dataset.py:270-292: inside a worker __getitem__ returns the decoded chunk as host PCM, the default
    collate makes ONE HostPcmBatch per batch (shared memory), in order, ragged last batch included.  Runs here, without a
    GPU: that the workers touch no GPU API is the point (pin_memory is off only because this container has no device to
    page-lock for; the -m gpu twin of this test runs the cell verbatim)."""
    from torch.utils.data import DataLoader
    from deep_audio_mixer_amd.data.dataset import HostPcmBatch
    sr = 8000
    rng = np.random.default_rng(5)
    files = {'A': _write_wav_song(tmp_path, 'A', sr * 4 + 100, sr, rng), 'B': _write_wav_song(tmp_path, 'B', sr * 3 + 5, sr, rng)}
    d = MultitrackAudioDataset(str(tmp_path), chunk_length=1, sr=sr, seed=11, augment_data=True, device='cpu')
    assert len(d) == 7
    loader = DataLoader(d, batch_size=3, shuffle=False, num_workers=2, pin_memory=False, drop_last=False, timeout=0,
                        worker_init_fn=None)
    got = list(loader)
    assert [type(b) for b in got] == [HostPcmBatch] * 3 and [b.clips.shape[0] for b in got] == [3, 3, 1]
    assert all(b.clips.dtype == torch.int16 and tuple(b.clips.shape[1:]) == (5, sr, 2) for b in got)
    assert all(b.clips.is_shared() for b in got)
    assert torch.cat([b.items for b in got]).tolist() == list(range(7))
    assert all(b.token == d._token and b.aug_seed == d._aug_seed and b.normalize is False for b in got)
    clips = torch.cat([b.clips for b in got]).numpy()
    for i in range(7):
        song_i, chunk_i = d._calculate_song_index(i)
        for k, t in enumerate(d.get_tracklist()):
            np.testing.assert_array_equal(clips[i, k], files[d.songlist[song_i]][t][chunk_i * sr:(chunk_i + 1) * sr])
    # the draws' read counters live in THIS process' dataset object, untouched by the workers' forks
    assert d._aug_reads == {}
    # a worker that would reach the GPU library raises a clear error instead of initialising a runtime in the fork

    class Wrong(torch.utils.data.Dataset):
        def __len__(self):
            return 2

        def __getitem__(self, i):
            return d.compute_features(np.zeros(4096))

    with pytest.raises(RuntimeError, match='DataLoader worker'):
        list(DataLoader(Wrong(), batch_size=1, num_workers=1))
    # a PcmItem unpacked inside a worker (= asking for features there) is the same error
    class Unpacks(torch.utils.data.Dataset):
        def __len__(self):
            return 1

        def __getitem__(self, i):
            x, gt = d[i]
            return x

    with pytest.raises(RuntimeError, match='DataLoader worker'):
        list(DataLoader(Unpacks(), batch_size=1, num_workers=1))


def test_pcm_items_collate_mixed_sample_types_and_pickle():
    """Items whose files hold different sample types collate to float64 in [-1, 1) (what soundfile.read yields,
    data/dataset.py:194); a HostPcmBatch survives pickling (spawned workers) without its device cache."""
    import pickle
    from deep_audio_mixer_amd.data.dataset import HostPcmBatch, PcmItem, collate_pcm_items
    from torch.utils.data import default_collate
    a = PcmItem(torch.tensor([[[16384, -32768]]], dtype=torch.int16), 4, 77, None, False, 'cuda')
    b = PcmItem(torch.tensor([[[0.25, -0.5]]], dtype=torch.float32), 5, 77, None, False, 'cuda')
    hb = default_collate([a, b])
    assert isinstance(hb, HostPcmBatch) and hb.clips.dtype == torch.float64 and hb.items.tolist() == [4, 5]
    np.testing.assert_array_equal(hb.clips.numpy(), [[[[0.5, -1.0]]], [[[0.25, -0.5]]]])
    same = collate_pcm_items([a, a])
    assert same.clips.dtype == torch.int16 and same.aug_seed is None
    back = pickle.loads(pickle.dumps(same))
    assert torch.equal(back.clips, same.clips) and back.token == 77 and back._dev is None
    with pytest.raises(ValueError, match='one shape'):
        collate_pcm_items([a, PcmItem(torch.zeros((1, 2, 2), dtype=torch.int16), 6, 77, None, False, 'cuda')])


def test_bench_watchdog_names_the_phase_and_exits_nonzero():
    """bench.py's backstop for a multi-rank hang: a rank that enters no new phase within the limit prints its rank and last
    phase and ends with exit code 3 (a plain os._exit from a watcher thread -- no GPU call, no re-exec); a rank that keeps
    moving is left alone."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys, time; sys.path.insert(0, %r); import bench\n'
            'w = bench.ProgressWatchdog(0.4, rank=5)\n'
            'for k in range(4):\n'
            '    w.phase("moving %%d" %% k); time.sleep(0.2)\n'
            'w.phase("all-reduce that never returns"); time.sleep(30)\n' % root)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert 'rank 5' in r.stderr and 'all-reduce that never returns' in r.stderr and 'moving 3' in r.stderr
    import bench
    os.environ['DAM_DIST_TIMEOUT_S'] = '12.5'
    try:
        assert bench.dist_timeout_s() == 12.5
    finally:
        del os.environ['DAM_DIST_TIMEOUT_S']
    assert bench.dist_timeout_s() == 300.0
    assert bench.ProgressWatchdog(0, rank=0).timeout_s == 0          # off: no thread, phase() still records
