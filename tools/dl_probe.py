"""Where does a DataLoader(num_workers=6, pin_memory=True) batch of the C3 shape spend its time?  (diagnostic)"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
import deep_audio_mixer_amd  # noqa: F401,E402
from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

cfg = bench.CONFIGS['C3']
songs, tracklist = bench._synthetic_songs(cfg, 4, 24, pcm16=True)
ds = MultitrackAudioDataset.from_arrays(songs, chunk_length=cfg['seconds'], sr=cfg['sr'], tracklist=tracklist, seed=1)
torch.cuda.init()
x = torch.zeros(1, device='cuda')
for pin in (False, True):
    dl = DataLoader(ds, batch_size=8, shuffle=False, num_workers=6, pin_memory=pin, drop_last=True)
    t0 = time.perf_counter()
    ts = []
    for b in dl:
        ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
    print('pin_memory=%s: per-batch arrival ms' % pin, ['%.1f' % (1e3 * t) for t in ts])
# pin + upload of one shared-memory batch
b = next(iter(DataLoader(ds, batch_size=8, num_workers=1, pin_memory=False)))
print('shared', b.clips.is_shared(), 'pinned', b.clips.is_pinned(), b.clips.shape, b.clips.dtype)
for k in range(4):
    t0 = time.perf_counter()
    p = b.clips.pin_memory()
    t1 = time.perf_counter()
    dev = torch.empty(p.shape, dtype=p.dtype, device='cuda')
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    dev.copy_(p, non_blocking=True)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print('pin %.2f ms, copy enqueue %.2f ms, copy done %.2f ms (is_pinned %s)' % (1e3 * (t1 - t0), 1e3 * (t3 - t2), 1e3 * (t4 - t2), p.is_pinned()))
    del p
own = torch.empty(b.clips.shape, dtype=b.clips.dtype, pin_memory=True)
for k in range(3):
    t0 = time.perf_counter()
    own.copy_(b.clips)
    t1 = time.perf_counter()
    dev.copy_(own, non_blocking=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('own pinned buffer: host copy %.2f ms, H2D %.2f ms' % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
