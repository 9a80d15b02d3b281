#!/usr/bin/env python3
"""tools/dl_rate_probe.py [workers] [repeat]: how fast does torch's DataLoader(num_workers=6, pin_memory=True) over the in-memory
MultitrackAudioDataset DELIVER page-locked batches to the main process when the consumer does nothing (no GPU step)?  The
ModelTrainer loop over that loader runs 5.7 ms per step against a 4.2 ms device step -- if this rate is ~5.7 ms per batch, the loop is
loader-bound whatever the device does.  Also with the pin thread's work replaced / removed: pin_memory=False (batches stay in the
workers' shared blocks), and the upload alone."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
import deep_audio_mixer_amd  # noqa: E402,F401
from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset  # noqa: E402
from torch.utils.data import DataLoader, Subset  # noqa: E402

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
repeat = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = bench.CONFIGS['C3']
B = cfg['batch']
torch.cuda.init()
torch.zeros(1, device='cuda')          # the GPU-owning process, as in training
songs, tracklist = bench._synthetic_songs(cfg, 8, 48, pcm16=True)
ds = MultitrackAudioDataset.from_arrays(songs, chunk_length=cfg['seconds'], sr=cfg['sr'], tracklist=tracklist, seed=1)
train_set = ds if repeat == 1 else Subset(ds, list(range(len(ds))) * repeat)      # (as bench.py: Subset forwards __getitems__)


def rate(pin, touch):
    loader = DataLoader(train_set, batch_size=B, shuffle=False, num_workers=workers, pin_memory=pin, drop_last=True)
    n, t_first = 0, None
    t0 = time.perf_counter()
    dev = None
    for batch in loader:
        if t_first is None:
            t_first = time.perf_counter()
        if touch == 'upload':
            dev = batch.to_device('cuda')
        n += 1
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return 1e3 * (t1 - t_first) / max(1, n - 1), 1e3 * (t_first - t0), n


for pin, touch in ((True, 'none'), (True, 'upload'), (False, 'none'), (False, 'upload'), (True, 'none')):
    ms, first, n = rate(pin, touch)
    print('pin_memory=%-5s consumer: %-6s  %.2f ms per batch after the first (first batch after %.0f ms, %d batches)' % (pin, touch, ms, first, n), flush=True)
