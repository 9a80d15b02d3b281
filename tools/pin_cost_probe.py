#!/usr/bin/env python3
"""tools/pin_cost_probe.py: what the DataLoader's pin thread spends per C3 batch (38 MB) in HostPcmBatch.pin_memory():
allocating page-locked memory, copying, and the time between its calls (receiving + unpickling the next batch)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
import deep_audio_mixer_amd  # noqa: E402,F401
from deep_audio_mixer_amd import staging  # noqa: E402
from deep_audio_mixer_amd.data import dataset as D  # noqa: E402
from torch.utils.data import DataLoader, Subset  # noqa: E402

cfg = bench.CONFIGS['C3']
torch.zeros(1, device='cuda')
songs, tracklist = bench._synthetic_songs(cfg, 4, 48, pcm16=True)
ds = D.MultitrackAudioDataset.from_arrays(songs, chunk_length=cfg['seconds'], sr=cfg['sr'], tracklist=tracklist, seed=1)
st = {'n': 0, 'alloc': 0.0, 'copy': 0.0, 'between': 0.0, 'last': None, 'keep': 0.0}
orig_copy = staging._host_copy
orig_empty = torch.empty


def pin(self):
    t0 = time.perf_counter()
    if st['last'] is not None:
        st['between'] += t0 - st['last']
    src = self.clips.contiguous()
    D._keep_shared_mapping(src)
    t1 = time.perf_counter()
    dst = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
    t2 = time.perf_counter()
    staging._host_copy(dst.view(-1).view(torch.uint8), src.view(-1).view(torch.uint8))
    t3 = time.perf_counter()
    if self.release is not None:
        self.release[0] = 0
    st['n'] += 1
    st['keep'] += t1 - t0
    st['alloc'] += t2 - t1
    st['copy'] += t3 - t2
    st['last'] = time.perf_counter()
    return D.HostPcmBatch(dst, self.items, self.token, self.aug_seed, self.normalize, self.device, self.n_fft, self.hop)


D.HostPcmBatch.pin_memory = pin
loader = DataLoader(Subset(ds, list(range(len(ds))) * 4), batch_size=8, shuffle=False, num_workers=6, pin_memory=True, drop_last=True)
n, tf = 0, None
for batch in loader:
    if tf is None:
        tf = time.perf_counter()
    n += 1
t1 = time.perf_counter()
k = max(1, st['n'])
print('%.2f ms per batch delivered; pin thread per batch: keep %.2f ms, allocate %.2f ms, copy %.2f ms, between calls %.2f ms (%d calls)'
      % (1e3 * (t1 - tf) / (n - 1), 1e3 * st['keep'] / k, 1e3 * st['alloc'] / k, 1e3 * st['copy'] / k, 1e3 * st['between'] / max(1, k - 1), k))
print('HOST_COPY_THREADS', staging.HOST_COPY_THREADS)
