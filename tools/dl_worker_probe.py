"""tools/dl_worker_probe.py <workers> [pin]: per-worker time inside MultitrackAudioDataset.__getitems__ (decoding one 38 MB C3 batch into the
worker's shared block) and between its calls, and the rate at which DataLoader(num_workers=N) delivers batches to a consumer that only
releases them -- is the notebooks' loader cell loader-bound on this box?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
import deep_audio_mixer_amd
from deep_audio_mixer_amd.data import dataset as D
from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
from torch.utils.data import DataLoader, Subset
cfg = bench.CONFIGS['C3']
songs, tracklist = bench._synthetic_songs(cfg, 4, 48, pcm16=True)
ds = MultitrackAudioDataset.from_arrays(songs, chunk_length=cfg['seconds'], sr=cfg['sr'], tracklist=tracklist, seed=1)
orig = MultitrackAudioDataset.__getitems__
stat = {'n': 0, 'decode': 0.0, 'last_end': None, 'between': 0.0}
def timed(self, indices):
    t0 = time.perf_counter()
    if stat['last_end'] is not None:
        stat['between'] += t0 - stat['last_end']
    out = orig(self, indices)
    t1 = time.perf_counter()
    stat['n'] += 1; stat['decode'] += t1 - t0; stat['last_end'] = t1
    if stat['n'] % 8 == 0:
        wi = torch.utils.data.get_worker_info()
        sys.stderr.write('worker %d: %d batches, __getitems__ %.1f ms, between calls %.1f ms\n' % (wi.id, stat['n'], 1e3*stat['decode']/stat['n'], 1e3*stat['between']/max(1,stat['n']-1)))
    return out
MultitrackAudioDataset.__getitems__ = timed
orig_wb = MultitrackAudioDataset._worker_block
wb = {'n': 0, 't': 0.0, 'fresh': 0}
def timed_wb(self, elem, shape):
    t0 = time.perf_counter()
    ring_before = len(self.__dict__.get('_shm_ring', []))
    out = orig_wb(self, elem, shape)
    wb['t'] += time.perf_counter() - t0; wb['n'] += 1
    if out[1] is None or len(self.__dict__.get('_shm_ring', [])) != ring_before: wb['fresh'] += 1
    if wb['n'] % 8 == 0:
        sys.stderr.write('  worker %d: _worker_block %.1f ms avg, fresh segments %d of %d\n' % (torch.utils.data.get_worker_info().id, 1e3*wb['t']/wb['n'], wb['fresh'], wb['n']))
    return out
MultitrackAudioDataset._worker_block = timed_wb
train_set = Subset(ds, list(range(len(ds))) * 2)
workers = int(sys.argv[1])
pin = len(sys.argv) > 2 and sys.argv[2] == 'pin'
if pin:
    torch.zeros(1, device='cuda')
loader = DataLoader(train_set, batch_size=8, shuffle=False, num_workers=workers, pin_memory=pin, drop_last=True)
n, tf = 0, None
for batch in loader:
    if tf is None: tf = time.perf_counter()
    if batch.release is not None: batch.release[0] = 0
    n += 1
t1 = time.perf_counter()
print('workers %d pin %s: %.2f ms per batch' % (workers, pin, 1e3*(t1-tf)/(n-1)))
