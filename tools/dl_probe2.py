"""DataLoader(num_workers=6, pin_memory=True) + upload + a stand-in 4 ms GPU step, phase by phase (diagnostic)."""
import sys
import threading
import time

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
import deep_audio_mixer_amd  # noqa: F401,E402
from deep_audio_mixer_amd.data import dataset as dsm  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

cfg = bench.CONFIGS['C3']
songs, tracklist = bench._synthetic_songs(cfg, 4, 24, pcm16=True)
ds = dsm.MultitrackAudioDataset.from_arrays(songs, chunk_length=cfg['seconds'], sr=cfg['sr'], tracklist=tracklist, seed=1)
torch.cuda.init()
pin_log = []
real_pin = dsm.HostPcmBatch.pin_memory


def logged_pin(self):
    t0 = time.perf_counter()
    out = real_pin(self)
    pin_log.append((threading.current_thread().name, 1e3 * (time.perf_counter() - t0)))
    return out


dsm.HostPcmBatch.pin_memory = logged_pin
mode = sys.argv[1] if len(sys.argv) > 1 else 'upload'
dev = [torch.empty((8, 9, 132300, 2), dtype=torch.int16, device='cuda') for _ in range(2)]
copy_stream = torch.cuda.Stream()
ev = [torch.cuda.Event(), torch.cuda.Event()]
cycles = int(4e-3 * 2.1e9)
for epoch in range(2):
    dl = DataLoader(ds, batch_size=8, shuffle=False, num_workers=6, pin_memory=True, drop_last=True)
    rows = []
    t_epoch = time.perf_counter()
    k = 0
    t0 = time.perf_counter()
    for b in dl:
        t1 = time.perf_counter()
        if mode != 'noupload':
            with torch.cuda.stream(copy_stream):
                dev[k & 1].copy_(b.clips, non_blocking=True)
                r = torch.cuda.Event()
                r.record(copy_stream)
            torch.cuda.current_stream().wait_event(r)
        torch.cuda._sleep(cycles)
        ev[k & 1].record()
        t2 = time.perf_counter()
        if k:
            ev[(k - 1) & 1].synchronize()
        t3 = time.perf_counter()
        rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
        k += 1
        t0 = time.perf_counter()
    torch.cuda.synchronize()
    print('epoch %d: %.1f ms per batch' % (epoch, 1e3 * (time.perf_counter() - t_epoch) / k))
    print('  loader / enqueue / wait ms:', ' '.join('%.1f/%.1f/%.1f' % r for r in rows))
print('pin calls:', ' '.join('%.1f' % t for _, t in pin_log), {n for n, _ in pin_log})
