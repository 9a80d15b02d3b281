"""ModelTrainer.fit over DataLoader(num_workers=6, pin_memory=True) with the pin thread's timeline logged.  (diagnostic)"""
import contextlib
import io
import os
import sys
import tempfile
import threading
import time

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
import deep_audio_mixer_amd  # noqa: F401,E402
from deep_audio_mixer_amd.data import dataset as dsm  # noqa: E402
from deep_audio_mixer_amd.model_trainer import ModelTrainer  # noqa: E402
from torch.utils.data import DataLoader, Subset  # noqa: E402

cfg = bench.CONFIGS['C3']
device = bench._setup_single()
songs, tracklist = bench._synthetic_songs(cfg, 4, 48, pcm16=True)
ds = dsm.MultitrackAudioDataset.from_arrays(songs, chunk_length=cfg['seconds'], sr=cfg['sr'], tracklist=tracklist, seed=1)
log = []
real_pin = dsm.HostPcmBatch.pin_memory
T0 = time.perf_counter()


def logged_pin(self):
    t0 = time.perf_counter()
    out = real_pin(self)
    log.append((t0 - T0, time.perf_counter() - t0))
    return out


dsm.HostPcmBatch.pin_memory = logged_pin
train = DataLoader(Subset(ds, list(range(len(ds))) * 4), batch_size=8, shuffle=False, num_workers=6, pin_memory=True, drop_last=True)
val = DataLoader(Subset(ds, list(range(8))), batch_size=8, num_workers=0)
model = bench.build_model(cfg, device)
opt = torch.optim.Adam(model.parameters(), weight_decay=1e-5)
trainer = ModelTrainer(model, torch.nn.MSELoss(), opt, device, model_name='probe')
with tempfile.TemporaryDirectory() as tmp:
    os.chdir(tmp)
    os.mkdir('weights')
    with contextlib.redirect_stdout(io.StringIO()):
        trainer.fit(train, val, 0, 1)
        log.clear()
        for k in trainer.host_times:
            trainer.host_times[k] = 0.0
        t0 = time.perf_counter()
        trainer.fit(train, val, 1, 1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
n = len(train)
print('epoch: %.1f ms per step; host ms per step:' % (1e3 * dt / n), {k: round(1e3 * v / n, 2) for k, v in trainer.host_times.items()})
gaps = [b[0] - a[0] for a, b in zip(log[:-1], log[1:])]
durs = [d for _, d in log]
print('pin calls: %d; duration ms median %.2f max %.2f; start-to-start gap ms median %.2f' % (
    len(log), 1e3 * sorted(durs)[len(durs) // 2], 1e3 * max(durs), 1e3 * sorted(gaps)[len(gaps) // 2]))
print('gaps ms:', ' '.join('%.1f' % (1e3 * g) for g in gaps[20:60]))
print('durs ms:', ' '.join('%.1f' % (1e3 * g) for g in durs[20:60]))
