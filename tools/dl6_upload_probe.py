#!/usr/bin/env python3
"""tools/dl6_upload_probe.py: ModelTrainer.fit over DataLoader(num_workers=6, pin_memory=True) with timing events around every upload
(ModelTrainer._upload) and in front of every step: when is the upload of batch k issued (host clock, relative to the previous step's start
on the device), how long does it take on the device, and does step k wait for it?  (rocprofv3's copy domain hangs with forked workers.)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
import deep_audio_mixer_amd  # noqa: E402,F401
from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset  # noqa: E402
from deep_audio_mixer_amd.model_trainer import ModelTrainer  # noqa: E402
from torch.utils.data import DataLoader, Subset  # noqa: E402

cfg = bench.CONFIGS['C3']
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
B = cfg['batch']
songs, tracklist = bench._synthetic_songs(cfg, 8, 48, pcm16=True)
ds = MultitrackAudioDataset.from_arrays(songs, chunk_length=cfg['seconds'], sr=cfg['sr'], tracklist=tracklist, seed=1)
train = DataLoader(Subset(ds, list(range(len(ds))) * 2), batch_size=B, shuffle=False, num_workers=6, pin_memory=True, drop_last=True)
val = DataLoader(Subset(ds, list(range(B))), batch_size=B, shuffle=False, num_workers=0)
model = bench.build_model(cfg, dev)
opt = torch.optim.Adam(model.parameters(), weight_decay=1e-5)
os.makedirs('/tmp/dl6w/weights', exist_ok=True)
os.chdir('/tmp/dl6w')
tr = ModelTrainer(model, torch.nn.MSELoss(), opt, dev, model_name='probe')
rec = []
orig_upload = tr._upload
orig_tdb = tr._train_device_batch


def upload(host):
    st = getattr(tr, '_pcm_stage', None)
    u0, u1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_host = time.perf_counter()
    if st is not None:
        u0.record(st['stream'])
    out = orig_upload(host)
    st = tr._pcm_stage
    u1.record(st['stream'])
    rec.append({'host': t_host, 'u0': u0 if st is not None else None, 'u1': u1})
    return out


def tdb(batch):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    rec[-1]['step'] = e
    rec[-1]['host_launch'] = time.perf_counter()
    return orig_tdb(batch)


tr._upload = upload
tr._train_device_batch = tdb
import io, contextlib  # noqa: E402
with contextlib.redirect_stdout(io.StringIO()):
    tr.fit(train, val, 0, 1)
torch.cuda.synchronize()
ok = [r for r in rec if r.get('u0') is not None and 'step' in r]
base = ok[20]['step']
print('batches %d' % len(ok))
for k in range(40, 52):
    r, p = ok[k], ok[k - 1]
    s_prev, s_k = base.elapsed_time(p['step']) * 1e3, base.elapsed_time(r['step']) * 1e3
    u0, u1 = base.elapsed_time(r['u0']) * 1e3, base.elapsed_time(r['u1']) * 1e3
    print('  step %2d starts %8.0f (period %5.0f); its upload on the device %8.0f .. %8.0f (%5.0f us) = %5.0f us into step %d; host: upload issued %6.0f us before the launch'
          % (k, s_k, s_k - s_prev, u0, u1, u1 - u0, u0 - s_prev, k - 1, (r['host_launch'] - r['host']) * 1e6))
tr.close()
