#!/usr/bin/env python3
"""tools/trainer_loop_probe.py: the ModelTrainer loop in miniature -- int16 PCM batches in page-locked memory, uploaded per step on a copy
stream into one of two device slots (ModelTrainer._upload), bound to the PCM-fed captured step, loss read one batch late -- with timing
events around every upload and in front of every step: when does the upload of batch k+1 run, and does step k+1 wait for it?"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
import deep_audio_mixer_amd  # noqa: E402,F401
from deep_audio_mixer_amd.engine import TrainStep  # noqa: E402
from deep_audio_mixer_amd.optim import Adam  # noqa: E402

cfg = bench.CONFIGS['C3']
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
S, B, hop = cfg['n_stems'], cfg['batch'], cfg['hop']
n = cfg['sr'] * cfg['seconds']
model = bench.build_model(cfg, dev)
opt = Adam(model.parameters(), weight_decay=1e-5)
step = TrainStep(model, opt, S, n, bench.CHANNELS, B, bench.N_FFT, hop, pcm_dtype=torch.int16, copy_mark=True)
host = [torch.randint(-3000, 3000, (B, S + 1, n, bench.CHANNELS), dtype=torch.int16).pin_memory() for _ in range(4)]
slots = [torch.empty_like(host[0], device=dev) for _ in range(2)]
slots[0].copy_(host[0])
step.bind_clips(slots[0])
step.capture(warmup=2)
copy_stream = torch.cuda.Stream(device=dev)
ready = [torch.cuda.Event() for _ in range(2)]
consumed = [torch.cuda.Event() for _ in range(2)]
loss_host = torch.zeros(2, dtype=torch.float32, pin_memory=True)
loss_ev = [torch.cuda.Event(), torch.cuda.Event()]
mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
steps = 24
ev_step, ev_up, host_t = [], [], []
pending = []
torch.cuda.synchronize()
t00 = time.perf_counter()
for k in range(steps):
    slot = k % 2
    cur = torch.cuda.current_stream(dev)
    consumed[slot].synchronize()
    if mode == 'gate' and k > 0:
        step.copy_mark.synchronize()
    th = time.perf_counter() - t00
    with torch.cuda.stream(copy_stream):
        u0, u1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        u0.record(copy_stream)
        slots[slot].copy_(host[k % 4], non_blocking=True)
        u1.record(copy_stream)
        ready[slot].record(copy_stream)
    cur.wait_event(ready[slot])
    step.bind_clips(slots[slot])
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    loss = step()
    consumed[slot].record(cur)
    loss_host[slot:slot + 1].copy_(loss.detach().reshape(1), non_blocking=True)
    loss_ev[slot].record()
    ev_step.append(e)
    ev_up.append((u0, u1))
    host_t.append(th)
    pending.append(slot)
    if len(pending) > 1:
        loss_ev[pending.pop(0)].synchronize()
torch.cuda.synchronize()
wall = time.perf_counter() - t00
base = ev_step[4]
print('mode %s: %.4f ms per step (wall over %d steps)' % (mode, 1e3 * wall / steps, steps))
for k in range(8, 16):
    ts = base.elapsed_time(ev_step[k]) * 1e3
    us, ue = base.elapsed_time(ev_up[k][0]) * 1e3, base.elapsed_time(ev_up[k][1]) * 1e3
    prev = base.elapsed_time(ev_step[k - 1]) * 1e3
    print('  step %2d starts %8.0f us (period %6.0f); its upload ran %8.0f .. %8.0f (%5.0f us), i.e. %6.0f us into step %d; issued by the host at %8.0f'
          % (k, ts, ts - prev, us, ue, ue - us, us - prev, k - 1, host_t[k] * 1e6))
