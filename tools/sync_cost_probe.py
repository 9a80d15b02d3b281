#!/usr/bin/env python3
"""tools/sync_cost_probe.py: what does each piece of the streamed leg's per-step stream synchronisation cost the captured C3 step?
Wall clock over 40 graph replays (HBM-resident clips in every variant, so no upload travels), the pieces added one by one between
the replays: an event record on the training stream, a wait for an event recorded on a copy stream, a tiny copy on that stream --
with torch's events (system-scope fence at every record) and with step marks (include/dam_hip.h: no system-scope fence)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import bench  # noqa: E402
import deep_audio_mixer_amd  # noqa: E402,F401
from deep_audio_mixer_amd import staging  # noqa: E402
from deep_audio_mixer_amd.engine import TrainStep  # noqa: E402
from deep_audio_mixer_amd.optim import Adam  # noqa: E402

cfg = bench.CONFIGS['C3']
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
S, B, hop = cfg['n_stems'], cfg['batch'], cfg['hop']
n = cfg['sr'] * cfg['seconds']
model = bench.build_model(cfg, dev)
opt = Adam(model.parameters(), weight_decay=1e-5)
step = TrainStep(model, opt, S, n, bench.CHANNELS, B, bench.N_FFT, hop, copy_mark=True)
clips = bench.synth_clips(2 * B, S, n, dev, 7)
bufs = [clips[:B], clips[B:]]
step.load_clips(bufs[0])
step.capture(warmup=2)
side = torch.cuda.Stream(device=dev)
host_word = torch.zeros(1024, dtype=torch.float32, pin_memory=True)
dev_word = torch.zeros(1024, dtype=torch.float32, device=dev)


class TorchEv:
    def __init__(self):
        self.e = torch.cuda.Event()

    def record(self, stream):
        self.e.record(stream)

    def wait(self, stream):
        stream.wait_event(self.e)


class MarkEv:
    def __init__(self):
        self.m = staging.StepMark()

    def record(self, stream):
        with torch.cuda.stream(stream):
            self.m.record()

    def wait(self, stream):
        self.m.wait(stream)


def region(kind, ev, k_steps=40):
    consumed, ready = [ev(), ev()], [ev(), ev()]

    def one(k):
        cur = torch.cuda.current_stream(dev)
        if 'record' in kind:
            consumed[k % 2].record(cur)
        if 'cross' in kind:                       # the copy stream waits for the training stream's event (buffer free)
            consumed[k % 2].wait(side)
        if 'wait' in kind:
            if 'copy' in kind:
                with torch.cuda.stream(side):
                    dev_word.copy_(host_word, non_blocking=True)
            ready[k % 2].record(side)
            ready[k % 2].wait(cur)
        step.bind_clips(bufs[k % 2])
        step()
    for k in range(4):
        one(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(k_steps):
        one(k)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k_steps


for rep in range(2):
    for kind in ('plain', 'record', 'wait', 'record+wait', 'record+wait+copy'):
        for name, ev in (('torch events', TorchEv), ('step marks', MarkEv)):
            if kind == 'plain' and name != 'torch events':
                continue
            print('%-18s %-13s %.4f ms per step' % (kind, '' if kind == 'plain' else name, region(kind, ev)), flush=True)

# ---- the other direction: a COPY stream waiting for an event of the TRAINING stream ("this staging buffer is free")
for rep in range(2):
    for kind in ('record+cross', 'record+cross+wait', 'record+cross+wait+copy'):
        for name, ev in (('torch events', TorchEv), ('step marks', MarkEv)):
            print('%-22s %-13s %.4f ms per step' % (kind, name, region(kind, ev)), flush=True)


# ---- ... against the HOST waiting for the training stream's event (no stream waits for the training stream)
def hostsync_region(lag, k_steps=40):
    evs = [torch.cuda.Event() for _ in range(k_steps + 8)]

    def one(k):
        if k >= lag:
            evs[k - lag].synchronize()
        step.bind_clips(bufs[k % 2])
        step()
        evs[k].record()
    for k in range(4):
        one(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(4, 4 + k_steps):
        one(k)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k_steps


for rep in range(2):
    for lag in (1, 2, 3):
        print('host waits for step k-%d before launching step k: %.4f ms per step' % (lag, hostsync_region(lag)), flush=True)


# ---- the reference loop's per-batch loss read (model_trainer.py:41,43), one batch late: what does the 4-byte D2H copy + event cost?
def loss_region(kind, k_steps=40):
    host = torch.zeros(2, dtype=torch.float32, pin_memory=True)
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    state = {'pending': None}

    def one(k):
        step.bind_clips(bufs[k % 2])
        loss = step()
        if kind == 'd2h+event':
            host[k % 2:k % 2 + 1].copy_(loss.detach().reshape(1), non_blocking=True)
            evs[k % 2].record()
            if state['pending'] is not None:
                evs[state['pending']].synchronize()
            state['pending'] = k % 2
        elif kind == 'event only':
            evs[k % 2].record()
            if state['pending'] is not None:
                evs[state['pending']].synchronize()
            state['pending'] = k % 2
    for k in range(4):
        one(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(4, 4 + k_steps):
        one(k)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k_steps


for rep in range(2):
    for kind in ('plain', 'event only', 'd2h+event'):
        print('loss read one batch late, %-11s %.4f ms per step' % (kind, loss_region(kind)), flush=True)
