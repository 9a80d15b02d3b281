#!/usr/bin/env python3
"""tools/copy_gate_probe.py [gate|end] [steps]: when does the streamed leg's upload of batch k+1 really start, on the device and on
the host?  C3 TrainStep fed by staging.BatchStager; timing events on the training stream in front of every step and on the copy stream
in front of / behind every copy; host clock stamps around next() and the graph launch.  Prints per step: the host's lead over the
device, and the copy's start / end relative to the start of the step that is running when it starts.  Also checks the DATA: every
step's loss against the same clips run HBM-resident (an upload that overtook the step still reading its buffer would change it)."""
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

mode = sys.argv[1] if len(sys.argv) > 1 else 'gate'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = bench.CONFIGS['C3']
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
S, B, hop = cfg['n_stems'], cfg['batch'], cfg['hop']
n = cfg['sr'] * cfg['seconds']


def make():
    model = bench.build_model(cfg, dev)
    opt = Adam(model.parameters(), weight_decay=1e-5)
    step = TrainStep(model, opt, S, n, bench.CHANNELS, B, bench.N_FFT, hop, copy_mark=(mode == 'gate'))
    return step


n_host = 8 * B
host = torch.empty((n_host, S + 1, n, bench.CHANNELS), dtype=torch.float32, pin_memory=True)
host.copy_(bench.synth_clips(n_host, S, n, dev, 7))

# reference: the same batches, resident
step = make()
res = host.to(dev)
step.load_clips(res[:B])
step.capture(warmup=2)
ref = []
for k in range(steps):
    j = (k % (n_host // B)) * B
    step.bind_clips(res[j:j + B])
    ref.append(step().clone())
torch.cuda.synchronize()
ref = [float(r) for r in ref]
step.close()
del step, res

step = make()
step.load_clips(host[:B].to(dev))
step.capture(warmup=2)
stager = staging.BatchStager(host, B, dev, gate=step.copy_mark)
orig_issue = stager._issue
ev_copy = {}


def issue(k):
    b = k % stager.N_BUFS
    lo = (k % stager.n_batches) * stager.batch
    with torch.cuda.stream(stager.stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stager.stream)
        stager.bufs[b].copy_(stager.host[lo:lo + stager.batch], non_blocking=True)
        e1.record(stager.stream)
        stager.ready[b].record(stager.stream)
    ev_copy[k] = (e0, e1, time.perf_counter())


stager._issue = issue
ev_step, host_t, got = [], [], []
torch.cuda.synchronize()
t00 = time.perf_counter()
for k in range(steps):
    t0 = time.perf_counter()
    clips = stager.next()
    t1 = time.perf_counter()
    step.bind_clips(clips)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    ev_step.append(e)
    got.append(step().clone())
    host_t.append((t0 - t00, t1 - t00, time.perf_counter() - t00))
torch.cuda.synchronize()
got = [float(g) for g in got]
print('mode %s: losses equal to the resident run: %s' % (mode, got == ref))
if got != ref:
    print('  resident', ref)
    print('  streamed', got)
base = ev_step[0]
for k in range(steps):
    ts = base.elapsed_time(ev_step[k]) * 1e3
    line = 'step %2d: host next() at %8.0f us, launched by %8.0f; device step start %8.0f' % (k, host_t[k][0] * 1e6, host_t[k][2] * 1e6, ts)
    c = ev_copy.get(k + 1)
    if c is not None and k >= 1:
        cs, ce = base.elapsed_time(c[0]) * 1e3, base.elapsed_time(c[1]) * 1e3
        run = max(j for j in range(steps) if base.elapsed_time(ev_step[j]) * 1e3 <= cs)
        line += '; copy of batch %2d: %8.0f .. %8.0f = %6.0f us into step %d (issued by the host at %8.0f)' % (
            k + 1, cs, ce, cs - base.elapsed_time(ev_step[run]) * 1e3, run, (c[2] - t00) * 1e6)
    print(line)

# ---- un-instrumented periods (wall clock over 40 steps, no timing events): what costs the streamed leg its 0.1 ms?
def region(kind, k_steps=40):
    """kind: resident (bind only) | stream (the stager as configured) | tiny (the stager's events and waits, but a 1-clip copy)."""
    if kind == 'resident':
        def one(k):
            step.bind_clips(stager.bufs[k % 2])
            step()
    else:
        st = staging.BatchStager(host, B, dev, gate=step.copy_mark)
        if kind == 'tiny':
            def tiny_issue(k):
                b = k % st.N_BUFS
                with torch.cuda.stream(st.stream):
                    st.bufs[b][:1, :1, :1024].copy_(st.host[:1, :1, :1024], non_blocking=True)
                    st.ready[b].record(st.stream)
            st._issue = tiny_issue

        def one(k):
            step.bind_clips(st.next())
            step()
    for k in range(4):
        one(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(k_steps):
        one(k)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k_steps


for kind in ('resident', 'stream', 'tiny', 'resident', 'stream', 'tiny'):
    print('period, %-8s (%s): %.4f ms' % (kind, mode, region(kind)))
sys.exit(0 if got == ref else 1)
