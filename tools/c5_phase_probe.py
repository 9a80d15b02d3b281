#!/usr/bin/env python3
"""Diagnostic: where the host-inclusive time of a C5 song goes (upload / graph / download)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_audio_mixer_amd
from deep_audio_mixer_amd import inference_utils, staging
from deep_audio_mixer_amd.models.model_resnet import ResNet18
dev = torch.device('cuda', 0)
S, sr, n = 8, 44100, 44100 * 180
torch.manual_seed(0)
model = ResNet18(n_stems=S, input_shape=(1025, 130)).to(dev).eval()
mixer = inference_utils.SongMixer(model, S, 2, n, torch.float32, 3 * sr, 'master', True, torch.float32)
rng = np.random.default_rng(0)
tracks = [(0.1 * rng.standard_normal((2, n))).astype(np.float32) for _ in range(S)]
mixer.run(tracks)
pipe = staging.pipe_for(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i, a in enumerate(tracks):
        pipe.upload(mixer.pcm[i], a)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    mixer.launch(); torch.cuda.synchronize(); t3 = time.perf_counter()
    out = pipe.download(mixer.out); t4 = time.perf_counter()
    g = mixer.gains.cpu().numpy(); t5 = time.perf_counter()
    print('upload enqueue %.1f ms, upload drain %.1f ms, graph %.1f ms, download %.1f ms, gains %.2f ms' %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3))
# variants
torch.cuda.synchronize(); t0 = time.perf_counter()
for i, a in enumerate(tracks):
    mixer.pcm[i].copy_(torch.from_numpy(a))
torch.cuda.synchronize(); print('plain copy_ upload: %.1f ms' % ((time.perf_counter() - t0) * 1e3))
pin = torch.empty((2, n), dtype=torch.float32, pin_memory=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
pin.copy_(mixer.out, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
res = np.empty((2, n), np.float32); t2 = time.perf_counter()
torch.from_numpy(res).copy_(pin); t3 = time.perf_counter()
res2 = pin.numpy().copy(); t4 = time.perf_counter()
print('D2H pinned %.1f ms, np.empty %.2f ms, pinned->fresh pageable %.1f ms, numpy copy %.1f ms' %
      ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
t0 = time.perf_counter(); torch.from_numpy(res).copy_(pin); print('pinned->touched pageable %.1f ms' % ((time.perf_counter() - t0) * 1e3))
print('--- replay after idle gaps')
for gap in (0.0, 0.005, 0.02, 0.1, 0.0, 0.0):
    time.sleep(gap)
    t0 = time.perf_counter(); mixer.launch(); torch.cuda.synchronize(); print('gap %.3f s: graph %.1f ms' % (gap, (time.perf_counter() - t0) * 1e3))
print('--- plain copies, 5 reps')
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i, a in enumerate(tracks):
        mixer.pcm[i].copy_(torch.from_numpy(a))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    mixer.launch(); torch.cuda.synchronize(); t2 = time.perf_counter()
    o = mixer.out.cpu(); t3 = time.perf_counter()
    print('upload %.1f ms, graph %.1f ms, download %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
print('--- pipe, 5 reps, per-stem upload times')
for rep in range(5):
    torch.cuda.synchronize(); ts = [time.perf_counter()]
    for i, a in enumerate(tracks):
        pipe.upload(mixer.pcm[i], a); ts.append(time.perf_counter())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    mixer.launch(); torch.cuda.synchronize(); t2 = time.perf_counter()
    o = pipe.download(mixer.out); t3 = time.perf_counter()
    print('upload %s ms, drain %.1f, graph %.1f ms, download %.1f ms' % (['%.1f' % ((b - a) * 1e3) for a, b in zip(ts[:-1], ts[1:])], (t1 - ts[-1]) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
