#!/usr/bin/env python3
"""Diagnostic: time of the BS.1770 meter on a 4-minute stereo stem (device).  The CPU figure quoted in DESIGN.md (scipy
lfilter restatement, 296 ms) comes from tests/test_loudness_gpu.py::test_full_song_length_property's oracle call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_audio_mixer_amd
from deep_audio_mixer_amd import loudness
rate, n = 44100, 44100 * 240
x = (0.1 * np.random.RandomState(0).randn(n, 2)).astype(np.float32)
xd = torch.from_numpy(x).cuda()
m = loudness.Meter(rate)
m.integrated_loudness(xd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    v = m.integrated_loudness(xd)
torch.cuda.synchronize()
dev = (time.perf_counter() - t0) / 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); z = m.block_energies(xd); e1.record(); torch.cuda.synchronize()
print('device: %.2f ms per 4-min stereo stem end to end (%.2f ms kernels+copy), %.3f LUFS' % (dev * 1e3, e0.elapsed_time(e1), v))
