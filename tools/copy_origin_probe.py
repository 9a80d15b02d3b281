#!/usr/bin/env python3
"""Diagnostic (renamed from find_copies.py): where do device-to-device copies in a training step come from?  Uses torch.profiler on one eager step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_audio_mixer_amd
from deep_audio_mixer_amd.models.model_resnet import ResNet18
from deep_audio_mixer_amd.optim import Adam
from deep_audio_mixer_amd.engine import TrainStep
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ResNet18(n_stems=8, input_shape=(1025, 130)).to(dev)
opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-5)
ts = TrainStep(model, opt, batch=2, n_stems=8, n_samples=132300, channels=2, use_graph=False)
ts.load_batch(torch.randn(2, 8, 132300, 2, device=dev) * 0.1, torch.randn(2, 132300, 2, device=dev) * 0.1)
for _ in range(2):
    ts._eager()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts._eager()
    torch.cuda.synchronize()
rows = [e for e in prof.events() if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::_foreach_copy_')]
print(len(rows), 'copy-like aten ops')
from collections import Counter
c = Counter()
for e in rows:
    st = [s for s in (e.stack or []) if 'deep-audio-mixer_amd' in s or 'torch/autograd' in s][:2]
    c[(e.name, str(e.input_shapes)[:40], tuple(st))] += 1
for k, v in c.most_common(25):
    print(v, k)
