#!/usr/bin/env python3
"""Diagnostic: the four variants of the layer1-shaped strip kernel launch that a training step issues, timed one by one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_audio_mixer_amd
from deep_audio_mixer_amd import ops
dev = torch.device('cuda', 0)
B, H, W, C = 8, 1025, 130, 16
x = torch.randn((B, H, W, C), device=dev); dy = torch.randn((B, H, W, C), device=dev); msk = torch.randn((B, H, W, C), device=dev)
wt = torch.randn((C, C, 3, 3), device=dev) * 0.05
wp, wpt = ops.pack_weights(wt), ops.pack_weights(wt, transpose=True)
sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
buf = ops.bn_partial_buffer(dev, C)
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('fwd plain              %.1f us' % t(lambda: ops.conv2d_fwd(x, wp, C, 3, 3, 1, 1, 1)))
print('fwd + stats            %.1f us' % t(lambda: ops.conv2d_fwd(x, wp, C, 3, 3, 1, 1, 1, bn_partial=buf)))
print('fwd + stats + affine   %.1f us' % t(lambda: ops.conv2d_fwd(x, wp, C, 3, 3, 1, 1, 1, bn_partial=buf, in_scale=sc, in_shift=sh, relu_in=True)))
print('dgrad plain            %.1f us' % t(lambda: ops.conv2d_dgrad(dy, wpt, C, H, W, 3, 3, 1, 1, 1)))
print('dgrad + res + mask     %.1f us' % t(lambda: ops.conv2d_dgrad(dy, wpt, C, H, W, 3, 3, 1, 1, 1, res=x, res_mask=msk)))
