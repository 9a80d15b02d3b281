#!/usr/bin/env python3
"""Runs single kernels of the hot path in isolation at the benchmark's shapes (for rocprofv3 --pmc passes).
usage: python tools/conv_probe.py [conv1|layer1|layer1_wgrad|layer3|bn|stft] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deep_audio_mixer_amd  # noqa: E402,F401
from deep_audio_mixer_amd import features, ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'layer1'
which, _, opts = which.partition(':')            # layer4|5|6[:stats][+affine]: forward with the statistics epilogue / fused input affine
opts = set(opts.split('+')) if opts else set()
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda', 0)
B, H, W = 8, 1025, 130
torch.manual_seed(0)
if which in ('layer1', 'layer1_wgrad', 'layer1_dgrad', 'layer1_dgrad_epi1', 'layer1_dgrad_epi2', 'layer1_dgrad_epi3'):
    x = torch.randn((B, H, W, 16), device=dev)
    wt = torch.randn((16, 16, 3, 3), device=dev) * 0.05
    wp, wpt = ops.pack_weights(wt), ops.pack_weights(wt, transpose=True)
    dy = torch.randn((B, H, W, 16), device=dev)
    if which == 'layer1':
        kw = {}
        if 'stats' in opts:
            kw['bn_partial'] = ops.bn_partial_buffer(dev, 16)
        if 'affine' in opts:
            kw.update(in_scale=torch.rand(16, device=dev) + 0.5, in_shift=torch.randn(16, device=dev) * 0.1, relu_in=True)
        fn = lambda: ops.conv2d_fwd(x, wp, 16, 3, 3, 1, 1, 1, **kw)
    elif which == 'layer1_dgrad':
        fn = lambda: ops.conv2d_dgrad(dy, wpt, 16, H, W, 3, 3, 1, 1, 1, res=dy, res_mask=x)
    elif which.startswith('layer1_dgrad_epi'):
        # the data gradients with the BatchNorm-backward sums epilogues: 1 = sums of bn1 (x rides in the residual operand), 2 = identity
        # shortcut (residual + sign bytes) + the sums of the upstream relu(bn(x)) (mask from its affine), 3 = upstream mask as sign bytes
        sc, sh = torch.rand(16, device=dev) + 0.5, torch.randn(16, device=dev) * 0.1
        mean, invstd = torch.randn(16, device=dev) * 0.1, torch.rand(16, device=dev) + 0.5
        y = torch.randn((B, H, W, 16), device=dev)
        _, bits = ops.bn_apply(y, sc, sh, relu=True, sign_bits=True)
        up = torch.randn((B, H, W, 16), device=dev)
        if which.endswith('1'):
            fn = lambda: ops.conv2d_dgrad(dy, wpt, 16, H, W, 3, 3, 1, 1, 1, bn_bwd=(x, mean, invstd, sc, sh))
        elif which.endswith('2'):
            fn = lambda: ops.conv2d_dgrad(dy, wpt, 16, H, W, 3, 3, 1, 1, 1, res=x, res_mask=y, res_mask_bits=bits, bn_bwd=(up, mean, invstd, sc, sh))
        else:
            _, ubits = ops.bn_apply(up, sc, sh, relu=True, sign_bits=True)
            fn = lambda: ops.conv2d_dgrad(dy, wpt, 16, H, W, 3, 3, 1, 1, 1, res=x, res_mask=y, res_mask_bits=bits,
                                          bn_bwd=(up, mean, invstd, None, None, ubits))
    else:
        fn = lambda: ops.conv2d_wgrad(x, dy, 16, 3, 3, 1, 1, 1)
elif which == 'layer3':
    x = torch.randn((B, 257, 33, 64), device=dev)
    wp = ops.pack_weights(torch.randn((64, 64, 3, 3), device=dev) * 0.05)
    fn = lambda: ops.conv2d_fwd(x, wp, 64, 3, 3, 1, 1, 1)
elif which in ('layer4', 'layer5', 'layer6'):
    hw, c = {'layer4': ((129, 17), 96), 'layer5': ((65, 9), 128), 'layer6': ((33, 5), 256)}[which]
    x = torch.randn((B, hw[0], hw[1], c), device=dev)
    wp = ops.pack_weights(torch.randn((c, c, 3, 3), device=dev) * 0.05)
    kw = {}
    if 'stats' in opts:
        kw['bn_partial'] = ops.bn_partial_buffer(dev, c)
    if 'affine' in opts:
        kw.update(in_scale=torch.rand(c, device=dev) + 0.5, in_shift=torch.randn(c, device=dev) * 0.1, relu_in=True)
    fn = lambda: ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1, **kw)
elif which in ('layer3s2', 'layer4s2', 'layer5s2', 'layer6s2'):       # the strided 3x3 convolution at the head of a stage
    hw, ci, co = {'layer3s2': ((513, 65), 32, 64), 'layer4s2': ((257, 33), 64, 96), 'layer5s2': ((129, 17), 96, 128),
                  'layer6s2': ((65, 9), 128, 256)}[which]
    x = torch.randn((B, hw[0], hw[1], ci), device=dev)
    wp = ops.pack_weights(torch.randn((co, ci, 3, 3), device=dev) * 0.05)
    fn = lambda: ops.conv2d_fwd(x, wp, co, 3, 3, 2, 1, 1)
elif which in ('layer2_wgrad', 'layer3_wgrad', 'layer4_wgrad', 'layer5_wgrad', 'layer6_wgrad'):
    hw, c = {'layer2_wgrad': ((513, 65), 32), 'layer3_wgrad': ((257, 33), 64), 'layer4_wgrad': ((129, 17), 96), 'layer5_wgrad': ((65, 9), 128),
             'layer6_wgrad': ((33, 5), 256)}[which]
    x = torch.randn((B, hw[0], hw[1], c), device=dev)
    dy = torch.randn((B, hw[0], hw[1], c), device=dev)
    fn = lambda: ops.conv2d_wgrad(x, dy, c, 3, 3, 1, 1, 1)
elif which in ('layer2s2_wgrad', 'layer3s2_wgrad', 'layer4s2_wgrad', 'layer5s2_wgrad', 'layer6s2_wgrad'):
    hw, ci, co = {'layer2s2_wgrad': ((1025, 130), 16, 32), 'layer3s2_wgrad': ((513, 65), 32, 64), 'layer4s2_wgrad': ((257, 33), 64, 96),
                  'layer5s2_wgrad': ((129, 17), 96, 128), 'layer6s2_wgrad': ((65, 9), 128, 256)}[which]
    x = torch.randn((B, hw[0], hw[1], ci), device=dev)
    dy = torch.randn((B, (hw[0] + 1) // 2, (hw[1] + 1) // 2, co), device=dev)
    fn = lambda: ops.conv2d_wgrad(x, dy, co, 3, 3, 2, 1, 1)
elif which in ('l2s2_pair', 'l3s2_pair', 'l4s2_pair', 'l5s2_pair', 'l6s2_pair'):                              # a down-sampling block's conv1 + shortcut + statistics, one launch
    hw, ci, co = {'l2s2_pair': ((1025, 130), 16, 32), 'l3s2_pair': ((513, 65), 32, 64), 'l4s2_pair': ((257, 33), 64, 96),
                  'l5s2_pair': ((129, 17), 96, 128), 'l6s2_pair': ((65, 9), 128, 256)}[which]
    x = torch.randn((B, hw[0], hw[1], ci), device=dev)
    wp = ops.pack_weights(torch.randn((co, ci, 3, 3), device=dev) * 0.05)
    wps = ops.pack_weights(torch.randn((co, ci, 1, 1), device=dev) * 0.05)
    fn = lambda: ops.conv_s2_pair_fwd(x, wp, wps, co)
elif which in ('l2s2_dgrad', 'l3s2_dgrad', 'l4s2_dgrad', 'l5s2_dgrad', 'l6s2_dgrad'):    # its data gradient (conv1's four parity classes + the shortcut's tap)
    hw, ci, co = {'l2s2_dgrad': ((1025, 130), 16, 32), 'l3s2_dgrad': ((513, 65), 32, 64), 'l4s2_dgrad': ((257, 33), 64, 96),
                  'l5s2_dgrad': ((129, 17), 96, 128), 'l6s2_dgrad': ((65, 9), 128, 256)}[which]
    hd, wd = (hw[0] + 1) // 2, (hw[1] + 1) // 2
    dy = torch.randn((B, hd, wd, co), device=dev)
    ds = torch.randn((B, hd, wd, co), device=dev)
    wpt = ops.pack_weights(torch.randn((co, ci, 3, 3), device=dev) * 0.05, transpose=True)
    wpst = ops.pack_weights(torch.randn((co, ci, 1, 1), device=dev) * 0.05, transpose=True)
    fn = lambda: ops.conv2d_dgrad(dy, wpt, ci, hw[0], hw[1], 3, 3, 2, 1, 1, pair_1x1=(ds, wpst))
elif which == 'conv1':
    x = torch.randn((B, 8, H, W), device=dev)
    wp = ops.pack_weights(torch.randn((16, 8, 3, 3), device=dev) * 0.05)
    fn = lambda: ops.conv2d_fwd(x, wp, 16, 3, 3, 1, 1, 1, in_nchw=True)
elif which == 'bn':
    x = torch.randn((B, H, W, 16), device=dev)
    gam, bet = torch.ones(16, device=dev), torch.zeros(16, device=dev)
    fn = lambda: ops.bn_stats(x, gam, bet, None, None, None, 0.1, 1e-5)
elif which == 'bn_apply':
    x = torch.randn((B, H, W, 16), device=dev)
    gam, bet = torch.ones(16, device=dev), torch.zeros(16, device=dev)
    fn = lambda: ops.bn_apply(x, gam, bet, relu=True, sign_bits=True)
elif which == 'stft':
    pcm = 0.1 * torch.randn((72, 132300, 2), device=dev)
    fn = lambda: features.stft_logmag(pcm, hop=1024)
else:
    raise SystemExit('unknown probe')
for _ in range(3):
    fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    fn()
b.record()
torch.cuda.synchronize()
print('%s: %.1f us per call' % (which, a.elapsed_time(b) * 1e3 / iters))
