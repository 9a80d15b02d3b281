#!/usr/bin/env python3
"""Diagnostic: runs the layer1 conv with the -DDAM_STAMPS library and prints phase durations (s_memtime cycles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('DAM_LIB_PATH', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libdam_hip_stamps.so'))
import numpy as np, torch
import deep_audio_mixer_amd
from deep_audio_mixer_amd import ops
dev = torch.device('cuda', 0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B, H, W = {16: (8, 1025, 130), 32: (8, 513, 65), 64: (8, 257, 33)}[C]
x = torch.randn((B, H, W, C), device=dev)
wp = ops.pack_weights(torch.randn((C, C, 3, 3), device=dev) * 0.05)
buf = torch.zeros(1024 * 16 * 3 * 4, dtype=torch.float32, device=dev)
for _ in range(3):
    y, parts = ops.conv2d_fwd(x, wp, C, 3, 3, 1, 1, 1, bn_partial=buf)
torch.cuda.synchronize()
buf.zero_()
y, parts = ops.conv2d_fwd(x, wp, C, 3, 3, 1, 1, 1, bn_partial=buf)
torch.cuda.synchronize()
st = buf.view(torch.int64).cpu().numpy().astype(np.uint64).reshape(-1, 2, 32)
nwg = int((st[:, 0, 0] != 0).sum())
names = {1: 'start', 2: 'sync1', 3: 'sync2', 4: 'issued', 5: 'mfma', 6: 'bar+commit', 7: 'epi+bar', 8: 'end', 9: 'w-req', 10: 'rows-req', 11: 'zeroed', 12: 'w-lds'}
t0all = None
for wg in (0, min(200, nwg - 1)):
    for role, rn in ((0, 'wave0'), (1, 'wave8 (loader)')):
        v = st[wg, role]
        v = v[v != 0]
        tags = (v >> np.uint64(56)).astype(int)
        t = (v & np.uint64((1 << 56) - 1)).astype(np.int64)
        if t0all is None: t0all = t[0]
        print('wg %d %s: start@%d' % (wg, rn, t[0] - t0all))
        print('   ' + ' '.join('%s+%d' % (names[k][:9], d) for k, d in zip(tags[1:], np.diff(t))))
allv = st[:nwg, 0]
ends = []
for wg in range(nwg):
    v = allv[wg]; v = v[v != 0]
    t = (v & np.uint64((1 << 56) - 1)).astype(np.int64)
    ends.append((t[0], t[-1]))
ends = np.array(ends)
print('kernel span (cycles): first start %d .. last stamp %d; median wg duration %d' % (0, ends[:, 1].max() - ends[:, 0].min(), np.median(ends[:, 1] - ends[:, 0])))
print('wg start spread: p50 %d p99 %d max %d' % tuple(np.percentile(ends[:, 0] - ends[:, 0].min(), [50, 99, 100])))
