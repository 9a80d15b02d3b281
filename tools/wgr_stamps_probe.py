#!/usr/bin/env python3
"""Diagnostic: one row-streaming weight gradient with the -DDAM_WGR_STAMPS library (tools/build_variant.sh wgr_stamps dam_wgrad.hip
-DDAM_WGR_STAMPS); prints per workgroup role the prologue, the median slot body / barrier wait and the epilogue in shader clocks.
usage: python tools/wgr_stamps_probe.py layer1|layer2|layer3"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('DAM_LIB_PATH', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libdam_wgr_stamps.so'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import deep_audio_mixer_amd  # noqa: E402,F401
from deep_audio_mixer_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'layer1'
hw, c, nblk = {'layer1': ((1025, 130), 16, 9), 'layer2': ((513, 65), 32, 18), 'layer3': ((257, 33), 64, 18)}[which]
dev = torch.device('cuda', 0)
x = torch.randn((8, hw[0], hw[1], c), device=dev)
dy = torch.randn((8, hw[0], hw[1], c), device=dev)
for _ in range(3):
    ops.conv2d_wgrad(x, dy, c, 3, 3, 1, 1, 1)
ws = ops._workspaces[(dev.type, dev.index)]
torch.cuda.synchronize()
ws.zero_()
ops.conv2d_wgrad(x, dy, c, 3, 3, 1, 1, 1)
torch.cuda.synchronize()
raw = ws.view(torch.int64).cpu().numpy().astype(np.uint64)
per = nblk * 256 // 2                                   # u64 words per slab
mask = np.uint64((1 << 56) - 1)
rows = {0: [], 1: []}
nwg = 0
for wg in range(raw.size // per):
    blk = raw[wg * per: wg * per + 64].reshape(2, 32)
    if blk[0, 0] >> np.uint64(56) != 1:
        continue
    nwg += 1
    for role in (0, 1):
        v = blk[role]; v = v[v != 0]
        tags = (v >> np.uint64(56)).astype(int); t = (v & mask).astype(np.int64)
        d = np.diff(t)
        body = d[tags[1:] == 5]; wait = d[tags[1:] == 7]
        rows[role].append((t[tags == 2][0] - t[0], np.median(body[:-1]) if len(body) > 1 else 0, np.median(wait[:-1]) if len(wait) > 1 else 0,
                           len(body), t[-1] - t[tags == 7][-1] if (tags == 7).any() else 0, t[-1] - t[0]))
print('%s: %d workgroups with stamps' % (which, nwg))
pro = []
for wg in range(raw.size // per):
    blk = raw[wg * per: wg * per + 64].reshape(2, 32)
    if blk[0, 0] >> np.uint64(56) != 1:
        continue
    v = blk[1]; v = v[v != 0]
    tags = (v >> np.uint64(56)).astype(int); t = (v & mask).astype(np.int64)
    if all((tags == k).any() for k in (9, 10, 11, 12, 2)):
        g = lambda k: t[tags == k][0] - t[0]
        pro.append((g(9), g(10) - g(9), g(11) - g(10), g(12) - g(11), g(2) - g(12)))
if pro:
    print('  loader prologue (median clocks): setup %d | requests issued %d | LDS zeroed + barrier %d | first rows landed and written %d | next requests + barrier %d'
          % tuple(np.median(np.array(pro), axis=0)))
for role, name in ((0, 'compute wave 0'), (1, 'loader wave 0')):
    a = np.array(rows[role], dtype=np.float64)
    print('  %-15s prologue %6.0f | slot body %6.0f  barrier wait %5.0f  (x %d slots) | after the last barrier %6.0f | total %7.0f (max %7.0f)'
          % ((name,) + tuple(np.median(a[:, i]) for i in range(3)) + (int(np.median(a[:, 3])), np.median(a[:, 4]), np.median(a[:, 5]), a[:, 5].max())))
