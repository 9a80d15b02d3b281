#!/usr/bin/env python3
"""Diagnostic: one thick-layer convolution with the -DDAM_PIPE_STAMPS library; prints the phase durations (s_memtime ticks) of a
few workgroups and the launch-wide picture.  usage: DAM_TILE=1x4 python tools/pipe_stamps_probe.py layer5"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('DAM_LIB_PATH', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libdam_pipe_stamps.so'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import deep_audio_mixer_amd  # noqa: E402,F401
from deep_audio_mixer_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'layer3'
dev = torch.device('cuda', 0)
hw, c = {'layer3': ((257, 33), 64), 'layer4': ((129, 17), 96), 'layer5': ((65, 9), 128), 'layer6': ((33, 5), 256)}[which]
x = torch.randn((8, hw[0], hw[1], c), device=dev)
wp = ops.pack_weights(torch.randn((c, c, 3, 3), device=dev) * 0.05)
for _ in range(3):
    ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1)
ws = ops._workspaces[(dev.type, dev.index)]
torch.cuda.synchronize()
ws.zero_()
ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1)
torch.cuda.synchronize()
st = ws.view(torch.int64).cpu().numpy().astype(np.uint64)
st = st[:st.size // 128 * 128].reshape(-1, 2, 64)
nwg = int((st[:, 0, 0] != 0).sum())
names = {1: 'start', 2: 'setup', 3: 'bar0', 4: 'commit0', 5: 'bar1', 6: 'work', 7: 'bar', 8: 'epilogue'}
mask = np.uint64((1 << 56) - 1)


def decode(v):
    v = v[v != 0]
    return (v >> np.uint64(56)).astype(int), (v & mask).astype(np.int64)


t00 = min(decode(st[w, 0])[1][0] for w in range(nwg))
for wg in sorted({0, nwg // 2, nwg - 1}):
    for role, rn in ((0, 'compute wave 0'), (1, 'loader wave')):
        tags, t = decode(st[wg, role])
        print('wg %d %s: start@%d' % (wg, rn, t[0] - t00))
        print('   ' + ' '.join('%s+%d' % (names[k], d) for k, d in zip(tags[1:], np.diff(t))))
se = np.array([[decode(st[w, 0])[1][0], decode(st[w, 0])[1][-1]] for w in range(nwg)])
print('%d workgroups; launch span %d ticks; workgroup duration p50 %d p90 %d max %d' %
      ((nwg, se[:, 1].max() - se[:, 0].min()) + tuple(np.percentile(se[:, 1] - se[:, 0], [50, 90, 100]).astype(int))))
print('start spread: p50 %d p90 %d max %d' % tuple(np.percentile(se[:, 0] - se[:, 0].min(), [50, 90, 100]).astype(int)))
work = []
for w in range(nwg):
    tags, t = decode(st[w, 0])
    d = np.diff(t)
    work.append([d[tags[1:] == 6].sum(), d[tags[1:] == 7].sum(), d[(tags[1:] == 2) | (tags[1:] == 3) | (tags[1:] == 5)].sum(),
                 d[tags[1:] == 8].sum()])
work = np.array(work)
print('compute wave 0, mean ticks per workgroup: MFMA phases %d, loop barriers %d, prologue (until first chunk is staged) %d, epilogue %d'
      % tuple(work.mean(axis=0).astype(int)))
