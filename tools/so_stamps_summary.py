#!/usr/bin/env python3
"""Per-slot cycle counts of the self-overlapped strip convolution from s_memtime stamps (DAM_STAMPS builds): for each variant
library tools/libdam_<name>.so given on the command line, the median length of a compute wave's slot body (tag 5: MFMAs +
fillers) and of its barrier wait (tag 7), over all workgroups and slots, for the layer1 (16 ch) and layer2 (32 ch) launches."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
import deep_audio_mixer_amd
from deep_audio_mixer_amd import ops
dev = torch.device('cuda', 0)
out = {}
for C, (B, H, W) in ((16, (8, 1025, 130)), (32, (8, 513, 65))):
    x = torch.randn((B, H, W, C), device=dev)
    wp = ops.pack_weights(torch.randn((C, C, 3, 3), device=dev) * 0.05)
    buf = torch.zeros(1024 * 32 * 3 * 4, dtype=torch.float32, device=dev)
    for _ in range(3):
        ops.conv2d_fwd(x, wp, C, 3, 3, 1, 1, 1, bn_partial=buf)
    torch.cuda.synchronize(); buf.zero_()
    ops.conv2d_fwd(x, wp, C, 3, 3, 1, 1, 1, bn_partial=buf)
    torch.cuda.synchronize()
    st = buf.view(torch.int64).cpu().numpy().astype(np.uint64).reshape(-1, 2, 32)
    body, wait, pro = [], [], []
    for wg in range(st.shape[0]):
        v = st[wg, 0]; v = v[v != 0]
        if len(v) < 6: continue
        tags = (v >> np.uint64(56)).astype(int); t = (v & np.uint64((1 << 56) - 1)).astype(np.int64)
        d = np.diff(t)
        pro.append(int(t[2] - t[0]))
        for k in range(3, len(tags)):
            if tags[k] == 5 and 3 < k < len(tags) - 2: body.append(int(d[k - 1]))
            if tags[k] == 7 and 3 < k < len(tags) - 2: wait.append(int(d[k - 1]))
    out[str(C)] = {'body': float(np.median(body)) if body else None, 'wait': float(np.median(wait)) if wait else None,
                   'prologue': float(np.median(pro)), 'n': len(body)}
print('STAMPS ' + json.dumps(out))
''' % ROOT
print('%-14s %s' % ('variant', ''.join('%36s' % ('C=%d: body / barrier wait / prologue (cycles)' % c) for c in (16, 32))))
for name in sys.argv[1:]:
    lib = os.path.join(ROOT, 'tools', 'libdam_%s.so' % name)
    r = subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, DAM_LIB_PATH=lib), capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('STAMPS ')]
    if not line:
        print(name, 'FAILED', r.stderr[-400:]); continue
    d = json.loads(line[0][7:])
    print('%-14s %s' % (name, ''.join('%36s' % ('%.0f / %.0f / %.0f' % (d[c]['body'], d[c]['wait'], d[c]['prologue'])) for c in ('16', '32'))))
