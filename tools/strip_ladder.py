#!/usr/bin/env python3
"""Ablation ladder of conv_strip_kernel: every timing-only variant built by tools/strip_ladder.sh is
run in its own process (DAM_LIB_PATH) on the launches of a C3 step; prints us per launch and clocks per MFMA per SIMD
(= duration x 2.4 GHz / (MFMAs of the launch / 1024 SIMDs)).  Results of the diagnostic variants are WRONG by construction."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = ['full', 'noyield', 'nostats', 'nogeom', 'nostore', 'nowriteout', 'noload', 'mfma_only', 'mfma_only_noyield', 'nomfma']

CHILD = r'''
import json, sys, torch
sys.path.insert(0, %r)
import deep_audio_mixer_amd
from deep_audio_mixer_amd import ops
dev = torch.device('cuda', 0)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
out = {}
for name, (B, H, W, c) in (('layer1', (8, 1025, 130, 16)), ('layer2', (8, 513, 65, 32)), ('layer2_b59', (59, 513, 65, 32))):
    x = torch.randn((B, H, W, c), device=dev); dy = torch.randn((B, H, W, c), device=dev)
    wt = torch.randn((c, c, 3, 3), device=dev) * 0.05
    wp, wpt = ops.pack_weights(wt), ops.pack_weights(wt, transpose=True)
    buf = ops.bn_partial_buffer(dev, c)
    sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    mean, invstd = torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5
    out[name + ' fwd plain'] = timeit(lambda: ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1))
    if B == 59: continue
    out[name + ' fwd+stats'] = timeit(lambda: ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1, bn_partial=buf))
    out[name + ' fwd+stats+affine'] = timeit(lambda: ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1, bn_partial=buf, in_scale=sc, in_shift=sh, relu_in=True))
    out[name + ' dgrad+sums (EPI1)'] = timeit(lambda: ops.conv2d_dgrad(dy, wpt, c, H, W, 3, 3, 1, 1, 1, bn_bwd=(x, mean, invstd, sc, sh)))
    if c == 16:
        bits = torch.randint(0, 16, (B, H, W, c // 4), device=dev, dtype=torch.uint8)
        out[name + ' dgrad+res+upsums (EPI2)'] = timeit(lambda: ops.conv2d_dgrad(dy, wpt, c, H, W, 3, 3, 1, 1, 1, res=x, res_mask=dy, res_mask_bits=bits, bn_bwd=(x, mean, invstd, sc, sh)))
print('LADDER ' + json.dumps(out))
''' % ROOT

def run_child(env_extra):
    r = subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, **env_extra), capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('LADDER ')]
    return json.loads(line[0][7:]) if line else r.stderr[-800:]


def table(rows):
    cols = list(next(iter(rows.values())).keys())
    mfma_per_simd = {c: (2.0 * (59 if 'b59' in c else 8) * (1025 * 130 if c.startswith('layer1') else 513 * 65) *
                         (16 * 144 if c.startswith('layer1') else 32 * 288)) / 2048 / 1024 for c in cols}
    print('us per launch')
    print('%-20s' % 'variant' + ''.join('%28s' % c for c in cols))
    for v, r in rows.items():
        print('%-20s' % v + ''.join('%28.1f' % r[c] for c in cols))
    print('clocks per MFMA per SIMD at 2.4 GHz (pipe rate: 32)')
    for v, r in rows.items():
        print('%-20s' % v + ''.join('%28.1f' % (r[c] * 1e-6 * 2.4e9 / mfma_per_simd[c]) for c in cols))


def main():
    rows = {}
    for v in VARIANTS:
        lib = os.path.join(ROOT, 'tools', 'libdam_ladder_%s.so' % v)
        if not os.path.exists(lib):
            continue
        r = run_child({'DAM_LIB_PATH': lib, 'DAM_STRIP_PINGPONG': '1'})
        if isinstance(r, str):
            print(v, 'FAILED', r)
            continue
        rows[v] = r
    table(rows)


if __name__ == '__main__':
    main()
