"""GPU: the slip-bound argument of the self-overlapped row-ring convolution (dam_conv_strip.hip, "WHY `tile & 3` CANNOT ALIAS"),
checked on the device.  libdam_hip_diag.so is the shipped library with that one source compiled with -DDAM_STRIP_DIAG_TAGS: the
loader waves tag every geometry-table entry with the tile it describes, every compute wave checks the tag of every entry it turns
into addresses.  A fresh child process (cold instruction cache on its first launches -- the condition under which round 3's
two-buffer table failed) runs the C3 step's 16- and 32-channel stages through it, forward and backward: stem -> layer1.0 ->
layer1.1 (plain, statistics, fused input affine, the data gradients with the BatchNorm-backward / residual / upstream-sum
epilogues) and a 513 x 65 x 32 identity block.  Not a single mismatch may be counted, and the checks must have run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import ctypes, json, sys
import torch
sys.path.insert(0, %(root)r)
import deep_audio_mixer_amd  # noqa: F401
from deep_audio_mixer_amd import _lib, layers, ops
L = _lib.lib()
out = (ctypes.c_uint32 * 2)()
assert L.dam_strip_diag_counters(out, 1) == 0, 'not a diagnostic build'
torch.manual_seed(0)
B = 8
def run_chain(cin, cout, H, W, stem):
    blks = [layers.BasicBlock(cout, cout, 1).cuda().train() for _ in range(2)]
    x = torch.randn(B, cin, H, W, device='cuda') if stem else torch.relu(torch.randn(B, H, W, cout, device='cuda')).requires_grad_(True)
    if stem:
        conv = torch.nn.Conv2d(cin, cout, 3, 1, 1, bias=False).cuda()
        bn = torch.nn.BatchNorm2d(cout).cuda().train()
        spec = layers.ConvSpec(cin, cout, 3, 1, 1, in_nchw=True)
        a = layers.ConvBnReluFn.apply(x, conv.weight, None, bn.weight, bn.bias, spec, bn, True)
    else:
        a = x
    for blk in blks:
        a = blk(a)
    a.backward(torch.randn_like(a))
    ops.wgrad_flush()
    torch.cuda.synchronize()
    return bool(torch.isfinite(a).all())
ok1 = run_chain(8, 16, 1025, 130, True)          # stem + layer1: 16 channels, 1025 x 130
ok2 = run_chain(32, 32, 513, 65, False)           # layer2's identity blocks: 32 channels, 513 x 65
assert L.dam_strip_diag_counters(out, 0) == 0
print(json.dumps({'checks': int(out[0]), 'mismatches': int(out[1]), 'finite': ok1 and ok2, 'upstream_hits': layers._UpstreamBn.hits}))
'''


def test_geometry_table_tags_never_mismatch():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, 'deep-audio-mixer_amd', 'libdam_hip_diag.so')
    if not os.path.exists(diag):
        from deep_audio_mixer_amd import build
        build.build_lib()
    assert os.path.exists(diag), 'libdam_hip_diag.so is built by deep-audio-mixer_amd/build.py'
    env = dict(os.environ, DAM_LIB_PATH=diag)
    r = subprocess.run([sys.executable, '-c', CHILD % {'root': root}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    # every compute wave checks every pixel block of every tile: 8 images x 133250 pixels / 16 per block, per launch
    assert out['checks'] > 500000, out
    assert out['mismatches'] == 0, out
    assert out['finite'] and out['upstream_hits'] >= 1, out


def test_shipped_library_has_no_diagnostics(dam_lib):
    import ctypes
    out = (ctypes.c_uint32 * 2)()
    assert dam_lib.dam_strip_diag_counters(out, 0) != 0
