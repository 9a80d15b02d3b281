#!/usr/bin/env python3
"""The dx-hat ladder: what would it cost the two consumers of a BatchNorm-backward result dc = a (dy . mask) +
b c + k -- the row-streaming weight gradient and the row-ring data gradient -- to form dc in their own loaders from dy and c,
so that the bn_bwd_apply launch (and the 68 MB tensor it writes and they re-read) disappears?  TIMING-ONLY builds
(tools/build_variant.sh <name> <src> -DDAM_DIAG_DXHAT=1|2; results are wrong): every dY / input plane a loader requests is
accompanied by a second plane from a third tensor (real HBM traffic); 1 = loads + one fma per quad, 2 = the full arithmetic
(mask recomputed from the second stream, three fma, four selects).

  python tools/dxhat_ladder.py            # parent: runs every build in its own process, prints the table
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILDS = [('shipped', None), ('wgrad +stream', 'tools/libdam_dxw1.so'), ('wgrad +stream+valu', 'tools/libdam_dxw2.so'),
          ('dgrad +stream', 'tools/libdam_dxs1.so'), ('dgrad +stream+valu', 'tools/libdam_dxs2.so')]


def child():
    import torch
    sys.path.insert(0, ROOT)
    import deep_audio_mixer_amd  # noqa: F401
    from deep_audio_mixer_amd import ops
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    out = {}

    def timed(fn, iters=20):
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / iters
    ops._workspace(dev, 60 << 20)          # the wgrad timing builds read their second stream at +32 M floats of the slab workspace
    for name, (B, H, W, C) in (('L1', (8, 1025, 130, 16)), ('L2', (8, 513, 65, 32))):
        x, dy, third = (torch.randn((B, H, W, C), device=dev) for _ in range(3))
        wt = torch.randn((C, C, 3, 3), device=dev) * 0.05
        wpt = ops.pack_weights(wt, transpose=True)
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        mean, invstd = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
        out[name + ' wgrad'] = timed(lambda: ops.conv2d_wgrad(x, dy, C, 3, 3, 1, 1, 1))
        out[name + ' wgrad +in_affine'] = timed(lambda: ops.conv2d_wgrad(x, dy, C, 3, 3, 1, 1, 1, in_scale=sc, in_shift=sh, relu_in=True))
        out[name + ' dgrad +bn sums'] = timed(lambda: ops.conv2d_dgrad(dy, wpt, C, H, W, 3, 3, 1, 1, 1, bn_bwd=(x, mean, invstd, sc, sh),
                                                                    _diag_bias=third))
        if C == 16:
            y = torch.randn((B, H, W, C), device=dev)
            _, bits = ops.bn_apply(y, sc, sh, relu=True, sign_bits=True)
            up = torch.randn((B, H, W, C), device=dev)
            out[name + ' dgrad +res +upstream sums'] = timed(lambda: ops.conv2d_dgrad(
                dy, wpt, C, H, W, 3, 3, 1, 1, 1, res=x, res_mask=y, res_mask_bits=bits, bn_bwd=(up, mean, invstd, sc, sh), _diag_bias=third))
        # what the fusion would delete: finalize + apply of the backward pass (records given)
        gamma = torch.rand(C, device=dev) + 0.5
        rec = torch.randn(ops._lib.lib().dam_bn_workspace_floats(C), device=dev)
        out[name + ' bn_bwd finalize+apply'] = timed(lambda: ops.bn_backward(dy, None, x, gamma, mean, invstd, True, mask_affine=(sc, sh),
                                                                            partials=(rec, 248)))
    print(json.dumps(out))


def main():
    rows = {}
    for name, lib in BUILDS:
        env = dict(os.environ)
        if lib:
            if not os.path.exists(os.path.join(ROOT, lib)):
                print('(no %s: build it with tools/build_variant.sh)' % lib)
                continue
            env['DAM_LIB_PATH'] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(name, 'FAILED', r.stderr[-1500:])
            continue
        rows[name] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    keys = list(next(iter(rows.values())).keys())
    print('%-34s' % 'us per launch (isolated, C3 shapes)' + ''.join('%22s' % n for n in rows))
    for k in keys:
        print('%-34s' % k + ''.join('%22.1f' % rows[n][k] for n in rows))


if __name__ == '__main__':
    child() if '--child' in sys.argv else main()
