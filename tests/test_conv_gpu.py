"""GPU: implicit-GEMM convolution (forward, data gradient, fused prologue/epilogue) through the C ABI
against torch's CPU conv2d in float64.  fp32 MFMA is an exact fp32 fma chain, so the tolerance is that
of fp32 accumulation: 2e-5 relative to the output's max-abs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope='module')
def ops(dam_lib):
    from deep_audio_mixer_amd import ops
    return ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def close(got, want, tol=TOL):
    got, want = got.double().cpu(), want.double()
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= tol * scale, (err, scale)


CASES = [  # B, Cin, Cout, H, W, k, stride, pad, dil, bias
    (2, 16, 16, 37, 23, 3, 1, 1, 1, False),
    (2, 16, 32, 41, 27, 3, 2, 1, 1, False),
    (2, 32, 64, 21, 14, 1, 2, 0, 1, False),
    (1, 96, 96, 19, 17, 3, 1, 1, 1, False),
    (1, 128, 256, 9, 5, 3, 2, 1, 1, False),
    (2, 256, 256, 5, 5, 3, 1, 1, 1, False),
    (1, 48, 64, 40, 33, 7, 1, 0, 1, True),
    (1, 64, 128, 30, 25, 9, 1, 0, 1, True),
    (2, 16, 32, 45, 31, 5, 1, 0, 1, True),
    (2, 32, 48, 45, 31, 5, 1, 0, 1, True),        # 48 output channels: the three-block weight-gradient tile
    (1, 16, 16, 40, 216, 3, 1, 1, 1, False),     # the reference's native 216-frame rows: wide-row strip loader
    (1, 32, 32, 30, 108, 3, 1, 1, 1, False),
    (2, 16, 32, 21, 130, 3, 2, 1, 1, True),      # too wide for the loader-wave kernel's patch buffers: two column ranges
    (1, 16, 32, 9, 216, 3, 2, 1, 1, False),      # three
]


@pytest.mark.parametrize('case', CASES)
def test_conv_fwd_dgrad(ops, case):
    B, Ci, Co, H, W, k, s, p, d, use_bias = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g) if use_bias else None
    want = F.conv2d(x.double(), w.double(), None if b is None else b.double(), s, p, d)
    wp = ops.pack_weights(w.cuda())
    y = ops.conv2d_fwd(nhwc(x).cuda(), wp, Co, k, k, s, p, d, bias=None if b is None else b.cuda())
    assert y.shape == (B, want.shape[2], want.shape[3], Co)
    close(nchw(y), want)
    # data gradient: dx = conv_transpose(dy)
    dy = torch.randn(want.shape, generator=g)
    want_dx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), s, p, d)
    wpt = ops.pack_weights(w.cuda(), transpose=True)
    dx = ops.conv2d_dgrad(nhwc(dy).cuda(), wpt, Ci, H, W, k, k, s, p, d)
    close(nchw(dx), want_dx)
    # weight gradient
    want_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), s, p, d)
    dw = ops.conv2d_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), Co, k, k, s, p, d)
    assert dw.shape == w.shape
    close(dw, want_dw, 5e-5)
    # residual-fused epilogue: + r * (m > 0)
    if k == 1 and s == 2:
        return      # classes without taps: no fused residual (the shortcut path accumulates instead)
    r, m = torch.randn(B, Ci, H, W, generator=g), torch.randn(B, Ci, H, W, generator=g)
    dx2 = ops.conv2d_dgrad(nhwc(dy).cuda(), wpt, Ci, H, W, k, k, s, p, d, res=nhwc(r).cuda(), res_mask=nhwc(m).cuda())
    close(nchw(dx2), want_dx + r.double() * (m > 0))


@pytest.mark.parametrize('S,k,s,p,d', [(8, 3, 1, 1, 1), (4, 3, 1, 1, 1), (2, 3, 2, 0, 1), (4, 3, 2, 0, 2)])
def test_first_layer_nchw(ops, S, k, s, p, d):
    g = torch.Generator().manual_seed(S)
    x = torch.randn(2, S, 67, 45, generator=g) * 20 - 20
    w = torch.randn(16, S, k, k, generator=g) / (S * k * k) ** 0.5
    b = torch.randn(16, generator=g)
    want = F.conv2d(x.double(), w.double(), b.double(), s, p, d)
    y = ops.conv2d_fwd(x.cuda(), ops.pack_weights(w.cuda()), 16, k, k, s, p, d, bias=b.cuda(), in_nchw=True)
    close(nchw(y), want)
    dy = torch.randn(want.shape, generator=g)
    want_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), s, p, d)
    dw = ops.conv2d_wgrad(x.cuda(), nhwc(dy).cuda(), 16, k, k, s, p, d, in_nchw=True)
    close(dw, want_dw, 5e-5)


def test_fused_bn_relu_prologue(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 33, 29, generator=g)
    sc, sh = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
    w = torch.randn(32, 32, 3, 3, generator=g) / 17
    a = F.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    want = F.conv2d(a, w.double(), None, 1, 1)
    y = ops.conv2d_fwd(nhwc(x).cuda(), ops.pack_weights(w.cuda()), 32, 3, 3, 1, 1, 1, in_scale=sc.cuda(),
                       in_shift=sh.cuda(), relu_in=True)
    close(nchw(y), want)
    dy = torch.randn(want.shape, generator=g)
    want_dw = torch.nn.grad.conv2d_weight(a, w.shape, dy.double(), 1, 1)
    dw = ops.conv2d_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), 32, 3, 3, 1, 1, 1, in_scale=sc.cuda(), in_shift=sh.cuda(),
                          relu_in=True)
    close(dw, want_dw, 5e-5)


def test_shortcut_dgrad_accumulates(ops):
    """1x1 stride-2 shortcut: only the even/even input class receives gradient; it is added in place."""
    g = torch.Generator().manual_seed(9)
    x_shape = (2, 16, 21, 13)
    w = torch.randn(32, 16, 1, 1, generator=g)
    dy = torch.randn(2, 32, 11, 7, generator=g)
    base = torch.randn(2, 21, 13, 16, generator=g)
    want = nchw(base).double() + torch.nn.grad.conv2d_input(x_shape, w.double(), dy.double(), 2, 0, 1)
    dx = base.cuda()
    ops.conv2d_dgrad(nhwc(dy).cuda(), ops.pack_weights(w.cuda(), transpose=True), 16, 21, 13, 1, 1, 2, 0, 1,
                     accumulate_into=dx)
    close(nchw(dx), want)


def test_full_resolution_layer1(ops):
    """ResNet layer1 conv at the BASELINE size (16->16, 1025x130): exact-shape check on a strided sample."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 16, 1025, 130, generator=g)
    w = torch.randn(16, 16, 3, 3, generator=g) / 12
    y = nchw(ops.conv2d_fwd(nhwc(x).cuda(), ops.pack_weights(w.cuda()), 16, 3, 3, 1, 1, 1)).cpu()
    want = F.conv2d(x, w, None, 1, 1)
    close(y, want, 5e-5)
    dy = torch.randn(2, 16, 1025, 130, generator=g)
    want_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), 1, 1)
    close(ops.conv2d_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), 16, 3, 3, 1, 1, 1), want_dw, 1e-4)


@pytest.mark.parametrize('B,C,H,W', [(2, 16, 23, 130), (1, 16, 9, 127), (2, 32, 21, 65), (1, 64, 19, 33), (3, 32, 4, 66),
                                     (1, 64, 37, 31), (1, 16, 9, 216), (2, 32, 7, 108), (1, 64, 11, 54), (1, 16, 3, 213),
                                     (8, 16, 165, 130), (2, 96, 33, 17), (1, 128, 65, 9), (2, 256, 33, 5)])
def test_wgrad_row_streaming_shapes(ops, B, C, H, W):
    """3x3 / stride 1 / pad 1 weight gradient at the widths of the ResNet stages (row-streaming kernel), ragged strips.
    (8, 16, 165, 130): 32 strips of 5 or 6 rows per image -- last slots of ONE row, which the one-block tile splits over its four
    waves by MFMA steps; 17-pixel rows: the four-loader-wave instantiation; 5-pixel rows: the tile kernel with a 176-pixel tile."""
    g = torch.Generator().manual_seed(B * 1000 + C + H + W)
    x = torch.randn(B, C, H, W, generator=g)
    dy = torch.randn(B, C, H, W, generator=g)
    want = torch.nn.grad.conv2d_weight(x.double(), (C, C, 3, 3), dy.double(), 1, 1)
    got = ops.conv2d_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), C, 3, 3, 1, 1, 1)
    assert got.shape == (C, C, 3, 3)
    close(got, want, 5e-5)


@pytest.mark.parametrize('B,Ci,Co,H,W', [(2, 16, 32, 41, 130), (1, 16, 32, 24, 129), (2, 32, 64, 37, 65), (1, 32, 64, 20, 66),
                                         (1, 64, 96, 33, 33), (2, 64, 96, 18, 34), (1, 32, 64, 11, 108), (1, 64, 96, 13, 54),
                                         (8, 16, 32, 131, 130), (1, 64, 96, 3, 33), (1, 16, 32, 1, 130),
                                         (2, 96, 128, 65, 17), (1, 96, 128, 22, 18), (2, 128, 256, 33, 9), (1, 128, 256, 12, 10)])
def test_wgrad_strided_row_streaming_shapes(ops, B, Ci, Co, H, W):
    """3x3 / stride 2 / pad 1 weight gradient of the down-sampling convolutions (row-streaming kernel with column-parity
    planes): odd and even heights and widths (bottom padding row / right padding column present or not), ragged strips,
    strips of a single slot, the widths of the ResNet stages at 130 and 216 frames."""
    g = torch.Generator().manual_seed(B * 1000 + Ci + H + W)
    x = torch.randn(B, Ci, H, W, generator=g)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(B, Co, Ho, Wo, generator=g)
    want = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, 3, 3), dy.double(), 2, 1)
    got = ops.conv2d_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), Co, 3, 3, 2, 1, 1)
    assert got.shape == (Co, Ci, 3, 3)
    close(got, want, 5e-5)


@pytest.mark.parametrize('B,Ci,Co,H,W,k,s,p', [(2, 16, 32, 41, 65, 3, 2, 1), (1, 32, 64, 33, 34, 3, 2, 1), (2, 96, 96, 9, 17, 3, 1, 1),
                                              (1, 16, 32, 9, 216, 3, 2, 1), (1, 16, 16, 21, 40, 3, 2, 1),
                                              (2, 16, 32, 41, 65, 1, 2, 0), (1, 64, 96, 21, 33, 1, 2, 0), (1, 48, 48, 20, 30, 3, 1, 1)])
def test_wgrad_direct_kernel_shapes(ops, B, Ci, Co, H, W, k, s, p):
    """Weight gradient without LDS staging: strided 3x3, narrow-row 3x3 and 1x1 shortcut shapes (ragged row ends, padding
    columns and rows, odd channel-block counts)."""
    g = torch.Generator().manual_seed(Ci + Co + H + W + k)
    x = torch.randn(B, Ci, H, W, generator=g)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(B, Co, Ho, Wo, generator=g)
    want = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, k, k), dy.double(), s, p)
    got = ops.conv2d_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), Co, k, k, s, p, 1)
    assert got.shape == (Co, Ci, k, k)
    close(got, want, 5e-5)


def test_stem_relayout_and_nhwc16_path(ops):
    """NCHW stem input re-laid as NHWC-16 (zero channels), and the stem convolution / weight gradient through it."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 8, 37, 130, generator=g)
    y = ops.nchw_to_nhwc16(x.cuda()).cpu()
    assert y.shape == (2, 37, 130, 16)
    assert torch.equal(y[..., :8], x.permute(0, 2, 3, 1)) and torch.count_nonzero(y[..., 8:]) == 0
    from deep_audio_mixer_amd.layers import ConvSpec
    spec = ConvSpec(8, 16, 3, 1, 1, in_nchw=True)
    assert spec.nhwc16 is not None
    w = torch.randn(16, 8, 3, 3, generator=g) / 9
    c = spec.nhwc16.fwd(ops.nchw_to_nhwc16(x.cuda()), w.cuda())
    close(nchw(c), F.conv2d(x.double(), w.double(), None, 1, 1))
    dy = torch.randn(2, 16, 37, 130, generator=g)
    dw = spec.nhwc16.wgrad(ops.nchw_to_nhwc16(x.cuda()), nhwc(dy).cuda())
    assert dw.shape == (16, 8, 3, 3)
    close(dw, torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), 1, 1), 5e-5)


def test_fused_bn_statistics_epilogue(ops):
    """The strip kernel's BatchNorm partial records, merged by dam_bn_finalize_f32, equal the two-pass statistics of the
    conv output (mean far from zero on purpose: dB-valued activations)."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 16, 150, 130, generator=g) * 5 + 40
    w = torch.randn(32, 16, 3, 3, generator=g) / 12
    want = F.conv2d(x.double(), w.double(), None, 1, 1)
    buf = ops.bn_partial_buffer(torch.device('cuda'), 32)
    y, parts = ops.conv2d_fwd(nhwc(x).cuda(), ops.pack_weights(w.cuda()), 32, 3, 3, 1, 1, 1, bn_partial=buf)
    assert parts > 0
    close(nchw(y), want)
    gamma, beta = torch.rand(32, device='cuda') + 0.5, torch.randn(32, device='cuda')
    rm, rv = torch.zeros(32, device='cuda'), torch.ones(32, device='cuda')
    nbt = torch.zeros((), dtype=torch.int64, device='cuda')
    mean, invstd, scale, shift = ops.bn_finalize(buf, parts, gamma, beta, rm, rv, nbt, 0.1, 1e-5)
    wm, wv = want.mean((0, 2, 3)), want.var((0, 2, 3), unbiased=False)
    assert (mean.double().cpu() - wm).abs().max() <= 1e-5 * wm.abs().max()
    assert ((invstd.double().cpu() - 1 / (wv + 1e-5).sqrt()).abs() * (wv + 1e-5).sqrt()).max() <= 2e-5
    n = want.numel() / 32
    assert torch.allclose(rv.double().cpu(), 0.9 + 0.1 * wv * n / (n - 1), rtol=1e-4)
    assert int(nbt.item()) == 1
    m2, i2, _, _ = ops.bn_stats(y, gamma, beta, None, None, None, 0.1, 1e-5)
    assert torch.allclose(mean, m2, rtol=1e-5, atol=1e-4) and torch.allclose(invstd, i2, rtol=1e-4)


def test_in_kernel_finalize_matches_finalize_launch(ops):
    """"The last workgroup finalizes" (dam_bn_fin.h): statistics + running-stat update produced inside the convolution
    launch equal the separate dam_bn_finalize_f32 launch, over repeated launches (the arrival counter returns to zero) and
    for the two-kernel statistics / backward paths that use the same hand-off."""
    g = torch.Generator().manual_seed(12)
    x = torch.randn(4, 16, 311, 130, generator=g) * 4 + 30
    w = torch.randn(16, 16, 3, 3, generator=g) / 12
    xd, wp = nhwc(x).cuda(), ops.pack_weights(w.cuda())
    gamma, beta = torch.rand(16, device='cuda') + 0.5, torch.randn(16, device='cuda')
    buf = ops.bn_partial_buffer(torch.device('cuda'), 16)
    y, parts = ops.conv2d_fwd(xd, wp, 16, 3, 3, 1, 1, 1, bn_partial=buf)
    rm0, rv0, nb0 = torch.zeros(16, device='cuda'), torch.ones(16, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda')
    want = ops.bn_finalize(buf, parts, gamma, beta, rm0, rv0, nb0, 0.1, 1e-5)
    rm, rv, nbt = torch.zeros(16, device='cuda'), torch.ones(16, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda')
    ops.INKERNEL_FINALIZE = True          # off by default (slower than the finalize launch, ops.py); still has to be right
    try:
        _check_in_kernel_finalize(ops, g, xd, wp, gamma, beta, buf, y, parts, want, rm, rv, nbt, rm0)
    finally:
        ops.INKERNEL_FINALIZE = False


def _check_in_kernel_finalize(ops, g, xd, wp, gamma, beta, buf, y, parts, want, rm, rv, nbt, rm0):
    for rep in range(3):
        y2, parts2, out4 = ops.conv2d_fwd(xd, wp, 16, 3, 3, 1, 1, 1, bn_partial=buf, bn=(gamma, beta, rm, rv, nbt, 0.1, 1e-5))
        assert parts2 == parts and torch.equal(y2, y)
        for got, w_ in zip(out4, want):
            assert torch.allclose(got, w_, rtol=1e-6, atol=1e-6)
        assert int(ops.arrival_counter(torch.device('cuda'))[0].item()) == 0
    assert int(nbt.item()) == 3
    assert torch.allclose(rm, rm0 * (1 + 0.9 + 0.81), rtol=1e-5)              # three momentum updates towards the same mean
    # the stand-alone statistics / backward kernels finalize in their last workgroup too: against float64
    for C, P in ((16, 70001), (96, 5000), (256, 330), (1024, 50)):
        t = torch.randn(P, C, generator=g) * 2 + 3
        gm, bt = torch.rand(C) + 0.5, torch.randn(C)
        mean, invstd, scale, shift = ops.bn_stats(t.cuda().view(1, P, 1, C), gm.cuda(), bt.cuda(), None, None, None, 0.1, 1e-5)
        td = t.double()
        close(mean, td.mean(0), 1e-6)
        close(invstd, 1 / (td.var(0, unbiased=False) + 1e-5).sqrt(), 1e-5)
        close(shift, bt.double() - td.mean(0) * gm.double() / (td.var(0, unbiased=False) + 1e-5).sqrt(), 2e-5)


@pytest.mark.parametrize('B,C,H,W,s', [(2, 64, 20, 33, 1), (2, 96, 11, 17, 1), (1, 128, 13, 9, 1), (2, 256, 7, 5, 1),
                                       (2, 64, 12, 12, 2), (1, 64, 9, 54, 1)])
def test_loader_wave_tile_kernel(ops, B, C, H, W, s, monkeypatch):
    """conv_pipe_kernel (thick 3x3 stages): every tile of its rule, few persistent workgroups (each walks many units, so the
    chunk stream crosses unit and image boundaries), the fused input affine, the residual epilogue -- and the tile kernel on
    the same inputs (DAM_NO_PIPE) as a second witness."""
    g = torch.Generator().manual_seed(C + W)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    want = F.conv2d(x.double(), w.double(), None, s, 1)
    a = F.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    want_aff = F.conv2d(a, w.double(), None, s, 1)
    wp, wpt = ops.pack_weights(w.cuda()), ops.pack_weights(w.cuda(), transpose=True)
    xd = nhwc(x).cuda()
    dy = torch.randn(want.shape, generator=g)
    want_dx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), s, 1)
    r, m = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    envs = [{}, {'DAM_PIPE_WGS': '3'}, {'DAM_TILE': '1x4'}, {'DAM_TILE': '2x2', 'DAM_PIPE_WGS': '2'}, {'DAM_TILE': '1x1'},
            {'DAM_TILE': '2x1', 'DAM_PIPE_WGS': '5'}, {'DAM_NO_PIPE': '1'}]
    for env in envs:
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            close(nchw(ops.conv2d_fwd(xd, wp, C, 3, 3, s, 1, 1)), want)
            close(nchw(ops.conv2d_fwd(xd, wp, C, 3, 3, s, 1, 1, in_scale=sc.cuda(), in_shift=sh.cuda(), relu_in=True)), want_aff)
            if s == 1:
                dx = ops.conv2d_dgrad(nhwc(dy).cuda(), wpt, C, H, W, 3, 3, 1, 1, 1, res=nhwc(r).cuda(), res_mask=nhwc(m).cuda())
                close(nchw(dx), want_dx + r.double() * (m > 0))


def test_deferred_weight_gradient_reductions(ops):
    """conv2d_wgrad(defer=True) + wgrad_flush(): one batched reduction launch, bitwise the immediate result (same slabs,
    same summation order), for a mix of the three slab kernels (row-streaming, direct, tile)."""
    g = torch.Generator().manual_seed(11)
    shapes = [(2, 16, 16, 23, 130, 3, 1, 1), (2, 32, 64, 33, 34, 3, 2, 1), (1, 256, 256, 7, 5, 3, 1, 1), (2, 32, 64, 21, 14, 1, 2, 0)]
    calls = []
    for B, Ci, Co, H, W, k, s, p in shapes:
        x = torch.randn(B, H, W, Ci, generator=g).cuda()
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        dy = torch.randn(B, Ho, Wo, Co, generator=g).cuda()
        calls.append((x, dy, Co, k, k, s, p, 1))
    now = [ops.conv2d_wgrad(*c) for c in calls]
    outs = [torch.full_like(t, float('nan')) for t in now]
    for c, o in zip(calls, outs):
        ops.conv2d_wgrad(*c, out=o, defer=True)
    assert all(torch.isnan(o).all() for o in outs)          # nothing is written before the flush
    ops.wgrad_flush()
    for a, b in zip(now, outs):
        assert torch.equal(a, b)
    ops.wgrad_flush()                                       # idempotent
    with pytest.raises(ValueError):
        ops.conv2d_wgrad(*calls[0], defer=True)


@pytest.mark.parametrize('B,C,H,W,s,Co', [(2, 96, 21, 17, 1, 96), (3, 128, 13, 9, 1, 128), (2, 256, 9, 5, 1, 256),
                                          (2, 64, 24, 20, 2, 96), (2, 64, 20, 33, 1, 64)])
def test_loader_wave_kernel_statistics_epilogue(ops, B, C, H, W, s, Co):
    """conv_pipe_kernel's per-wave BatchNorm records (with the fused input affine in front, as conv2 of a block runs it),
    merged by dam_bn_finalize_f32, against the two-pass statistics of the float64 convolution; mean far from zero."""
    g = torch.Generator().manual_seed(C + W + s)
    x = torch.randn(B, C, H, W, generator=g) * 3 + 7
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    want = F.conv2d(x.double(), w.double(), None, s, 1)
    buf = ops.bn_partial_buffer(torch.device('cuda'), Co)
    y, parts = ops.conv2d_fwd(nhwc(x).cuda(), ops.pack_weights(w.cuda()), Co, 3, 3, s, 1, 1, bn_partial=buf)
    close(nchw(y), want)
    assert parts > 0, 'this shape is expected to take the loader-wave kernel with its statistics epilogue'
    gamma, beta = torch.rand(Co, device='cuda') + 0.5, torch.randn(Co, device='cuda')
    mean, invstd, scale, shift = ops.bn_finalize(buf, parts, gamma, beta, None, None, None, 0.1, 1e-5)
    wm, wv = want.mean((0, 2, 3)), want.var((0, 2, 3), unbiased=False)
    assert (mean.double().cpu() - wm).abs().max() <= 1e-5 * wm.abs().max()
    assert ((invstd.double().cpu() - 1 / (wv + 1e-5).sqrt()).abs() * (wv + 1e-5).sqrt()).max() <= 2e-5


@pytest.mark.parametrize('B,C,H,W', [(2, 16, 41, 130), (3, 16, 23, 127), (2, 32, 37, 65), (1, 32, 20, 66), (1, 16, 9, 216),
                                     (1, 32, 7, 108), (1, 64, 19, 33), (2, 96, 33, 17), (2, 128, 17, 9), (1, 256, 9, 5),
                                     (8, 64, 65, 33)])
def test_dgrad_bn_backward_sums_epilogue(ops, B, C, H, W):
    """conv2d_dgrad(bn_bwd=...): the data gradient is bitwise the plain one, and the records it leaves are the two sums of
    the BatchNorm backward pass of relu(bn(x)) -- checked against float64 and through bn_backward(partials=) against the
    separate pass.  Both kernels that have the epilogue are covered (row-ring kernel: 16 / 32 channels; loader-wave tile kernel:
    thick layers); a thick layer whose tiles would leave more than 1024 records returns partials None (last case)."""
    g = torch.Generator().manual_seed(B * 100 + C + H + W)
    dy = torch.randn(B, H, W, C, generator=g).cuda()
    w = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    x = (torch.randn(B, H, W, C, generator=g) * 1.5 + 0.3).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (0.3 * torch.randn(C, generator=g)).cuda()
    mean = x.double().mean((0, 1, 2))
    invstd = 1.0 / torch.sqrt(x.double().var((0, 1, 2), unbiased=False) + 1e-5)
    mean, invstd = mean.float(), invstd.float()
    msc = gamma * invstd
    msh = beta - mean * msc
    wpt = ops.pack_weights(w.cuda(), transpose=True)
    plain = ops.conv2d_dgrad(dy, wpt, C, H, W, 3, 3, 1, 1, 1)
    dx, partials = ops.conv2d_dgrad(dy, wpt, C, H, W, 3, 3, 1, 1, 1, bn_bwd=(x, mean, invstd, msc, msh))
    assert torch.equal(dx, plain)
    if partials is None:
        assert (B, C) == (8, 64)
        return
    assert (B, C) != (8, 64)
    rec, parts = partials
    assert 0 < parts <= 1024
    sums = rec[:parts * C * 2].view(parts, C, 2).double().sum(0).cpu()
    mask = (x.double() * msc.double() + msh.double()) > 0
    dz = dx.double() * mask
    xhat = (x.double() - mean.double()) * invstd.double()
    want_a, want_b = dz.sum((0, 1, 2)).cpu(), (dz * xhat).sum((0, 1, 2)).cpu()
    n = B * H * W
    assert (sums[:, 0] - want_a).abs().max() <= 2e-5 * dz.abs().max().item() * n ** 0.5 + 1e-3
    assert (sums[:, 1] - want_b).abs().max() <= 2e-5 * (dz * xhat).abs().max().item() * n ** 0.5 + 1e-3
    for tr in (True, False):
        a = ops.bn_backward(dx, None, x, gamma, mean, invstd, tr, mask_affine=(msc, msh))
        b = ops.bn_backward(dx, None, x, gamma, mean, invstd, tr, mask_affine=(msc, msh), partials=partials)
        for u, v in zip(a, b):
            close(v, u.double().cpu(), 2e-5)


def test_full_size_strided_wgrad_and_backward_sums(ops):
    """BASELINE size (C3: batch 8, 1025x130): the strided 16->32 weight gradient (stride-2 row-streaming kernel, 232 strips of
    nine two-row slots) against float64, and the BatchNorm-backward sums of the layer1 data-gradient epilogue (248 records)
    against float64 sums of the kernel's own output."""
    g = torch.Generator().manual_seed(77)
    B, H, W = 8, 1025, 130
    x = torch.randn(B, 16, H, W, generator=g)
    dy = torch.randn(B, 32, 513, 65, generator=g)
    want = torch.nn.grad.conv2d_weight(x.double(), (32, 16, 3, 3), dy.double(), 2, 1)
    got = ops.conv2d_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), 32, 3, 3, 2, 1, 1)
    close(got, want, 1e-4)
    # sums epilogue: dx of a 16->16 3x3 convolution's data gradient, x = the BatchNorm input
    dy1 = torch.randn(B, H, W, 16, generator=g).cuda()
    w = torch.randn(16, 16, 3, 3, generator=g) / 12
    c1 = (torch.randn(B, H, W, 16, generator=g) * 2 - 0.5).cuda()
    gamma, beta = (torch.rand(16, generator=g) + 0.5).cuda(), (0.3 * torch.randn(16, generator=g)).cuda()
    mean = c1.double().mean((0, 1, 2)).float()
    invstd = (1.0 / torch.sqrt(c1.double().var((0, 1, 2), unbiased=False) + 1e-5)).float()
    msc = gamma * invstd
    msh = beta - mean * msc
    wpt = ops.pack_weights(w.cuda(), transpose=True)
    dx, partials = ops.conv2d_dgrad(dy1, wpt, 16, H, W, 3, 3, 1, 1, 1, bn_bwd=(c1, mean, invstd, msc, msh))
    assert partials is not None and partials[1] == 248
    assert torch.equal(dx, ops.conv2d_dgrad(dy1, wpt, 16, H, W, 3, 3, 1, 1, 1))
    rec, parts = partials
    sums = rec[:parts * 32].view(parts, 16, 2).double().sum(0)
    dz = dx.double() * ((c1.double() * msc.double() + msh.double()) > 0)
    xhat = (c1.double() - mean.double()) * invstd.double()
    n = B * H * W
    for col, ref in ((0, dz.sum((0, 1, 2))), (1, (dz * xhat).sum((0, 1, 2)))):
        assert (sums[:, col] - ref).abs().max().item() <= 1e-5 * n ** 0.5 * dz.abs().max().item() + 1e-2


@pytest.mark.parametrize('B,C,H,W', [(2, 16, 41, 130), (3, 16, 23, 127), (1, 16, 9, 216), (2, 32, 21, 65), (1, 64, 19, 33)])
def test_dgrad_identity_shortcut_with_upstream_sums(ops, B, C, H, W):
    """conv2d_dgrad(res=, res_mask=, res_mask_bits=, bn_bwd=): the data gradient that carries an identity shortcut is the
    gradient reaching the block input, i.e. relu(bn(x)) of the layer upstream (the stem in front of the first block).  The
    result is bitwise the float-mask launch; the 16-channel row-ring kernel also leaves the upstream BatchNorm's two sums,
    the other kernels return partials None."""
    g = torch.Generator().manual_seed(B * 10 + C + H + W)
    dy = torch.randn(B, H, W, C, generator=g).cuda()
    w = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    r = torch.randn(B, H, W, C, generator=g).cuda()
    m = torch.randn(B, H, W, C, generator=g).cuda()
    bits = ((m > 0).view(B, H, W, C // 4, 4).to(torch.int32) * torch.tensor([1, 2, 4, 8], device='cuda', dtype=torch.int32)
            ).sum(-1).to(torch.uint8).contiguous()
    x = (torch.randn(B, H, W, C, generator=g) * 1.3 + 0.2).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (0.3 * torch.randn(C, generator=g)).cuda()
    mean = x.double().mean((0, 1, 2)).float()
    invstd = (1.0 / torch.sqrt(x.double().var((0, 1, 2), unbiased=False) + 1e-5)).float()
    msc = gamma * invstd
    msh = beta - mean * msc
    wpt = ops.pack_weights(w.cuda(), transpose=True)
    plain = ops.conv2d_dgrad(dy, wpt, C, H, W, 3, 3, 1, 1, 1, res=r, res_mask=m)
    dx, partials = ops.conv2d_dgrad(dy, wpt, C, H, W, 3, 3, 1, 1, 1, res=r, res_mask=m, res_mask_bits=bits,
                                    bn_bwd=(x, mean, invstd, msc, msh))
    assert torch.equal(dx, plain)
    if C != 16:
        assert partials is None
        return
    rec, parts = partials
    sums = rec[:parts * C * 2].view(parts, C, 2).double().sum(0).cpu()
    dz = dx.double() * ((x.double() * msc.double() + msh.double()) > 0)
    xhat = (x.double() - mean.double()) * invstd.double()
    n = B * H * W
    for col, ref in ((0, dz.sum((0, 1, 2)).cpu()), (1, (dz * xhat).sum((0, 1, 2)).cpu())):
        assert (sums[:, col] - ref).abs().max().item() <= 2e-5 * n ** 0.5 * dz.abs().max().item() + 1e-3
    with pytest.raises(ValueError):
        ops.conv2d_dgrad(dy, wpt, C, H, W, 3, 3, 1, 1, 1, res=r, res_mask=m, bn_bwd=(x, mean, invstd, msc, msh))
    # the upstream mask as sign bytes (the upstream layer is a residual block's relu(bn2(c2) + shortcut))
    um = torch.randn(B, H, W, C, generator=g).cuda()
    ubits = ((um > 0).view(B, H, W, C // 4, 4).to(torch.int32) * torch.tensor([1, 2, 4, 8], device='cuda', dtype=torch.int32)
             ).sum(-1).to(torch.uint8).contiguous()
    dx3, partials3 = ops.conv2d_dgrad(dy, wpt, C, H, W, 3, 3, 1, 1, 1, res=r, res_mask=m, res_mask_bits=bits,
                                      bn_bwd=(x, mean, invstd, None, None, ubits))
    assert torch.equal(dx3, plain) and partials3 is not None
    rec, parts = partials3
    sums = rec[:parts * C * 2].view(parts, C, 2).double().sum(0).cpu()
    dz = dx3.double() * (um > 0)
    for col, ref in ((0, dz.sum((0, 1, 2)).cpu()), (1, (dz * xhat).sum((0, 1, 2)).cpu())):
        assert (sums[:, col] - ref).abs().max().item() <= 2e-5 * n ** 0.5 * dz.abs().max().item() + 1e-3


@pytest.mark.parametrize('C,H,W', [(128, 65, 9), (256, 33, 5)])
def test_loader_wave_kernel_k_split_matches_unsplit(ops, monkeypatch, C, H, W):
    """The deep stages' one-block tile splits K over the wave pairs (32-pixel units, partial sums handed over through LDS).
    At the benchmark's shapes (batch 8), with the fused input affine and the statistics epilogue: output and merged statistics
    agree with the unsplit form (DAM_PIPE_KS=1) to float32 summation-order noise, and with float64."""
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(8, H, W, C, generator=g).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5))
    wp = ops.pack_weights(w.cuda())
    sc, sh = (torch.rand(C, generator=g) + 0.5).cuda(), (0.2 * torch.randn(C, generator=g)).cuda()
    gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')

    def run():
        buf = ops.bn_partial_buffer(x.device, C).clone()
        y, parts = ops.conv2d_fwd(x, wp, C, 3, 3, 1, 1, 1, in_scale=sc, in_shift=sh, relu_in=True, bn_partial=buf)
        assert parts > 0
        return y, ops.bn_finalize(buf, parts, gamma, beta, None, None, None, 0.1, 1e-5)

    y2, st2 = run()
    monkeypatch.setenv('DAM_PIPE_KS', '1')
    y1, st1 = run()
    monkeypatch.delenv('DAM_PIPE_KS')
    a = torch.relu(x.double().cpu() * sc.double().cpu() + sh.double().cpu()).permute(0, 3, 1, 2)
    want = torch.nn.functional.conv2d(a, w.double(), padding=1).permute(0, 2, 3, 1)
    scale = want.abs().max().item()
    assert (y2.double().cpu() - want).abs().max().item() <= 2e-6 * scale
    assert (y2 - y1).abs().max().item() <= 2e-6 * scale
    assert not torch.equal(y2, y1) or C < 32           # (two different summation orders: identical bits would mean one path ran twice)
    for a2, a1 in zip(st2[:2], st1[:2]):                # save_mean, save_invstd
        assert (a2 - a1).abs().max().item() <= 2e-6 * a1.abs().max().item()


@pytest.mark.parametrize('B,C,H,W', [(2, 16, 21, 130), (2, 32, 19, 65), (1, 64, 23, 33), (2, 128, 17, 9)])
def test_fused_input_affine_without_relu(ops, B, C, H, W):
    """in_scale / in_shift with relu_in = 0 (the loaders apply the ReLU as max(., lower bound) with the bound at -inf then):
    forward and weight gradient of every loader family (strip, loader-wave tile, row-streaming weight gradient) against float64."""
    g = torch.Generator().manual_seed(7 * C + H)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    dy = torch.randn(B, C, H, W, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, 0.3 * torch.randn(C, generator=g)
    a = x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)           # negative values stay
    assert (a < 0).any()
    want = torch.nn.functional.conv2d(a, w.double(), padding=1)
    got = ops.conv2d_fwd(nhwc(x).cuda(), ops.pack_weights(w.cuda()), C, 3, 3, 1, 1, 1, in_scale=sc.cuda(), in_shift=sh.cuda(),
                         relu_in=False)
    close(nchw(got), want, 2e-5)
    want_dw = torch.nn.grad.conv2d_weight(a, w.shape, dy.double(), 1, 1)
    got_dw = ops.conv2d_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), C, 3, 3, 1, 1, 1, in_scale=sc.cuda(), in_shift=sh.cuda(), relu_in=False)
    close(got_dw, want_dw, 5e-5)



def test_pack_weights_multi_equals_single(dam_lib):
    """dam_conv_pack_weights_multi_f32 (one launch for every convolution of a model, a thread per (n, k) position walking its taps)
    against dam_conv_pack_weights_f32 tensor by tensor, bit for bit: 3x3, 1x1, 5x5, 9x9, channel counts that are not multiples
    of 16 (zero padding), forward and transposed images."""
    from deep_audio_mixer_amd import ops
    L = dam_lib
    g = torch.Generator().manual_seed(5)
    shapes = [(16, 8, 3, 3), (16, 16, 3, 3), (32, 16, 1, 1), (96, 64, 3, 3), (48, 32, 5, 5), (128, 64, 9, 9), (256, 256, 3, 3), (20, 4, 3, 3)]
    ws = [torch.randn(s, generator=g).cuda() for s in shapes]
    rows, outs, want = [], [], []
    for w in ws:
        o, i, kh, kw = w.shape
        for transpose in (False, True):
            n_out, k_in = (i, o) if transpose else (o, i)
            n = L.dam_conv_packed_weight_count(n_out, k_in, kh, kw)
            buf = torch.full((n,), float('nan'), device='cuda')
            outs.append(buf)
            want.append(ops.pack_weights(w, transpose=transpose))
            rows.append([w.data_ptr(), buf.data_ptr(), o, i, kh, kw, int(transpose), n])
    desc = torch.tensor(rows, dtype=torch.int64).cuda()
    ops.pack_weights_multi(desc, len(rows), max(r[7] for r in rows))
    torch.cuda.synchronize()
    for got, exp, r in zip(outs, want, rows):
        assert torch.equal(got, exp), r[2:7]


@pytest.mark.parametrize('B,Ci,Co,H,W', [(2, 16, 32, 41, 27), (2, 16, 32, 40, 28), (8, 16, 32, 1025, 130), (2, 32, 64, 513, 65),
                                          (1, 32, 64, 34, 17), (1, 16, 32, 3, 3), (2, 16, 32, 21, 131),
                                          # the wide blocks (weights streamed from L2): the ResNet's layer4.0 / 5.0 / 6.0, ragged, two blocks per wave
                                          (8, 64, 96, 257, 33), (8, 96, 128, 129, 17), (8, 128, 256, 65, 9), (1, 64, 96, 7, 5),
                                          (2, 64, 96, 513, 130), (1, 32, 32, 9, 12)])
@pytest.mark.parametrize('pair', [False, True], ids=['alone', 'with_shortcut'])
def test_strided_dgrad_one_launch(ops, monkeypatch, B, Ci, Co, H, W, pair):
    """dam_dgrad_s2_3x3_f32: the data gradient of a 3x3 / stride-2 / pad-1 convolution's input -- all four output parity classes
    from one read of dy, optionally + the 1x1 / stride-2 shortcut's gradient at the (even, even) pixels -- against
    torch.nn.grad.conv2d_input in float64, odd and even sizes, ragged pixel segments, and bit for bit the same as ... no: within
    float32 summation order of the parity-class launches it replaces (DAM_NO_DGRAD_S2)."""
    g = torch.Generator().manual_seed(B * 100 + H + W + pair)
    Hd, Wd = (H + 1) // 2, (W + 1) // 2
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    dy = torch.randn(B, Co, Hd, Wd, generator=g)
    want = torch.nn.grad.conv2d_input((B, Ci, H, W), w.double(), dy.double(), 2, 1, 1)
    wpt = ops.pack_weights(w.cuda(), transpose=True)
    kw = {}
    if pair:
        wsc = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
        ds = torch.randn(B, Co, Hd, Wd, generator=g)
        want = want + torch.nn.grad.conv2d_input((B, Ci, H, W), wsc.double(), ds.double(), 2, 0, 1)
        kw['pair_1x1'] = (nhwc(ds).cuda(), ops.pack_weights(wsc.cuda(), transpose=True))
    assert ops.DGRAD_S2
    taken = ops.dgrad_s2_launches
    dx = ops.conv2d_dgrad(nhwc(dy).cuda(), wpt, Ci, H, W, 3, 3, 2, 1, 1, **kw)
    assert ops.dgrad_s2_launches == taken + 1, 'the shape fell back to the parity-class launches'
    close(nchw(dx), want, 2e-5)
    monkeypatch.setattr(ops, 'DGRAD_S2', False)                       # the launches it replaces: same numbers to rounding
    dx_old = ops.conv2d_dgrad(nhwc(dy).cuda(), wpt, Ci, H, W, 3, 3, 2, 1, 1, **kw)
    close(dx, dx_old.cpu(), 1e-5)


@pytest.mark.parametrize('B,Ci,Co,H,W', [(2, 16, 32, 41, 27), (2, 16, 32, 40, 28), (8, 16, 32, 1025, 130), (8, 32, 64, 513, 65),
                                          (1, 32, 64, 34, 17), (1, 16, 32, 3, 3), (2, 16, 32, 21, 131), (1, 32, 64, 2, 2),
                                          # the wide blocks (weights streamed from L2): the ResNet's layer4.0 / 5.0 / 6.0, ragged
                                          (8, 64, 96, 257, 33), (8, 96, 128, 129, 17), (8, 128, 256, 65, 9), (1, 64, 96, 7, 5),
                                          (3, 32, 48, 19, 23)])
def test_downsampling_pair_forward_one_launch(ops, B, Ci, Co, H, W):
    """dam_conv_s2_pair_fwd_f32: a down-sampling block's conv1 (3x3 / stride 2 / pad 1) and its 1x1 / stride-2 shortcut convolution
    from one read of x, with the BatchNorm statistics records of both outputs -- against torch's conv2d in float64, odd and even
    sizes, units that straddle rows and images; the records merged (dam_bn_finalize_pair_f32) against the float64 mean / variance of
    the float64 outputs, and their pixel counts against the tensor's."""
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    x = torch.randn(B, Ci, H, W, generator=g) + 0.3 * torch.randn(1, Ci, 1, 1, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    wsc = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    want1 = F.conv2d(x.double(), w.double(), None, 2, 1)
    wants = F.conv2d(x.double(), wsc.double(), None, 2, 0)
    got = ops.conv_s2_pair_fwd(nhwc(x).cuda(), ops.pack_weights(w.cuda()), ops.pack_weights(wsc.cuda()), Co)
    assert got is not None, 'the shape fell back to the separate launches'
    c1, cs, (p1, p2, parts) = got
    close(nchw(c1), want1)
    close(nchw(cs), wants)
    npx = want1.numel() // Co
    for rec in (p1, p2):
        r = rec[:parts * Co * 3].view(parts, Co, 3).double().cpu()
        assert torch.equal(r[:, :, 0].sum(0), torch.full((Co,), float(npx), dtype=torch.float64))
        assert (r[:, :, 2] >= 0).all()
    mk = lambda: (torch.ones(Co).cuda(), torch.zeros(Co).cuda(), torch.zeros(Co).cuda(), torch.ones(Co).cuda(),
                  torch.zeros((), dtype=torch.int64).cuda(), 0.1, 1e-5)
    bn_a, bn_b = mk(), mk()
    sa, sb = ops.bn_finalize_pair(p1, p2, parts, bn_a, bn_b)
    for (mean, invstd, scale, shift), want, bn in ((sa, want1, bn_a), (sb, wants, bn_b)):
        var, mu = torch.var_mean(want, dim=(0, 2, 3), unbiased=False)
        close(mean, mu, 1e-5)
        close(invstd, 1.0 / torch.sqrt(var + 1e-5), 1e-5)
        close(scale, 1.0 / torch.sqrt(var + 1e-5), 1e-5)
        assert int(bn[4].item()) == 1
        if npx > 1:
            close(bn[2], 0.1 * mu, 1e-5)
            close(bn[3], 0.9 + 0.1 * var * npx / (npx - 1), 1e-5)
    # without statistics (evaluation-style call): same outputs
    c1b, csb, none = ops.conv_s2_pair_fwd(nhwc(x).cuda(), ops.pack_weights(w.cuda()), ops.pack_weights(wsc.cuda()), Co, stats=False)
    assert none is None and torch.equal(c1b, c1) and torch.equal(csb, cs)


@pytest.mark.parametrize('B,Ci,Co,H,W', [(2, 16, 32, 41, 27), (2, 16, 32, 40, 28), (2, 32, 64, 34, 17), (4, 32, 64, 257, 33), (1, 16, 32, 3, 3),
                                          (1, 64, 96, 7, 5)])
def test_strided_dgrad_takes_the_upstream_batchnorm_sums(ops, B, Ci, Co, H, W):
    """dam_dgrad_s2_3x3_f32 with bn_bwd: dx reaches y = relu(bn(u) + shortcut) of the block in front; the launch also leaves
    sum(dz) and sum(dz * uhat), dz = dx * (y > 0), per channel -- against float64 sums over the dx it wrote; the wide layers
    (weights streamed) report "not produced"."""
    g = torch.Generator().manual_seed(B * 10 + H + W)
    Hd, Wd = (H + 1) // 2, (W + 1) // 2
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    wsc = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    dy, ds = torch.randn(B, Hd, Wd, Co, generator=g).cuda(), torch.randn(B, Hd, Wd, Co, generator=g).cuda()
    u = (torch.randn(B, H, W, Ci, generator=g) * 1.5 + 0.4).cuda()
    mean, invstd = (torch.randn(Ci, generator=g) * 0.3).cuda(), (torch.rand(Ci, generator=g) + 0.5).cuda()
    y = torch.randn(B, H, W, Ci, generator=g).cuda()
    _, bits = ops.bn_apply(y, torch.ones(Ci).cuda(), torch.zeros(Ci).cuda(), relu=True, sign_bits=True)
    pair = (ds, ops.pack_weights(wsc.cuda(), transpose=True))
    wpt = ops.pack_weights(w.cuda(), transpose=True)
    dx, sums = ops.conv2d_dgrad(dy, wpt, Ci, H, W, 3, 3, 2, 1, 1, pair_1x1=pair, bn_bwd=(u, mean, invstd, None, None, bits))
    plain = ops.conv2d_dgrad(dy, wpt, Ci, H, W, 3, 3, 2, 1, 1, pair_1x1=pair)
    assert torch.equal(dx, plain)
    if Ci > 32:
        assert sums is None
        return
    rec, parts = sums
    r = rec[:parts * Ci * 2].view(parts, Ci, 2).double().sum(0).cpu()
    dz = (dx.double() * (y > 0)).cpu()
    uhat = ((u.double() - mean.double()) * invstd.double()).cpu()
    close(r[:, 0], dz.sum((0, 1, 2)), 2e-5)
    want2 = (dz * uhat).sum((0, 1, 2))
    assert (r[:, 1] - want2).abs().max().item() <= 2e-5 * (dz * uhat).abs().sum((0, 1, 2)).max().item()
    # the records feed bn_backward(partials=): same result as its own pass over dx and u
    gam = (torch.rand(Ci, generator=g) + 0.5).cuda()
    a = ops.bn_backward(dx, None, u, gam, mean, invstd, True, mask_bits=bits, partials=sums)
    b = ops.bn_backward(dx, None, u, gam, mean, invstd, True, mask_bits=bits)
    for t1, t2 in zip(a, b):
        close(t1, t2.cpu(), 2e-5)


def test_wgrad_batched_tile_launches(ops):
    """dam_wgrad_queue_set_batching (ABI 14): deferred weight gradients of ONE geometry that take the tile kernel -- the three
    256-channel convolutions of the 33 x 5 stage, models/model_resnet.py:71 -- are recorded and run as ONE launch at the flush
    (blockIdx.z = job), with the fused input affine of the conv2 form on some jobs; a job of another geometry in between
    launches what is pending; five equal jobs split 4 + 1.  Every gradient against torch's float64 weight gradient."""
    import ctypes
    from deep_audio_mixer_amd import _lib
    assert ops.WGRAD_BATCH
    g = torch.Generator().manual_seed(77)
    dev = torch.device('cuda', torch.cuda.current_device())

    def job(B, Ci, Co, H, W, affine):
        x = torch.randn(B, Ci, H, W, generator=g)
        dy = torch.randn(B, Co, H, W, generator=g)
        sc = sh = None
        xin = x
        if affine:
            sc, sh = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g)
            xin = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        want = torch.nn.grad.conv2d_weight(xin.double(), (Co, Ci, 3, 3), dy.double(), 1, 1)
        out = torch.full((Co, Ci, 3, 3), float('nan'), device=dev)
        return dict(x=nhwc(x).cuda(), dy=nhwc(dy).cuda(), sc=None if sc is None else sc.cuda(), sh=None if sh is None else sh.cuda(),
                    want=want, out=out, co=Co)
    # geometry A x 3 (one with the affine -> it may not share a launch with the plain ones), geometry B, geometry A x 5
    jobs = [job(8, 256, 256, 33, 5, False), job(8, 256, 256, 33, 5, False), job(8, 256, 256, 33, 5, True),
            job(2, 128, 128, 12, 5, False)] + [job(8, 256, 256, 33, 5, False) for _ in range(5)]
    q = ops._wgrad_queue(dev)
    L = _lib.lib()
    for j in jobs:
        ops.conv2d_wgrad(j['x'], j['dy'], j['co'], 3, 3, 1, 1, 1, in_scale=j['sc'], in_shift=j['sh'], relu_in=j['sc'] is not None,
                         out=j['out'], defer=True)
    assert L.dam_wgrad_queue_pending(ctypes.addressof(q[0])) > 0
    torch.cuda.synchronize()
    assert all(bool(torch.isnan(j['out']).all()) for j in jobs)          # nothing has been reduced yet
    ops.wgrad_flush(dev)
    assert L.dam_wgrad_queue_pending(ctypes.addressof(q[0])) == 0 and q[2] == []
    for j in jobs:
        close(j['out'], j['want'], 5e-5)
    # switching the mode needs an empty queue
    ops.conv2d_wgrad(jobs[0]['x'], jobs[0]['dy'], 256, 3, 3, 1, 1, 1, out=jobs[0]['out'], defer=True)
    assert L.dam_wgrad_queue_set_batching(ctypes.addressof(q[0]), 0) == -1
    ops.wgrad_flush(dev)
    assert L.dam_wgrad_queue_set_batching(ctypes.addressof(q[0]), 0) == 0
    jobs[1]['out'].fill_(float('nan'))
    ops.conv2d_wgrad(jobs[1]['x'], jobs[1]['dy'], 256, 3, 3, 1, 1, 1, out=jobs[1]['out'], defer=True)
    ops.wgrad_flush(dev)
    close(jobs[1]['out'], jobs[1]['want'], 5e-5)
    assert L.dam_wgrad_queue_set_batching(ctypes.addressof(q[0]), 1) == 0
