"""GPU: BatchNorm forward statistics / apply and the three mask modes of the backward (saved output, recomputed from the fused
affine, none) through the C ABI against torch autograd in float64.  Tolerance: fp32 accumulation over up to 1e5 pixels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops(dam_lib):
    from deep_audio_mixer_amd import ops
    return ops


def close(got, want, tol):
    got, want = got.double().cpu(), want.double()
    scale = want.abs().max().item() + 1e-30
    assert (got - want).abs().max().item() <= tol * scale


@pytest.mark.parametrize('B,H,W,C', [(2, 37, 23, 16), (1, 129, 65, 32), (3, 9, 5, 256), (8, 128, 130, 16)])
def test_bn_relu_forward_backward_mask_modes(ops, B, H, W, C):
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, H, W, C, generator=g) * 3 + 5).requires_grad_(False)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    dy = torch.randn(B, H, W, C, generator=g)
    # reference: float64 autograd, training-mode statistics, relu(bn(x))
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mean = xr.mean(dim=(0, 1, 2)); var = xr.var(dim=(0, 1, 2), unbiased=False)
    yr = torch.relu((xr - mean) / torch.sqrt(var + 1e-5) * gr + br)
    yr.backward(dy.double())
    xd = x.cuda()
    rm, rv, nbt = torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.zeros((), dtype=torch.int64).cuda()
    sm, si, sc, sh = ops.bn_stats(xd, gamma.cuda(), beta.cuda(), rm, rv, nbt, 0.1, 1e-5)
    a = ops.bn_apply(xd, sc, sh, relu=True)
    close(a, yr.detach(), 2e-5)
    for mode in ('saved', 'affine'):
        if mode == 'saved':
            dx, dgm, dbt = ops.bn_backward(dy.cuda(), a, xd, gamma.cuda(), sm, si, True)
        else:
            dx, dgm, dbt = ops.bn_backward(dy.cuda(), None, xd, gamma.cuda(), sm, si, True, mask_affine=(sc, sh))
        close(dx, xr.grad, 1e-4); close(dgm, gr.grad, 1e-4); close(dbt, br.grad, 1e-4)
    # the two mask modes must agree bit for bit (same fma as the forward)
    d1 = ops.bn_backward(dy.cuda(), a, xd, gamma.cuda(), sm, si, True)
    d2 = ops.bn_backward(dy.cuda(), None, xd, gamma.cuda(), sm, si, True, mask_affine=(sc, sh))
    assert all(torch.equal(p, q) for p, q in zip(d1, d2))
    # no mask: plain bn
    xr2 = x.double().requires_grad_(True)
    y2 = (xr2 - xr2.mean(dim=(0, 1, 2))) / torch.sqrt(xr2.var(dim=(0, 1, 2), unbiased=False) + 1e-5) * gamma.double() + beta.double()
    y2.backward(dy.double())
    dx0, _, _ = ops.bn_backward(dy.cuda(), None, xd, gamma.cuda(), sm, si, True)
    close(dx0, xr2.grad, 1e-4)



def test_backward_pair_is_bitwise_two_single_calls(ops):
    """dam_bn_backward_pair_f32 (a block's bn2 + its shortcut BatchNorm: shared dy and mask) against two single calls."""
    g = torch.Generator().manual_seed(4)
    for shape in [(3, 37, 29, 32), (2, 9, 5, 256), (2, 65, 33, 64)]:
        C = shape[-1]
        dy, y = torch.randn(shape, generator=g).cuda(), torch.randn(shape, generator=g).cuda()
        xs = [torch.randn(shape, generator=g).cuda() for _ in range(2)]
        par = [(torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda(), torch.rand(C, generator=g).cuda() + 0.5)
               for _ in range(2)]
        for training in (True, False):
            single = [ops.bn_backward(dy, y, x, gm, mu, iv, training) for x, (gm, mu, iv) in zip(xs, par)]
            pair = ops.bn_backward_pair(dy, y, (xs[0], *par[0], None, None), (xs[1], *par[1], None, None), training)
            for s, p in zip(single, pair):
                for a, b in zip(s, p):
                    assert torch.equal(a, b)


def test_stats_pair_is_bitwise_two_single_calls(ops):
    g = torch.Generator().manual_seed(6)
    for shape in [(3, 37, 29, 32), (2, 9, 5, 256)]:
        C = shape[-1]
        xs = [torch.randn(shape, generator=g).cuda() * (i + 1) + i for i in range(2)]

        def params():
            return [(torch.rand(C, generator=torch.Generator().manual_seed(i)).cuda() + 0.5, torch.randn(C, generator=torch.Generator().manual_seed(10 + i)).cuda(),
                     torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.zeros((), dtype=torch.int64).cuda(), 0.1, 1e-5) for i in range(2)]
        pa, pb = params(), params()
        single = [ops.bn_stats(x, *p) for x, p in zip(xs, pa)]
        pair = ops.bn_stats_pair(xs[0], pb[0], xs[1], pb[1])
        for s, p in zip(single, pair):
            for a, b in zip(s, p):
                assert torch.equal(a, b)
        for p, q in zip(pa, pb):                 # running statistics and the batch counter moved the same way
            assert torch.equal(p[2], q[2]) and torch.equal(p[3], q[3]) and int(p[4]) == int(q[4]) == 1


def test_sign_bytes_equal_the_float_mask(ops):
    """bn_apply(sign_bits=True) + bn_backward(mask_bits=) / bn_backward_pair(mask_bits=): bitwise the y_mask forms."""
    g = torch.Generator().manual_seed(8)
    for shape in [(3, 37, 29, 32), (2, 9, 5, 256)]:
        C = shape[-1]
        x, res = torch.randn(shape, generator=g).cuda(), torch.randn(shape, generator=g).cuda()
        sc, sh = torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda()
        y = ops.bn_apply(x, sc, sh, relu=True, res=res)
        y2, bits = ops.bn_apply(x, sc, sh, relu=True, res=res, sign_bits=True)
        assert torch.equal(y, y2) and bits.shape == shape[:-1] + (C // 4,)
        want = (y.view(*shape[:-1], C // 4, 4) > 0).to(torch.uint8)
        want = want[..., 0] | (want[..., 1] << 1) | (want[..., 2] << 2) | (want[..., 3] << 3)
        assert torch.equal(bits, want)
        dy = torch.randn(shape, generator=g).cuda()
        xs = [torch.randn(shape, generator=g).cuda() for _ in range(2)]
        par = [(torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda(), torch.rand(C, generator=g).cuda() + 0.5)
               for _ in range(2)]
        a = ops.bn_backward(dy, y, xs[0], *par[0], True)
        b = ops.bn_backward(dy, None, xs[0], *par[0], True, mask_bits=bits)
        assert all(torch.equal(p, q) for p, q in zip(a, b))
        pa = ops.bn_backward_pair(dy, y, (xs[0], *par[0], None, None), (xs[1], *par[1], None, None), True)
        pb = ops.bn_backward_pair(dy, None, (xs[0], *par[0], None, None), (xs[1], *par[1], None, None), True, mask_bits=bits)
        for s, t in zip(pa, pb):
            assert all(torch.equal(p, q) for p, q in zip(s, t))


@pytest.mark.parametrize('B,H,W,C', [(8, 128, 130, 16), (2, 37, 23, 16), (1, 129, 65, 32), (2, 33, 17, 48), (3, 40, 33, 64),
                                     (8, 129, 17, 96), (8, 65, 9, 128), (8, 33, 5, 256), (1, 3, 3, 16)])
@pytest.mark.parametrize('form', ['plain', 'res', 'res_affine'])
def test_finalize_inside_the_apply_launch(ops, B, H, W, C, form):
    """dam_bn_stats_partial_f32 + dam_bn_finalize_apply_f32 (every workgroup merges the records of the channels it applies):
    outputs, the four per-channel results, the running statistics and the sign bytes against float64, and against the
    separate finalize + apply launches at float32 rounding."""
    g = torch.Generator().manual_seed(3 * C + H)
    x = torch.randn(B, H, W, C, generator=g) * 3 + 5 * torch.randn(C, generator=g)      # mean^2 comparable to / above var
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    res = torch.randn(B, H, W, C, generator=g) if form != 'plain' else None
    rs, rh = (torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)) if form == 'res_affine' else (None, None)
    xd = x.cuda()
    cu = lambda t: None if t is None else t.cuda()

    def bn_args():
        return (gamma.cuda(), beta.cuda(), torch.zeros(C).cuda() + 0.25, torch.ones(C).cuda() * 2, torch.zeros((), dtype=torch.int64).cuda(), 0.1, 1e-5)
    # float64 reference
    xr = x.double()
    mean, var = xr.mean(dim=(0, 1, 2)), xr.var(dim=(0, 1, 2), unbiased=False)
    yr = (xr - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
    if res is not None:
        yr = yr + (res.double() * rs.double() + rh.double() if rs is not None else res.double())
    yr = torch.relu(yr)
    n = B * H * W
    # fused
    a1 = bn_args()
    rec, parts = ops.bn_stats_partial(xd)
    assert parts >= 1
    (m, i, sc, sh), (y, bits) = ops.bn_finalize_apply(rec, parts, a1, xd, relu=True, res=cu(res), res_scale=cu(rs), res_shift=cu(rh),
                                                       sign_bits=True)
    close(y, yr, 2e-5)
    close(m, mean, 1e-6)
    close(i, 1.0 / torch.sqrt(var + 1e-5), 1e-5)
    close(a1[2], 0.9 * 0.25 + 0.1 * mean, 1e-6)
    close(a1[3], 0.9 * 2 + 0.1 * var * n / max(n - 1, 1), 1e-5)
    assert int(a1[4]) == 1
    want = (y.view(B, H, W, C // 4, 4) > 0).to(torch.uint8)
    assert torch.equal(bits, want[..., 0] | (want[..., 1] << 1) | (want[..., 2] << 2) | (want[..., 3] << 3))
    # separate launches: same numbers to float32 rounding
    a2 = bn_args()
    m2, i2, sc2, sh2 = ops.bn_stats(xd, *a2)
    y2 = ops.bn_apply(xd, sc2, sh2, relu=True, res=cu(res), res_scale=cu(rs), res_shift=cu(rh))
    close(m, m2.cpu(), 1e-6); close(i, i2.cpu(), 2e-6); close(sc, sc2.cpu(), 2e-6); close(sh, sh2.cpu(), 3e-6)
    close(y, y2.cpu(), 1e-5)
    close(a1[2], a2[2].cpu(), 1e-6); close(a1[3], a2[3].cpu(), 2e-6)


def test_fused_backward_equals_separate_launches(ops, monkeypatch):
    """dam_bn_backward_f32 / _pair_f32 merge their records inside the apply launch; DAM_BN_FUSED_FIN=0 (read once per process:
    checked through a child process) keeps the three-launch form.  Here: every mask mode on shapes that exercise one and many
    slices, ragged ranges and the record cap, against float64."""
    g = torch.Generator().manual_seed(12)
    for shape in [(8, 257, 33, 64), (8, 513, 65, 32), (4, 129, 17, 96), (8, 33, 5, 256), (1, 5, 3, 16)]:
        C = shape[-1]
        x, dy = torch.randn(shape, generator=g) * 2 + 1, torch.randn(shape, generator=g)
        gamma = torch.rand(C, generator=g) + 0.5
        xr = x.double().requires_grad_(True)
        gr = gamma.double().requires_grad_(True)
        br = torch.zeros(C, dtype=torch.float64, requires_grad=True)
        mean, var = xr.mean(dim=(0, 1, 2)), xr.var(dim=(0, 1, 2), unbiased=False)
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        yr = torch.relu((xr - mean) * invstd * gr + br)
        yr.backward(dy.double())
        sc = (gamma.double() * invstd.detach()).float().cuda()
        sh = (-mean.detach() * gamma.double() * invstd.detach()).float().cuda()
        y = ops.bn_apply(x.cuda(), sc, sh, relu=True)
        _, bits = ops.bn_apply(x.cuda(), sc, sh, relu=True, sign_bits=True)
        for kw in (dict(y_mask=y), dict(mask_affine=(sc, sh)), dict(mask_bits=bits)):
            y_mask = kw.pop('y_mask', None)
            dx, dgm, dbt = ops.bn_backward(dy.cuda(), y_mask, x.cuda(), gamma.cuda(), mean.detach().float().cuda(),
                                           invstd.detach().float().cuda(), True, **kw)
            close(dx, xr.grad, 1e-4); close(dgm, gr.grad, 1e-4); close(dbt, br.grad, 1e-4)
