"""GPU: the three models (all-HIP forward and backward through the C ABI) against
  (a) the golden vectors produced by the reference itself in float64 (tests/golden/models.*), and
  (b) the CPU oracle for the non-reference shapes of BASELINE.json's configs (S = 2 / 8 stems, 3 s clips).
north_star tolerance: gains within 1e-4 relative (fp32); gradients are checked per parameter tensor."""
import json
import os

import numpy as np
import pytest
import torch

from _inputs import model_input
from oracle import models_ref

pytestmark = pytest.mark.gpu
GAIN_RTOL = 1e-4


@pytest.fixture(scope='module')
def dam(dam_lib):
    import deep_audio_mixer_amd.models.model_resnet as mr
    import deep_audio_mixer_amd.models.model_scalar_1s as m1
    import deep_audio_mixer_amd.models.model_scalar_2s as m2
    return {'resnet18': (mr.ResNet18, models_ref.RefResNet18),
            'scalar1s': (m1.MixingModelScalar1s, models_ref.RefMixingModelScalar1s),
            'scalar2s': (m2.MixingModelScalar2s, models_ref.RefMixingModelScalar2s)}


def ref_named_grads(model):
    """Gradients of the product model keyed by the REFERENCE parameter names."""
    out = {}
    for n, p in model.named_parameters():
        if not n.startswith('_heads.'):
            out[n] = p.grad
    h = model._heads
    for i in range(h.n_stems):
        out['conv_head%d.weight' % (i + 1)] = h.conv_w.grad[i]
        out['conv_head%d.bias' % (i + 1)] = h.conv_b.grad[i:i + 1]
        out['fc_head%d.weight' % (i + 1)] = h.fc_w.grad[i]
        out['fc_head%d.bias' % (i + 1)] = h.fc_b.grad[i:i + 1]
    return out


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return np.abs(got - want).max() / (np.abs(want).max() + 1e-30)


def no_dropout(m):
    for mod in m.modules():
        if hasattr(mod, 'dropout_p'):
            mod.dropout_p = -1          # oracle blocks
        elif getattr(mod, 'dropout', None) is not None and not isinstance(mod, torch.nn.Dropout):
            mod.dropout = None          # product blocks
    return m


@pytest.mark.parametrize('name', ['resnet18', 'scalar1s', 'scalar2s'])
def test_against_reference_golden(dam, name, golden_dir):
    data = np.load(os.path.join(golden_dir, 'models.npz'))
    meta = json.load(open(os.path.join(golden_dir, 'models.json')))
    ctor, ref_ctor = dam[name]
    x, gt = model_input(*meta[name]['shape'], seed=meta[name]['seed'])
    ref = models_ref.closed_form_fill(ref_ctor())
    model = no_dropout(ctor())
    model.load_state_dict(ref.state_dict())            # reference-keyed checkpoint loads
    model = model.cuda().train()
    xc, gtc = torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda()
    masked, gains = model(xc)
    assert masked.shape == gtc.shape and len(gains) == 4 and gains[0].shape == (x.shape[0], 1)
    loss = torch.nn.functional.mse_loss(masked, gtc)
    loss.backward()
    key = name + '_f64_train'
    g = torch.cat(gains, 1).detach().cpu().numpy()
    assert rel_err(g, data[key + '_gains']) <= GAIN_RTOL
    assert rel_err(masked.detach().cpu().numpy()[:, ::41, ::7], data[key + '_masked_sample']) <= GAIN_RTOL
    assert abs(loss.item() - data[key + '_loss']) <= 2e-4 * abs(data[key + '_loss'])
    grads = ref_named_grads(model)
    names = meta[name + '_param_names']
    norms = np.array([grads[n].double().norm().item() for n in names])
    want = data[key + '_gradnorm']
    # gradient tolerances are calibrated on the reference itself: its own float32 and float64 runs differ by
    # 4e-3 in per-tensor gradient norms and 1.7e-2 (of the tensor's max) in sampled entries (ReLU decisions flip);
    # conv biases in front of a training-mode BatchNorm have an exactly-zero true gradient, hence the absolute term
    bad = [(n, a, b) for n, a, b in zip(names, norms, want) if abs(a - b) > 1e-2 * b + 1e-4 * want.max()]
    assert not bad, bad[:5]
    samples = np.array([[grads[n].flatten()[(j * grads[n].numel()) // 5].item() for j in range(5)] for n in names])
    scale = np.abs(data[key + '_gradsample']).max(axis=1, keepdims=True) + 1e-4 * want.max()
    assert np.max(np.abs(samples - data[key + '_gradsample']) / scale) <= 4e-2
    sd = model.state_dict()
    bn = np.concatenate([sd[k].double().cpu().numpy().ravel() for k in meta[name + '_bn_names']])
    np.testing.assert_allclose(bn, data[key + '_bn_running'], rtol=2e-4, atol=1e-5)
    # eval mode uses the running statistics
    model2 = no_dropout(ctor())
    model2.load_state_dict(ref.state_dict())
    model2 = model2.cuda().eval()
    with torch.no_grad():
        _, gains = model2(xc)
    assert rel_err(torch.cat(gains, 1).cpu().numpy(), data[name + '_f64_eval_gains']) <= GAIN_RTOL
    # the state_dict written back is reference-shaped
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == meta[name]['state_dict']


@pytest.mark.parametrize('name,n_stems,shape', [('resnet18', 8, (1025, 130)), ('scalar2s', 4, (1025, 130)),
                                                 ('scalar1s', 2, (1025, 63))])
def test_baseline_configs_against_oracle(dam, name, n_stems, shape):
    """BASELINE.json configs C1-C3 (stem counts / clip lengths the reference cannot express)."""
    ctor, ref_ctor = dam[name]
    torch.manual_seed(3)
    ref = no_dropout(ref_ctor(n_stems=n_stems, input_shape=shape)).double().train()
    model = no_dropout(ctor(n_stems=n_stems, input_shape=shape))
    model.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    model = model.cuda().train()
    x, gt = model_input(2, n_stems, shape[0], shape[1], seed=21)
    torch.set_num_threads(16)
    masked_r, gains_r = ref(torch.from_numpy(x).double())
    loss_r = torch.nn.functional.mse_loss(masked_r, torch.from_numpy(gt).double())
    loss_r.backward()
    loss, masked, gains = model.forward_mse(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda())
    loss.backward()
    assert rel_err(torch.cat(gains, 1).detach().cpu().numpy(), torch.cat(gains_r, 1).detach().numpy()) <= GAIN_RTOL
    assert abs(loss.item() - loss_r.item()) <= 2e-4 * loss_r.item()
    assert rel_err(masked.cpu().numpy(), masked_r.detach().numpy()) <= GAIN_RTOL
    grads = ref_named_grads(model)
    gmax = max(p.grad.norm().item() for p in ref.parameters())
    for n, p in ref.named_parameters():
        a, b = grads[n].double().cpu().flatten(), p.grad.flatten()
        assert (a - b).norm().item() <= 2e-2 * b.norm().item() + 1e-5 * gmax, n


@pytest.mark.parametrize('name,shape', [('resnet18', (2, 4, 257, 64)), ('scalar1s', (2, 4, 257, 87)), ('scalar2s', (2, 4, 257, 93))])
def test_gradients_tight_when_no_relu_flips(dam, name, shape):
    """Composition-level gradient check at float32 tightness.  Measured (tools/grad_accuracy_probe.py): without a ReLU
    decision flipping, the HIP path's activation gradients agree with the float64 oracle to ~1.5e-6 through all twelve
    blocks -- and so does the CPU float32 oracle; ONE flipped decision (|pre-activation| below the ~1e-6 forward rounding
    error: a handful of the ~10 M activations of a full-size clip) moves every gradient upstream of it by 2-5e-3, in the
    CPU float32 run exactly as in the HIP run.  That is why the checks against the float64 goldens sit at 1e-2.
    Here the flips are taken out of the comparison instead: moderate shapes (few activations near zero), four seeds, and
    for every parameter tensor the BEST agreement over the seeds -- a tensor only needs the layers behind it to be
    flip-free in one of the runs -- must reach float32 level: 3x the CPU float32 oracle's own best distance to the
    float64 truth, floor 2e-5 of the tensor norm.  A wrong tap, halo, stride or mask is off by O(1) in every run."""
    ctor, ref_ctor = dam[name]
    torch.set_num_threads(16)
    s, hw = shape[1], shape[2:]
    ref32 = no_dropout(models_ref.closed_form_fill(ref_ctor(n_stems=s, input_shape=hw))).train()
    ref64 = no_dropout(models_ref.closed_form_fill(ref_ctor(n_stems=s, input_shape=hw))).double().train()
    model = no_dropout(ctor(n_stems=s, input_shape=hw))
    model.load_state_dict(ref32.state_dict())
    model = model.cuda().train()
    state = {k: v.clone() for k, v in ref32.state_dict().items()}
    names = [n for n, _ in ref64.named_parameters()]
    best_hip, best_cpu = {n: np.inf for n in names}, {n: np.inf for n in names}
    for seed in (31, 32, 33, 34):
        x, gt = model_input(*shape, seed=seed)
        for m in (ref32, ref64, model):
            m.load_state_dict(state)              # BatchNorm running statistics back to the start; same parameters
            m.zero_grad()
        for ref, dt in ((ref32, torch.float32), (ref64, torch.float64)):
            masked_r, _ = ref(torch.from_numpy(x).to(dt))
            torch.nn.functional.mse_loss(masked_r, torch.from_numpy(gt).to(dt)).backward()
        model.forward_mse(torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda())[0].backward()
        grads = ref_named_grads(model)
        p32 = dict(ref32.named_parameters())
        gmax = max(p.grad.norm().item() for p in ref64.parameters())
        for n, p in ref64.named_parameters():
            truth = p.grad.flatten()
            scale = truth.norm().item() + 1e-5 * gmax   # conv biases in front of a training-mode BatchNorm: true gradient 0
            e_hip = (grads[n].detach().double().cpu().flatten() - truth).norm().item() / scale
            e_cpu = (p32[n].grad.double().flatten() - truth).norm().item() / scale
            assert e_hip <= 1e-1, (n, seed, e_hip)    # with a flip or two in the run: sanity bound only
            best_hip[n], best_cpu[n] = min(best_hip[n], e_hip), min(best_cpu[n], e_cpu)
    rows = sorted(((best_hip[n] / max(best_cpu[n], 2e-5 / 3), n, best_hip[n], best_cpu[n]) for n in names), reverse=True)
    print('%s: worst (ratio, tensor, best hip err, best cpu-f32 err): %s'
          % (name, [(round(r, 2), n, '%.1e' % a, '%.1e' % b) for r, n, a, b in rows[:4]]))
    print('%s: median best hip err %.1e, cpu-f32 %.1e' % (name, np.median([r[2] for r in rows]), np.median([r[3] for r in rows])))
    bad = [(n, a, b) for r, n, a, b in rows if a > max(3 * b, 2e-5)]
    assert not bad, bad[:5]


def test_forward_mse_equals_unfused(dam):
    ctor, _ = dam['resnet18']
    torch.manual_seed(0)
    m = ctor(n_stems=4, input_shape=(257, 64)).cuda().train()
    x, gt = model_input(2, 4, 257, 64, seed=2)
    xc, gtc = torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    masked, _ = m(xc)
    torch.nn.functional.mse_loss(masked, gtc).backward()
    g1 = [p.grad.clone() for p in m.parameters()]
    m.load_state_dict(state)
    m.zero_grad()
    loss, masked2, _ = m.forward_mse(xc, gtc)
    loss.backward()
    assert torch.equal(masked, masked2)
    for a, p in zip(g1, m.parameters()):
        assert (a - p.grad).norm() <= 1e-5 * a.norm() + 1e-12


def test_wrong_inputs_fail_loudly(dam):
    ctor, _ = dam['resnet18']
    m = ctor().cuda()
    with pytest.raises(RuntimeError, match='expected scalar type Float'):     # reference behaviour, SURVEY F4
        m(torch.zeros(1, 4, 1025, 216, dtype=torch.float64, device='cuda'))
    with pytest.raises(RuntimeError, match='GPU only'):
        m(torch.zeros(1, 4, 1025, 216))
    with pytest.raises(ValueError, match='flattened_dim'):                     # wrong clip length for this head width
        m(torch.zeros(1, 4, 1025, 130, device='cuda'))


def test_dropout_kernel_statistics(dam_lib):
    """ConvBlock2d dropout (training mode only): keep-rate, 1/(1-p) scaling, fresh mask per call, backward uses the
    forward's mask, eval mode is the identity."""
    from deep_audio_mixer_amd.layers import DropoutFn
    x = torch.ones(64, 1024, device='cuda', requires_grad=True)
    y1 = DropoutFn.apply(x, 0.3)
    y2 = DropoutFn.apply(x, 0.3)
    keep = (y1 > 0).float().mean().item()
    assert abs(keep - 0.7) < 0.01
    assert torch.allclose(y1[y1 > 0], torch.full_like(y1[y1 > 0], 1 / 0.7))
    assert not torch.equal(y1 > 0, y2 > 0)
    y1.sum().backward()
    assert torch.equal(x.grad > 0, y1 > 0) and torch.allclose(x.grad[x.grad > 0], torch.full_like(x.grad[x.grad > 0], 1 / 0.7))
    from deep_audio_mixer_amd.models.model_scalar_1s import MixingModelScalar1s
    torch.manual_seed(0)
    m = MixingModelScalar1s(n_stems=2, input_shape=(129, 80)).cuda()
    xin = torch.randn(2, 2, 129, 80, device='cuda')
    m.train()
    a, b = m(xin)[0], m(xin)[0]
    assert not torch.equal(a, b)                       # dropout active while training
    m.eval()
    a, b = m(xin)[0], m(xin)[0]
    assert torch.equal(a, b)


@pytest.mark.parametrize('name,shape', [('resnet18', (2, 4, 257, 64)), ('scalar1s', (2, 4, 257, 87))])
def test_folded_inference_path(dam, name, shape):
    """eval + no_grad takes the folded form (BatchNorm scale in the weights, shift + shortcut + ReLU in the convolution
    epilogue); with autograd enabled the same eval forward runs the BatchNorm kernels.  Both must agree, and the folded
    images must follow an in-place parameter update."""
    ctor, _ = dam[name]
    torch.manual_seed(5)
    model = no_dropout(ctor(n_stems=shape[1], input_shape=shape[2:])).cuda()
    x, gt = model_input(*shape, seed=9)
    xc = torch.from_numpy(x).cuda()
    model.train()
    for _ in range(2):                                   # non-trivial running statistics
        model(xc)
    model.eval()

    def both():
        with torch.no_grad():
            folded = torch.cat(model(xc)[1], 1)
        plain = torch.cat(model(xc)[1], 1).detach()      # autograd on: the unfolded eval path
        return folded, plain
    folded, plain = both()
    assert rel_err(folded.cpu().numpy(), plain.cpu().numpy()) <= 2e-5
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.01)
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.add_(0.01)
    folded2, plain2 = both()
    assert rel_err(folded2.cpu().numpy(), plain2.cpu().numpy()) <= 2e-5
    assert rel_err(folded2.cpu().numpy(), folded.cpu().numpy()) > 1e-4      # the update was seen
    # updates made by this library's own kernels (fused Adam over the flat buffer, running statistics from the statistics
    # finalize) do not move torch's version counters: the folded images must follow them too
    from deep_audio_mixer_amd.optim import Adam
    opt = Adam(model.parameters(), lr=1e-2)
    model.train()
    for _ in range(2):
        opt.zero_grad()
        loss = model.forward_mse(xc, torch.from_numpy(gt).cuda())[0]
        loss.backward()
        opt.step()
    model.eval()
    folded3, plain3 = both()
    assert rel_err(folded3.cpu().numpy(), plain3.cpu().numpy()) <= 2e-5
    assert rel_err(folded3.cpu().numpy(), folded2.cpu().numpy()) > 1e-4
