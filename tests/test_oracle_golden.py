"""CPU: the oracle restatement against the golden vectors produced by the reference itself."""
import json
import os

import numpy as np
import pytest
import torch

from _inputs import make_audio, model_input
from oracle import features_ref, inference_ref, models_ref


@pytest.fixture(scope='module')
def feat(golden_dir):
    return (np.load(os.path.join(golden_dir, 'features.npz')),
            json.load(open(os.path.join(golden_dir, 'features.json'))))


def test_features_match_reference(feat):
    data, meta = feat
    assert meta['tracklist'] == features_ref.TRACKLIST
    for c in meta['cases']:
        a = make_audio(c['kind'], c['n'], c['seed'])
        dt = np.float32 if c['dtype'] == 'f32' else np.float64
        n_fft = c.get('n_fft', 2048)
        f = features_ref.compute_features(a.astype(dt), n_fft, c['hop'], dt)
        assert list(f.shape) == c['shape'] == [n_fft // 2 + 1, 1 + c['n'] // c['hop']]
        tol = 2e-3 if dt is np.float32 else 1e-7
        want = data[c['key'] + '_sample']
        got = f[::37, ::5]
        if c['kind'] == 'silence':
            assert np.all(f == -100.0) and np.all(want == -100.0)
            continue
        if dt is np.float64:
            np.testing.assert_allclose(got, want, rtol=0, atol=tol)
            np.testing.assert_allclose(f.max(0), data[c['key'] + '_colmax'], rtol=0, atol=tol)
            if c['key'] + '_full' in data:
                np.testing.assert_allclose(f, data[c['key'] + '_full'], rtol=0, atol=tol)
        else:   # the reference's f32 path rounds differently inside the FFT: compare relative to frame peak
            lin_g, lin_w = 10 ** (got / 20.0), 10 ** (want / 20.0)
            assert np.max(np.abs(lin_g - lin_w)) <= 3e-6 * lin_w.max()


def test_stereo_to_mono(feat):
    st = np.random.default_rng(7).standard_normal((4000, 2))
    np.testing.assert_array_equal(features_ref.stereo_to_mono(st), feat[0]['mono'])


def test_inference_pieces(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'inference.json')))
    for c in g['interpolate_mask']:
        np.testing.assert_array_equal(inference_ref.interpolate_mask(c['mask'], c['n']), np.array(c['out']))
    np.testing.assert_allclose([inference_ref.scalar_db_to_amplitude(v) for v in g['db']], g['db_to_amplitude'],
                               rtol=1e-15)
    assert inference_ref.savgol_window(60) == 15 and inference_ref.savgol_window(64) == 17


MODELS = {'resnet18': models_ref.RefResNet18, 'scalar1s': models_ref.RefMixingModelScalar1s,
          'scalar2s': models_ref.RefMixingModelScalar2s}


@pytest.mark.parametrize('name', ['resnet18', 'scalar1s', 'scalar2s'])
def test_models_match_reference(name, golden_dir):
    data = np.load(os.path.join(golden_dir, 'models.npz'))
    meta = json.load(open(os.path.join(golden_dir, 'models.json')))
    m = MODELS[name]()
    sd = m.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == meta[name]['state_dict']
    assert sum(p.numel() for p in m.parameters()) == meta[name]['n_params']
    assert [n for n, _ in m.named_parameters()] == meta[name + '_param_names']
    x, gt = model_input(*meta[name]['shape'], seed=meta[name]['seed'])
    torch.set_num_threads(8)
    # float64, train mode: forward, loss, gradients, BN running statistics
    m = models_ref.closed_form_fill(MODELS[name]()).double()
    for mod in m.modules():
        if hasattr(mod, 'dropout_p'):
            mod.dropout_p = -1
    m.train()
    masked, gains = m(torch.from_numpy(x).double())
    loss = torch.nn.functional.mse_loss(masked, torch.from_numpy(gt).double())
    loss.backward()
    key = name + '_f64_train'
    np.testing.assert_allclose(torch.cat(gains, 1).detach().numpy(), data[key + '_gains'], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(loss.item(), data[key + '_loss'], rtol=1e-10)
    np.testing.assert_allclose(masked.detach().numpy()[:, ::41, ::7], data[key + '_masked_sample'], rtol=1e-9, atol=1e-8)
    norms = np.array([p.grad.norm().item() for p in m.parameters()])
    np.testing.assert_allclose(norms, data[key + '_gradnorm'], rtol=1e-7, atol=1e-9)
    sd = m.state_dict()
    bn = np.concatenate([sd[k].numpy().ravel() for k in meta[name + '_bn_names']])
    np.testing.assert_allclose(bn, data[key + '_bn_running'], rtol=1e-10, atol=1e-12)
    # float32, eval mode
    m = models_ref.closed_form_fill(MODELS[name]()).eval()
    with torch.no_grad():
        _, gains = m(torch.from_numpy(x))
    np.testing.assert_allclose(torch.cat(gains, 1).numpy(), data[name + '_f32_eval_gains'], rtol=1e-4, atol=1e-4)
