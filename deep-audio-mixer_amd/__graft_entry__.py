"""Driver entry points: build() compiles every HIP source for gfx950 (works without a GPU) and imports the
package; smoke() runs one tiny training step of the hot path on cuda:0 and checks it against the CPU oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build() -> None:
    import deep_audio_mixer_amd  # noqa: F401
    from deep_audio_mixer_amd import build as _build, _lib
    path = _build.build_lib()
    lib = _lib.lib()
    assert lib.dam_arch() == b'gfx950', 'libdam_hip.so was not built for gfx950'
    # the oracle is pure numpy / PyTorch-CPU (nothing to compile); importing it checks it is intact
    from oracle import features_ref, inference_ref, models_ref  # noqa: F401
    import deep_audio_mixer_amd.models.model_resnet  # noqa: F401
    import deep_audio_mixer_amd.models.model_scalar_1s  # noqa: F401
    import deep_audio_mixer_amd.models.model_scalar_2s  # noqa: F401
    import deep_audio_mixer_amd.model_trainer  # noqa: F401
    print('built', path)


def smoke() -> None:
    """One small invocation of the whole hot path on cuda:0 (STFT front-end -> ResNet18 forward -> fused MSE ->
    backward -> Adam), checked against the oracle (numpy STFT + PyTorch-CPU model in float64)."""
    import numpy as np
    import torch
    build()
    from deep_audio_mixer_amd import features
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    from oracle import features_ref, models_ref
    assert torch.cuda.is_available(), 'smoke() needs cuda:0'
    dev = torch.device('cuda', 0)
    n_stems, n, hop = 2, 33 * 1024, 1024              # ~0.77 s clips -> 1025 x 34 features
    rng = np.random.default_rng(0)
    stems = (0.1 * rng.standard_normal((2, n_stems, n, 2))).astype(np.float32)
    mix = stems.sum(1)
    x = features.stft_logmag(torch.from_numpy(stems.reshape(-1, n, 2)).to(dev), hop=hop).view(2, n_stems, 1025, -1)
    gt = features.stft_logmag(torch.from_numpy(mix).to(dev), hop=hop)
    want = features_ref.compute_features(stems[1, 0].astype(np.float64).mean(1), 2048, hop)
    lin = np.abs(10 ** (x[1, 0].cpu().numpy() / 20.0) - 10 ** (want / 20.0)).max() / (10 ** (want / 20.0)).max()
    assert lin < 2e-6, 'front-end mismatch %g' % lin
    t = x.shape[-1]
    torch.manual_seed(0)
    ref = models_ref.RefResNet18(n_stems=n_stems, input_shape=(1025, t)).double().train()
    model = ResNet18(n_stems=n_stems, input_shape=(1025, t))
    model.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    model = model.to(dev).train()
    opt = Adam(model.parameters(), weight_decay=1e-5)
    loss, masked, gains = model.forward_mse(x, gt)
    loss.backward()
    opt.step()
    masked_r, gains_r = ref(x.double().cpu())
    loss_r = torch.nn.functional.mse_loss(masked_r, gt.double().cpu())
    g, gr = torch.cat(gains, 1).cpu().double(), torch.cat(gains_r, 1).detach()
    rel = ((g - gr).abs().max() / gr.abs().max()).item()
    assert rel < 1e-4, 'gain mismatch %g' % rel
    assert abs(loss.item() - loss_r.item()) < 2e-4 * abs(loss_r.item())
    print('smoke ok: gains rel err %.2e, loss %.4f (oracle %.4f)' % (rel, loss.item(), loss_r.item()))


if __name__ == '__main__':
    build()
    if len(sys.argv) > 1 and sys.argv[1] == 'smoke':
        smoke()
