import sys, time, torch
sys.path.insert(0, '/root/repo')
import deep_audio_mixer_amd
from deep_audio_mixer_amd.models.model_scalar_2s import MixingModelScalar2s
from deep_audio_mixer_amd.models.model_scalar_1s import MixingModelScalar1s
from deep_audio_mixer_amd.optim import Adam
dev = torch.device('cuda', 0)
for name, cls, shape in (('scalar_2s', MixingModelScalar2s, (1025, 130)), ('scalar_1s', MixingModelScalar1s, (1025, 87))):
    try:
        m = cls(n_stems=4, input_shape=shape).to(dev)
    except TypeError:
        m = cls().to(dev)
    opt = Adam(m.parameters(), lr=1e-4, weight_decay=1e-5)
    x = torch.randn(8, 4, *shape, device=dev); gt = torch.randn(8, *shape, device=dev)
    def step():
        opt.zero_grad()
        loss = m.forward_mse(x, gt)[0]
        loss.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print(name, shape, '%.2f ms per eager step (batch 8)' % ((time.perf_counter() - t0) * 100))
