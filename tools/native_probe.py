import sys, time, torch
sys.path.insert(0, '/root/repo')
import deep_audio_mixer_amd
from deep_audio_mixer_amd.models.model_resnet import ResNet18
from deep_audio_mixer_amd.optim import Adam
dev = torch.device('cuda', 0)
m = ResNet18().to(dev).train()          # reference-native: 4 stems, 1025 x 216
opt = Adam(m.parameters(), lr=1e-4, weight_decay=1e-5)
x = torch.randn(8, 4, 1025, 216, device=dev); gt = torch.randn(8, 1025, 216, device=dev)
def step():
    opt.zero_grad(); loss = m.forward_mse(x, gt)[0]; loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print('ResNet18 reference-native (4 x 1025 x 216, batch 8): %.2f ms per eager step' % ((time.perf_counter() - t0) * 100))
