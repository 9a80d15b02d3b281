import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import bench
from oracle import features_ref, models_ref
import deep_audio_mixer_amd
from deep_audio_mixer_amd.engine import TrainStep
from deep_audio_mixer_amd.optim import Adam
from deep_audio_mixer_amd import features, ops
cfg = bench.CONFIGS['C3']; S, B, hop = cfg['n_stems'], cfg['batch'], cfg['hop']; n = cfg['sr'] * cfg['seconds']
dev = torch.device('cuda', 0)
model = bench.build_model(cfg, dev)
opt = Adam(model.parameters(), weight_decay=1e-5)
clips = bench.synth_clips(3 * B, S, n, dev, 1234)
# step 0 eagerly (same arithmetic as the graph)
x0, gt0 = features.stft_logmag_clips(clips[:B], 2048, hop)
opt.zero_grad(); loss = model.forward_mse(x0, gt0)[0]; loss.backward(); opt.step()
print('step0 loss', loss.item())
state = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu().clone()) for k, v in model.state_dict().items()}
x1, gt1 = features.stft_logmag_clips(clips[B:2 * B], 2048, hop)
opt.zero_grad()
loss, masked, gains = model.forward_mse(x1, gt1)
loss.backward()
h = model._heads
hip = {'conv_w': h.conv_w.grad.double().cpu(), 'conv_b': h.conv_b.grad.double().cpu(), 'fc_w': h.fc_w.grad.double().cpu(), 'fc_b': h.fc_b.grad.double().cpu()}
torch.set_num_threads(16)
ref = models_ref.RefResNet18(n_stems=S, input_shape=(1025, 130)).double().train(); ref.load_state_dict(state)
host = clips.cpu().numpy()
for feats_from in ('numpy', 'hip'):
    if feats_from == 'numpy':
        items = [features_ref.clip_features(host[B + b], 2048, hop, np.float32) for b in range(B)]
        x = torch.from_numpy(np.stack([i[0] for i in items])).double(); gt = torch.from_numpy(np.stack([i[1] for i in items])).double()
    else:
        x, gt = x1.double().cpu(), gt1.double().cpu()
    ref.load_state_dict(state); ref.zero_grad()
    m_r, g_r = ref(x)
    l_r = torch.nn.functional.mse_loss(m_r, gt); l_r.backward()
    print('oracle features from', feats_from, ': loss hip %.6f oracle %.6f' % (loss.item(), l_r.item()))
    gr = torch.cat(g_r, 1).detach()
    print('   gains rel err', ((torch.cat(gains, 1).double().cpu() - gr).abs().max() / gr.abs().max()).item())
    for s in range(S):
        cw = getattr(ref, 'conv_head%d' % (s + 1)).weight.grad.flatten(); cb = getattr(ref, 'conv_head%d' % (s + 1)).bias.grad
        fw = getattr(ref, 'fc_head%d' % (s + 1)).weight.grad.flatten(); fb = getattr(ref, 'fc_head%d' % (s + 1)).bias.grad
        print('   head %d: conv_w %.2e conv_b %.2e (|ref| %.2e) fc_w %.2e fc_b %.2e' % (
            s + 1, ((hip['conv_w'][s] - cw).norm() / cw.norm()).item(), ((hip['conv_b'][s] - cb).abs() / cb.abs()).item(), cb.abs().item(),
            ((hip['fc_w'][s] - fw).norm() / fw.norm()).item(), ((hip['fc_b'][s] - fb).abs() / fb.abs()).item()))
# feature difference
items = [features_ref.clip_features(host[B + b], 2048, hop, np.float32) for b in range(B)]
xn = np.stack([i[0] for i in items])
d = np.abs(x1.cpu().numpy() - xn)
print('feature abs diff (dB): max %.3e mean %.3e; count > 1e-2: %d of %d' % (d.max(), d.mean(), int((d > 1e-2).sum()), d.size))
