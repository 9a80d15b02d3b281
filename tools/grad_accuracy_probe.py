#!/usr/bin/env python3
"""Diagnostic: where does the HIP ResNet18's float32 error (vs the float64 oracle) enter?  Per-block activation error in
the forward pass and per-block activation-gradient error in the backward pass, next to the CPU float32 oracle's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import deep_audio_mixer_amd
from deep_audio_mixer_amd.models.model_resnet import ResNet18
from deep_audio_mixer_amd.layers import ConvBnReluFn
from oracle import models_ref
from _inputs import model_input

shape = tuple(int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (2, 4, 1025, 216)
seed = int(os.environ.get('PROBE_SEED', 11))
x, gt = model_input(*shape, seed=seed)
torch.set_num_threads(16)
refs = {}
for name, dt in (('f32', torch.float32), ('f64', torch.float64)):
    ref = models_ref.closed_form_fill(models_ref.RefResNet18(n_stems=shape[1], input_shape=shape[2:])).to(dt).train()
    acts = {}
    def hook(key):
        def f(mod, inp, out):
            out.retain_grad(); acts[key] = out
        return f
    ref.bn1.register_forward_hook(hook('stem_bn'))
    for i in range(1, 7):
        for b in range(2):
            getattr(ref, 'layer%d' % i)[b].register_forward_hook(hook('l%d.%d' % (i, b)))
    masked, gains = ref(torch.from_numpy(x).to(dt))
    loss = torch.nn.functional.mse_loss(masked, torch.from_numpy(gt).to(dt))
    loss.backward()
    refs[name] = (ref, acts, loss.item(), torch.cat(gains, 1).detach())
ref32 = refs['f32'][0]
model = ResNet18(n_stems=shape[1], input_shape=shape[2:])
model.load_state_dict(ref32.state_dict())
model = model.cuda().train()
xc, gtc = torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda()
model._pack_weights()
hip = {}
out = ConvBnReluFn.apply(xc, model.conv1.weight, None, model.bn1.weight, model.bn1.bias, model._stem, model.bn1, True)
out.retain_grad(); hip['stem'] = out
for i in range(1, 7):
    for b in range(2):
        out = getattr(model, 'layer%d' % i)[b](out)
        out.retain_grad(); hip['l%d.%d' % (i, b)] = out
loss, masked, g = model._heads.forward_mse(out, xc, gtc)
loss.backward()
print('loss hip %.6f f32 %.6f f64 %.6f' % (loss.item(), refs['f32'][2], refs['f64'][2]))
g64 = refs['f64'][3]
print('gains rel err: hip %.2e  f32 %.2e' % ((g.detach().cpu().double() - g64).abs().max() / g64.abs().max(),
                                             (refs['f32'][3].double() - g64).abs().max() / g64.abs().max()))
def rel(a, b):
    return ((a - b).norm() / b.norm()).item()
print('%-8s %12s %12s %14s %14s' % ('block', 'act hip', 'act f32', 'dact hip', 'dact f32'))
for key in ['l%d.%d' % (i, b) for i in range(1, 7) for b in range(2)]:
    a64 = refs['f64'][1][key]
    a32 = refs['f32'][1][key]
    h = hip[key]
    hn = h.detach().permute(0, 3, 1, 2)[:, :a64.shape[1]].cpu().double()
    hg = h.grad.permute(0, 3, 1, 2)[:, :a64.shape[1]].cpu().double()
    print('%-8s %12.2e %12.2e %14.2e %14.2e' % (key, rel(hn, a64.detach()), rel(a32.detach().double(), a64.detach()),
                                                rel(hg, a64.grad), rel(a32.grad.double(), a64.grad)))
