import sys, torch
sys.path.insert(0, '/root/repo')
import deep_audio_mixer_amd
from deep_audio_mixer_amd import ops
dev = torch.device('cuda', 0)
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = 0
for ci, co, h, w in ((16, 32, 1025, 130), (32, 64, 513, 65), (64, 96, 257, 33), (96, 128, 129, 17), (128, 256, 65, 9)):
    x = torch.randn((8, h, w, ci), device=dev)
    wt = torch.randn((co, ci, 1, 1), device=dev) * 0.1
    wp, wpt = ops.pack_weights(wt), ops.pack_weights(wt, transpose=True)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    dy = torch.randn((8, ho, wo, co), device=dev)
    dx = torch.zeros((8, h, w, ci), device=dev)
    a = t(lambda: ops.conv2d_fwd(x, wp, co, 1, 1, 2, 0, 1))
    b = t(lambda: ops.conv2d_dgrad(dy, wpt, ci, h, w, 1, 1, 2, 0, 1, accumulate_into=dx))
    mb = (8 * ho * wo * (ci + co) * 4) / 1e6
    print('1x1 s2 %3d->%3d @%dx%d: fwd %.1f us, dgrad(accumulate) %.1f us  (min traffic %.0f MB)' % (ci, co, h, w, a, b, mb))
    tot += a + b
print('total %.0f us' % tot)
