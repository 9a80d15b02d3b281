import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_audio_mixer_amd
from deep_audio_mixer_amd import features
rng = np.random.default_rng(6)
n = 3 * 16000 + 17
for n_fft, hop in ((2048, 1024), (1024, 256)):
    for ch in (2, 1):
        for nn in (n, n + 1):
            s16 = rng.integers(-20000, 20000, (3, nn, ch), dtype=np.int16)
            f = s16.astype(np.float32) / np.float32(32768.0)
            a = features.stft_logmag(torch.from_numpy(s16).cuda(), n_fft, hop)
            b = features.stft_logmag(torch.from_numpy(f).cuda(), n_fft, hop)
            d = (a - b).abs()
            bad = (d > 0).nonzero()
            print(n_fft, hop, 'ch', ch, 'n', nn, 'max diff', d.max().item(), 'n diff', int((d > 0).sum()), 'of', d.numel(),
                  'first bad', bad[:3].tolist(), 'frames with diff', sorted(set(bad[:, 2].tolist()))[:10] if len(bad) else [])
