#!/usr/bin/env python3
"""Diagnostic: BASELINE config C5 -- full-song inference, 8 stems, 3 minutes @ 44.1 kHz stereo, 3 s chunks, ResNet18 (eval)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_audio_mixer_amd
from deep_audio_mixer_amd import inference_utils
from deep_audio_mixer_amd.models.model_resnet import ResNet18
from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
dev = torch.device('cuda', 0)
sr, secs, stems = 44100, 180, ['s%d' % i for i in range(8)]
rng = np.random.default_rng(0)
tracks = {k: (0.1 * rng.standard_normal((2, sr * secs))).astype(np.float32) for k in stems}
ds = MultitrackAudioDataset.from_arrays({'song': {k: tracks[k].T for k in stems + ['mix']} if False else {**{k: tracks[k].T for k in stems}, 'mix': tracks['s0'].T}},
                                        tracklist=stems + ['mix'], chunk_length=3, sr=sr)
torch.manual_seed(0)
model = ResNet18(n_stems=8, input_shape=(1025, 130)).to(dev).eval()
for _ in range(2):
    out = inference_utils.mix_song_to_master(ds, model, tracks, chunk_length=3, sr=sr)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = inference_utils.mix_song_to_master(ds, model, tracks, chunk_length=3, sr=sr)
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
pcm = torch.stack([torch.from_numpy(tracks[k]) for k in stems]).to(dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
g = inference_utils.predict_chunk_gains(model, pcm, 8, secs // 3, 3 * sr)
torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
print('3-minute 8-stem song: %.1f ms end to end (host arrays in, master out), %.1f ms for front-end + 59 chunk forwards on the device' % (t_all * 1e3, t_dev * 1e3))
print('= %.0f stem-spectrogram-frames/s (inference, device part)' % (59 * 8 * 130 / t_dev))
