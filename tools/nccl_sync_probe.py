#!/usr/bin/env python3
"""tools/nccl_sync_probe.py: what does the stream synchronisation of torch.distributed's RCCL collectives cost the captured C3 step?
ONE rank (world_size 1 on the one GPU of the box: the all-reduce moves nothing, ProcessGroupNCCL's event traffic is the same as with 8
ranks): between graph replays, an asynchronous all_reduce of the gradient bucket (RCCL's stream waits for an event of the training
stream, the training stream later waits for RCCL's) against a synchronous one (launched on the training stream itself since torch 2.8)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
import deep_audio_mixer_amd  # noqa: E402,F401
from deep_audio_mixer_amd.engine import TrainStep  # noqa: E402
from deep_audio_mixer_amd.optim import Adam  # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', str(bench.free_port()))
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
cfg = bench.CONFIGS['C3']
S, B, hop = cfg['n_stems'], cfg['batch'], cfg['hop']
n = cfg['sr'] * cfg['seconds']
model = bench.build_model(cfg, dev)
opt = Adam(model.parameters(), weight_decay=1e-5)
step = TrainStep(model, opt, S, n, bench.CHANNELS, B, bench.N_FFT, hop)
clips = bench.synth_clips(2 * B, S, n, dev, 7)
step.load_clips(clips[:B])
step.capture(warmup=2)
step.bind_rotation([clips[:B], clips[B:]])
g = opt.flat_grad
small, big = g[:g.numel() // 8], g[g.numel() // 8:]


def region(kind, k_steps=40):
    def one():
        step()
        if kind == 'async x2':
            w1 = dist.all_reduce(big, async_op=True)
            w0 = dist.all_reduce(small, async_op=True)
            w1.wait()
            w0.wait()
        elif kind == 'async x1':
            dist.all_reduce(g, async_op=True).wait()
        elif kind == 'sync x2':
            dist.all_reduce(big)
            dist.all_reduce(small)
        elif kind == 'sync x1':
            dist.all_reduce(g)
    for _ in range(4):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k_steps):
        one()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k_steps


for rep in range(2):
    for kind in ('plain', 'async x1', 'async x2', 'sync x1', 'sync x2'):
        print('%-10s %.4f ms per step' % (kind, region(kind)), flush=True)
dist.destroy_process_group()
