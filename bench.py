#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric on MI355X.

  metric : stem-spectrogram-frames/sec (train), model_resnet, 8-stem 3 s @ 44.1 kHz (BASELINE config C3/C4)
  step   : STFT/log-mag front-end (9 tracks per clip) + ResNet18 forward + MSE + backward + gradient bucket
           (+ RCCL all-reduce when N > 1) + Adam(+L2), batch 8 clips per GPU, float32 end to end;
           PCM already resident in HBM when the timed region starts (synthetic clips)
  unit   : one 1025-bin STFT column of one input stem (8 stems x 130 frames = 1040 per clip)

  python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0, with the `roofline` object of the dominant kernel (measured live with HIP
events on the launch stream) and the `cpu_baseline` object (the CPU oracle timed on the host cores, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_STEMS, SR, SECONDS, CHANNELS = 8, 44100, 3, 2
N_SAMPLES = SR * SECONDS
HOP, N_FFT = 1024, 2048
BATCH = 8
PEAK_F32_MFMA = 157.3e12        # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM = 8.0e12               # MI355X_MICROARCH.md: HBM3E spec


def synth_clips(n_clips, device, seed):
    """stems 0.1*N(0,1), mix = sum_s linspace(0.5, 1.5, S)[s] * stem_s (stereo float32)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    stems = 0.1 * torch.randn((n_clips, N_STEMS, N_SAMPLES, CHANNELS), generator=g, device=device, dtype=torch.float32)
    w = torch.linspace(0.5, 1.5, N_STEMS, device=device).view(1, N_STEMS, 1, 1)
    return stems, (stems * w).sum(1)


def time_kernel(fn, iters=30, warm=5):
    """Average device time of one launch sequence, HIP events on the current (launch) stream."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


def roofline_probe(device):
    """Times the dominant kernels in isolation at the benchmark's shapes.

    Dominant kernel: the implicit-GEMM convolution (forward/dgrad; conv_strip_kernel for the thin full-resolution layers,
    conv_igemm_kernel otherwise).  Its heaviest instance is a ResNet layer1 convolution: 16 -> 16 channels, 3x3, on
    8 x 1025 x 130 pixels.  Algorithmic FLOPs per launch =
    2 * B*H*W * Cout * 9*Cin; the bound is the fp32 matrix pipe (157.3 TFLOP/s).
    """
    from deep_audio_mixer_amd import ops
    out = {}
    B, H, W = BATCH, N_FFT // 2 + 1, 1 + N_SAMPLES // HOP
    for name, cin, cout, h, w_ in (('layer1_conv3x3_16x16', 16, 16, H, W), ('layer3_conv3x3_64x64', 64, 64, 257, 33)):
        x = torch.randn((B, h, w_, cin), device=device)
        wt = torch.randn((cout, cin, 3, 3), device=device) * 0.05
        wp = ops.pack_weights(wt)
        dy = torch.randn((B, h, w_, cout), device=device)
        flops = 2.0 * B * h * w_ * cout * 9 * cin
        t_f = time_kernel(lambda: ops.conv2d_fwd(x, wp, cout, 3, 3, 1, 1, 1))
        t_w = time_kernel(lambda: ops.conv2d_wgrad(x, dy, cout, 3, 3, 1, 1, 1))
        out[name] = {'fwd_s': t_f, 'fwd_tflops': flops / t_f / 1e12, 'wgrad_s': t_w, 'wgrad_tflops': flops / t_w / 1e12,
                     'flops_per_launch': flops,
                     'alg_bytes_per_launch': 4.0 * B * h * w_ * (cin + cout)}
        if cin == 16:
            # the nine launches of this kernel in one training step, as the step issues them (stem + two BasicBlocks of
            # layer1): 3 forward with BatchNorm statistics, 2 forward with statistics and the fused input affine, 2 plain
            # data gradients, 2 data gradients with the residual / mask epilogue -- what a kernel trace of the step averages
            wpt = ops.pack_weights(wt, transpose=True)
            sc, sh = torch.rand(cin, device=device) + 0.5, torch.randn(cin, device=device)
            buf = ops.bn_partial_buffer(device, cout)
            msk = torch.randn((B, h, w_, cin), device=device)

            def step_mix():
                for _ in range(3):
                    ops.conv2d_fwd(x, wp, cout, 3, 3, 1, 1, 1, bn_partial=buf)
                for _ in range(2):
                    ops.conv2d_fwd(x, wp, cout, 3, 3, 1, 1, 1, bn_partial=buf, in_scale=sc, in_shift=sh, relu_in=True)
                for _ in range(2):
                    ops.conv2d_dgrad(dy, wpt, cin, h, w_, 3, 3, 1, 1, 1)
                for _ in range(2):
                    ops.conv2d_dgrad(dy, wpt, cin, h, w_, 3, 3, 1, 1, 1, res=x, res_mask=msk)
            t_mix = time_kernel(step_mix) / 9.0
            out[name]['step_mix_s'] = t_mix
            out[name]['step_mix_tflops'] = flops / t_mix / 1e12
    return out


def cpu_baseline(seconds_budget=25.0):
    """The CPU oracle (oracle/: numpy front-end + PyTorch-CPU ResNet18 S=8, Adam) on the host cores: the same
    step on a bounded sample (1 warm-up + up to 3 timed steps of batch 8)."""
    import numpy as np
    from oracle import features_ref, models_ref
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = min(avail, 16)          # a 1-GPU box grants a 16-core share; more threads only oversubscribe
    torch.set_num_threads(cores)
    rng = np.random.default_rng(1234)
    stems = (0.1 * rng.standard_normal((BATCH, N_STEMS, N_SAMPLES, CHANNELS))).astype(np.float32)
    mix = (stems * np.linspace(0.5, 1.5, N_STEMS, dtype=np.float32)[None, :, None, None]).sum(1)
    torch.manual_seed(0)
    model = models_ref.RefResNet18(n_stems=N_STEMS, input_shape=(N_FFT // 2 + 1, 1 + N_SAMPLES // HOP)).train()
    opt = torch.optim.Adam(model.parameters(), weight_decay=1e-5)

    def step():
        feats = [[features_ref.compute_features(stems[b, s].mean(1), N_FFT, HOP, np.float32) for s in range(N_STEMS)]
                 for b in range(BATCH)]
        gts = [features_ref.compute_features(mix[b].mean(1), N_FFT, HOP, np.float32) for b in range(BATCH)]
        x, gt = torch.from_numpy(np.asarray(feats)), torch.from_numpy(np.asarray(gts))
        models_ref.train_step_ref(model, opt, x, gt)

    step()
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 and (time.perf_counter() - t_all) < seconds_budget:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    frames = BATCH * N_STEMS * (1 + N_SAMPLES // HOP)
    return {'value': frames / med, 'unit': 'stem-spectrogram-frames/s', 'cores': cores, 'kind': 'port',
            'sample': '%d timed steps (median) of batch %d: numpy STFT front-end + PyTorch-CPU ResNet18(S=8) fwd+MSE+bwd+Adam, f32'
                      % (len(times), BATCH), 's_per_step': med}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    n_dev = torch.cuda.device_count()
    local = local % max(n_dev, 1)                      # rehearsal only: more ranks than GPUs share a device (gloo)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('DAM_DIST_BACKEND', 'nccl')     # 'nccl' is RCCL on ROCm; 'gloo' for 1-GPU rehearsals
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=device)
        else:
            torch.distributed.init_process_group(backend)

    import deep_audio_mixer_amd  # noqa: F401
    from deep_audio_mixer_amd import build
    if rank == 0:
        build.build_lib()
    if world > 1:
        torch.distributed.barrier()
    from deep_audio_mixer_amd.engine import TrainStep
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam

    torch.manual_seed(0)
    frames_t = 1 + N_SAMPLES // HOP
    model = ResNet18(n_stems=N_STEMS, input_shape=(N_FFT // 2 + 1, frames_t)).to(device).train()
    if world > 1:     # identical replicas: broadcast rank 0's parameters and buffers
        for t in list(model.parameters()) + list(model.buffers()):
            torch.distributed.broadcast(t.data, 0)
    opt = Adam(model.parameters(), weight_decay=1e-5, world_size=world)
    n_resident = 4 * BATCH
    stems, mix = synth_clips(n_resident, device, 1234 + rank)
    step = TrainStep(model, opt, N_STEMS, N_SAMPLES, CHANNELS, BATCH, N_FFT, HOP, use_graph=not args.no_graph)
    step.load_batch(stems[:BATCH], mix[:BATCH])
    step.capture(warmup=2)

    def run(k, first):
        for i in range(k):
            j = ((first + i) % (n_resident // BATCH)) * BATCH
            step.load_batch(stems[j:j + BATCH], mix[j:j + BATCH])      # device-to-device: inputs stay in HBM
            step()

    run(args.warmup, 0)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    loss = step.loss.item()
    frames_per_step = step.frames_per_step * world
    value = frames_per_step * args.steps / dt

    result = {
        'metric': 'stem-spectrogram-frames/sec (train)', 'value': value, 'unit': 'stem-spectrogram-frames/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'C3: model_resnet (ResNet18, 8 stems) train step incl. STFT front-end, 8-stem 3 s @ 44.1 kHz '
                               'stereo clips, n_fft 2048 hop 1024 -> 8x1025x130 per clip',
                   'batch_per_gpu': BATCH, 'global_batch': BATCH * world, 'parallelism': 'dp%d' % world,
                   'hip_graph': not args.no_graph, 'final_loss': loss},
    }
    if rank == 0 and not args.no_roofline:
        probe = roofline_probe(device)
        k = probe['layer1_conv3x3_16x16']
        # fwd+bwd algorithmic FLOPs of the whole step (29.5 GFLOP per clip) over the step time
        # traffic: HBM bytes per launch from the committed PMC passes on this kernel at this shape
        # (profiles/r01_layer1_conv_wgrad_pmc.csv: FETCH_SIZE 36,648 KB x 2 [gfx950 halves wide coalesced reads] + WRITE_SIZE 66,625 KB)
        # achieved = algorithmic FLOPs per launch / the average duration of ALL launches of this kernel in a step (the nine
        # layer1-shaped launches timed as the step issues them: what `rocprofv3 --kernel-trace --stats` of this command
        # averages for conv_strip_kernel<4,1,1,true>); the plain forward launch alone is roofline.forward_only
        result['roofline'] = {'bound': 'mfma', 'achieved': k['step_mix_tflops'], 'peak': PEAK_F32_MFMA / 1e12,
                              'unit': 'TFLOP/s', 'frac': k['step_mix_tflops'] * 1e12 / PEAK_F32_MFMA, 'traffic': 139.9e6,
                              'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC on the forward launch, profiles/r01_layer1_conv_wgrad_pmc.csv)',
                              'kernel': 'conv_strip_kernel<4,1,1,true> (ResNet layer1 3x3 conv 16->16, 8x1025x130 px: 5 forward + 4 dgrad launches per step)',
                              'avg_launch_s': k['step_mix_s'], 'flops_per_launch': k['flops_per_launch'],
                              'forward_only': {'avg_launch_s': k['fwd_s'], 'achieved': k['fwd_tflops'],
                                               'frac': k['fwd_tflops'] * 1e12 / PEAK_F32_MFMA},
                              'hbm_alg_bytes_per_launch': k['alg_bytes_per_launch'],
                              'hbm_alg_GBps': k['alg_bytes_per_launch'] / k['step_mix_s'] / 1e9,
                              'hbm_frac_of_8TBps': k['alg_bytes_per_launch'] / k['step_mix_s'] / PEAK_HBM}
        result['roofline_extra'] = {
            'whole_step_tflops': 29.5e9 * BATCH * world * args.steps / dt / 1e12 / world,
            'whole_step_frac_of_fp32_mfma_peak': 29.5e9 * BATCH * args.steps / dt / PEAK_F32_MFMA,
            'kernels': probe}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
