#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric on MI355X.

  metric : stem-spectrogram-frames/sec (train), model_resnet, 8-stem 3 s @ 44.1 kHz (BASELINE config C3/C4)
  step   : STFT/log-mag front-end (9 tracks per clip, one launch) + ResNet18 forward + MSE + backward + gradient
           buckets (+ RCCL all-reduce overlapped with backward when N > 1) + Adam(+L2), batch 8 clips per GPU, float32
           end to end; PCM already resident in HBM when the timed region starts (synthetic clips)
  unit   : one 1025-bin STFT column of one input stem (8 stems x 130 frames = 1040 per clip)

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C1|C2|C3|C5]

With --gpus N > 1 and no WORLD_SIZE in the environment this process starts the N ranks itself
(`python -m torch.distributed.run --nproc-per-node N ... bench.py`) BEFORE touching the GPU and relays their output; it
never falls back to one rank.  Launched by torch.distributed.run it is one rank (RANK / LOCAL_RANK / WORLD_SIZE).

Prints ONE JSON line on rank 0, with the `roofline` object of the config's dominant kernel (measured live with HIP
events on the launch stream), the `cpu_baseline` object (the CPU oracle timed on the host cores, N = 1 only) and
`pcie_inclusive` (the same steps fed from page-locked host memory through an overlapped copy stream).
Other configs (--config) print the same shape of line for BASELINE.json's other configurations.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FFT = 2048
PEAK_F32_MFMA = 157.3e12        # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM = 8.0e12               # MI355X_MICROARCH.md: HBM3E spec

# BASELINE.json configs resolved to shapes (SURVEY section 8, table "Configs")
CONFIGS = {
    'C1': dict(model='scalar_1s', n_stems=2, sr=16000, seconds=1, hop=256, batch=8, gflop_fwd=9.65,
               workload='C1: model_scalar_1s train step incl. STFT front-end, 2-stem 1 s @ 16 kHz stereo clips, n_fft 2048 '
                        'hop 256 -> 2x1025x63 per clip'),
    'C2': dict(model='scalar_2s', n_stems=4, sr=44100, seconds=3, hop=1024, batch=4, gflop_fwd=36.89,
               workload='C2: model_scalar_2s train step incl. STFT front-end (HIP conv stack as well), 4-stem 3 s @ 44.1 kHz '
                        'stereo clips, n_fft 2048 hop 1024 -> 4x1025x130 per clip'),
    'C3': dict(model='resnet', n_stems=8, sr=44100, seconds=3, hop=1024, batch=8, gflop_fwd=9.89,
               workload='C3: model_resnet (ResNet18, 8 stems) train step incl. STFT front-end, 8-stem 3 s @ 44.1 kHz '
                        'stereo clips, n_fft 2048 hop 1024 -> 8x1025x130 per clip'),
    'C5': dict(model='resnet', n_stems=8, sr=44100, seconds=3, hop=1024, batch=59, gflop_fwd=9.89, song_seconds=180,
               workload='C5: inference_utils full-song inference, 8-stem 3-min @ 44.1 kHz stereo, 59 chunks of 3 s as one '
                        'eval-mode ResNet18 batch, one hipGraph from PCM in HBM to the peak-normalised master'),
}
CHANNELS = 2
N_HOST_CLIPS = 512              # dataset of 512 clips held in pinned host memory


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class ProgressWatchdog:
    """An N-rank run must not hang a whole node silently: every rank names the phase it enters (`phase(...)`); if NO new
    phase is entered for `timeout_s` seconds a daemon thread prints the rank and its last phase to stderr and ends the
    process with exit code 3 (os._exit from a plain thread: no GPU call, no re-exec; torch.distributed.run then stops the
    other ranks and the parent returns non-zero).  The collectives' own timeout (init_process_group(timeout=)) is set to
    the same figure; this is the backstop for a hang that is not inside a collective.  timeout_s <= 0: off."""

    def __init__(self, timeout_s, rank=0):
        import threading
        self.timeout_s, self.rank = float(timeout_s), rank
        self.last, self.since, self.history = 'start', time.monotonic(), []
        self._lock = threading.Lock()
        if self.timeout_s > 0:
            threading.Thread(target=self._watch, daemon=True).start()

    def phase(self, name):
        with self._lock:
            self.history.append((self.last, round(time.monotonic() - self.since, 3)))
            self.last, self.since = name, time.monotonic()

    def _watch(self):
        while True:
            time.sleep(min(1.0, max(0.05, self.timeout_s / 4)))
            with self._lock:
                idle, last = time.monotonic() - self.since, self.last
            if idle > self.timeout_s:
                sys.stderr.write('bench.py: rank %d made no progress for %.0f s in phase "%s" (phases so far: %s) -- giving up '
                                 '(exit 3)\n' % (self.rank, idle, last, ', '.join('%s %.1fs' % h for h in self.history[-8:])))
                sys.stderr.flush()
                os._exit(3)


def dist_timeout_s():
    """Seconds a rank may sit in one phase / one collective (DAM_DIST_TIMEOUT_S, default 300)."""
    try:
        return float(os.environ.get('DAM_DIST_TIMEOUT_S', '300'))
    except ValueError:
        return 300.0


def visible_gpu_count():
    """GPUs this process may use, WITHOUT touching a GPU runtime (the parent of an N-rank run must not initialise the
    device before it starts its children): the KFD topology nodes that have SIMDs (CPUs have none) and whose DRM render
    node is accessible to this user / container, narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES
    when one of them is set.  Returns None when /sys/class/kfd is not readable (not a ROCm host)."""
    import glob
    nodes = []
    for path in sorted(glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties'),
                       key=lambda q: int(q.split('/')[-2]) if q.split('/')[-2].isdigit() else 0):
        try:
            props = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get('simd_count', '0')) <= 0:
            continue
        minor = int(props.get('drm_render_minor', '-1'))
        dev = '/dev/dri/renderD%d' % minor
        if minor >= 0 and os.path.exists(dev) and os.access(dev, os.R_OK | os.W_OK):
            nodes.append(minor)
    if not nodes and not os.path.isdir('/sys/class/kfd/kfd/topology/nodes'):
        return None
    n = len(nodes)
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            ids = [t for t in v.split(',') if t.strip() != '']
            n = min(n, len(ids))
    return n


def spawn_ranks(args, argv):
    """Parent of an N-rank run: no GPU runtime call is made in this process -- the GPUs are counted from sysfs."""
    n_dev = visible_gpu_count()
    if n_dev is None:
        sys.stderr.write('bench.py: --gpus %d but /sys/class/kfd is not readable here (no ROCm GPU): refusing to run fewer '
                         'ranks\n' % args.gpus)
        return 2
    rehearsal = os.environ.get('DAM_DIST_BACKEND', 'nccl') != 'nccl'       # gloo rehearsal: ranks may share a GPU
    if n_dev < args.gpus and not rehearsal:
        sys.stderr.write('bench.py: --gpus %d but only %d GPU(s) visible; refusing to run fewer ranks\n' % (args.gpus, n_dev))
        return 2
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.run(cmd, env=env).returncode


def synth_clips(n_clips, n_stems, n_samples, device, seed):
    """stems 0.1*N(0,1), mix = sum_s linspace(0.5, 1.5, S)[s] * stem_s (stereo float32).
    Returns [n_clips, S+1, n, ch]: every clip's stems followed by its mix."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    clips = torch.empty((n_clips, n_stems + 1, n_samples, CHANNELS), device=device, dtype=torch.float32)
    clips[:, :n_stems] = 0.1 * torch.randn((n_clips, n_stems, n_samples, CHANNELS), generator=g, device=device)
    w = torch.linspace(0.5, 1.5, n_stems, device=device).view(1, n_stems, 1, 1)
    clips[:, n_stems] = (clips[:, :n_stems] * w).sum(1)
    return clips


def time_kernel(fn, iters=30, warm=5, graph=False):
    """Average device time of one launch sequence, HIP events on the current (launch) stream.  graph=True: the sequence is
    captured once and replayed, as the training step replays it -- no host time between its launches (eager launches of
    ~55 us kernels leave the interpreter ~5 % behind the device on some boxes, which would be timed as kernel time)."""
    import torch
    for _ in range(warm):
        fn()
    run = fn
    if graph:
        from deep_audio_mixer_amd import staging
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with staging.capture_guard:
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                fn()
        run = g.replay
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        run()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


def build_model(cfg, device, train=True):
    import torch
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.models.model_scalar_1s import MixingModelScalar1s
    from deep_audio_mixer_amd.models.model_scalar_2s import MixingModelScalar2s
    ctor = {'resnet': ResNet18, 'scalar_1s': MixingModelScalar1s, 'scalar_2s': MixingModelScalar2s}[cfg['model']]
    t = 1 + cfg['sr'] * cfg['seconds'] // cfg['hop']
    torch.manual_seed(0)
    m = ctor(n_stems=cfg['n_stems'], input_shape=(N_FFT // 2 + 1, t)).to(device)
    return m.train() if train else m.eval()


# ------------------------------------------------------------------------------------------------ roofline probes
def roofline_resnet_layer1(device, batch, t_frames):
    """Dominant kernel of C3 / C5: the implicit-GEMM convolution of the thin full-resolution layers (conv_strip_kernel;
    ResNet layer1: 16 -> 16 channels, 3x3, on batch x 1025 x T pixels).  Algorithmic FLOPs per launch =
    2 * B*H*W * Cout * 9*Cin; algorithmic HBM bytes = 4 * B*H*W * (Cin + Cout); the bound is the
    fp32 matrix pipe (157.3 TFLOP/s), with the HBM figure beside it because the intensity (36 FLOP/B) sits at the ridge."""
    import torch
    from deep_audio_mixer_amd import ops
    B, H, W, c = batch, N_FFT // 2 + 1, t_frames, 16
    x = torch.randn((B, H, W, c), device=device)
    wt = torch.randn((c, c, 3, 3), device=device) * 0.05
    wp, wpt = ops.pack_weights(wt), ops.pack_weights(wt, transpose=True)
    dy = torch.randn((B, H, W, c), device=device)
    flops = 2.0 * B * H * W * c * 9 * c
    out = {'flops_per_launch': flops, 'alg_bytes_per_launch': 4.0 * B * H * W * 2 * c}
    out['fwd_s'] = time_kernel(lambda: ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1), warm=100)
    out['wgrad_s'] = time_kernel(lambda: ops.conv2d_wgrad(x, dy, c, 3, 3, 1, 1, 1), warm=40)
    # the launches of this kernel in one training step, as the step issues them (stem + two BasicBlocks of layer1):
    # 3 forward with BatchNorm statistics, 2 forward with statistics and the fused input affine, 2 plain data gradients,
    # 2 data gradients with the residual / mask epilogue -- what a kernel trace of the step averages
    sc, sh = torch.rand(c, device=device) + 0.5, torch.randn(c, device=device)
    mean, invstd = torch.randn(c, device=device), torch.rand(c, device=device) + 0.5
    buf = ops.bn_partial_buffer(device, c)
    msk = torch.randn((B, H, W, c), device=device)
    bits = torch.randint(0, 16, (B, H, W, c // 4), device=device, dtype=torch.uint8)
    x2 = torch.randn((B, H, W, c), device=device)

    # exactly the nine launches of this kernel in a captured C3 step (profiles/r0N_C3_kernel_trace_summary.txt):
    #   EPI 0 x 5: three forwards with the BatchNorm statistics epilogue + two that also apply the producer's affine+ReLU on load
    #   EPI 1 x 2: conv2's data gradient with bn1's backward sums in the epilogue
    #   EPI 2 x 1: layer1.0's conv1 data gradient: residual + sign-byte mask + the stem BatchNorm's sums (mask from its affine)
    #   EPI 3 x 1: layer1.1's conv1 data gradient: residual + sign-byte mask + layer1.0.bn2's sums (mask from sign bytes)
    def step_mix():
        for _ in range(3):
            ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1, bn_partial=buf)
        for _ in range(2):
            ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1, bn_partial=buf, in_scale=sc, in_shift=sh, relu_in=True)
        for _ in range(2):
            ops.conv2d_dgrad(dy, wpt, c, H, W, 3, 3, 1, 1, 1, bn_bwd=(x, mean, invstd, sc, sh))
        ops.conv2d_dgrad(dy, wpt, c, H, W, 3, 3, 1, 1, 1, res=x2, res_mask=msk, res_mask_bits=bits, bn_bwd=(x, mean, invstd, sc, sh))
        ops.conv2d_dgrad(dy, wpt, c, H, W, 3, 3, 1, 1, 1, res=x2, res_mask=msk, res_mask_bits=bits,
                         bn_bwd=(x, mean, invstd, None, None, bits))
    # 40 warm-up rounds (20 ms of launches): timed after 5 the same sequence reads 60 us per launch instead of 55-56 (clocks and
    # memory-side cache still cold) -- the figure the step's kernel trace shows is the warm one (56.6-57.5 us from box to box).
    # The same nine launches captured and replayed are reported beside it: no host time between the launches, but the capture's
    # private memory pool makes it the less repeatable of the two (56.0-62.1 us over this round's runs).
    out['step_mix_s'] = time_kernel(step_mix, warm=40) / 9.0
    try:
        out['step_mix_graph_s'] = time_kernel(step_mix, warm=5, graph=True) / 9.0
    except Exception as e:
        sys.stderr.write('bench: roofline probe could not be captured (%r)\n' % (e,))
        out['step_mix_graph_s'] = None
    return out


def roofline_conv(device, batch, h, w, cin, cout, k):
    """A valid k x k convolution (scalar models' conv_b5: the tile kernel, conv_igemm_kernel)."""
    import torch
    from deep_audio_mixer_amd import ops
    x = torch.randn((batch, h, w, cin), device=device)
    wt = torch.randn((cout, cin, k, k), device=device) * 0.02
    wp = ops.pack_weights(wt)
    ho, wo = h - k + 1, w - k + 1
    dy = torch.randn((batch, ho, wo, cout), device=device)
    flops = 2.0 * batch * ho * wo * cout * k * k * cin
    return {'flops_per_launch': flops, 'alg_bytes_per_launch': 4.0 * batch * (h * w * cin + ho * wo * cout),
            'fwd_s': time_kernel(lambda: ops.conv2d_fwd(x, wp, cout, k, k, 1, 0, 1), iters=10, warm=3),
            'wgrad_s': time_kernel(lambda: ops.conv2d_wgrad(x, dy, cout, k, k, 1, 0, 1), iters=10, warm=3)}


def pmc_traffic_layer1():
    """HBM bytes per launch of the dominant kernel's forward launch (batch 8) from the COMMITTED PMC passes of the last profile
    collection (profiles/r05_pmc_summary.csv, tools/collect_profiles.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc
    runs of tools/conv_probe.py layer1; KiB per dispatch -- WRITE_SIZE 66,625 is exactly the launch's 68,224,000 output bytes / 1024 --
    FETCH_SIZE x 2 = the guide's gfx950 correction for wide coalesced reads; earlier rounds' lines multiplied by 1000: 140.0 MB).  bench.py cannot read the counters of its own launches: the figure belongs to that separate run of the same kernel,
    not to this one.  -> (bytes, source) or (None, reason)."""
    import csv
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r05_pmc_summary.csv')
    try:
        got = {}
        for r in csv.DictReader(open(path)):
            if r['pass'] in ('pmc_layer1_3', 'pmc_layer1_4') and 'conv_strip_kernel' in r['kernel'] and r['counter'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                got[r['counter']] = float(r['mean_per_dispatch'])
        fetch, write = got['FETCH_SIZE'], got['WRITE_SIZE']
    except (OSError, KeyError, ValueError):
        return None, 'profiles/r05_pmc_summary.csv not readable here: no PMC figure for this line'
    total = (2.0 * fetch + write) * 1024.0
    return total, ('profiles/r05_pmc_summary.csv, passes pmc_layer1_3 / pmc_layer1_4 (separate rocprofv3 --pmc runs of the same '
                   'forward launch, batch 8): FETCH_SIZE %.0f KiB x 2 (gfx950: wide coalesced reads count half) + WRITE_SIZE %.0f KiB '
                   '= %.1f MB = %.2f x the 136.4 MB algorithmic bytes; read from the committed file, not re-measured by this run'
                   % (fetch, write, total / 1e6, total / 136448000.0))


def roofline_object(name, cfg, device, t_frames):
    if cfg['model'] == 'resnet':
        k = roofline_resnet_layer1(device, cfg['batch'], t_frames)
        train = name != 'C5'
        t = k['step_mix_s'] if train else k['fwd_s']
        tf = k['flops_per_launch'] / t / 1e12
        kern = ('conv_strip_kernel<4,1,1,true,false,EPI,SO> (ResNet layer1 3x3 conv 16->16, %dx1025x%d px: ' % (cfg['batch'], t_frames) +
                ('its 9 launches per step: 5 forward with statistics, 4 data gradients with the BatchNorm-backward / residual / '
                 'upstream-sum epilogues)' if train else 'forward launches of the eval-mode chunk batch)'))
        # traffic: measured in a SEPARATE rocprofv3 --pmc run (bench.py cannot read PMC counters of its own launches);
        # the figure below was taken at batch 8 on the forward launch of this kernel at the commit named in traffic_source
        traffic, traffic_src = pmc_traffic_layer1() if cfg['batch'] == 8 else (None, None)
        obj = {'bound': 'mfma', 'achieved': tf, 'peak': PEAK_F32_MFMA / 1e12, 'unit': 'TFLOP/s', 'frac': tf * 1e12 / PEAK_F32_MFMA,
               'traffic': traffic, 'traffic_source': traffic_src,
               'kernel': kern, 'avg_launch_s': t, 'flops_per_launch': k['flops_per_launch'],
               'avg_launch_graph_replay_s': k.get('step_mix_graph_s') if train else None,
               'forward_only': {'avg_launch_s': k['fwd_s'], 'achieved': k['flops_per_launch'] / k['fwd_s'] / 1e12,
                                'frac': k['flops_per_launch'] / k['fwd_s'] / PEAK_F32_MFMA},
               'wgrad': {'avg_launch_s': k['wgrad_s'], 'achieved': k['flops_per_launch'] / k['wgrad_s'] / 1e12},
               'hbm_alg_bytes_per_launch': k['alg_bytes_per_launch'],
               'hbm_alg_GBps': k['alg_bytes_per_launch'] / t / 1e9,
               'hbm_frac_of_8TBps': k['alg_bytes_per_launch'] / t / PEAK_HBM}
        return obj
    # scalar models: conv_b5 (9x9, 64 -> 128) is 60 % of the forward FLOPs
    f, t = N_FFT // 2 + 1, t_frames
    dil = 2 if cfg['model'] == 'scalar_2s' else 1
    f, t = (f - dil * 2 - 1) // 2 + 1, (t - dil * 2 - 1) // 2 + 1          # conv_b1 (k3 s2, dilated in the 2 s model)
    for kk in (5, 5, 7):
        f, t = f - kk + 1, t - kk + 1
    k = roofline_conv(device, cfg['batch'], f, t, 64, 128, 9)
    tf = k['flops_per_launch'] / k['fwd_s'] / 1e12
    return {'bound': 'mfma', 'achieved': tf, 'peak': PEAK_F32_MFMA / 1e12, 'unit': 'TFLOP/s', 'frac': tf * 1e12 / PEAK_F32_MFMA,
            'traffic': None, 'kernel': 'conv_igemm_kernel (conv_b5: 9x9 valid conv 64->128 on %dx%dx%d px, forward launch)'
                                       % (cfg['batch'], f, t),
            'avg_launch_s': k['fwd_s'], 'flops_per_launch': k['flops_per_launch'],
            'wgrad': {'avg_launch_s': k['wgrad_s'], 'achieved': k['flops_per_launch'] / k['wgrad_s'] / 1e12},
            'hbm_alg_bytes_per_launch': k['alg_bytes_per_launch']}


# ------------------------------------------------------------------------------------------------ CPU baseline
def host_cores():
    """(threads to use, os.cpu_count()): the cgroup CPU quota when there is one (a 1-GPU box grants a share of the
    host), else the affinity mask."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = None
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            quota = max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    return (min(avail, quota) if quota else avail), (os.cpu_count() or avail)


class _CpuClipFeatures:
    """Dataset of the 6-worker leg: item = the oracle's features of one clip (data/dataset.py:185-210 on the CPU)."""

    def __init__(self, clips, hop):
        self.clips, self.hop = clips, hop

    def __len__(self):
        return len(self.clips)

    def __getitem__(self, i):
        import torch
        from oracle import features_ref
        import numpy as np
        x, gt = features_ref.clip_features(self.clips[i], N_FFT, self.hop, np.float32)
        return torch.from_numpy(x), torch.from_numpy(gt)


def cpu_baseline(cfg):
    """the CPU oracle (oracle/: numpy front-end + PyTorch-CPU model, Adam) on the host cores,
    same synthetic clips, float32; 2 warm-up + 5 timed steps, medians; legs: (a) front-end only, single process and with
    the reference's DataLoader(num_workers=6) arrangement, (b) model train step only, (c) end to end."""
    import numpy as np
    import torch
    from oracle import features_ref, models_ref
    cores, visible = host_cores()
    torch.set_num_threads(cores)
    S, B, hop = cfg['n_stems'], cfg['batch'], cfg['hop']
    n = cfg['sr'] * cfg['seconds']
    t = 1 + n // hop
    rng = np.random.default_rng(1234)
    clips = np.empty((B, S + 1, n, CHANNELS), dtype=np.float32)
    clips[:, :S] = 0.1 * rng.standard_normal((B, S, n, CHANNELS))
    clips[:, S] = (clips[:, :S] * np.linspace(0.5, 1.5, S, dtype=np.float32)[None, :, None, None]).sum(1)
    torch.manual_seed(0)
    ctor = {'resnet': models_ref.RefResNet18, 'scalar_1s': models_ref.RefMixingModelScalar1s,
            'scalar_2s': models_ref.RefMixingModelScalar2s}[cfg['model']]
    model = ctor(n_stems=S, input_shape=(N_FFT // 2 + 1, t)).train()
    opt = torch.optim.Adam(model.parameters(), weight_decay=1e-5)

    def step():
        t0 = time.perf_counter()
        items = [features_ref.clip_features(clips[b], N_FFT, hop, np.float32) for b in range(B)]
        x, gt = torch.from_numpy(np.stack([i[0] for i in items])), torch.from_numpy(np.stack([i[1] for i in items]))
        t1 = time.perf_counter()
        models_ref.train_step_ref(model, opt, x, gt)
        return t1 - t0, time.perf_counter() - t1

    warm, timed = 2, 5
    t_all = time.perf_counter()
    for _ in range(warm):
        step()
    parts = []
    while len(parts) < timed and (len(parts) < 3 or time.perf_counter() - t_all < 40.0):
        parts.append(step())
    med = lambda v: sorted(v)[len(v) // 2]
    fe, mo, e2e = med([p[0] for p in parts]), med([p[1] for p in parts]), med([p[0] + p[1] for p in parts])
    frames = B * S * t
    legs = {'front_end_single_process_s': fe, 'model_train_step_s': mo, 'end_to_end_s': e2e,
            'front_end_single_process_frames_per_s': frames / fe, 'model_train_step_frames_per_s': frames / mo}
    try:        # the reference's arrangement: 6 DataLoader workers computing features (training.ipynb cell 6)
        # steady state only: prefetch_factor 2 x 6 workers = 12 batches may be finished before the clock starts, so the first
        # 12 batches are skipped and the NEXT 48 are timed (at most ~3 batches' worth of work was under way at t0: <= 6 %)
        n_skip, n_timed = 12, 48
        torch.set_num_threads(1)                    # worker processes inherit it: six single-threaded workers, as torch sets up
        loader = torch.utils.data.DataLoader(_CpuClipFeatures([clips[b % B] for b in range((n_skip + n_timed) * B)], hop),
                                             batch_size=B, num_workers=6, shuffle=False)
        it = iter(loader)
        for _ in range(n_skip):
            next(it)
        t0 = time.perf_counter()
        k = sum(1 for _ in it)
        fe6 = (time.perf_counter() - t0) / max(k, 1)
        torch.set_num_threads(cores)
        legs['front_end_6_workers_s'] = fe6
        legs['front_end_6_workers_frames_per_s'] = frames / fe6
        legs['front_end_6_workers_batches_timed'] = k
        # sanity: six workers cannot beat six times one process (numpy's FFT is single-threaded)
        legs['front_end_6_workers_speedup_vs_single'] = fe / fe6
        if fe / fe6 > 6.5:
            legs['front_end_6_workers_warning'] = 'speed-up above 6x: prefetched batches leaked into the timed span'
    except Exception as e:      # worker processes unavailable on this host: the leg is reported as missing, not invented
        legs['front_end_6_workers_error'] = repr(e)[:200]
    return {'value': frames / e2e, 'unit': 'stem-spectrogram-frames/s', 'cores': cores, 'cores_visible': visible,
            'kind': 'port',
            'sample': '%d warm-up + %d timed steps (medians) of batch %d: numpy STFT front-end + PyTorch-CPU %s(S=%d) '
                      'fwd+MSE+bwd+Adam, f32, single process, %d torch threads' % (warm, len(parts), B, cfg['model'], S, cores),
            's_per_step': e2e, 'legs': legs}


# ------------------------------------------------------------------------------------------------ runs
def dist_env():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def replica_sync_proof(model, opt, device, dev_index, world):
    """After the timed steps every rank all-gathers (a) a checksum of the optimizer's flat PARAMETER buffer -- identical
    only if every step's gradients were summed over all ranks: each rank trains on its own clips -- (b) a checksum of the
    first BatchNorm's running mean (local statistics: expected to DIFFER between ranks, reported, not required), (c) which
    physical device it ran on (index, PCI bus id, uuid where the runtime gives one).  World size 1: all None."""
    import torch
    if world <= 1:
        return {'replicas_in_sync': None, 'checksums': None, 'devices': None}
    flat = opt._flat.detach()
    ints = flat.view(torch.int32).to(torch.int64)
    # order-sensitive 64-bit checksum of the raw parameter bits (position-weighted sum, exact in int64 arithmetic mod 2^64)
    w = torch.arange(1, ints.numel() + 1, device=device, dtype=torch.int64)
    mine = torch.stack([(ints * w).sum(), ints.sum(),
                        next(b for n, b in model.named_buffers() if n.endswith('running_mean')).detach().double().sum().mul(1e6).round().to(torch.int64)])
    if torch.distributed.get_backend() != 'nccl':      # gloo rehearsal: its CUDA path stalls behind busy streams
        torch.cuda.synchronize()
        mine = mine.cpu()
    allc = [torch.zeros_like(mine) for _ in range(world)]
    torch.distributed.all_gather(allc, mine)
    props = torch.cuda.get_device_properties(dev_index)
    ident = {'device_index': dev_index, 'name': props.name, 'uuid': str(getattr(props, 'uuid', '')),
             'pci_bus_id': getattr(props, 'pci_bus_id', None), 'pci_device_id': getattr(props, 'pci_device_id', None)}
    idents = [None] * world
    torch.distributed.all_gather_object(idents, ident)
    rows = [[int(v) for v in c.tolist()] for c in allc]
    in_sync = all(r[:2] == rows[0][:2] for r in rows)
    return {'replicas_in_sync': bool(in_sync),
            'checksums': {'flat_params_weighted': [r[0] for r in rows], 'flat_params_sum': [r[1] for r in rows],
                          'bn_running_mean_sum_x1e6 (local statistics, may differ)': [r[2] for r in rows]},
            'devices': idents}


def run_train(name, cfg, args):
    import torch
    rank, world, local = dist_env()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    n_dev = torch.cuda.device_count()
    backend = os.environ.get('DAM_DIST_BACKEND', 'nccl')     # 'nccl' is RCCL on ROCm; 'gloo' for 1-GPU rehearsals
    if world > 1 and backend == 'nccl' and n_dev < world:
        raise SystemExit('%d ranks but %d GPUs: RCCL needs one GPU per rank' % (world, n_dev))
    dev_index = local % max(n_dev, 1)                  # gloo rehearsal only: more ranks than GPUs share a device
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    dog = ProgressWatchdog(dist_timeout_s() if world > 1 else 0, rank)
    if world > 1:
        import datetime
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dog.phase('init_process_group(%s)' % backend)
        limit = datetime.timedelta(seconds=max(1.0, dist_timeout_s()))
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=device, timeout=limit)
        else:
            torch.distributed.init_process_group(backend, timeout=limit)
    dog.phase('build + first barrier')

    import deep_audio_mixer_amd  # noqa: F401
    from deep_audio_mixer_amd import build
    if rank == 0:
        build.build_lib()
    if world > 1:
        torch.distributed.barrier()
    from deep_audio_mixer_amd.engine import TrainStep
    from deep_audio_mixer_amd.optim import Adam
    from deep_audio_mixer_amd import staging

    S, B, hop = cfg['n_stems'], cfg['batch'], cfg['hop']
    n = cfg['sr'] * cfg['seconds']
    t_frames = 1 + n // hop
    dog.phase('model + broadcast')
    model = build_model(cfg, device)
    if world > 1:     # identical replicas: broadcast rank 0's parameters and buffers
        for t in list(model.parameters()) + list(model.buffers()):
            torch.distributed.broadcast(t.data, 0)
    opt = Adam(model.parameters(), weight_decay=1e-5, world_size=world)
    n_resident = 4 * B
    clips = synth_clips(n_resident, S, n, device, 1234 + rank)
    step = TrainStep(model, opt, S, n, CHANNELS, B, N_FFT, hop, use_graph=not args.no_graph, overlap=not args.no_overlap,
                     copy_mark=not args.no_copy_mark)
    step.load_clips(clips[:B])
    dog.phase('eager warm-up steps + graph capture')
    step.capture(warmup=2)

    resident = [clips[j:j + B] for j in range(0, n_resident, B)]
    if not args.bind_per_step:
        # inputs stay where they are in HBM: the front-end walks a device table of the resident batches with the optimizer's
        # device-side step count (DAM_PCM_ROTATE) -- nothing is launched between the graph replays
        step.bind_rotation(resident)

    def run(k, first):
        for i in range(k):
            if args.bind_per_step:               # A/B: the address word re-pointed per step (a fill launch between the replays)
                step.bind_clips(resident[(first + i) % len(resident)])
            step()

    def timed(fn, k, first):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(k, first)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = tt.item()
        return dt

    def local_time(fn, k, first):
        """This rank's own clock around k steps (no barrier inside): per-rank spread of the N-rank line."""
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(k, first)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    # N > 1: how should the staged step exchange its buckets ON THIS NODE?  'overlapped' hides the big bucket's all-reduce behind
    # graph A2 but makes RCCL's stream wait for an event of the training stream (0.10-0.13 ms per step for the event traffic alone,
    # measured with one rank: profiles/r05_nccl_sync_probe.txt) and puts RCCL's kernels beside A2's 248-workgroup kernels; 'inline'
    # runs synchronous collectives on the training stream (nothing waits across streams, nothing overlaps).  Both are timed over
    # the same three graphs before the warm-up; every rank sees the same (max-over-ranks) times and keeps the faster one.
    reduce_cal = None
    three_graphs = world > 1 and step.staged and step._graphs is not None and len(step._graphs) == 3
    if three_graphs:
        dog.phase('reduce-mode calibration')
        if args.reduce_mode == 'auto':
            reduce_cal, n_cal = {}, max(5, args.steps // 2)
            for mode in ('overlapped', 'inline'):
                step.reduce_mode = mode
                run(2, 0)
                reduce_cal[mode] = 1e3 * timed(run, n_cal, 0) / n_cal
            step.reduce_mode = min(reduce_cal, key=reduce_cal.get)
        else:
            step.reduce_mode = args.reduce_mode
    chosen_mode = step.reduce_mode if three_graphs else None
    dog.phase('warm-up steps')
    run(args.warmup, 0)
    step.measure_exposed = world > 1
    dog.phase('timed region (%d steps)' % args.steps)
    dt = timed(run, args.steps, args.warmup)
    exposed = step.exposed_wait_ms()
    step.measure_exposed = False
    loss = step.loss.item()
    frames_per_step = step.frames_per_step * world
    value = frames_per_step * args.steps / dt
    # the same region repeated (diagnostic: run-to-run spread of a 0.1 s region; `value` stays the region above)
    dog.phase('repeat regions + per-rank clocks')
    reps = [1e3 * timed(run, args.steps, args.warmup + (r + 1) * args.steps) / args.steps for r in range(args.repeat)]
    rank_ms = 1e3 * local_time(run, args.steps, 0) / args.steps
    if world > 1:
        allr = [None] * world
        torch.distributed.all_gather_object(allr, (rank_ms, exposed))
    else:
        allr = [(rank_ms, exposed)]
    gflop_step = cfg['gflop_fwd'] * 2.98 * B            # fwd+bwd algorithmic FLOPs per rank and step (SURVEY 8d ratio)
    # N > 1, staged step: does overlapping the big bucket's all-reduce with graph A2 PAY?  The strip kernels of A2 are sized 248
    # workgroups for 256 CUs and a kernel that shares their CUs slows them (side-stream weight gradients, every
    # big kernel 1.3-1.7x longer) -- RCCL's kernels will sit exactly there.  So the same invocation times two more regions over
    # the SAME three graphs: the all-reduces beside A2 (as `value` was measured) and both started only after A2, each with
    # HIP events around graph A2.  One SCALE record then says which schedule the node prefers and what the overlap cost A2.
    overlap_ab = None
    if three_graphs:
        dog.phase('overlap A/B regions')
        ab = {}
        for label, ov in (('overlapped', True), ('serialized', False), ('inline', True)):
            step.reduce_mode = 'inline' if label == 'inline' else 'overlapped'
            step.overlap_reduce, step.measure_a2, step.measure_exposed = ov, True, True
            d = timed(run, args.steps, args.warmup)
            a2, ex = step.a2_ms(), step.exposed_wait_ms()
            got = [None] * world
            torch.distributed.all_gather_object(got, (a2, ex))
            ab[label] = {'ms_per_step': 1e3 * d / args.steps, 'graph_A2_ms': [g[0] for g in got],
                         'exposed_allreduce_wait_ms': [g[1] for g in got]}
        step.overlap_reduce, step.measure_a2, step.measure_exposed = True, False, False
        step.reduce_mode = chosen_mode
        a2o, a2s = max(ab['overlapped']['graph_A2_ms']), max(ab['serialized']['graph_A2_ms'])
        overlap_ab = dict(ab, overlap_pays=bool(ab['overlapped']['ms_per_step'] < min(ab['serialized']['ms_per_step'],
                                                                                      ab['inline']['ms_per_step'])),
                          graph_A2_slowdown_from_overlap=(a2o / a2s if a2s else None),
                          note='same three graphs; "serialized" starts both asynchronous all-reduces after graph A2 has been '
                               'enqueued; "inline" runs synchronous all-reduces on the training stream (no stream waits for the '
                               'training stream; exposed_allreduce_wait_ms is then the small bucket\'s whole exchange); '
                               'graph_A2_ms = HIP events around graph A2 per rank (events are recorded in these regions only)')
    dog.phase('replica sync proof')
    sync = replica_sync_proof(model, opt, device, dev_index, world)

    result = {
        'metric': 'stem-spectrogram-frames/sec (train)', 'value': value, 'unit': 'stem-spectrogram-frames/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': cfg['workload'], 'batch_per_gpu': B, 'global_batch': B * world, 'parallelism': 'dp%d' % world,
                   'hip_graph': not args.no_graph, 'final_loss': loss, 'world_size': world, 'device_index': dev_index,
                   'dist_backend': (backend if world > 1 else None),
                   'rccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version()) if world > 1 and backend == 'nccl' else None,
                   'grad_buckets': opt.n_buckets, 'allreduce_overlap': bool(step.staged) and chosen_mode != 'inline',
                   # how `value` exchanged its gradient buckets, and the calibration that chose it (ms per step, max over ranks)
                   'reduce_mode': chosen_mode, 'reduce_mode_calibration_ms': reduce_cal,
                   # N > 1: proof that the ranks were N distinct devices and that the all-reduce averaged -- replicas that
                   # each saw DIFFERENT clips hold bit-identical parameters after the timed steps (None on one rank)
                   'replicas_in_sync': sync['replicas_in_sync'], 'replica_checksums': sync['checksums'],
                   'rank_devices': sync['devices'],
                   # the reference loop reads loss.item() every batch (model_trainer.py:41,43); the timed region here does not
                   # (throughput metric): the host only enqueues graph replays.  `--via-trainer` times the loop WITH the sync.
                   'sync_per_step': False},
    }
    if reps:
        srt = sorted(reps)
        result['repeat'] = {'regions': len(reps), 'steps_per_region': args.steps, 'ms_per_step_median': srt[len(srt) // 2],
                            'ms_per_step_min': srt[0], 'ms_per_step_max': srt[-1]}
    if world > 1:
        result['per_rank'] = {'ms_per_step_min': min(a[0] for a in allr), 'ms_per_step_max': max(a[0] for a in allr),
                              'ms_per_step': [a[0] for a in allr],
                              # device time between the end of backward (graph A2) and both buckets having arrived: the part of
                              # the all-reduce that did NOT hide behind backward, per step
                              'exposed_allreduce_wait_ms': [a[1] for a in allr]}
    if overlap_ab is not None:
        result['overlap_ab'] = overlap_ab
    if world > 1:
        result['config']['dist_timeout_s'] = dist_timeout_s()
    if args.breakdown and world > 1:
        dog.phase('breakdown')
        result['breakdown'] = ddp_breakdown(step, device)
    # the same steps fed from page-locked HOST memory (512 clips in pinned host memory): batch k+1 is uploaded on
    # a copy stream while step k runs.  PCIe-inclusive: reported beside `value`, never as `value`.
    if not args.no_host_stream:
        dog.phase('host-streamed region')
        n_host = max(2 * B, min(N_HOST_CLIPS, args.host_clips) // B * B)
        host = torch.empty((n_host, S + 1, n, CHANNELS), dtype=torch.float32, pin_memory=True)
        for lo in range(0, n_host, n_resident):
            hi = min(lo + n_resident, n_host)
            host[lo:hi].copy_(synth_clips(hi - lo, S, n, device, 99 + rank + lo))
        stager = staging.BatchStager(host, B, device, gate=step.copy_mark)

        if not args.bind_per_step:
            # the staging buffers are walked by the same device table: buffer stager.k % 3 is the next one to be read
            step.bind_rotation(stager.bufs, first=stager.k % len(stager.bufs))

        def run_streamed(k, first):
            for _ in range(k):
                got = stager.next()              # the staging buffer the upload landed in is read in place
                if args.bind_per_step:
                    step.bind_clips(got)
                step()
        run_streamed(max(2, args.warmup), 0)
        dts = timed(run_streamed, args.steps, 0)
        result['pcie_inclusive'] = {'value': frames_per_step * args.steps / dts, 'ms_per_step': 1e3 * dts / args.steps,
                                    'host_clips_pinned': n_host, 'h2d_bytes_per_step': float(host[:B].numel() * 4),
                                    'copy_gate': ('step mark: the upload of batch k+1 starts when step k-1 reaches the backward '
                                                  'pass of its shallow layers' if step.copy_mark is not None else 'end of step k-1'),
                                    'note': 'batch k+1 uploaded from page-locked host memory on a copy stream beside the steps'}
        del host, stager
    dog.phase('roofline probe / cpu baseline (rank 0)')
    if rank == 0 and not args.no_roofline:
        result['roofline'] = roofline_object(name, cfg, device, t_frames)
        result['roofline_extra'] = {'whole_step_tflops': gflop_step * 1e9 * args.steps / dt / 1e12,
                                    'whole_step_frac_of_fp32_mfma_peak': gflop_step * 1e9 * args.steps / dt / PEAK_F32_MFMA}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(cfg)
    if rank == 0:
        result['config']['phases_s'] = {h[0]: h[1] for h in dog.history[1:]}
        print(json.dumps(result), flush=True)
    if world > 1:
        dog.phase('final barrier + destroy_process_group')
        # the other ranks wait here while rank 0 runs its roofline probe: the watchdog covers it, the barrier has the timeout
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _setup_single(device_index=0):
    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    torch.cuda.set_device(device_index)
    import deep_audio_mixer_amd  # noqa: F401
    from deep_audio_mixer_amd import build
    build.build_lib()
    return torch.device('cuda', device_index)


def _synthetic_songs(cfg, n_songs, chunks_per_song, seed=1234, pcm16=False):
    """In-memory songs of SURVEY 8(d)'s synthetic clips: {song: {track: float32 [n, 2]}}, S stems + 'mix'; pcm16: the same
    signals quantised to 16-bit PCM (what MedleyDB / MUSDB18-HQ stems are), int16 [n, 2]."""
    import numpy as np
    S, n = cfg['n_stems'], cfg['sr'] * cfg['seconds']
    rng = np.random.default_rng(seed)
    g = np.linspace(0.5, 1.5, S, dtype=np.float32)
    tracklist = ['stem%d' % i for i in range(S)] + ['mix']
    songs = {}
    for j in range(n_songs):
        stems = [(0.1 * rng.standard_normal((chunks_per_song * n, CHANNELS))).astype(np.float32) for _ in range(S)]
        tracks = stems + [sum(gi * st for gi, st in zip(g, stems))]
        if pcm16:
            tracks = [np.clip(np.rint(t * 32768.0), -32768, 32767).astype(np.int16) for t in tracks]
        songs['song%02d' % j] = dict(zip(tracklist, tracks))
    return songs, tracklist


def run_via_trainer(name, cfg, args):
    """The reference API at the measured speed: ModelTrainer.fit over MultitrackAudioDataset.batch_loader
    (in-memory songs -> decode threads -> page-locked staging -> H2D -> one front-end launch per batch), the loop body
    captured by ModelTrainer after its first batches, WITH the reference's per-batch loss.item() and progress prints."""
    import contextlib
    import io
    import tempfile
    import torch
    device = _setup_single()
    from deep_audio_mixer_amd import staging as _staging
    from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
    from deep_audio_mixer_amd.model_trainer import ModelTrainer
    from deep_audio_mixer_amd.optim import Adam
    S, B = cfg['n_stems'], cfg['batch']
    n_songs, chunks = 8, 48                                      # 384 clips = 48 batches of 8 per epoch (1.8 GB of 16-bit PCM)
    songs, tracklist = _synthetic_songs(cfg, n_songs, chunks, pcm16=not args.float_pcm)
    ds = MultitrackAudioDataset.from_arrays(songs, chunk_length=cfg['seconds'], sr=cfg['sr'], tracklist=tracklist, seed=1)
    # default: the loader runs the front-end per batch on the copy stream, beside the consumer's step, and the features are copied
    # into the captured step (15 us).  --pcm-loader: the loader hands ModelTrainer the uploaded PCM (PcmBatch), the captured step
    # contains the front-end and reads it in place -- measured 0.07-0.3 ms per step SLOWER (the 50 us front-end moves from beside
    # the step into it; profiles/r04_via_trainer_ab.txt)
    pcm_fed = args.pcm_loader
    if args.dataloader_workers:
        # training.ipynb cell 6 as written: torch's DataLoader with worker PROCESSES and pin_memory=True.  The workers decode on
        # the host (no GPU API), the batches arrive as page-locked HostPcmBatch objects; ModelTrainer uploads them beside the
        # running step and binds them to the PCM-fed captured step
        from torch.utils.data import DataLoader, Subset
        pcm_fed = True
        # an epoch of the in-memory set is 48 batches = 0.2 s of GPU time; the loader forks its six workers at every epoch start
        # (as any DataLoader without persistent_workers does), which a real epoch -- MedleyDB at this clip length: some 800
        # batches -- amortises.  --epoch-repeat walks the same clips R times per epoch; the epoch-start cost is reported beside it
        train_set = Subset(ds, list(range(len(ds))) * args.epoch_repeat) if args.epoch_repeat > 1 else ds
        train = DataLoader(train_set, batch_size=B, shuffle=False, num_workers=args.dataloader_workers, pin_memory=True,
                           drop_last=True, timeout=0, worker_init_fn=None)
        val = DataLoader(Subset(ds, list(range(B))), batch_size=B, shuffle=False, num_workers=args.dataloader_workers,
                         pin_memory=True, drop_last=False, timeout=0, worker_init_fn=None)
    else:
        train = ds.batch_loader(B, drop_last=True, workers=args.workers, pcm=pcm_fed)
        val = ds.batch_loader(B, indices=list(range(B)), workers=args.workers, pcm=pcm_fed)
    model = build_model(cfg, device)
    # training.ipynb cell 11, as written: torch's own Adam -- ModelTrainer adopts it into the fused launch
    opt = Adam(model.parameters(), weight_decay=1e-5) if args.own_adam else torch.optim.Adam(model.parameters(), weight_decay=1e-5)
    trainer = ModelTrainer(model, torch.nn.MSELoss(), opt, device, model_name='bench')
    epoch_s = []
    inner = trainer._train_epoch

    def timed_epoch(loader):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = inner(loader)
        torch.cuda.synchronize()
        epoch_s.append(time.perf_counter() - t0)
        return out
    trainer._train_epoch = timed_epoch
    # steady state: a timing event in front of every training step; the median distance between consecutive steps of the TIMED
    # epochs is what the loop costs once an epoch's loader has started (ms_per_step spreads each epoch's start-up -- with worker
    # processes: twelve forks and the first batch -- over its batches)
    step_ev = []
    inner_tdb = trainer._train_device_batch

    def marked_tdb(batch):
        if len(epoch_s) >= 1:                  # (epoch 0 holds the eager batches and the capture)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            step_ev.append((len(epoch_s), e))
        return inner_tdb(batch)
    trainer._train_device_batch = marked_tdb
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        os.mkdir('weights')
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                trainer.fit(train, val, 0, 1)                    # epoch 0: two eager batches, the capture, replays
                n_epochs = max(1, args.steps // len(train))
                for k in trainer.host_times:
                    trainer.host_times[k] = 0.0
                t0 = time.perf_counter()
                tl, vl = trainer.fit(train, val, 1, n_epochs)
                torch.cuda.synchronize()
                fit_s = time.perf_counter() - t0
        finally:
            os.chdir(cwd)
    steps = n_epochs * len(train)
    train_s = sum(epoch_s[1:])
    gaps = sorted(a[1].elapsed_time(b[1]) for a, b in zip(step_ev[:-1], step_ev[1:]) if a[0] == b[0])
    steady_ms = gaps[len(gaps) // 2] if gaps else None
    t_frames = 1 + cfg['sr'] * cfg['seconds'] // cfg['hop']
    frames = B * S * t_frames
    print(json.dumps({
        'metric': 'stem-spectrogram-frames/sec (train, via ModelTrainer.fit)', 'value': frames * steps / train_s,
        'unit': 'stem-spectrogram-frames/s', 'n_gpus': 1, 'steps': steps, 'ms_per_step': 1e3 * train_s / steps,
        # median device-side distance between consecutive steps inside the timed epochs (HIP events in front of every step)
        'steady_state_ms_per_step': steady_ms,
        'higher_is_better': True, 'dtype': 'f32', 'data': 'synthetic', 'diagnostic': True,
        'config': {'workload': cfg['workload'] + ' -- through ModelTrainer.fit(%s) from in-memory songs (%s)' % (
                       'torch DataLoader(num_workers=%d, pin_memory=True)' % args.dataloader_workers if args.dataloader_workers
                       else 'Dataset.batch_loader(8)', 'float32' if args.float_pcm else '16-bit PCM'),
                   'dataloader_workers': args.dataloader_workers, 'epoch_repeat': args.epoch_repeat,
                   'epoch_start_ms_per_epoch': 1e3 * trainer.host_times.get('epoch_start', 0.0) / n_epochs,
                   'fork_guard': {k: v for k, v in _staging._fork_guard.items() if k != 'wanted'},
                   'sync_per_step': 'every batch\'s loss is read on the host and logged, one batch late (ModelTrainer._run)',
                   'decode_threads': args.workers, 'pcm': 'float32' if args.float_pcm else 'int16 (16-bit PCM)',
                   'optimizer_passed': 'deep_audio_mixer_amd.optim.Adam' if args.own_adam else 'torch.optim.Adam (adopted by ModelTrainer)',
                   'loader': 'torch.utils.data.DataLoader worker processes -> HostPcmBatch (page-locked) -> upload on a copy stream -> front-end inside the captured step' if args.dataloader_workers
                   else 'PcmBatch (front-end inside the captured step)' if pcm_fed else 'features (front-end per batch on the copy stream)',
                   'graph_steps': trainer.graph_steps, 'epoch_s': epoch_s[1:],
                   'host_ms_per_step': {k: 1e3 * v / steps for k, v in trainer.host_times.items()},
                   'eager_steps': trainer.eager_steps, 'epochs_timed': n_epochs, 'batches_per_epoch': len(train),
                   'ms_per_step_incl_validation_and_checkpoint': 1e3 * fit_s / steps, 'final_train_loss': tl[-1]}}), flush=True)


def run_ingest(name, cfg, args):
    """Ingest rate: 16-bit stereo WAV files on tmpfs -> iter_batches (decode threads read the file's own
    int16 samples straight into page-locked staging -> H2D on a copy stream -> ONE front-end launch per batch) at the
    config's clip shape, against what the training step consumes.  Also the float32-staging variant (host conversion, twice
    the PCIe bytes) for comparison."""
    import shutil
    import tempfile
    import wave
    import numpy as np
    import torch
    device = _setup_single()
    from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
    S, B, sr = cfg['n_stems'], cfg['batch'], cfg['sr']
    n = sr * cfg['seconds']
    n_songs, chunks = 4, 24                                      # 96 clips = 12 batches of 8
    root = tempfile.mkdtemp(dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    tracklist = ['bass', 'drums', 'vocals', 'other', 'gtr', 'keys', 'perc', 'fx'][:S] + ['mix']
    rng = np.random.default_rng(3)
    try:
        for j in range(n_songs):
            name_j = 'Song%d' % j
            d = os.path.join(root, name_j, name_j + '_STEMS_JOINED')
            os.makedirs(d)
            for t in tracklist:
                path = (os.path.join(root, name_j, name_j + '_MIX.wav') if t == 'mix'
                        else os.path.join(d, '%s_STEM_%s.wav' % (name_j, t.upper())))
                with wave.open(path, 'wb') as w:
                    w.setnchannels(CHANNELS), w.setsampwidth(2), w.setframerate(sr)
                    w.writeframes(rng.integers(-3000, 3000, (chunks * n, CHANNELS), dtype=np.int16).tobytes())
        t_frames = 1 + n // cfg['hop']
        frames = B * S * t_frames
        out = {}
        # two rounds of both legs, alternating (a single 12-batch pass per leg measured mostly which leg ran second: thread
        # start-up and page-cache state differ by 2 x from run to run); each leg: one warm pass, then >= 4 passes timed; the
        # faster round of a leg is reported, both are kept
        passes = max(4, args.steps // 12)
        for rnd in range(2):
            for label, force_f32 in (('int16_staging', False), ('float32_staging', True)):
                ds = MultitrackAudioDataset(root, chunk_length=cfg['seconds'], sr=sr, tracklist=tracklist, seed=1)
                if force_f32:
                    ds.staging_format = lambda: (np.dtype(np.float32), CHANNELS)
                for _ in ds.iter_batches(B, workers=args.workers, drop_last=True):    # warm pass: page cache, tables, pinned
                    pass
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                k = 0
                for _ in range(passes):
                    for x, gt in ds.iter_batches(B, workers=args.workers, drop_last=True):
                        k += 1
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                bytes_h2d = B * (S + 1) * n * CHANNELS * (4 if force_f32 else 2)
                rec = {'frames_per_s': frames * k / dt, 'ms_per_batch': 1e3 * dt / k, 'batches': k,
                       'h2d_bytes_per_batch': bytes_h2d, 'h2d_GBps': bytes_h2d * k / dt / 1e9}
                prev = out.get(label)
                rounds = (prev['ms_per_batch_rounds'] if prev else []) + [rec['ms_per_batch']]
                if prev is None or rec['ms_per_batch'] < prev['ms_per_batch']:
                    out[label] = rec
                out[label]['ms_per_batch_rounds'] = rounds
        print(json.dumps({
            'metric': 'stem-spectrogram-frames/sec (ingest: WAV -> features)', 'value': out['int16_staging']['frames_per_s'],
            'unit': 'stem-spectrogram-frames/s', 'n_gpus': 1, 'higher_is_better': True, 'data': 'synthetic 16-bit stereo WAV on tmpfs',
            'diagnostic': True,
            'config': {'workload': 'iter_batches(%d) at the clip shape of %s' % (B, name), 'decode_threads': args.workers,
                       'host_cores': host_cores()[0], **out}}), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


def ddp_breakdown(step, device, iters=10):
    """Where a multi-rank step spends its time: every phase bracketed by device synchronisation (diagnostic only)."""
    import torch
    if step._graphs is None or len(step._graphs) != 3:
        return None
    g, opt = step._graphs, step.opt
    acc = {'graph_a1_ms': 0.0, 'allreduce_bucket1_ms': 0.0, 'graph_a2_ms': 0.0, 'allreduce_bucket0_ms': 0.0, 'graph_b_ms': 0.0}

    def phase(key, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        acc[key] += (time.perf_counter() - t0) * 1e3 / iters
    for _ in range(iters):
        phase('graph_a1_ms', g[0].replay)
        phase('allreduce_bucket1_ms', lambda: opt.all_reduce_grads(1))
        phase('graph_a2_ms', g[1].replay)
        phase('allreduce_bucket0_ms', lambda: opt.all_reduce_grads(0))
        phase('graph_b_ms', g[2].replay)
    acc['bucket_bytes'] = [int(opt.bucket_view(b).numel() * 4) for b in range(opt.n_buckets)]
    return acc


def run_inference(name, cfg, args):
    """C5: one "step" = one whole song through the captured graph (PCM resident in HBM): strided front-end over 59 x 8
    chunk tracks -> ResNet18 (eval) on the chunk batch -> gains -> Savitzky-Golay -> fused mixdown + peak normalise."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    import deep_audio_mixer_amd  # noqa: F401
    from deep_audio_mixer_amd import build, inference_utils
    build.build_lib()
    S, sr = cfg['n_stems'], cfg['sr']
    n = sr * cfg['song_seconds']
    chunk = sr * cfg['seconds']
    model = build_model(cfg, device, train=False)
    mixer = inference_utils.SongMixer(model, S, CHANNELS, n, torch.float32, chunk, 'master', True, torch.float32)
    g = torch.Generator(device=device).manual_seed(1234)
    mixer.pcm.copy_(0.1 * torch.randn(mixer.pcm.shape, generator=g, device=device))
    for _ in range(max(1, args.warmup)):
        mixer.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mixer.launch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t_frames = 1 + chunk // cfg['hop']
    frames = mixer.n_proc * S * t_frames
    result = {
        'metric': 'stem-spectrogram-frames/sec (inference)', 'value': frames * args.steps / dt,
        'unit': 'stem-spectrogram-frames/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': cfg['workload'], 'chunks': mixer.n_proc, 'hip_graph': mixer.graph is not None,
                   'savgol_window': mixer.window},
    }
    # PCIe-inclusive: host arrays in, normalised master out (page-locked double-buffered staging both ways)
    rng = np.random.default_rng(0)
    tracks = [(0.1 * rng.standard_normal((CHANNELS, n))).astype(np.float32) for _ in range(S)]
    mixer.run(tracks)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        mixer.run(tracks)
    dth = (time.perf_counter() - t0) / reps
    result['pcie_inclusive'] = {'ms_per_song': 1e3 * dth, 'value': frames / dth,
                                'h2d_bytes': float(S * CHANNELS * n * 4), 'd2h_bytes': float(CHANNELS * n * 4)}
    if not args.no_roofline:
        result['roofline'] = roofline_object(name, cfg, device, t_frames)
        result['roofline_extra'] = {'whole_step_tflops': cfg['gflop_fwd'] * mixer.n_proc * 1e9 * args.steps / dt / 1e12}
    print(json.dumps(result), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='C3')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-overlap', action='store_true', help='N > 1: one un-overlapped all-reduce of the whole flat buffer')
    ap.add_argument('--reduce-mode', choices=('auto', 'overlapped', 'inline'), default='auto',
                    help='N > 1: all-reduces beside backward on RCCL\'s stream, on the training stream, or whichever is faster here')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-host-stream', action='store_true')
    ap.add_argument('--bind-per-step', action='store_true', help='A/B: re-point the front-end per step instead of the device-side rotation')
    ap.add_argument('--no-copy-mark', action='store_true', help='A/B: the streamed leg\'s uploads start at step ends (no step mark)')
    ap.add_argument('--host-clips', type=int, default=N_HOST_CLIPS)
    ap.add_argument('--breakdown', action='store_true', help='N > 1: add per-phase timings of the step (diagnostic)')
    ap.add_argument('--repeat', type=int, default=5, help='extra timed regions of --steps steps: median / min / max in the line')
    ap.add_argument('--via-trainer', action='store_true',
                    help='diagnostic line (never the headline): ms per step of ModelTrainer.fit fed by Dataset.batch_loader')
    ap.add_argument('--ingest', action='store_true',
                    help='diagnostic line: frames/s of WAV files -> decode threads -> pinned -> H2D -> STFT (iter_batches)')
    ap.add_argument('--workers', type=int, default=8, help='--ingest / --via-trainer: decode threads')
    ap.add_argument('--float-pcm', action='store_true', help='--via-trainer: float32 in-memory songs instead of 16-bit PCM')
    ap.add_argument('--pcm-loader', action='store_true', help='--via-trainer: loader yields uploaded PCM, front-end inside the captured step')
    ap.add_argument('--dataloader-workers', type=int, default=0,
                    help='--via-trainer: torch DataLoader(num_workers=N, pin_memory=True) over the Dataset (training.ipynb cell 6 as written)')
    ap.add_argument('--epoch-repeat', type=int, default=1, help='--dataloader-workers: the clip set is walked R times per epoch')
    ap.add_argument('--own-adam', action='store_true', help='--via-trainer: pass optim.Adam instead of torch.optim.Adam')
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    world_env = os.environ.get('WORLD_SIZE')
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))            # before any GPU call in this process
    world = int(world_env) if world_env is not None else 1
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    cfg = CONFIGS[args.config]
    if args.via_trainer or args.ingest:
        if world != 1 or args.config == 'C5':
            raise SystemExit('--via-trainer / --ingest: one GPU, a training config')
        (run_via_trainer if args.via_trainer else run_ingest)(args.config, cfg, args)
        return
    if args.config == 'C5':
        if world != 1:
            raise SystemExit('C5 (one song) runs on one GPU; songs shard over ranks as independent replicas')
        run_inference(args.config, cfg, args)
    else:
        run_train(args.config, cfg, args)


if __name__ == '__main__':
    main()
