#!/usr/bin/env python3
"""Diagnostic: host <-> device transfer rates on this box (pageable vs page-locked, host memcpy into staging)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
dev = torch.device('cuda', 0)
n = 64 << 20       # bytes
src = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)
src_t = torch.from_numpy(src)
pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
pin2 = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d = torch.empty(n, dtype=torch.uint8, device=dev)
out = torch.empty(n, dtype=torch.uint8)
print('torch threads', torch.get_num_threads(), 'cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
def t(fn, reps=5, sync=True):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    if sync: torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
def rate(name, dt): print('%-46s %7.2f ms  %6.1f GB/s' % (name, dt * 1e3, n / dt / 1e9))
rate('pageable -> pinned  (torch copy_)', t(lambda: pin.copy_(src_t)))
rate('pageable -> pinned  (numpy copyto)', t(lambda: np.copyto(pin.numpy(), src)))
rate('pinned -> pinned    (torch copy_)', t(lambda: pin2.copy_(pin)))
rate('pageable -> pageable (torch copy_)', t(lambda: out.copy_(src_t)))
rate('pinned -> device    (copy_ non_blocking)', t(lambda: d.copy_(pin, non_blocking=True)))
rate('pageable -> device  (copy_)', t(lambda: d.copy_(src_t)))
rate('device -> pinned    (copy_ non_blocking)', t(lambda: pin.copy_(d, non_blocking=True)))
rate('device -> pageable  (copy_)', t(lambda: out.copy_(d)))
rate('pinned -> pageable  (torch copy_)', t(lambda: out.copy_(pin)))
rate('pinned -> pageable  (numpy copyto)', t(lambda: np.copyto(out.numpy(), pin.numpy())))
for th in (1, 4, 8, 16):
    torch.set_num_threads(th)
    rate('pageable -> pinned  (torch copy_, %2d threads)' % th, t(lambda: pin.copy_(src_t)))
import deep_audio_mixer_amd
from deep_audio_mixer_amd.staging import PinnedPipe
torch.set_num_threads(16)
pipe = PinnedPipe(dev)
big = np.random.default_rng(1).standard_normal(127_000_000 // 4).astype(np.float32)
dbig = torch.empty(big.shape, dtype=torch.float32, device=dev)
nb = big.nbytes
t0 = time.perf_counter(); pipe.upload(dbig, big); torch.cuda.synchronize(); dt = time.perf_counter() - t0
t0 = time.perf_counter(); pipe.upload(dbig, big); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('PinnedPipe.upload   %.1f MB: %.2f ms  %.1f GB/s' % (nb / 1e6, dt * 1e3, nb / dt / 1e9))
t0 = time.perf_counter(); o = pipe.download(dbig); dt = time.perf_counter() - t0
print('PinnedPipe.download %.1f MB: %.2f ms  %.1f GB/s' % (nb / 1e6, dt * 1e3, nb / dt / 1e9))
t0 = time.perf_counter(); dd = torch.from_numpy(big).to(dev); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('plain .to(device)   %.1f MB: %.2f ms  %.1f GB/s' % (nb / 1e6, dt * 1e3, nb / dt / 1e9))
t0 = time.perf_counter(); hh = dbig.cpu(); dt = time.perf_counter() - t0
print('plain .cpu()        %.1f MB: %.2f ms  %.1f GB/s' % (nb / 1e6, dt * 1e3, nb / dt / 1e9))
