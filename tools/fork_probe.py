"""GPU stall caused by ONE os.fork() of a GPU-initialised process, against device memory held, and whether
madvise(MADV_DONTFORK) on the GPU driver's mappings removes it.  (diagnostic)
usage: fork_probe.py <device_GB> [dontfork]"""
import ctypes
import os
import sys
import threading
import time

import torch

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dontfork = len(sys.argv) > 2 and sys.argv[2] == 'dontfork'
libc = ctypes.CDLL('libc.so.6', use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
MADV_DONTFORK = 10


def gpu_maps():
    out = []
    for line in open('/proc/self/maps'):
        f = line.split()
        if len(f) >= 6 and (f[5].startswith('/dev/dri') or f[5].startswith('/dev/kfd')):
            lo, hi = (int(x, 16) for x in f[0].split('-'))
            out.append((lo, hi, f[5]))
    return out


torch.cuda.init()
devs = [torch.empty(64 << 20, dtype=torch.uint8, device='cuda') for _ in range(int(gb * 16))]
x = torch.zeros(1 << 20, device='cuda')
torch.cuda.synchronize()
maps = gpu_maps()
print('device GB %.1f: %d driver mappings, %.2f GB of address space; vmas total %d' % (
    gb, len(maps), sum(h - l for l, h, _ in maps) / 2**30, sum(1 for _ in open('/proc/self/maps'))))
if dontfork:
    bad = 0
    for lo, hi, _ in maps:
        if libc.madvise(lo, hi - lo, MADV_DONTFORK) != 0:
            bad += 1
    print('madvise(DONTFORK) failed on %d of %d' % (bad, len(maps)))
stalls, stop = [], threading.Event()
T0 = time.perf_counter()


def heartbeat():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop.is_set():
            t0 = time.perf_counter()
            x.add_(1)
            s.synchronize()
            dt = time.perf_counter() - t0
            if dt > 0.005:
                stalls.append((round(t0 - T0, 3), round(1e3 * dt, 1)))


th = threading.Thread(target=heartbeat, daemon=True)
th.start()
time.sleep(0.2)
forks = []
for k in range(4):
    t0 = time.perf_counter()
    pid = os.fork()
    if pid == 0:
        os._exit(0)
    t1 = time.perf_counter()
    os.waitpid(pid, 0)
    forks.append((round(t0 - T0, 3), round(1e3 * (t1 - t0), 1)))
    time.sleep(0.25)
stop.set()
th.join()
print('forks (at s, fork() ms):', forks)
print('GPU stalls > 5 ms (at s, ms):', stalls)
