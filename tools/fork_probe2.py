"""What must a forked child DO to stall the parent's GPU?  (diagnostic)
usage: fork_probe2.py <child: sleep|touch|alloc|torchop|exit> """
import os
import sys
import threading
import time

import numpy as np
import torch

what = sys.argv[1] if len(sys.argv) > 1 else 'sleep'
heap = np.ones(1 << 27, dtype=np.int16)          # 256 MB of parent heap
t_host = torch.ones(1 << 26)                      # 256 MB torch CPU tensor
torch.cuda.init()
devs = [torch.empty(64 << 20, dtype=torch.uint8, device='cuda') for _ in range(16)]
x = torch.zeros(1 << 20, device='cuda')
torch.cuda.synchronize()
stalls, stop = [], threading.Event()
T0 = time.perf_counter()


def heartbeat():
    s = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        while not stop.is_set():
            t0 = time.perf_counter()
            e0.record()
            x.add_(1)
            e1.record()
            t1 = time.perf_counter()
            if os.environ.get('PROBE_POLL'):
                while not e1.query():
                    pass
            else:
                s.synchronize()
            t2 = time.perf_counter()
            if t2 - t0 > 0.01:
                stalls.append((round(t0 - T0, 3), 'enqueue %.1f ms, sync %.1f ms, gpu %.2f ms' % (1e3 * (t1 - t0), 1e3 * (t2 - t1), e0.elapsed_time(e1))))
            time.sleep(0.0005)


pinned = [torch.empty(48 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
if os.environ.get('PROBE_GUARD'):
    sys.path.insert(0, '.')
    import deep_audio_mixer_amd  # noqa: F401
    from deep_audio_mixer_amd import staging
    t0 = time.perf_counter()
    print('guard marked (mappings, bytes):', staging.dontfork_pinned_host_memory(), '%.2f ms' % (1e3 * (time.perf_counter() - t0)))
th = threading.Thread(target=heartbeat, daemon=True)
th.start()
time.sleep(0.3)
for k in range(3):
    t0 = time.perf_counter()
    pid = os.fork()
    if pid == 0:
        if what == 'sleep':
            time.sleep(0.3)
        elif what == 'touch':                       # COW faults in the child on inherited heap
            heap[::2048] = 2
            time.sleep(0.2)
        elif what == 'read':
            s = int(heap[::2048].sum())
            time.sleep(0.2)
        elif what == 'alloc':                       # fresh memory only
            a = np.ones(1 << 26, dtype=np.int16)
            time.sleep(0.2)
        elif what == 'torchop':
            torch.set_num_threads(1)
            b = t_host[:1 << 22].clone()
            time.sleep(0.2)
        elif what == 'shm':
            torch.set_num_threads(1)
            b = t_host[:1 << 22].clone().share_memory_()
            time.sleep(0.2)
        os._exit(0)
    t1 = time.perf_counter()
    print('fork %d at %.3f s: fork() %.1f ms' % (k, t0 - T0, 1e3 * (t1 - t0)), flush=True)
    os.waitpid(pid, 0)
    time.sleep(0.3)
stop.set()
th.join()
print('stalls > 10 ms:', *stalls, sep='\n   ')
