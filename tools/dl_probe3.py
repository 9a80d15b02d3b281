"""What stalls the GPU when a DataLoader forks its workers from a GPU-initialised process?  (diagnostic)
usage: dl_probe3.py [fork|forkserver|persistent] [pinned_mb] [host_gb]"""
import sys
import threading
import time

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

mode = sys.argv[1] if len(sys.argv) > 1 else 'fork'
pinned_mb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
host_gb = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0


class DS(Dataset):
    def __init__(self):
        self.a = np.zeros(int(host_gb * (1 << 30)) // 2, dtype=np.int16)     # stands for the in-memory songs
        self.a[::2048] = 1

    def __len__(self):
        return 96

    def __getitem__(self, i):
        return torch.from_numpy(self.a[i * 1000:i * 1000 + 2381400].copy())


if __name__ == '__main__':
    ds = DS()
    torch.cuda.init()
    pins = [torch.empty(32 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(pinned_mb // 32)]
    devs = [torch.empty(64 << 20, dtype=torch.uint8, device='cuda') for _ in range(16)]
    torch.cuda.synchronize()
    T0 = time.perf_counter()
    beats, stop = [], threading.Event()

    def heartbeat():
        s = torch.cuda.Stream()
        x = torch.zeros(1024, device='cuda')
        with torch.cuda.stream(s):
            while not stop.is_set():
                t0 = time.perf_counter()
                x.add_(1)
                s.synchronize()
                t1 = time.perf_counter()
                if t1 - t0 > 0.02:
                    beats.append((t0 - T0, t1 - t0))
                time.sleep(0.001)
    th = threading.Thread(target=heartbeat, daemon=True)
    th.start()
    kw = {}
    if mode == 'forkserver':
        kw['multiprocessing_context'] = 'forkserver'
    if mode == 'persistent':
        kw['persistent_workers'] = True
    dl = DataLoader(ds, batch_size=8, shuffle=False, num_workers=6, pin_memory=True, **kw)
    for epoch in range(3):
        t0 = time.perf_counter()
        it = iter(dl)
        t1 = time.perf_counter()
        n = 0
        for b in it:
            if n == 0:
                t2 = time.perf_counter()
            n += 1
        t3 = time.perf_counter()
        print('epoch %d: starts at %.3f s; iter() %.1f ms, first batch after %.1f ms, epoch %.1f ms' % (
            epoch, t0 - T0, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t0)), flush=True)
        time.sleep(0.3)
    stop.set()
    th.join()
    print('GPU heartbeat stalls > 20 ms: (at s, for ms)', [(round(a, 3), round(1e3 * d, 1)) for a, d in beats])
