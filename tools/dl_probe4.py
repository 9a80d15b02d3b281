"""Which DataLoader operation stalls the GPU: forking the workers, their work, or their shutdown?  (diagnostic)"""
import sys
import threading
import time

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

dev_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
pin = len(sys.argv) > 2 and sys.argv[2] == 'pin'


class DS(Dataset):
    def __init__(self):
        self.a = np.zeros(1 << 28, dtype=np.int16)

    def __len__(self):
        return 96

    def __getitem__(self, i):
        return torch.from_numpy(self.a[i * 1000:i * 1000 + 2381400].copy())


if __name__ == '__main__':
    ds = DS()
    torch.cuda.init()
    devs = [torch.empty(64 << 20, dtype=torch.uint8, device='cuda') for _ in range(int(16 * dev_gb))]
    x = torch.zeros(1 << 20, device='cuda')
    torch.cuda.synchronize()
    T0 = time.perf_counter()
    stalls, stop = [], threading.Event()

    def heartbeat():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            while not stop.is_set():
                t0 = time.perf_counter()
                x.add_(1)
                s.synchronize()
                dt = time.perf_counter() - t0
                if dt > 0.01:
                    stalls.append((round(t0 - T0, 3), round(1e3 * dt, 1)))
                time.sleep(0.0005)
    th = threading.Thread(target=heartbeat, daemon=True)
    th.start()

    def mark(what):
        print('%8.3f s  %s' % (time.perf_counter() - T0, what), flush=True)
    dl = DataLoader(ds, batch_size=8, shuffle=False, num_workers=6, pin_memory=pin)
    for rnd in range(2):
        time.sleep(0.5)
        mark('iter() begins')
        it = iter(dl)
        mark('iter() done (workers forked)')
        time.sleep(0.5)
        mark('consume begins')
        n = sum(1 for _ in it)
        mark('consume done (%d batches)' % n)
        time.sleep(0.5)
        mark('shutdown begins')
        it._shutdown_workers()
        del it
        mark('shutdown done')
    time.sleep(0.5)
    stop.set()
    th.join()
    print('GPU stalls > 10 ms (at s, ms):', stalls)
