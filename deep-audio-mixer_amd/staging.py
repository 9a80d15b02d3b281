"""Host <-> HBM staging for the ingest side of the path.

The reference hands pageable numpy arrays to ``tensor.to(device)`` (model_trainer.py:34, inference_utils.py:123): the
driver then bounces them through its own small staging buffer, single-threaded, at a few GB/s.  Here the bounce is
explicit: two page-locked buffers, the host copy into buffer k+1 (a few numpy-copy threads) runs while the DMA
engine moves buffer k on a dedicated copy stream, and the compute stream only waits on an event.  The same pipe in
reverse brings results back.  PyTorch supplies the page-locked memory, the streams and the events; nothing is
computed here.
"""
import ctypes
import os
import threading

import numpy as np
import torch
import torch.utils.data

PIECE_BYTES = 32 << 20

# Held by whoever captures a hipGraph (engine.TrainStep.capture, inference_utils.SongMixer) and, per batch, by background threads
# around their GPU calls (MultitrackAudioDataset.iter_batches' feeder): a capture and a background thread's event waits / H2D copies /
# allocator misses then never overlap.  Captures also run in torch's "thread_local" error mode, so this is the second fence.
capture_guard = threading.RLock()


_copy_pool = None
HOST_COPY_THREADS = 4


def _wait(event):
    """Host-side wait for a recorded event (DMA pieces that take well under a millisecond): poll, do not sleep."""
    while not event.query():
        pass


def _host_copy(dst, src):
    """dst, src: equally long contiguous uint8 CPU tensors.  A few plain threads doing numpy copies (the GIL is released
    inside), NOT torch's copy_: that fans out over every OpenMP thread torch was given -- 128 on a box whose cgroup grants
    a 16-core share -- and the spinning team burns the process's CPU quota: the kernel then throttles the whole process
    for the rest of its 100 ms period (measured as one or two 70-95 ms stalls per song, landing in whatever host call
    came next; tools/c5_phase_probe.py)."""
    global _copy_pool
    d, s = dst.numpy(), src.numpy()
    n = d.shape[0]
    if n < (4 << 20):
        np.copyto(d, s)
        return
    if _copy_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _copy_pool = ThreadPoolExecutor(max_workers=HOST_COPY_THREADS)
    step = -(-n // HOST_COPY_THREADS)
    list(_copy_pool.map(lambda lo: np.copyto(d[lo:lo + step], s[lo:lo + step]), range(0, n, step)))


class PinnedPipe:
    """Double-buffered page-locked staging between host arrays and device tensors on a private copy stream."""

    def __init__(self, device, piece_bytes=PIECE_BYTES, n_buffers=2):
        self.device = torch.device(device)
        self.piece_bytes = int(piece_bytes)
        self.bufs = [torch.empty(self.piece_bytes, dtype=torch.uint8, pin_memory=True) for _ in range(n_buffers)]
        self.free = [torch.cuda.Event() for _ in range(n_buffers)]      # recorded when the DMA out of / into buffer i is done
        self.stream = torch.cuda.Stream(device=self.device)
        self._turn = 0

    def _next(self):
        i = self._turn
        self._turn = (i + 1) % len(self.bufs)
        _wait(self.free[i])                     # the previous transfer through this buffer has finished
        return i

    def upload(self, dst, src):
        """dst: contiguous device tensor; src: numpy array (or CPU tensor) of the same dtype and element count, any
        pageable memory.  Returns after the last piece has been ENQUEUED; the current stream is made to wait for it."""
        src_t = torch.from_numpy(np.ascontiguousarray(src)) if isinstance(src, np.ndarray) else src.contiguous()
        if src_t.dtype != dst.dtype or src_t.numel() != dst.numel() or not dst.is_contiguous():
            raise ValueError('upload: dtype / size mismatch or non-contiguous destination')
        s8, d8 = src_t.view(-1).view(torch.uint8), dst.view(-1).view(torch.uint8)
        n = s8.numel()
        # the copy stream must not overtake earlier work on the destination
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        for lo in range(0, n, self.piece_bytes):
            hi = min(lo + self.piece_bytes, n)
            i = self._next()
            _host_copy(self.bufs[i][:hi - lo], s8[lo:hi])               # pageable -> page-locked
            with torch.cuda.stream(self.stream):
                d8[lo:hi].copy_(self.bufs[i][:hi - lo], non_blocking=True)
                self.free[i].record(self.stream)
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return dst

    def download(self, src, out=None):
        """src: contiguous device tensor -> numpy array (a fresh pageable array, or `out`).  Synchronous."""
        if not src.is_contiguous():
            raise ValueError('download: non-contiguous source')
        res = torch.empty(src.shape, dtype=src.dtype) if out is None else (
            torch.from_numpy(out) if isinstance(out, np.ndarray) else out)
        r8, s8 = res.view(-1).view(torch.uint8), src.view(-1).view(torch.uint8)
        n = s8.numel()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        nb = len(self.bufs)
        pieces = [(lo, min(lo + self.piece_bytes, n)) for lo in range(0, n, self.piece_bytes)]
        for ev in self.free:
            _wait(ev)                                                   # nothing else is in flight through the buffers

        def drain(k):                                                   # page-locked -> pageable once piece k has landed
            lo, hi = pieces[k]
            _wait(self.free[k % nb])
            _host_copy(r8[lo:hi], self.bufs[k % nb][:hi - lo])

        for k, (lo, hi) in enumerate(pieces):
            if k >= nb:
                drain(k - nb)                                           # overlaps the DMA of piece k-1
            with torch.cuda.stream(self.stream):
                self.bufs[k % nb][:hi - lo].copy_(s8[lo:hi], non_blocking=True)
                self.free[k % nb].record(self.stream)
        for k in range(max(0, len(pieces) - nb), len(pieces)):
            drain(k)
        return res.numpy() if out is None else out


_pipes = {}


def pipe_for(device):
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _pipes:
        _pipes[key] = PinnedPipe(device)
    return _pipes[key]


class StepMark:
    """A point inside a (captured) training step that a copy stream can wait for: include/dam_hip.h, dam_step_mark_*.
    ``record()`` on the current stream -- inside a capture it becomes an event-record node every replay executes;
    ``wait(stream)`` makes a stream outside the graph wait for the latest record enqueued so far."""

    def __init__(self):
        from . import _lib
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().dam_step_mark_create(ctypes.byref(h)), 'dam_step_mark_create')
        self._h, self.records = h, 0

    def record(self):
        from . import _lib
        _lib.check(_lib.lib().dam_step_mark_record(self._h, _lib.stream()), 'dam_step_mark_record')
        self.records += 1

    def wait(self, stream):
        from . import _lib
        if self.records:
            _lib.check(_lib.lib().dam_step_mark_wait(self._h, stream.cuda_stream), 'dam_step_mark_wait')

    def synchronize(self):
        """The host waits for the latest record enqueued so far (directly or through a graph launch)."""
        from . import _lib
        if self.records:
            _lib.check(_lib.lib().dam_step_mark_synchronize(self._h), 'dam_step_mark_synchronize')

    def __del__(self):
        try:
            from . import _lib
            if self._h and _lib._lib is not None and not torch.cuda._is_in_bad_fork():
                _lib._lib.dam_step_mark_destroy(self._h)
        except Exception:        # noqa: BLE001 -- interpreter shutdown
            pass


class BatchStager:
    """Feeds device batches from a page-locked host dataset [N, ...]: while the consumer works on batch k, batch k+1
    travels on a private copy stream into another of three device buffers (the role the reference gives to
    DataLoader(pin_memory=True) + ``.to(device)``, model_trainer.py:34 -- here without blocking the training stream).

    ``next()`` returns the device tensor of the next batch; the caller enqueues exactly the work that reads it (one step)
    on the current stream before calling ``next()`` again.

    Who waits for whom (profiles/r05_sync_cost_probe.txt): the TRAINING stream waits for the copy stream's event on the
    device -- that direction is free.  The other direction is not: a stream that waits for an event of the training stream
    costs the training stream 0.09 ms per step on this stack, whatever the event's flags, while a HOST wait for the same
    event costs nothing.  So the buffer a copy is about to overwrite is known free on the host: three buffers, and the host
    waits for the step two batches back (``consumed``) -- the device always has the step in between to work on.

    gate (the consuming TrainStep's ``copy_mark``): the host waits for the step enqueued last to reach its mark -- the
    backward pass of the shallow layers -- before it enqueues the next upload: a 1.3 ms copy beside the forward pass slows the
    forward's latency-bound launches (BatchNorm finalize kernels 113 -> 177 us per step, profiles/r05_pcie_trace.txt), beside
    that part of backward it costs a third.  The next step is launched right behind, two milliseconds before the device needs it."""

    N_BUFS = 3

    def __init__(self, host, batch, device, gate=None):
        if not host.is_pinned():
            raise ValueError('BatchStager needs page-locked host memory (torch.empty(..., pin_memory=True))')
        self.host, self.batch, self.device = host, batch, torch.device(device)
        self.n_batches = host.shape[0] // batch
        if self.n_batches < 1:
            raise ValueError('the host dataset holds less than one batch')
        n = self.N_BUFS
        self.bufs = [torch.empty((batch,) + tuple(host.shape[1:]), dtype=host.dtype, device=self.device) for _ in range(n)]
        self.ready = [torch.cuda.Event() for _ in range(n)]
        self.consumed = [torch.cuda.Event() for _ in range(n)]
        self.stream = torch.cuda.Stream(device=self.device)
        self.k, self.gate = 0, gate
        self._issue(0)

    def _issue(self, k):
        b = k % self.N_BUFS
        lo = (k % self.n_batches) * self.batch
        with torch.cuda.stream(self.stream):
            self.bufs[b].copy_(self.host[lo:lo + self.batch], non_blocking=True)
            self.ready[b].record(self.stream)

    def next(self):
        k, cur, n = self.k, torch.cuda.current_stream(self.device), self.N_BUFS
        if k > 0:
            self.consumed[(k - 1) % n].record(cur)          # everything enqueued so far has read batch k-1's buffer
        cur.wait_event(self.ready[k % n])
        if k + 1 >= n:                                      # batch k+1 goes where batch k-2 was: that step must be over
            self.consumed[(k - 2) % n].synchronize()        # (the device still has step k-1 to work on)
        if self.gate is not None and k > 0:
            self.gate.synchronize()                         # ... and step k-1 has reached its backward pass
        self._issue(k + 1)
        self.k = k + 1
        return self.bufs[k % n]


# ---- fork guard (include/dam_hip.h: dam_host_dontfork_pinned) ---------------------------------------------------------------
def dontfork_pinned_host_memory():
    """Marks every page-locked, GPU-mapped host buffer of this process MADV_DONTFORK -> (mappings, bytes) marked.  A no-op
    (0, 0) where there is nothing to protect or nothing may be asked of the GPU runtime: before it is initialised, and in a
    forked child."""
    if not torch.cuda.is_initialized() or torch.cuda._is_in_bad_fork() or torch.utils.data.get_worker_info() is not None:
        return 0, 0
    from . import _lib
    n, b = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(_lib.lib().dam_host_dontfork_pinned(ctypes.byref(n), ctypes.byref(b)), 'dam_host_dontfork_pinned')
    return n.value, b.value


_fork_guard = {'installed': False, 'wanted': [], 'calls': 0, 'mappings': 0, 'bytes': 0}


def _before_fork():
    """os.register_at_fork(before=...): runs in the forking (GPU-owning) process for every os.fork() -- the start of each
    DataLoader worker.  Without it every fork write-protects torch's pinned blocks (user-pointer memory to the GPU driver),
    the driver evicts the process's queues and the device idles for hundreds of milliseconds per fork."""
    g = _fork_guard
    if os.environ.get('DAM_FORK_GUARD', '1') == '0' or not any(w() for w in g['wanted']):
        return
    try:
        n, b = dontfork_pinned_host_memory()
    except Exception:            # noqa: BLE001 -- a fork must never fail because of the guard
        return
    g['calls'] += 1
    g['mappings'], g['bytes'] = n, b


def install_fork_guard(wanted):
    """wanted: a callable -> bool, asked at every fork (data.dataset: "is a MultitrackAudioDataset alive in this process?").
    With ``DAM_FORK_GUARD=0`` in the environment the guard stays off (a program that hands PINNED tensors to forked
    children to read needs that: the children would not inherit them)."""
    g = _fork_guard
    g['wanted'].append(wanted)
    if not g['installed']:
        os.register_at_fork(before=_before_fork)
        g['installed'] = True
