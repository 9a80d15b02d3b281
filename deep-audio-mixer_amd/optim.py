"""Adam(+L2) over one flat parameter buffer -- the optimizer of the reference's training loop
(training.ipynb cell 11: torch.optim.Adam(model.parameters(), weight_decay=1e-5), stepped at
model_trainer.py:37) as a single HIP launch, and the gradient buckets of the data-parallel path.

``Adam(params, lr, betas, eps, weight_decay)`` keeps torch.optim.Adam's constructor and semantics
(L2 folded into the gradient, bias correction, no amsgrad).  On construction the parameters are
re-pointed at views of one contiguous buffer; ``step()`` gathers the gradients into one flat bucket,
optionally all-reduces it over the process group (RCCL) and runs ``dam_adam_l2_step_f32``.  The step
counter and the hyper-parameters live on the device: the whole step is hipGraph-capturable and a captured
graph still follows ``param_groups`` edits (LR schedulers) -- ``sync_hyper()`` refreshes six floats.

``state_dict()`` / ``load_state_dict()`` speak torch.optim.Adam's format (per-parameter ``step`` /
``exp_avg`` / ``exp_avg_sq``), so optimizer checkpoints move between the two.

Data parallel: the flat gradient buffer is cut into contiguous BUCKETS at parameter boundaries
(``set_bucket_boundaries``); the step engine all-reduces a bucket as soon as the backward pass has produced
it, overlapping the transfer of the deep layers' gradients (85 % of ResNet18's bytes, ready after the first
tenth of backward) with the rest of backward (engine.TrainStep).
"""
import torch

from . import ops


class _HostStagedAllReduce:
    """All-reduce of a CUDA bucket over a HOST transport (gloo: CPU rehearsals of the multi-rank step, several ranks
    sharing one GPU).  ProcessGroupGloo accepts CUDA tensors but its internal staging stalls for seconds when the
    producing stream is still busy (measured: 2.6 s per step for a 12.6 MB bucket, against 3 ms once the stream has
    been synchronised first -- gpurun_out/r2_ddp2.log), so the staging is done here: D2H into
    page-locked memory behind an event, the collective on the CPU tensor, H2D back.  RCCL never takes this path."""

    def __init__(self, t, group, host):
        self.t, self.group, self.host = t, group, host
        self.host.copy_(t, non_blocking=True)
        self.ev = torch.cuda.Event()
        self.ev.record()

    def wait(self):
        while not self.ev.query():
            pass
        torch.distributed.all_reduce(self.host, op=torch.distributed.ReduceOp.SUM, group=self.group)
        self.t.copy_(self.host, non_blocking=True)
        return True


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False,
                 process_group=None, world_size=1):
        if amsgrad:
            raise ValueError('amsgrad is not implemented (the reference does not use it)')
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError('one parameter group expected (the reference passes model.parameters())')
        self._params = [p for p in self.param_groups[0]['params'] if p.requires_grad]
        dev = self._params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('move the model to the GPU before constructing the optimizer (as the reference does); '
                               'there is no CPU fallback')
        n = sum(p.numel() for p in self._params)
        self._flat = torch.empty(n, dtype=torch.float32, device=dev)
        self._offsets, off = [], 0
        for p in self._params:            # parameters become views of the flat buffer (same Parameter objects)
            k = p.numel()
            self._flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self._flat[off:off + k].view(p.shape)
            self._offsets.append(off)
            off += k
        self._offsets.append(off)
        self._grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self._exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self._exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._step = torch.zeros(1, dtype=torch.int64, device=dev)
        self._derived = torch.zeros(2, dtype=torch.float32, device=dev)
        self._hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self._hyper_host = None
        self._host_grad = None
        self.slots_bound = False
        self.process_group, self.world_size = process_group, world_size
        self._bucket_params = [(0, len(self._params))]           # parameter index ranges, in flat-buffer order
        self.sync_hyper()

    # ---- hyper-parameters on the device
    def _hyper_tuple(self):
        g = self.param_groups[0]
        b1, b2 = float(g['betas'][0]), float(g['betas'][1])
        return (float(g['lr']), b1, b2, float(g['eps']), float(g['weight_decay']), 1.0 / self.world_size, 1.0 - b1, 1.0 - b2)

    def sync_hyper(self):
        """Uploads {lr, betas, eps, weight_decay, 1/world} if param_groups changed since the last call (six floats;
        call before replaying a captured step -- engine.TrainStep does)."""
        h = self._hyper_tuple()
        if h != self._hyper_host:
            self._hyper.copy_(torch.tensor(h, dtype=torch.float32), non_blocking=False)
            self._hyper_host = h

    # ---- gradient buckets
    @property
    def flat_grad(self):
        return self._grad

    def set_bucket_boundaries(self, first_params):
        """Cuts the flat gradient buffer in front of each given parameter (parameter objects that start a new bucket)."""
        ids = {id(p): i for i, p in enumerate(self._params)}
        cuts = sorted({ids[id(p)] for p in first_params} - {0})
        edges = [0] + cuts + [len(self._params)]
        self._bucket_params = [(a, b) for a, b in zip(edges[:-1], edges[1:])]

    @property
    def n_buckets(self):
        return len(self._bucket_params)

    def bucket_params(self, b):
        lo, hi = self._bucket_params[b]
        return self._params[lo:hi]

    def bucket_view(self, b):
        lo, hi = self._bucket_params[b]
        return self._grad[self._offsets[lo]:self._offsets[hi]]

    def bind_grad_slots(self):
        """Hands every parameter its slice of the flat gradient buffer (``p._dam_grad``): the backward kernels of
        layers.py then write gradients straight into the bucket -- no .grad tensors, no copies, no gather launch.
        Gradients OVERWRITE (one backward per step); ``p.grad`` stays None.  Undo with ``unbind_grad_slots()``."""
        for i, p in enumerate(self._params):
            p._dam_grad = self._grad[self._offsets[i]:self._offsets[i + 1]].view(p.shape)
        self.slots_bound = True

    def unbind_grad_slots(self):
        for p in self._params:
            if hasattr(p, '_dam_grad'):
                del p._dam_grad
        self.slots_bound = False

    def zero_grad(self, set_to_none=True):
        """torch's zero_grad, plus: reductions a failed backward left recorded are dropped (ops.wgrad_abandon)."""
        if self._flat.is_cuda:
            ops.wgrad_abandon(self._flat.device)
        return super().zero_grad(set_to_none=set_to_none)

    def gather_grads(self, bucket=None, grads=None):
        """One launch: every p.grad (or the given list of gradient tensors, bucket order) -> its slice of the flat
        buffer (missing grads count as zero).  bucket=None: all parameters."""
        if self._flat.is_cuda:
            ops.side_stream_join(self._flat.device)
            ops.wgrad_flush(self._flat.device)            # slab reductions of the in-place weight gradients
        lo, hi = (0, len(self._params)) if bucket is None else self._bucket_params[bucket]
        views, srcs = [], []
        for i in range(lo, hi):
            p = self._params[i]
            g = p.grad if grads is None else grads[i - lo]
            v = self._grad[self._offsets[i]:self._offsets[i + 1]]
            if g is None and self.slots_bound and hasattr(p, '_dam_grad'):
                continue                                  # the backward kernels wrote this slice themselves
            if g is None:
                v.zero_()
            else:
                views.append(v.view(p.shape))
                srcs.append(g)
        if views:
            torch._foreach_copy_(views, srcs)
        return self._grad

    def all_reduce_grads(self, bucket=None, async_op=False):
        """Sum one bucket (default: the whole flat buffer) over the data-parallel group (RCCL all-reduce over xGMI);
        the 1/world average is folded into the Adam launch.  Returns the work handle when async_op."""
        if self._flat.is_cuda:
            ops.side_stream_join(self._flat.device)
            ops.wgrad_flush(self._flat.device)
        if self.world_size > 1:
            t = self._grad if bucket is None else self.bucket_view(bucket)
            if torch.distributed.get_backend(self.process_group) == 'gloo':
                if self._host_grad is None:
                    self._host_grad = torch.empty(self._grad.shape, dtype=torch.float32, pin_memory=True)
                lo = t.storage_offset() - self._grad.storage_offset()
                work = _HostStagedAllReduce(t, self.process_group, self._host_grad[lo:lo + t.numel()])
                if async_op:
                    return work
                work.wait()
                return None
            return torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=self.process_group,
                                                async_op=async_op)
        return None

    def launch_update(self):
        """The Adam launch alone (gradients already in the flat buffer, already reduced)."""
        if self._flat.is_cuda:
            ops.side_stream_join(self._flat.device)
            ops.wgrad_flush(self._flat.device)
        ops.params_changed()
        h = self._hyper_tuple()
        ops.adam_l2_step(self._flat, self._grad, self._exp_avg, self._exp_avg_sq, self._step, self._derived, h[0], h[1],
                         h[2], h[3], h[4], h[5], hyper=self._hyper)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.sync_hyper()
        self.gather_grads()
        self.all_reduce_grads()
        self.launch_update()
        return loss

    # ---- torch.optim.Adam-format checkpoints
    def state_dict(self):
        step = self._step.to(torch.float32).reshape(()).cpu()
        state = {}
        for i, p in enumerate(self._params):
            lo, hi = self._offsets[i], self._offsets[i + 1]
            state[i] = {'step': step.clone(), 'exp_avg': self._exp_avg[lo:hi].view(p.shape).clone(),
                        'exp_avg_sq': self._exp_avg_sq[lo:hi].view(p.shape).clone()}
        g = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        g['params'] = list(range(len(self._params)))
        return {'state': state, 'param_groups': [g]}

    def load_state_dict(self, state_dict):
        groups = state_dict['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self._params):
            raise ValueError('loaded state dict has a different number of parameter groups / parameters')
        for k, v in groups[0].items():
            if k in ('lr', 'betas', 'eps', 'weight_decay'):
                self.param_groups[0][k] = tuple(v) if k == 'betas' else v
        state = state_dict['state']
        steps = set()
        for i, p in enumerate(self._params):
            st = state.get(i, state.get(str(i)))
            lo, hi = self._offsets[i], self._offsets[i + 1]
            if st is None:                       # torch leaves parameters that never saw a gradient without state
                self._exp_avg[lo:hi].zero_()
                self._exp_avg_sq[lo:hi].zero_()
                continue
            if tuple(st['exp_avg'].shape) != tuple(p.shape):
                raise ValueError('optimizer state of parameter %d has shape %s, expected %s'
                                 % (i, tuple(st['exp_avg'].shape), tuple(p.shape)))
            self._exp_avg[lo:hi].copy_(st['exp_avg'].reshape(-1))
            self._exp_avg_sq[lo:hi].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(st['step']))
        if len(steps) > 1:
            raise ValueError('per-parameter step counts differ (%s): one flat buffer has one step' % sorted(steps))
        self._step.fill_(steps.pop() if steps else 0)
        self._hyper_host = None
        self.sync_hyper()
