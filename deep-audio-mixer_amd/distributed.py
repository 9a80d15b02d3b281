"""Data-parallel training over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference is single-device; the semantics defined here are: clips (chunks) are sharded
rank-strided over the replicas, BatchNorm statistics stay local to a replica, gradients are summed over ranks
in ONE flat bucket (12.6 MB for ResNet18: a single all-reduce, sized for xGMI's per-link bandwidth rather than
many small ones) and divided by the world size, rank 0 logs and writes checkpoints.  An N-GPU step therefore equals
a reference step over N micro-batches with averaged gradients.
"""
import os

import torch
import torch.distributed as dist
from torch.utils.data import Sampler


def init_process_group(backend=None, device=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torch.distributed.run sets them)."""
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=device or torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_indices(n_items, rank, world, drop_last=True):
    """Rank-strided partition of the global chunk index space (i = rank mod world); with drop_last every rank
    gets the same number of items so that collectives line up."""
    n = (n_items // world) * world if drop_last else n_items
    return list(range(rank, n, world))


class DistributedChunkSampler(Sampler):
    """Sequential rank-strided sampler (the reference loaders use shuffle=False, training.ipynb cell 6)."""

    def __init__(self, dataset_len, rank=None, world=None):
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.indices = shard_indices(dataset_len, self.rank, self.world)

    def __iter__(self):
        return iter(self.indices)

    def __len__(self):
        return len(self.indices)


def broadcast_module(module, src=0, group=None):
    """Identical replicas: rank `src`'s parameters and buffers everywhere."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src, group=group)


def backward_late(loss, late_params, boundary):
    """Stage 1 of the cut backward pass: gradients of `loss` w.r.t. the parameters BEHIND the bucket boundary (the deep
    end of the trunk + heads) and w.r.t. the boundary activation; nothing in front of the boundary is touched.
    Returns (list of parameter gradients, d loss / d boundary)."""
    grads = torch.autograd.grad(loss, list(late_params) + [boundary], allow_unused=True)
    return list(grads[:-1]), grads[-1]          # (None for parameters whose gradient went straight into a bound slot)


def backward_early(boundary, d_boundary, early_params):
    """Stage 2: continues backward from the boundary activation into the layers in front of it; accumulates into the
    .grad of `early_params` (meant to run while the late bucket is on the wire)."""
    torch.autograd.backward([boundary], [d_boundary], inputs=list(early_params))


class GradBucket:
    """One flat gradient bucket for an arbitrary list of parameters (works on CPU/gloo and GPU/RCCL), for use with
    any torch optimizer:   loss.backward(); bucket.all_reduce_mean(); optimizer.step().
    (deep_audio_mixer_amd.optim.Adam carries its own flat bucket and folds the 1/world into its kernel.)"""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        if any(hasattr(p, '_dam_grad') for p in self.params):
            raise RuntimeError('these parameters have gradient slots bound (optim.Adam.bind_grad_slots / engine.TrainStep): '
                               'their .grad stays None, a GradBucket over them would all-reduce zeros -- unbind first '
                               '(TrainStep.close()) or use optim.Adam\'s own buckets')
        self.group = group
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)

    def _views(self):
        off = 0
        for p in self.params:
            k = p.numel()
            yield p, self.flat[off:off + k].view(p.shape)
            off += k

    def fill(self, grads=None):
        """p.grad (or the given gradients, parameter order) -> the flat bucket."""
        for i, (p, v) in enumerate(self._views()):
            g = p.grad if grads is None else grads[i]
            if g is None:
                v.zero_()
            else:
                v.copy_(g)

    def start_all_reduce(self):
        """Asynchronous SUM over the group; returns the work handle (None on a single rank)."""
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    def finish(self, work=None):
        """Waits for start_all_reduce, divides by the world size and writes the mean back into the .grad fields."""
        if work is not None:
            work.wait()
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world > 1:
            self.flat.div_(world)
        for p, v in self._views():
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)
        return self.flat

    def all_reduce_mean(self):
        self.fill()
        return self.finish(self.start_all_reduce())


def all_gather_gains(local_gains, group=None):
    """Inference over a sharded chunk axis: gathers [n_local, S] gain blocks of equal size from every rank and
    interleaves them back into chunk order (rank-strided sharding)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_gains
    world = dist.get_world_size(group)
    parts = [torch.empty_like(local_gains) for _ in range(world)]
    dist.all_gather(parts, local_gains.contiguous(), group=group)
    return torch.stack(parts, 1).reshape(-1, local_gains.shape[-1])
