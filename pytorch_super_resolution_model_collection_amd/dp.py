"""Data-parallel training over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm), replicated flat parameters, sharded patch minibatch,
gradient all-reduce on the flat gradient buffer.

The reference has no distributed code at all; the single call site this adds is
"after loss.backward(), before optimizer.step()" (edsr.py:154->155, vdsr.py:146->149,
srgan.py:286->287 and :309->310).

Why this shape on MI355X: xGMI is point-to-point (7 links per GPU), so a few LARGE collectives
beat many small ones (each ring step pays per-link latency); the flat gradient buffer makes the
whole model one message (EDSR 6.07 MB, VDSR 2.67 MB) or a handful of multi-MB buckets (SRGAN-D
153 MB).  The mean over ranks is obtained by seeding every rank's backward with 1/world_size
(`loss_seed`), so no extra averaging pass touches the gradients after the all-reduce.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK
    (the variables `python -m torch.distributed.run` sets). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # SRK_DIST_BACKEND=gloo lets the multi-rank code path be exercised on a single-GPU box
            backend = os.environ.get("SRK_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available() and os.environ.get("SRK_SINGLE_GPU"):
            local = 0  # test mode: every rank shares device 0
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of a batch of n_items for `rank`; the remainder
    goes to the first ranks so shard sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(tensor, rank, world):
    lo, hi = shard_range(tensor.shape[0], rank, world)
    return tensor[lo:hi]


class DataParallel(object):
    """Gradient exchange for a FlatParams-like object (anything with flat `.data` and `.grad`)."""

    def __init__(self, flat, group=None, bucket_bytes=32 << 20):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bucket_elems = max(1, int(bucket_bytes) // 4)
        # upstream gradient that makes the all-reduced SUM the mean over ranks
        self.loss_seed = torch.full((), 1.0 / self.world, dtype=torch.float32, device=flat.data.device)

    def broadcast_params(self, src=0):
        """Identical initial replicas: rank `src`'s flat parameter buffer to everyone."""
        if self.world > 1:
            dist.broadcast(self.flat.data, src=src, group=self.group)
            # the parameters changed behind the optimizer's back: invalidate packed-filter caches / the PackPlan
            if hasattr(self.flat, "mark_changed"):
                self.flat.mark_changed()
            else:
                from .layers import bump_weight_epoch
                bump_weight_epoch()

    def buckets(self):
        n = self.flat.grad.numel()
        return [(lo, min(lo + self.bucket_elems, n)) for lo in range(0, n, self.bucket_elems)]

    def allreduce_grads(self):
        """SUM all-reduce of the flat gradient buffer, in a few multi-MB buckets issued
        back-to-back (async) so RCCL pipelines them over the xGMI links."""
        from . import ops
        ops.join_side_streams()  # weight gradients forked onto the side stream must have landed
        if self.world == 1:
            return
        g = self.flat.grad
        works = [dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for lo, hi in self.buckets()]
        for w in works:
            w.wait()

    def allreduce_scalar(self, t):
        """Mean of a logging scalar (the loss) over ranks — off the critical path."""
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t = t / self.world
        return t
