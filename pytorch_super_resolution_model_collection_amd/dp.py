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

# the host driver of these boxes only supports dmabuf IPC: without this RCCL / cross-process GPU memory sharing fails with
# `hipIpcGetMemHandle: invalid argument` (must be in the environment before the HIP runtime initialises)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK
    (the variables `python -m torch.distributed.run` sets). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    # SRK_DP_FORCE_COMM=1 with ONE rank: a real process group (RCCL on a GPU box) so that every collective of the
    # data-parallel paths runs on a single-GPU machine (tests, `bench.py` dry runs of the N > 1 code path)
    forced = world == 1 and bool(os.environ.get("SRK_DP_FORCE_COMM"))
    if (world > 1 or forced) and not dist.is_initialized():
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

    def __init__(self, flat, group=None, bucket_bytes=32 << 20, min_bucket_bytes=256 << 10, trunk_chunk_layers=11):
        self.flat = flat
        self.group = group
        self.comm_enabled = True
        self.min_bucket_elems = max(1, int(min_bucket_bytes) // 4)   # smaller final runs wait for a neighbour
        self.trunk_chunk_layers = int(trunk_chunk_layers)             # layers per grouped weight-gradient launch under DP
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # `active`: the exchange really runs.  SRK_DP_FORCE_COMM=1 (tests) keeps every collective in the path with a
        # process group of ONE rank -- on a single-GPU box that is the only way to put RCCL itself (communicator init,
        # async all-reduce kernels between graph replays, stream / event ordering) under the data-parallel step
        self.active = self.world > 1 or (dist.is_initialized() and bool(os.environ.get("SRK_DP_FORCE_COMM")))
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bucket_elems = max(1, int(bucket_bytes) // 4)
        # upstream gradient that makes the all-reduced SUM the mean over ranks
        self.loss_seed = torch.full((), 1.0 / self.world, dtype=torch.float32, device=flat.data.device)

    def broadcast_params(self, src=0):
        """Identical initial replicas: rank `src`'s flat parameter buffer to everyone."""
        if self.active:
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

    # -- overlapped exchange ------------------------------------------------------------------------------------
    # After loss.backward() the data-gradient chain is done and the conv weight gradients are still pending
    # (ops.DEFER_WGRAD) as a short list of grouped launches in backward order: reconstruction conv, upsamplers, the
    # residual trunk in chunks, input conv.  exchange() launches them one group at a time and, right after each
    # group, hands the part of the flat gradient buffer that just became final to RCCL (async: ProcessGroupNCCL runs
    # it on its own stream behind an event), so a bucket travels over xGMI while the next group computes.  xGMI is
    # point-to-point, so a few multi-hundred-KB..MB messages (<= 8 per EDSR step) beat one message per tensor; the
    # trunk is cut into `trunk_chunks` pieces so that only the last ~2 MB bucket is exposed.
    def plan(self, groups, min_elems=None):
        """For launch groups (ops.pending_wgrad_groups order) -> per group the list of [lo, hi) ranges of the flat
        gradient buffer to send after it.  A parameter is final after the last group that writes it (parameters no
        pending record writes are final already); ranges are maximal runs of final, unsent parameters, held back while
        shorter than `min_elems` unless nothing will follow."""
        flat = self.flat
        if min_elems is None:
            min_elems = self.min_bucket_elems
        base, total = flat.grad.data_ptr(), flat.grad.numel()
        spans = []   # (lo, hi, ready_after_group) per parameter, in buffer order
        ready = {}
        for k, recs in enumerate(groups):
            for r in recs:
                for t in (r[6], r[7]):
                    if t is None:
                        continue
                    off = (t.data_ptr() - base) // 4
                    if 0 <= off < total and t.device == flat.grad.device:
                        ready[off] = k
        offs = list(flat.offsets) + [total]
        for i in range(len(flat.offsets)):
            spans.append((offs[i], offs[i + 1], ready.get(offs[i], -1)))
        sent = [False] * len(spans)
        out = []
        for k in range(len(groups)):
            last = k == len(groups) - 1
            ranges, i = [], 0
            while i < len(spans):
                if sent[i] or spans[i][2] > k:
                    i += 1
                    continue
                j = i
                while j < len(spans) and not sent[j] and spans[j][2] <= k:
                    j += 1
                lo, hi = spans[i][0], spans[j - 1][1]
                if last or hi - lo >= min_elems:
                    ranges.append((lo, hi))
                    for q in range(i, j):
                        sent[q] = True
                i = j
            out.append(ranges)
        if not groups:
            out.append([(0, total)])
        return out

    def send(self, ranges, works):
        if not self.comm_enabled:   # measurement only (bench.py: step time without the exchange = exposed communication)
            return
        g = self.flat.grad
        for lo, hi in ranges:
            for b0 in range(lo, hi, self.bucket_elems):
                works.append(dist.all_reduce(g[b0:min(b0 + self.bucket_elems, hi)], op=dist.ReduceOp.SUM, group=self.group,
                                             async_op=True))

    def exchange(self):
        """Launch the pending weight gradients group by group and all-reduce (SUM; the 1/world seed of the backward
        makes it the mean) every part of the flat gradient buffer as soon as it is final.  Returns when the current
        stream is ordered after the last bucket."""
        from . import ops
        if not self.active:
            ops.join_side_streams()
            return
        groups = ops.pending_wgrad_groups(self.trunk_chunk_layers)
        if not groups:      # nothing deferred (side-stream mode, or a model without conv layers)
            ops.join_side_streams()
            works = []
            self.send([(0, self.flat.grad.numel())], works)
            for w in works:
                w.wait()
            return
        sends = self.plan(groups)
        works = []
        ops.drop_pending_wgrads()
        for recs, ranges in zip(groups, sends):
            ops.launch_wgrad_group(recs)
            self.send(ranges, works)
        for w in works:
            w.wait()

    def allreduce_grads(self):
        """SUM all-reduce of the flat gradient buffer (see exchange: overlapped with the pending weight gradients)."""
        self.exchange()

    def allreduce_scalar(self, t):
        """Mean of a logging scalar (the loss) over ranks — off the critical path."""
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t = t / self.world
        return t
