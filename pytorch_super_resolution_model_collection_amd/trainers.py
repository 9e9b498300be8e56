"""Train-step bodies of the reference's trainers on the MI355X path.

Each `*_step` builder returns a closure step(*batch) -> device loss tensor(s) that performs
    zero_grad -> forward -> loss -> backward -> [RCCL all-reduce] -> [clip] -> optimizer step
with the same op order, loss and hyper-parameters as the cited reference lines, and with NO host
synchronisation (the reference syncs every iteration through loss.data[0], edsr.py:158).
`GraphedStep` captures the forward+backward and the optimizer parts into hipGraphs so a step is
two graph launches (+ the all-reduce) instead of ~200 kernel launches from Python.
"""
import contextlib
import gc

import torch

from . import ops
from .layers import bump_weight_epoch
from .optim import FlatParams, make_optimizer


@contextlib.contextmanager
def _no_gc_during_capture():
    """hipGraph objects must not be destroyed while a stream is capturing (`hipErrorStreamCaptureUnsupported`, raised
    from a destructor = process abort).  An earlier GraphedStep / GraphedSegments that is only reachable through a
    reference cycle (bound methods in its segment list) is freed by the cyclic garbage collector at an arbitrary
    allocation — e.g. in the middle of the next capture.  Collect first, keep the collector off while capturing."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()
        quiesce_gc()     # (what the capture built stays for the life of the step)


def quiesce_gc():
    """Take the cyclic collector out of the steady-state step.  A process that has imported torch tracks ~170 k container
    objects; one automatic full (generation-2) collection walks all of them: 73 - 83 ms measured (tools/dp_step_times.py),
    landing inside whichever train step crosses the threshold -- the 83 ms step among 1.6 ms ones of round 5's driver run,
    and under data parallelism a straggler every other rank waits for.  After set-up (model, optimizer, captured graphs)
    everything alive is long-lived: collect once, then gc.freeze() moves it to the permanent generation, so later
    collections only look at what a step allocates (< 1 ms).  Reference counting still frees frozen objects."""
    gc.collect()
    gc.freeze()


def _static_buffers(example_inputs):
    """Static input buffers of a captured step.  4-D image batches are kept channels_last: the copy of a new batch into
    the buffer (one strided copy either way) then delivers the layout the first convolution reads, and the captured step
    holds no layout kernel."""
    out = []
    for t in example_inputs:
        if t.dim() == 4 and t.is_floating_point():
            out.append(torch.empty_like(t, memory_format=torch.channels_last).copy_(t))
        else:
            out.append(torch.empty_like(t).copy_(t))
    return out


def _backward(loss, dp):
    """loss.backward() of the reference (edsr.py:154 ...).  Single GPU: the deferred weight gradients are launched
    (grouped) when the autograd engine finishes the pass.  Data parallel: they stay pending so that dp.exchange() can
    interleave the grouped launches with the gradient buckets; the backward is seeded with 1/world."""
    # (premasked_gradients: a train step reads parameter gradients only, so activation gradients may travel pre-masked)
    if dp is not None and dp.active:
        with ops.manual_wgrad_flush(), ops.premasked_gradients():
            loss.backward(dp.loss_seed)
    else:
        with ops.premasked_gradients():
            ops.backward(loss)   # seeded with the persistent ones tensor: no fill, no scale pass (ops.unit_seed)
        ops.join_side_streams()  # (also flushes weight gradients recorded outside an engine callback)


def _seeded(dp):
    """Context for computing a loss that `_backward(loss, dp)` will seed: under data parallelism the 1/world seed is
    folded into the loss gradient by the loss kernel itself (ops.loss_seed)."""
    if dp is not None and dp.active:
        return ops.loss_seed(1.0 / dp.world, dp.loss_seed)
    return contextlib.nullcontext()


def mse_step(model, opt, dp=None, clip=None):
    """srcnn.py:127-131 / fsrcnn.py:153-157 / vdsr.py:143-150 (clip = 0.4)."""
    def step(inp, target):
        opt.zero_grad()
        with _seeded(dp):
            loss = ops.mse_loss(model(inp), target)
        _backward(loss, dp)
        if dp is not None:
            dp.allreduce_grads()
        if clip is not None:
            opt.clip_grad_norm(clip)
        opt.step()
        return loss
    return step


def l1_step(model, opt, dp=None):
    """edsr.py:151-155"""
    def step(inp, target):
        opt.zero_grad()
        with _seeded(dp):
            loss = ops.l1_loss(model(inp), target)
        _backward(loss, dp)
        if dp is not None:
            dp.allreduce_grads()
        opt.step()
        return loss
    return step


def lapsrn_step(model, opt, dp=None):
    """lapsrn.py:190-199: two Charbonnier losses, two backward calls into the same gradients."""
    def step(inp, target2x, target4x):
        opt.zero_grad()
        hr2, hr4 = model(inp)
        with _seeded(dp):
            l1 = ops.charbonnier_loss(hr2, target2x)
            l2 = ops.charbonnier_loss(hr4, target4x)
        seed = dp.loss_seed if (dp is not None and dp.active) else None
        if seed is not None:
            with ops.manual_wgrad_flush():
                with ops.premasked_gradients():
                    torch.autograd.backward([l1, l2], [seed, seed])
        else:
            with ops.premasked_gradients():
                ops.backward([l1, l2])
        if dp is not None:
            dp.allreduce_grads()
        opt.step()
        return l1, l2
    return step


def srgan_step(G, D, g_opt, d_opt, g_dp=None, d_dp=None, feature_extractor=None, lazy_pack=False, prune_dead_grads=False):
    """srgan.py:249-310 with [B,1] labels.  As in the reference the D step back-propagates through G
    (G is not detached, srgan.py:279) and the G step accumulates into D's gradients, which the
    next D step's zero_grad discards.
    `feature_extractor` (models.FeatureExtractor): adds the reference's VGG content term 6e-3 * MSE(vgg(norm(recon.data)),
    vgg(norm(hr)).detach()) to the reported G loss (srgan.py:301-308).  Both operands are detached in the reference,
    so the term changes the logged scalar only — never a gradient; without an extractor the step
    returns mse + 1e-3 * GAN, which has the same gradients.
    lazy_pack: the two zero_grad() calls skip the filter pack of a model whose plan is current (optim.zero_grad,
    repack="stale": 2 instead of 4 whole-model packs per step) -- for steps replayed by a graph that knows both
    FlatParams (GraphedFn(flats=[...]) / GraphedSegments).
    prune_dead_grads (NOT the reference's execution, off by default and in bench.py's c5): the reference computes two sets
    of gradients nobody reads -- G's from D_loss.backward() (G_optimizer.zero_grad() clears them before the G step,
    srgan.py:287-291) and D's parameter gradients from G_loss.backward() (cleared by the next iteration's
    D_optimizer.zero_grad(), srgan.py:272).  With the flag the D step sees G's output detached and the G step runs D
    with its parameters frozen (data gradients only).  Parameters, optimizer states and BatchNorm statistics after the
    step are the faithful step's (up to the summation order inside the grouped weight-gradient launches, whose split
    depends on how many layers share a launch); D's `.grad` then holds the D step's gradients only.  Single GPU / eager
    DP path only (the DP graph segments keep the reference's execution)."""
    from . import utils
    repack = "stale" if lazy_pack else "always"
    d_params = [p for p in D.parameters()] if prune_dead_grads else []

    def freeze_d(flag):
        for p in d_params:
            p.requires_grad_(not flag)

    def step(lr_img, hr_img):
        b = lr_img.shape[0]
        real = ops.const_rows(1.0, b, lr_img.device)     # persistent label rows and srk_axpby loss sums: the captured
        fake = ops.const_rows(0.0, b, lr_img.device)     # step holds no ATen arithmetic node
        d_opt.zero_grad(repack=repack)
        recon_d = G(lr_img)      # (in grad mode either way: the same kernels and precision class as the reference path)
        d_loss = ops.loss_sum(ops.bce_loss(D(hr_img), real),
                              ops.bce_loss(D(recon_d.detach() if prune_dead_grads else recon_d), fake))
        del recon_d
        _backward(d_loss, d_dp)
        if d_dp is not None:
            d_dp.allreduce_grads()
        d_opt.step()
        g_opt.zero_grad(repack=repack)
        recon = G(lr_img)
        if prune_dead_grads:
            freeze_d(True)
            try:
                gan_loss = ops.bce_loss(D(recon), real)
            finally:
                freeze_d(False)
        else:
            gan_loss = ops.bce_loss(D(recon), real)
        g_loss = ops.loss_sum(ops.mse_loss(recon, hr_img), gan_loss, 1.0, 1e-3)
        if feature_extractor is not None:
            with torch.no_grad():   # srgan.py:301-305 (the inputs are already normalised once, as in the reference)
                real_feature = feature_extractor(utils.norm(hr_img, vgg=True))
                fake_feature = feature_extractor(utils.norm(recon.detach(), vgg=True))
                vgg_loss = ops.mse_loss(fake_feature, real_feature)
            g_loss = ops.loss_sum(g_loss, vgg_loss, 1.0, 6e-3)
        _backward(g_loss, g_dp)
        if g_dp is not None:
            g_dp.allreduce_grads()
        g_opt.step()
        return d_loss, g_loss
    return step


def srgan_segments(G, D, g_opt, d_opt, g_dp=None, d_dp=None, lazy_pack=False):
    """The adversarial step of `srgan_step` cut at its two gradient exchanges, for GraphedSegments:
    [(D forward/backward, d_dp), (D update + G forward/backward, g_dp), (G update, None)].  lazy_pack: see srgan_step."""
    out = {}
    repack = "stale" if lazy_pack else "always"

    def seg_d(lr_img, hr_img):
        b = lr_img.shape[0]
        real = ops.const_rows(1.0, b, lr_img.device)
        fake = ops.const_rows(0.0, b, lr_img.device)
        d_opt.zero_grad(repack=repack)
        out["d"] = ops.loss_sum(ops.bce_loss(D(hr_img), real), ops.bce_loss(D(G(lr_img)), fake))
        _backward(out["d"], d_dp)
        return out["d"]

    def seg_g(lr_img, hr_img):
        real = ops.const_rows(1.0, lr_img.shape[0], lr_img.device)
        d_opt.step()
        g_opt.zero_grad(repack=repack)
        recon = G(lr_img)
        out["g"] = ops.loss_sum(ops.mse_loss(recon, hr_img), ops.bce_loss(D(recon), real), 1.0, 1e-3)
        _backward(out["g"], g_dp)
        return out["g"]

    def seg_u(lr_img, hr_img):
        g_opt.step()
        return out["d"], out["g"]

    return [(seg_d, d_dp), (seg_g, g_dp), (seg_u, None)]


def _capture(graph, pool=None):
    """torch.cuda.graph with capture_error_mode="thread_local": ProcessGroupNCCL's watchdog thread polls the events of
    earlier collectives (hipEventQuery) at its own pace, and under the default "global" mode a query that lands while
    THIS thread is capturing is an illegal call that takes the process down -- seen on MI355X as an intermittent abort
    of the first capture after a broadcast / all-reduce (tests/test_dp_gpu.py, single-rank RCCL).  Work the autograd
    thread launches on the capturing stream is captured in either mode."""
    ops.amax_new_step()   # running maxima computed before the capture must not be baked into it
    return torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local")


class _Splitter(object):
    """Captures a function as a SEQUENCE of hipGraphs cut at the collectives it issues through ops._collective (SyncBN:
    one all-reduce of the [2C] sums per BatchNorm call and direction): items = [("graph", g) | ("eager", fn), ...].
    The backward pass runs on the calling thread while capturing (torch.autograd.set_multithreading_enabled(False)):
    a capture has to be ended by the thread that began it."""

    def __init__(self, pool):
        self.pool, self.items, self.g = pool, [], None

    def begin(self):
        self.g = torch.cuda.CUDAGraph()
        self.g.capture_begin(pool=self.pool, capture_error_mode="thread_local")

    def end(self):
        self.g.capture_end()
        self.pool = self.g.pool()
        self.items.append(("graph", self.g))
        self.g = None

    def collective(self, fn):
        self.end()
        fn()                       # (keeps the process group's sequence numbers in step with the other ranks)
        self.items.append(("eager", fn))
        self.begin()

    def capture(self, fn, args):
        ops.amax_new_step()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        ops._SPLITTER[0] = self
        try:
            with torch.cuda.stream(side), torch.autograd.set_multithreading_enabled(False):
                self.begin()
                try:
                    out = fn(*args)
                finally:
                    self.end()
        finally:
            ops._SPLITTER[0] = None
        cur.wait_stream(side)
        return out


def _has_sync_bn(segments):
    for _, dp in segments:
        mod = getattr(getattr(dp, "flat", None), "module", None)
        if mod is not None and any(getattr(m, "sync_group", None) is not None for m in mod.modules()):
            return True
    return False


def sync_batchnorm(module, group=None):
    """SyncBN (SURVEY.md 8e caveat): every BatchNorm of `module` all-reduces its [2C] sums over `group` (None: the default
    process group), so a data-parallel step normalises with the statistics of the GLOBAL batch, like the single-process
    reference; no-op without an initialised process group.  Returns the number of layers switched."""
    import torch.distributed as dist
    from .layers import BatchNorm1d, BatchNorm2d
    if not dist.is_initialized():
        return 0
    grp = group if group is not None else dist.group.WORLD
    n = 0
    for m in module.modules():
        if isinstance(m, (BatchNorm2d, BatchNorm1d)):
            m.sync_group = grp
            n += 1
    return n


class GraphedSegments(object):
    """A data-parallel train step as hipGraphs split at the gradient exchanges.

    `segments` = [(fn(*inputs), dp or None), ...]: every fn is captured as one graph; a segment with a DataParallel
    ends in a backward pass whose weight gradients were left pending (ops.manual_wgrad_flush): each pending launch
    group becomes a small graph of its own, and at replay the groups run one after the other with the RCCL all-reduce
    of the bucket each one completes issued (eagerly, async) right behind it — the bucket travels while the next
    group computes.  Call with new batches (copied into the static buffers); returns what the last fn returned."""

    def __init__(self, segments, example_inputs, warmup=2, eager_step=None):
        self.static = _static_buffers(example_inputs)
        self.segments = segments
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.plan, pool = [], None
        split = _has_sync_bn(segments)    # SyncBN: the segment's graph is cut again at every statistics all-reduce
        with _no_gc_during_capture():
            for fn, dp in segments:
                if split:
                    sp = _Splitter(pool)
                    self.out = sp.capture(fn, self.static)
                    pool, g = sp.pool, sp.items
                else:
                    g = torch.cuda.CUDAGraph()
                    with _capture(g, pool=pool):
                        self.out = fn(*self.static)
                    pool = g.pool()
                    g = [("graph", g)]
                wgraphs, sends, keep = [], [], None
                if dp is not None and dp.active:
                    keep = ops.pending_wgrad_groups(dp.trunk_chunk_layers)   # holds x / dy / mask tensors of graph `g` alive
                    sends = dp.plan(keep)
                    ops.drop_pending_wgrads()
                    for recs in keep:
                        wg = torch.cuda.CUDAGraph()
                        with _capture(wg, pool=pool):
                            ops.launch_wgrad_group(recs)
                        wgraphs.append(wg)
                else:
                    ops.join_side_streams()
                self.plan.append((g, dp, wgraphs, sends, keep))
        self._flats = [dp.flat for _, dp in segments if dp is not None and hasattr(dp.flat, "mark_changed")]
        self._seen = {}
        for f in self._flats:   # same host bookkeeping as after a replay
            f.mark_changed()
        _note_epochs(self._flats, self._seen)

    def _eager(self):
        out = None
        for fn, dp in self.segments:
            out = fn(*self.static)
            if dp is not None:
                dp.exchange()
        return out

    def __call__(self, *batch):
        for s, b in zip(self.static, batch):
            if b is not s:
                s.copy_(b, non_blocking=True)
        _repack_touched(self._flats, self._seen)
        for items, dp, wgraphs, sends, _ in self.plan:
            for kind, obj in items:
                if kind == "graph":
                    obj.replay()
                else:
                    obj()
            if dp is not None and dp.active:
                works = []
                if not wgraphs:
                    dp.send(sends[0] if sends else [(0, dp.flat.grad.numel())], works)
                for wg, ranges in zip(wgraphs, sends):
                    wg.replay()
                    dp.send(ranges, works)
                for w in works:
                    w.wait()
        bump_weight_epoch()
        for f in self._flats:
            f.mark_changed()
        _note_epochs(self._flats, self._seen)
        return self.out


def build(kind, model, lr, dp_group=None, use_dp=False):
    """(flat, optimizer, dp, step) for one of 'srcnn' | 'fsrcnn' | 'vdsr' | 'edsr' | 'lapsrn' | 'espcn'."""
    from .dp import DataParallel
    flat = FlatParams(model)
    opt = make_optimizer(kind, flat, lr)
    dp = DataParallel(flat, dp_group) if use_dp else None
    if dp is not None:
        dp.broadcast_params()
    if kind == "edsr":
        step = l1_step(model, opt, dp)
    elif kind == "lapsrn":
        step = lapsrn_step(model, opt, dp)
    elif kind == "vdsr":
        step = mse_step(model, opt, dp, clip=0.4)
    else:
        step = mse_step(model, opt, dp)
    return flat, opt, dp, step


class GraphedStep(object):
    """hipGraph capture of a train step with static input buffers.

    Single GPU: one graph = zero_grad + filter packing + forward + loss + backward (data-gradient chain, then the
    grouped weight gradients) + [clip] + optimizer.  Data parallel: GraphedSegments — graph A (through the data-gradient
    chain), one small graph per weight-gradient group with its gradient bucket's RCCL all-reduce issued behind it, then
    graph B = [clip] + optimizer.  Call with new batches; they are copied into the static buffers.
    """

    def __init__(self, model, opt, loss_fn, example_inputs, dp=None, clip=None, warmup=3):
        self.model, self.opt, self.dp, self.clip = model, opt, dp, clip
        self.loss_fn = loss_fn
        self.seg = None
        if dp is not None and dp.active:
            self.seg = GraphedSegments([(self._fwd_bwd_args, dp), (self._update_args, None)], example_inputs, warmup=warmup)
            self.static = self.seg.static
            return
        self.static = _static_buffers(example_inputs)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._fwd_bwd()
                self._update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph_a = torch.cuda.CUDAGraph()
        with _no_gc_during_capture():
            with _capture(self.graph_a):
                self.loss = self._fwd_bwd()
                self._update()

    def close(self):
        """Drop the captured graphs now (outside any capture) instead of whenever the garbage collector finds them."""
        self.seg = None
        self.graph_a = None

    def _fwd_bwd_args(self, *static):
        self.static = list(static)
        self.loss = self._fwd_bwd()
        return self.loss

    def _update_args(self, *static):
        self._update()
        return self.loss

    def _fwd_bwd(self):
        self.opt.zero_grad()
        with _seeded(self.dp):
            loss = self.loss_fn(self.model(self.static[0]), *self.static[1:])
        _backward(loss, self.dp)
        return loss

    def _update(self):
        if self.clip is not None:
            self.opt.clip_grad_norm(self.clip)
        self.opt.step()

    def __call__(self, *batch):
        if self.seg is not None:
            return self.seg(*batch)
        for s, b in zip(self.static, batch):
            if b is not s:
                s.copy_(b, non_blocking=True)
        self.graph_a.replay()
        # the replayed optimizer kernel changed the weights without running optim.step()'s host bookkeeping: packed
        # filters cached by no-grad forwards (layers._PackCache) and the PackPlan must not be reused
        self.opt.flat.mark_changed()
        return self.loss


def _repack_touched(flats, seen):
    """Before a replay: a captured multi-model step packs a model's filters only where ITS OWN updates made them stale
    (optim.zero_grad skips a current plan), so parameters somebody else changed since the last replay -- anything that
    went through FlatParams.mark_changed(): load_state_dict (post-hook registered by FlatParams), DataParallel's
    broadcast, an eager optimizer step in between; a raw write to `.data` needs an explicit mark_changed() -- are
    re-packed here, eagerly, in front of the graph."""
    for f in flats:
        if seen.get(id(f)) != f.epoch and not f.plan.current():
            f.plan.pack()


def _note_epochs(flats, seen):
    for f in flats:
        seen[id(f)] = f.epoch


class GraphedFn(object):
    """hipGraph capture of an arbitrary single-GPU step function of tensors (e.g. the two-model, two-optimizer SRGAN
    step, `srgan_step` without data parallelism): static input buffers, `warmup` eager calls on a side stream, one
    graph; call with new batches (copied into the static buffers), returns the captured output tensors.  Everything
    the step does must be stream work (no host reads of device values), which holds for all steps of this package."""

    def __init__(self, fn, example_inputs, warmup=3, flats=()):
        self.fn = fn
        self.flats = list(flats)   # FlatParams the step updates (their PackPlans are invalidated after every replay)
        self.static = _static_buffers(example_inputs)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn(*self.static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with _no_gc_during_capture():
            with _capture(self.graph):
                self.out = fn(*self.static)
        self._seen = {}
        for f in self.flats:   # same host bookkeeping as after a replay
            f.mark_changed()
        _note_epochs(self.flats, self._seen)

    def __call__(self, *batch):
        for s, b in zip(self.static, batch):
            if b is not s:
                s.copy_(b, non_blocking=True)
        _repack_touched(self.flats, self._seen)
        self.graph.replay()
        # weights changed inside the graph: invalidate the packed-filter caches of no-grad forwards (see GraphedStep)
        bump_weight_epoch()
        for f in self.flats:
            f.mark_changed()
        _note_epochs(self.flats, self._seen)
        return self.out


class AutoGraph(object):
    """A train step that turns itself into a hipGraph.  The FIRST batch of a shape runs through `eager` (a real step, and
    every lazy initialisation happens outside a capture); the next batch of that shape is captured by
    `make_graph(tensors)` (a GraphedStep / GraphedFn / GraphedSegments built with warmup=0 on that batch) and replayed
    from then on.  Batches of another shape (the ragged last one of an epoch) run eagerly.  Learning rates are device
    scalars the optimizer kernels read (optim._Group: `group['lr'] /= 2` reaches them), so a decay needs no re-capture.
    `enabled=False`: always eager."""

    def __init__(self, eager, make_graph, enabled=True):
        self.eager, self.make_graph, self.enabled = eager, make_graph, enabled
        self.graph, self.seen = None, None

    def __call__(self, *tensors):
        if not self.enabled or not all(t.is_cuda for t in tensors):
            return self.eager(*tensors)
        shapes = tuple(tuple(t.shape) for t in tensors)
        g = self.graph
        if g is not None and g[1] == shapes:
            return g[0](*tensors)
        if self.seen != shapes:                 # first batch of this shape: eager (and remember the shape)
            if self.seen is None or g is None:
                self.seen = shapes
            return self.eager(*tensors)
        self.close()
        gs = self.make_graph(tensors)
        self.graph = (gs, shapes)
        return gs(*tensors)

    def close(self):
        if self.graph is not None and hasattr(self.graph[0], "close"):
            self.graph[0].close()
        self.graph = None

