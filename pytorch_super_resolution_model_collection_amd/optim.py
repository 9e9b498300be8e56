"""Flat fp32 parameter / gradient storage and the optimizers of the reference's trainers
(torch.optim.SGD / Adam call sites: srcnn.py:79, fsrcnn.py:105-106, vdsr.py:86-90, espcn.py:79,
edsr.py:93, lapsrn.py:135, srgan.py:147-149; clip_grad_norm: vdsr.py:149) on srk_sgd_step /
srk_adam_step / srk_grad_norm_clip.

MI355X-first layout: all parameters of a model live in ONE contiguous fp32 buffer and all
gradients in another (288 GB of HBM make replication free), so
  * zero_grad is one memset, the optimizer is one kernel launch over the whole model,
  * the data-parallel gradient exchange is one (or a few large) RCCL all-reduce(s) on the flat
    gradient buffer instead of 74 small ones (EDSR),
  * the weight-gradient kernels accumulate straight into the gradient buffer (`_srk_grad` views),
    so autograd never launches an ATen add for parameter gradients.
nn.Parameter objects stay in place (their .data / .grad become views), so state_dict() and the
reference's checkpoint files keep working.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .layers import bump_weight_epoch

_ALIGN = 4  # floats (16 bytes) — keeps every parameter view 16-byte aligned for the vector kernels


class FlatParams(object):
    """Re-homes the (unique) parameters of `module` into flat `data` / `grad` buffers."""

    def __init__(self, module):
        seen, params, names = set(), [], []
        for name, p in module.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            params.append(p)
            names.append(name)
        if not params:
            raise ValueError("module has no parameters")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatParams: move the module to the GPU first (module.to('cuda'))")
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.module, self.params, self.names, self.offsets, self.numel = module, params, names, offs, total
        self.data = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(params, offs):
            v = self.data[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
        bump_weight_epoch()
        self.epoch = 0           # bumped whenever the flat parameters change (optimizer step, load_state_dict)
        # ONE allocation [packed filters ... | max|w| scratch words | flat gradients | running-maximum chunk]: everything a
        # train step has to find zeroed (the last three) is adjacent, so zero_grad is a single fill node instead of three
        # (the strong-scaled EDSR shard is ~70 kernels of 5 - 100 us: every launch counts)
        self._store = None
        self.plan = PackPlan(module, self)     # lays the packed buffers out and calls _alloc_store
        if self._store is None:
            self._alloc_store(0, 0)
        for p, o in zip(params, offs):
            g = self.grad[o:o + p.numel()].view(p.shape)
            p._srk_grad = g      # kernels accumulate here (ops._Conv2d.backward etc.)
            p.grad = g           # what user code / hooks see
        # load_state_dict copies into the parameter views in place: the packed filters (PackPlan, no-grad caches) and
        # the graphs that skip a current plan (trainers._repack_touched compares `epoch`) must see that as a change
        self._load_hook = None
        if hasattr(module, "register_load_state_dict_post_hook"):
            import weakref
            me = weakref.ref(self)

            def _loaded(mod, incompatible_keys):
                f = me()
                if f is not None:
                    f.mark_changed()

            self._load_hook = module.register_load_state_dict_post_hook(_loaded)

    AMAX_TENSORS = 128   # running-maximum buffers (ops._amax_alloc) a step gets from the zeroed chunk of the store

    def _alloc_store(self, packed_bytes, zero_from):
        """[packed_bytes of PackPlan buffers, of which everything from `zero_from` on is per-step scratch | gradients |
        running-maximum chunk] -> the PackPlan's part."""
        dev = self.data.device
        goff = (int(packed_bytes) + 255) // 256 * 256
        gbytes = self.numel * 4
        abytes = self.AMAX_TENSORS * _lib.AMAX_FLOATS * 4
        self._store = torch.zeros(goff + (gbytes + 255) // 256 * 256 + abytes, dtype=torch.uint8, device=dev)
        self.grad = self._store[goff:goff + gbytes].view(torch.float32)
        aoff = goff + (gbytes + 255) // 256 * 256
        self.amax_chunk = self._store[aoff:aoff + abytes].view(torch.float32)
        self._zero_region = self._store[(zero_from if packed_bytes else goff):]
        return self._store[:max(int(packed_bytes), 256)] if packed_bytes else None

    def zero_grad(self):
        """Gradients, the PackPlan's max|w| scratch words and the running-maximum chunk: one fill."""
        from . import ops
        ops.join_side_streams()  # no weight gradient of the previous step may still be accumulating
        self._zero_region.zero_()

    def mark_changed(self):
        """Call after the flat parameters changed by anything other than optimizer.step() — a hipGraph replay of a
        captured step, a direct write to `.data`, a broadcast: invalidates the PackPlan and every cached packed filter."""
        bump_weight_epoch()
        self.epoch += 1

    def named_grads(self):
        return {n: p._srk_grad for n, p in zip(self.names, self.params)}


class PackPlan(object):
    """Packs the filters of EVERY conv layer of a model (forward + data-gradient layouts, fp32 +
    bf16x3 planes, pixel-shuffle bias permutation) with ONE kernel launch per training step
    (srk_pack_weights_batched) instead of four small launches per layer.  Layers pick their views up
    in layers.Conv2d.run when the plan is current (`flat.epoch`)."""

    def __init__(self, module, flat):
        from .layers import Conv2d, ConvTranspose2d
        lib = _lib.load()
        self.flat = flat
        base = flat.data.data_ptr()
        rows, self.layers, off = [], [], 0

        def take(nbytes):
            nonlocal off
            o = off
            off += (int(nbytes) + 255) // 256 * 256
            return o

        seen = set()
        for m in module.modules():
            if not isinstance(m, (Conv2d, ConvTranspose2d)) or id(m.weight) in seen:
                continue
            seen.add(id(m.weight))
            tr = isinstance(m, ConvTranspose2d)
            cin, cout = (m.weight.shape[0], m.weight.shape[1]) if tr else (m.weight.shape[1], m.weight.shape[0])
            kh, kw = m.weight.shape[2], m.weight.shape[3]
            ps_r = int(getattr(m, "_ps_r", 0))
            nf = int(lib.srk_packed_weight_bytes(cout, cin, kh, kw, 0))
            nb = int(lib.srk_packed_weight_bytes(cout, cin, kh, kw, 1))
            fo, bo = take(nf), take(nb)
            w_off = (m.weight.data_ptr() - base) // 4
            b_off, bp_off = -1, -1
            if m.bias is not None and ps_r > 1:
                b_off = (m.bias.data_ptr() - base) // 4
                bp_off = take(cout * 4)
            rows.append([w_off, fo, bo, cout, cin, kh, kw, int(tr), ps_r, b_off, bp_off, -1, -1, 0])
            self.layers.append((m, fo, nf, bo, nb, bp_off, cout, ps_r))
        self.n = len(rows)
        if self.n == 0:
            return
        dev = flat.data.device
        # one zeroed 4-byte scratch word per layer for the parallel max|w| pass (fp16 planes of the forward buffers)
        # (64 bytes apart: atomics on one cache line serialise)
        self.scratch_off = take(64 * len(rows))
        for i, r in enumerate(rows):
            r[11] = self.scratch_off + 64 * i
        self.buf = flat._alloc_store(max(off, 256), self.scratch_off)   # (scratch words last: zeroed together with the gradients)
        self.scratch = self.buf[self.scratch_off:self.scratch_off + 64 * len(rows)]
        # fast path (k_pack_fast): plain Conv2d filters are packed tile by tile from LDS, (Cout / 8) * ceil(Cin / 32)
        # blocks per layer; everything else (first layers, 64 -> 3 convs, deconvs, 9x9 kernels) takes the generic kernel
        fast = []
        if os.environ.get("SRK_PACK_FAST", "1") != "0":
            for i, r in enumerate(rows):
                cout, cin, kh, kw, tr = r[3], r[4], r[5], r[6], r[7]
                # (channel counts that leave no padding in either prepared layout: the kernel writes real channels only,
                #  and the zero groups of a padded contraction axis must exist -- they meet zero activations, 0 x junk)
                if not tr and kh * kw <= 25 and (cout % 64 == 0 or cout == 32) and (cin % 64 == 0 or cin in (16, 32, 48)) \
                        and r[1] >= 0 and r[2] >= 0:
                    r[12] = len(fast)
                    icc = (cin + 31) // 32
                    fast += [(i, lb) for lb in range((cout // 8) * icc)]
        self.n_fast = len(fast)
        self.fast_blocks = torch.tensor(fast, dtype=torch.int32, device=dev) if fast else None
        self.table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.epoch = -1
        # grid of the generic kernel: sized by the largest filter it still packs itself
        biggest = max([r[3] * r[4] * r[5] * r[6] for r in rows if r[12] < 0] or [256])
        self.blocks = max(1, min(int(os.environ.get("SRK_PACK_BLOCKS", "512")), (biggest + 255) // 256))
        for m, fo, nf, bo, nb, bp_off, cout, ps_r in self.layers:
            wpf = self.buf[fo:fo + (nf + 3) // 4 * 4].view(torch.float32)
            wpb = self.buf[bo:bo + (nb + 3) // 4 * 4].view(torch.float32)
            bp = self.buf[bp_off:bp_off + cout * 4].view(torch.float32) if bp_off >= 0 else None
            m._plan = [self, ps_r, wpf, bp, wpb, -1, -1]

    def pack(self, scratch_zeroed=False):
        """scratch_zeroed: the caller just zeroed the scratch words (FlatParams.zero_grad's fill covers them)."""
        if self.n == 0:
            return
        lib = _lib.load()
        if not scratch_zeroed:
            self.scratch.zero_()
        check(lib.srk_pack_weights_batched(ptr(self.flat.data), ptr(self.buf), ptr(self.table), self.n, -self.blocks,
                                           ptr(self.fast_blocks), self.n_fast, stream_ptr()), "srk_pack_weights_batched")
        self.epoch = self.flat.epoch
        from . import ops
        for lay in self.layers:  # host-side edits of a parameter (load_state_dict, init) invalidate its views
            m = lay[0]
            m._plan[5] = ops._ver(m.weight)
            m._plan[6] = -1 if m.bias is None else ops._ver(m.bias)

    def current(self):
        return self.n > 0 and self.epoch == self.flat.epoch


class _Group(dict):
    """param_groups entry: `group['lr'] /= 2` (edsr.py:131-133) must reach the device scalar."""

    def __init__(self, owner, **kw):
        super(_Group, self).__init__(**kw)
        self._owner = owner

    def __setitem__(self, k, v):
        super(_Group, self).__setitem__(k, v)
        if k == "lr":
            self._owner._set_lr(v)


class _FlatOptimizer(object):
    def __init__(self, params, lr):
        if not isinstance(params, FlatParams):
            params = FlatParams(params)
        self.flat = params
        dev = params.data.device
        self.lr_dev = torch.tensor([float(lr)], dtype=torch.float32, device=dev)
        self.scale_dev = torch.ones(1, dtype=torch.float32, device=dev)   # gradient scale (clip / DP average)
        self.norm_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._norm_ws = None
        self.param_groups = [_Group(self, lr=float(lr))]

    def _set_lr(self, v):
        self.lr_dev.fill_(float(v))

    def zero_grad(self, set_to_none=False, repack="always"):
        """Start of a train step: clear the flat gradient buffer (one memset) and re-pack every conv
        filter of the model for this step's forward/backward (one launch).
        repack="stale": skip the pack when the plan is still current -- SRGAN's discriminator at the start of a step was
        packed for the generator phase of the previous one and has not changed since.  Only for callers that tell their
        graph which FlatParams it updates (trainers.GraphedFn(flats=...)): inside a replay nobody can see that somebody
        rewrote the parameters in place, the unconditional pack is what makes that case work by itself."""
        from . import ops
        ops.amax_new_step(self.flat.amax_chunk)   # (zeroed by the fill below)
        self.flat.zero_grad()
        if repack != "stale" or not self.flat.plan.current():
            self.flat.plan.pack(scratch_zeroed=True)

    def clip_grad_norm(self, max_norm):
        """torch.nn.utils.clip_grad_norm(params, max_norm) (vdsr.py:149): computes the global L2
        norm on the device and stores min(1, max_norm/(norm+1e-6)) where the next step() reads it
        (the flat gradient buffer itself is left unscaled). Returns the device norm tensor."""
        from . import ops
        ops.join_side_streams()
        lib = _lib.load()
        if self._norm_ws is None:
            self._norm_ws = torch.empty(int(lib.srk_grad_norm_workspace_bytes()), dtype=torch.uint8,
                                        device=self.flat.data.device)
        check(lib.srk_grad_norm_clip(ptr(self.flat.grad), self.flat.numel, float(max_norm), ptr(self.norm_dev),
                                     ptr(self.scale_dev), ptr(self._norm_ws), stream_ptr()), "srk_grad_norm_clip")
        return self.norm_dev

    def reset_grad_scale(self):
        self.scale_dev.fill_(1.0)


class SGD(_FlatOptimizer):
    """torch.optim.SGD(lr, momentum, weight_decay, nesterov) — dampening 0 as in every reference call."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, nesterov=False):
        super(SGD, self).__init__(params, lr)
        self.momentum, self.weight_decay, self.nesterov = float(momentum), float(weight_decay), bool(nesterov)
        # zero-initialised buffer: mom*0 + g == g, i.e. torch's "buf = clone(grad)" first step
        self.buf = torch.zeros_like(self.flat.data) if momentum != 0.0 else None

    def step(self):
        from . import ops
        ops.join_side_streams()  # weight gradients forked onto the side stream
        lib = _lib.load()
        f = self.flat
        check(lib.srk_sgd_step(ptr(f.data), ptr(f.grad), ptr(self.buf), f.numel, 0.0, self.momentum,
                               self.weight_decay, int(self.nesterov), 0, ptr(self.lr_dev), ptr(self.scale_dev),
                               stream_ptr()), "srk_sgd_step")
        f.mark_changed()


class Adam(_FlatOptimizer):
    """torch.optim.Adam(lr, betas, eps, weight_decay) without amsgrad."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super(Adam, self).__init__(params, lr)
        self.betas, self.eps, self.weight_decay = (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.exp_avg = torch.zeros_like(self.flat.data)
        self.exp_avg_sq = torch.zeros_like(self.flat.data)
        self.step_dev = torch.zeros(2, dtype=torch.int32, device=self.flat.data.device)   # {step count, kernel's ticket}

    def step(self):
        from . import ops
        ops.join_side_streams()  # weight gradients forked onto the side stream
        lib = _lib.load()
        f = self.flat
        check(lib.srk_adam_step(ptr(f.data), ptr(f.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), f.numel, 0.0,
                                self.betas[0], self.betas[1], self.eps, self.weight_decay, ptr(self.step_dev),
                                ptr(self.lr_dev), ptr(self.scale_dev), stream_ptr()), "srk_adam_step")
        f.mark_changed()


def make_optimizer(kind, flat, lr):
    """The reference's per-model optimizer choices."""
    if kind == "srcnn":      # srcnn.py:79
        return SGD(flat, lr)
    if kind == "fsrcnn":     # fsrcnn.py:105-106
        return SGD(flat, lr, momentum=0.9)
    if kind == "vdsr":       # vdsr.py:86-90
        return SGD(flat, lr, momentum=0.9, weight_decay=1e-4)
    if kind in ("espcn", "lapsrn"):   # espcn.py:79, lapsrn.py:135
        return Adam(flat, lr)
    if kind in ("edsr", "srgan_g"):   # edsr.py:93, srgan.py:147
        return Adam(flat, lr, betas=(0.9, 0.999), eps=1e-8)
    if kind == "srgan_d":    # srgan.py:149
        return SGD(flat, lr / 100, momentum=0.9, nesterov=True)
    raise ValueError("unknown optimizer kind %r" % (kind,))
