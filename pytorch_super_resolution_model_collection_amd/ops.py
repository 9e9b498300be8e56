"""autograd.Function wrappers around the C ABI (include/srk.h).

Every function here enqueues hand-written HIP kernels on torch's current stream and returns
torch tensors that merely OWN the device memory.  Activations are logically NCHW (the reference's
module surface, e.g. edsr.py:146-152) but stored channels_last (= dense NHWC, the kernels'
layout).  Nothing in this file computes on the CPU and nothing falls back to ATen operators.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (ACT_BY_NAME, ACT_LRELU, ACT_NONE, ACT_PRELU, ACT_RELU, ALGO_AUTO, BwdMask, ConvDesc, Epilogue,
                   check, ptr, require_cuda, stream_ptr)

CL = torch.channels_last

# Convolution arithmetic used when a ConvCfg leaves algo = AUTO, per role:
#   infer     forward under torch.no_grad()
#   train_fwd forward that will be differentiated
#   bwd       data gradient
# Modes:
#   "mixed"  (default; round 3: infer = the fp32-faithful class too -- f16x3 costs ~10 % over bf16x3 on the c2 net, and a
#            drop-in for an fp32 reference should not default to narrower products; "bf16x3" is the fast option)
#            train_fwd = fp32-faithful class: f16x3 where it pays, else bf16x6 (exact 3-way operand split, 6 bf16 MFMAs per
#            product: measured 3.5e-7 rms / 8.8e-7 max error vs fp64 — tighter than the fp32 MFMA
#            kernel's 4.3e-7 / 1.2e-6 — at 1.5-2x its speed; shapes it does not cover run exact fp32),
#            bwd = bf16x3.  The training forward stays fp32-faithful so that ReLU / LeakyReLU masks are
#            decided on fp32-accurate values (a 5e-6 forward perturbation flips ~1e-5 of the units of
#            a bias-free ReLU net, and one flipped unit moves that layer's gradient by
#            ~1/sqrt(units) ~ 1e-2); gradients then only carry the ~5e-6 arithmetic error of the
#            bf16x3 data-gradient kernels.
#            train_fwd_tail = bf16x3: the training forward of the layers a model marks `_linear_tail` -- nothing
#            but convolutions, additions and pixel shuffles between their output and the loss (EDSR: body-end conv,
#            the two upsampler convs, the reconstruction conv; VDSR: the reconstruction conv), so no mask depends on
#            their rounding and the ~5e-6 of the backward kernels is what their forward may carry as well (EDSR step
#            7.11 -> 6.67 ms).  "bf16x6" keeps the fp32-faithful products there too; SRK_LINEAR_TAIL_X3=0 likewise.
#   "bf16x3" everything on the 3-term bf16 split (fastest; forward error ~1e-5, gradients subject to
#            the mask-flip sensitivity above).
#   "fp32"   everything exact fp32 (summation-order-level agreement with ATen/oneDNN).
_MODES = {"mixed": {"infer": _lib.ALGO_MFMA_BF16X6, "train_fwd": _lib.ALGO_MFMA_BF16X6, "bwd": ALGO_AUTO,
                    "train_fwd_tail": ALGO_AUTO},
          # fp32-faithful products everywhere they exist (inference included): what "mixed" costs when bf16x3's ~5e-6 is
          # not acceptable for the forward either
          "bf16x6": {"infer": _lib.ALGO_MFMA_BF16X6, "train_fwd": _lib.ALGO_MFMA_BF16X6, "bwd": ALGO_AUTO,
                     "train_fwd_tail": _lib.ALGO_MFMA_BF16X6},
          # ... and in both gradient kernels too (data gradient on the six-MFMA split, weight gradient on the exact fp32 MFMA
          # kernels): every product of a train step at fp32 accuracy -- the reference point for what the 3-MFMA backward costs
          "faithful": {"infer": _lib.ALGO_MFMA_BF16X6, "train_fwd": _lib.ALGO_MFMA_BF16X6, "bwd": _lib.ALGO_MFMA_BF16X6,
                       "train_fwd_tail": _lib.ALGO_MFMA_BF16X6},
          "bf16x3": {"infer": ALGO_AUTO, "train_fwd": ALGO_AUTO, "bwd": ALGO_AUTO, "train_fwd_tail": ALGO_AUTO},
          "fp32": {"infer": _lib.ALGO_MFMA, "train_fwd": _lib.ALGO_MFMA, "bwd": _lib.ALGO_MFMA,
                   "train_fwd_tail": _lib.ALGO_MFMA}}
_PRECISION = {"mode": "mixed"}


def set_precision(mode):
    """Select the convolution arithmetic: 'mixed' (default), 'bf16x3', 'bf16x6', 'faithful' or 'fp32' (see above)."""
    if mode not in _MODES:
        raise ValueError("precision must be one of %s" % sorted(_MODES))
    _PRECISION["mode"] = mode


def get_precision():
    return _PRECISION["mode"]


def backward_arithmetic():
    """What the data- and weight-gradient products run on in the current mode (bench.py states it next to every training
    number)."""
    mode = _PRECISION["mode"]
    if mode == "fp32":
        return "exact fp32 MFMA"
    if mode == "faithful":
        return "bf16x6 data gradients (6 MFMAs, ~3e-7 rms), exact fp32 MFMA weight gradients"
    return "bf16x3 (3 MFMAs, ~5e-6 rms)"


def _algo_for(cfg, role):
    return cfg.algo if cfg.algo != ALGO_AUTO else _MODES[_PRECISION["mode"]][role]


# ------------------------------------------------------------------------------------------------
# fp32-faithful products with three MFMAs: SRK_ALGO_MFMA_F16X3
# ------------------------------------------------------------------------------------------------
# Wherever a mode asks for the fp32-faithful class (ALGO_MFMA_BF16X6: six bf16 MFMAs per product) and the fp16 kernels
# cover the layer, the forward runs f16x3 instead: both operands scaled by a power of two so that their largest magnitude
# sits at 2^13..2^14, split into two fp16 planes (2 x 11 bits), three MFMAs, exact descale -- 1.2e-7 rms against 0.8e-7
# for plain fp32 products and 7.6e-8 for bf16x6 (tools/precision_check.py), at the bf16x3 rate.  The scale of the
# activations needs an upper bound of max|x| ON THE DEVICE: every kernel that can (conv_bfd.hip, conv_res2.hip) leaves
# the running maximum of what it stored in a 1 KB buffer of 16 slots (`y_amax`), attached to the output tensor as `_srk_amax`;
# tensors without one (network inputs, outputs of other kernels) get it from one srk_absmax pass.  SRK_F16X3=0: off.
F16X3 = os.environ.get("SRK_F16X3", "1") != "0"
F16X3_ALWAYS = os.environ.get("SRK_F16X3", "1") == "2"   # tests: take the srk_absmax pass for every untagged input
_AMAX_CHUNK_TENSORS = 128
_AMAX = {}   # device index -> [chunk tensor, slots used]
_AMAX_EPOCH = [0]


def amax_new_step(chunk=None):
    """Start of a train step (optimizer.zero_grad) or of a graph capture: the next running maximum comes from a fresh
    zeroed chunk (inside a captured step the zero fill is part of the graph, so every replay starts from zero), and
    maxima attached to tensors before this point are no longer trusted -- a captured step must recompute the maximum of
    its static input buffers inside the graph, where every replay sees the batch of that replay.
    chunk: a float32 tensor of a multiple of AMAX_FLOATS elements that the CALLER zeroes on the stream before the step's
    first kernel (optim.FlatParams keeps one next to its gradients: one fill for both); the step's first buffers come
    from it, further ones from chunks allocated (and zeroed) here."""
    _AMAX.clear()
    _AMAX_EPOCH[0] += 1
    if chunk is not None:
        idx = chunk.device.index if chunk.device.index is not None else torch.cuda.current_device()
        _AMAX[idx] = [chunk, 0, chunk.numel() // _lib.AMAX_FLOATS]


def _amax_alloc(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ent = _AMAX.get(idx)
    if ent is None or ent[1] >= ent[2]:
        ent = _AMAX[idx] = [torch.zeros(_AMAX_CHUNK_TENSORS * _lib.AMAX_FLOATS, dtype=torch.float32,
                                        device=torch.device("cuda", idx)), 0, _AMAX_CHUNK_TENSORS]
    lo = ent[1] * _lib.AMAX_FLOATS
    ent[1] += 1
    return ent[0][lo:lo + _lib.AMAX_FLOATS]


F16X3_MIN_PIXELS = int(os.environ.get("SRK_F16X3_MIN_PIXELS", str(256 * 2 * 256)))


def _f16x3_pays(d):
    """Where three MFMAs instead of six are worth the running-maximum bookkeeping: problems large enough to be bound by
    the matrix work (conv_bfd.hip's large-block configurations: >= two 256-pixel tiles per CU; first layers of that size
    leave the fp32 MFMA kernel for the row-packed one) and kernels with more than nine taps.  The 64-pixel blocks of a small 3x3
    problem are a latency chain -- measured on the SRGAN step (16 patches): 18.7 us with f16x3 against 19.1 us with
    bf16x6 per conv, and the maxima cost more than that."""
    return (d.KH * d.KW > 9 or F16X3_ALWAYS
            or d.N * d.OH * d.OW * ((d.Cout + 63) // 64) >= F16X3_MIN_PIXELS)


def _ver(t):
    """Version counter of `t` for the "unchanged since it was tagged" tests below.  Tensors created under
    torch.inference_mode() track none (reading `_version` raises): the constant stands in, i.e. a tag on an inference
    tensor is trusted for the tensor's lifetime -- nothing in this package writes an activation in place."""
    return -1 if t.is_inference() else t._version


def _tag_amax(t, slots):
    t._srk_amax = (slots, _ver(t), _AMAX_EPOCH[0])


def amax_of(x, compute=True):
    """The running-maximum buffer of |x| on the device: the producer's, or one srk_absmax pass.  compute=False: only
    where that pass is known to pay -- x carries the marker a conv kernel without amax support leaves on its output
    (the first layer of a net: one pass buys f16x3 for the whole trunk behind it); anything else (a BatchNorm output in
    front of every SRGAN conv) returns None and the caller keeps the six-MFMA arithmetic."""
    a = getattr(x, "_srk_amax", None)
    fresh = a is not None and a[1] == _ver(x) and a[2] == _AMAX_EPOCH[0]
    if fresh and a[0] is not None:
        return a[0]
    if not compute and not fresh:
        return None
    slots = _amax_alloc(x.device)
    xs = x if x.is_contiguous() or _is_nhwc_dense(x) else x.contiguous()
    check(_lib.load().srk_absmax(ptr(xs), xs.numel(), ptr(slots), stream_ptr()), "srk_absmax")
    try:
        _tag_amax(x, slots)
    except Exception:  # noqa: BLE001 -- (a tensor subclass without a __dict__: recomputed next time)
        pass
    return slots


_BOUND_SLOTS = {}   # (device index, bound) -> slots filled with the bound: no kernel runs for a declared maximum


def declare_absmax(x, bound):
    """The caller's promise that |x| <= bound everywhere (e.g. 1.0 for a ToTensor image batch, dataset.py:71): the
    fp32-faithful f16x3 kernels then scale x by that bound and the srk_absmax pass over x is not run.  The C ABI asks for
    slots whose maximum is >= max|x| (include/srk.h, srk_epilogue.x_amax), so a true bound is as good as the maximum (a loose
    one costs mantissa bits of the 22 the two fp16 planes hold: 2^-k of them for a bound 2^k too large); a FALSE one lets the
    scaled fp16 planes overflow to inf.  Returns x, tagged."""
    require_cuda(x)
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    key = (idx, float(bound))
    slots = _BOUND_SLOTS.get(key)
    if slots is None:
        slots = _BOUND_SLOTS[key] = torch.full((_lib.AMAX_FLOATS,), float(bound), dtype=torch.float32,
                                               device=torch.device("cuda", idx))
    _tag_amax(x, slots)
    return x


def _empty_cl(n, c, h, w, like):
    return torch.empty((n, c, h, w), dtype=torch.float32, device=like.device, memory_format=CL)


def _is_nhwc_dense(x):
    """Memory is dense NHWC (strides of size-1 dims are irrelevant — views may rewrite them)."""
    n, c, h, w = x.shape
    want = (h * w * c, 1, w * c, c)
    return all(sz == 1 or st == e for sz, st, e in zip(x.shape, x.stride(), want))


def _is_nchw_dense(x):
    n, c, h, w = x.shape
    want = (c * h * w, h * w, w, 1)
    return all(sz == 1 or st == e for sz, st, e in zip(x.shape, x.stride(), want))


def to_nhwc(x):
    """Logical NCHW tensor -> same values, channels_last storage (srk_nchw_to_nhwc if a copy is needed)."""
    require_cuda(x)
    if x.dim() != 4:
        raise RuntimeError("expected a 4-D NCHW tensor, got shape %s" % (tuple(x.shape),))
    if _is_nhwc_dense(x):
        return x
    n, c, h, w = x.shape
    if not _is_nchw_dense(x):
        raise RuntimeError("input must be NCHW-contiguous or channels_last (got strides %s)" % (x.stride(),))
    y = _empty_cl(n, c, h, w, x)
    lib = _lib.load()
    check(lib.srk_nchw_to_nhwc(ptr(x), ptr(y), n, c, h, w, stream_ptr()), "srk_nchw_to_nhwc")
    a = getattr(x, "_srk_amax", None)   # a permutation keeps the maximum (or the "worth a pass" marker)
    if a is not None and a[1] == _ver(x) and a[2] == _AMAX_EPOCH[0]:
        _tag_amax(y, a[0])
    return y


# ---- pre-masked gradients ---------------------------------------------------------------------------------------------
# A conv with a fused ReLU multiplies its incoming dy by (y > 0) in BOTH backward kernels, reading y twice more.  When the
# consumer of y is another conv whose data gradient runs on the wave-specialised kernel, that kernel applies the mask to
# the dx it writes instead (srk_conv2d_backward_data_relu: y is its own input x, read once at the output tile) and marks
# the tensor: `_srk_premasked = (address of y, version of dx)`.  The upstream backward skips its masks only if the
# gradient it receives carries the mark for ITS y and is untouched since (autograd's fan-in accumulation adds in place
# and bumps the version; a fresh sum carries no mark).  ReLU only: its 0/1 mask is idempotent, so a mark that got lost
# merely costs the second application.  SRK_PREMASK=0 switches the protocol off.
# The gradient of an INTERMEDIATE activation is not what autograd defines while the protocol runs (it is already
# multiplied by the ReLU gradient of its producer), so the protocol is active only inside `premasked_gradients()` -- the
# backward passes of this package's own training steps (trainers._backward), where only parameter gradients are read.
# A plain loss.backward(), torch.autograd.grad(loss, activation) or tensor.retain_grad() outside it sees standard gradients.
PREMASK = os.environ.get("SRK_PREMASK", "1") != "0"
PREMASK_STATS = {"masked_dx": 0, "masks_skipped": 0}   # (tests: how often each half of the protocol ran)
_PREMASK_DEPTH = [0]


class premasked_gradients(object):
    """Backward passes inside this context may hand pre-masked gradients from layer to layer (see PREMASK)."""

    def __enter__(self):
        _PREMASK_DEPTH[0] += 1
        return self

    def __exit__(self, *a):
        _PREMASK_DEPTH[0] -= 1


def _premask_on():
    return PREMASK and _PREMASK_DEPTH[0] > 0


def _is_relu_output(x):
    return getattr(x, "_srk_relu_out", None) == _ver(x)


def _premasked_for(dy, y):
    t = getattr(dy, "_srk_premasked", None)
    return _premask_on() and t is not None and t == (y.data_ptr(), _ver(dy))


def to_nchw(x):
    """channels_last tensor -> NCHW-contiguous copy (srk_nhwc_to_nchw). Used before Linear layers."""
    require_cuda(x)
    n, c, h, w = x.shape
    if _is_nchw_dense(x):
        return x if x.is_contiguous() else x.contiguous()
    x = to_nhwc(x)
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    check(lib.srk_nhwc_to_nchw(ptr(x), ptr(y), n, c, h, w, stream_ptr()), "srk_nhwc_to_nchw")
    return y


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return to_nchw(x)

    @staticmethod
    def backward(ctx, g):
        return to_nhwc(g if (_is_nchw_dense(g) or _is_nhwc_dense(g)) else g.contiguous())


def flatten_nchw(x):
    """out.view(B, -1) of the reference (srgan.py:75) — flatten in (C,H,W) order."""
    y = _ToNCHW.apply(x)
    return y.reshape(y.shape[0], -1)


# ------------------------------------------------------------------------------------------------
# Convolution
# ------------------------------------------------------------------------------------------------
class ConvCfg(object):
    """Static configuration of one Conv2d / ConvTranspose2d call."""
    __slots__ = ("stride", "pad", "transposed", "out_pad", "act", "slope", "ps_r", "algo", "tail")

    def __init__(self, stride=1, pad=0, transposed=False, out_pad=0, act=ACT_NONE, slope=0.0, ps_r=0,
                 algo=ALGO_AUTO):
        self.stride, self.pad, self.transposed, self.out_pad = int(stride), int(pad), bool(transposed), int(out_pad)
        self.act, self.slope, self.ps_r, self.algo = int(act), float(slope), int(ps_r), int(algo)
        self.tail = False   # training forward of a layer with no nonlinearity between its output and the loss


def _weight_dims(weight, transposed):
    if transposed:
        cin, cout, kh, kw = weight.shape
    else:
        cout, cin, kh, kw = weight.shape
    return cout, cin, kh, kw


def _make_desc(x_shape, weight, cfg, role="infer"):
    lib = _lib.load()
    n, c, h, w = x_shape
    cout, cin, kh, kw = _weight_dims(weight, cfg.transposed)
    if c != cin:
        raise RuntimeError("conv: input has %d channels, weight expects %d" % (c, cin))
    oh = lib.srk_conv_out_dim(h, kh, cfg.stride, cfg.pad, int(cfg.transposed), cfg.out_pad)
    ow = lib.srk_conv_out_dim(w, kw, cfg.stride, cfg.pad, int(cfg.transposed), cfg.out_pad)
    if oh <= 0 or ow <= 0:
        raise RuntimeError("conv: empty output for input %s kernel %dx%d" % (tuple(x_shape), kh, kw))
    algo = _algo_for(cfg, role)
    return ConvDesc(n, h, w, cin, oh, ow, cout, kh, kw, cfg.stride, cfg.pad, int(cfg.transposed), cfg.out_pad, algo)


def pack_weight_fwd(weight, transposed, ps_r):
    lib = _lib.load()
    cout, cin, kh, kw = _weight_dims(weight, transposed)
    nbytes = int(lib.srk_packed_weight_bytes(cout, cin, kh, kw, 0))
    wp = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=weight.device)
    check(lib.srk_pack_weight_fwd(ptr(weight), ptr(wp), cout, cin, kh, kw, int(transposed), int(ps_r), stream_ptr()),
          "srk_pack_weight_fwd")
    return wp


def pack_weight_bwd(weight, transposed, ps_r=0):
    lib = _lib.load()
    cout, cin, kh, kw = _weight_dims(weight, transposed)
    nbytes = int(lib.srk_packed_weight_bytes(cout, cin, kh, kw, 1))
    wp = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=weight.device)
    check(lib.srk_pack_weight_bwd(ptr(weight), ptr(wp), cout, cin, kh, kw, int(transposed), int(ps_r), stream_ptr()),
          "srk_pack_weight_bwd")
    return wp


def pack_bias_ps(bias, ps_r):
    if bias is None or ps_r <= 1:
        return bias
    lib = _lib.load()
    bp = torch.empty_like(bias)
    check(lib.srk_pack_bias_ps(ptr(bias), ptr(bp), bias.numel(), int(ps_r), stream_ptr()), "srk_pack_bias_ps")
    return bp


# BatchNorm column sums from the producing conv's epilogue (srk_epilogue.bn_partial): a block that runs conv -> BatchNorm
# in training asks for them around the conv call; a kernel that keeps them (k_c64: the per-tile 64 -> 64 3x3 kernel of
# SRGAN's generator) tags its output, and _BatchNorm.forward then skips its own pass over the activation.
BN_PARTIAL = os.environ.get("SRK_BN_PARTIAL", "1") != "0"
_BN_REQ = [0]


class bn_partial_request(object):
    """with ops.bn_partial_request(on): y = conv.run(x)   -- y may carry `_srk_bn_partial` afterwards"""

    def __init__(self, on=True):
        self.on = bool(on) and BN_PARTIAL

    def __enter__(self):
        if self.on:
            _BN_REQ[0] += 1
        return self

    def __exit__(self, *exc):
        if self.on:
            _BN_REQ[0] -= 1
        return False


def conv_forward_raw(x, wp, bias_p, weight_shape_src, cfg, prelu_w=None, residual=None, role="infer", x_nchw=False):
    """Launch srk_conv2d_forward on already-packed weights. x must be NHWC-dense (or NCHW with x_nchw)."""
    lib = _lib.load()
    d = _make_desc(x.shape, weight_shape_src, cfg, role)
    d.x_nchw = int(bool(x_nchw))
    r = cfg.ps_r if cfg.ps_r > 1 else 1
    y = _empty_cl(d.N, d.Cout // (r * r), d.OH * r, d.OW * r, x)
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise RuntimeError("conv: residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    ep = Epilogue(ptr(bias_p), ptr(prelu_w), ptr(residual), cfg.slope, cfg.act,
                  0 if prelu_w is None else prelu_w.numel(), cfg.ps_r, None, None, None)
    bnp = None
    if (_BN_REQ[0] > 0 and role != "infer" and d.Cin == 64 and d.Cout == 64 and d.KH == 3 and d.KW == 3 and r == 1
            and cfg.act == ACT_NONE):
        # (the shape k_c64 covers; whether THAT kernel runs is the library's decision -- asked after the call)
        bnp = torch.empty((d.N * ((d.OH + 7) // 8) * ((d.OW + 7) // 8), 2 * d.Cout), dtype=torch.float64, device=x.device)
        ep.bn_partial = ptr(bnp)
    ya = None
    if d.algo == _lib.ALGO_MFMA_F16X3:      # asked for by name (ConvCfg.algo): the maximum is computed if nobody left one
        ep.x_amax = ptr(amax_of(x))
        ya = _amax_alloc(y.device)
        ep.y_amax = ptr(ya)
    elif F16X3 and d.Cout >= 8 and (d.Cin >= 8 or d.Cin <= 4) and _f16x3_pays(d):
        # fp32-faithful class on a layer the fp16 kernels cover: three MFMAs per product instead of six
        if d.algo == _lib.ALGO_MFMA_BF16X6 and lib.srk_conv2d_f16x3_supported(ctypes.byref(d), ctypes.byref(ep), ptr(y)):
            # (a first layer's input is the network input: one pass over a <= 4-channel tensor)
            xa = amax_of(x, compute=F16X3_ALWAYS or d.Cin <= 4)
            if xa is not None:
                ep.x_amax = ptr(xa)
                d.algo = _lib.ALGO_MFMA_F16X3
        if d.algo in (_lib.ALGO_MFMA_F16X3, _lib.ALGO_MFMA_BF16X6) and not os.environ.get("SRK_NO_YAMAX"):
            # (a layer of the fp32-faithful class feeds layers of that class: leave them the maximum of the output)
            ya = _amax_alloc(y.device)      # filled by the kernels of conv_bfd.hip (checked below)
            ep.y_amax = ptr(ya)
    if d.x_nchw and d.algo not in (ALGO_AUTO, _lib.ALGO_MFMA_BF16X3, _lib.ALGO_MFMA_F16X3):
        x = to_nhwc(x)      # only the bf16x3 / f16x3 first-layer kernels read an NCHW input in place
        d.x_nchw = 0
    # what the dispatched kernel did comes back in a host struct of the caller's (srk_conv_result)
    res = _lib.ConvResult()
    check(lib.srk_conv2d_forward_ex(ctypes.byref(d), ptr(x), ptr(wp), ptr(y), ctypes.byref(ep), ctypes.byref(res), stream_ptr()),
          "srk_conv2d_forward")
    if bnp is not None:
        rows_p = int(res.bn_partial_rows)
        if rows_p > 0:
            y._srk_bn_partial = (bnp, rows_p, _ver(y))
    if ya is not None and res.wrote_amax:
        _tag_amax(y, ya)
    elif F16X3 and d.algo in (_lib.ALGO_MFMA, _lib.ALGO_MFMA_BF16X6):
        _tag_amax(y, None)   # a faithful-class conv whose kernel leaves no maximum: worth one srk_absmax pass downstream
    return y


# Weight gradients on a second HIP stream.  In flat-buffer mode (optim.FlatParams) the weight-gradient kernels
# accumulate straight into the model's gradient buffer and autograd never looks at their result, so a layer's wgrad
# (+ slab reduce) can run concurrently with the same layer's data-gradient kernel and with the backward of the layers
# below it.
# Measured on MI355X / ROCm 7.2 (tools/edsr_small_batch.py, tools/graph_conc.py): hipGraph replay does not run the
# forked branch concurrently (a two-branch graph of small kernels replays slower than the same kernels in one chain),
# so the strong-scaled shard (EDSR x4, 16 patches per GPU) gains only 2 % (2.72 -> 2.65 ms) and the full batch loses
# 8 % (9.04 -> 9.76 ms, two co-resident kernels fighting over LDS / cache).  OFF by default; SRK_WGRAD_STREAM=1 or
# ops.WGRAD_SIDE_STREAM = True enables it.  Everything forked is joined by join_side_streams(), which runs at the end
# of every backward pass (autograd engine callback) and in FlatParams / the optimizers / DataParallel.
WGRAD_SIDE_STREAM = os.environ.get("SRK_WGRAD_STREAM", "0") == "1"
_SIDE = {}   # device index -> [stream, tensors kept alive until the join]


def _side(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ent = _SIDE.get(idx)
    if ent is None:
        ent = [torch.cuda.Stream(device=idx), []]
        _SIDE[idx] = ent
    return ent


def join_side_streams():
    """Every weight gradient of the backward passes so far is enqueued on / ordered before the current stream: launches
    the deferred grouped weight gradients that are still pending (flush_wgrads) and makes the current stream wait for
    any forked onto the side stream.  Runs automatically at the end of every backward pass (autograd engine callback),
    so `loss.backward(); p.grad...` is safe; optimizers, zero_grad, clipping and the gradient exchange call it too."""
    if _PENDING:
        flush_wgrads()
    for idx, ent in _SIDE.items():
        if ent[1]:
            torch.cuda.current_stream(idx).wait_stream(ent[0])
            ent[1] = []


# ------------------------------------------------------------------------------------------------
# Deferred, grouped weight gradients
# ------------------------------------------------------------------------------------------------
# The data-gradient chain (layer L's dx feeds layer L-1) is the critical path of a backward pass; a layer's WEIGHT
# gradient feeds nothing until the optimizer / the gradient exchange.  In flat-buffer mode (optim.FlatParams: kernels
# accumulate straight into the flat gradient buffer, autograd never sees dw) the conv backward therefore only RECORDS
# (x, dy, mask) and the weight gradients of a whole backward pass are launched afterwards, grouped by geometry:
# srk_conv2d_backward_weight_grouped runs e.g. the 33 body convs of EDSR as ONE launch + one reduce launch in which every
# block walks ~40 tiles of one layer, instead of 66 launches of one-tile blocks that each write a 73 KB partial slab
# (strong-scaled shard of 16 patches: 0.89 ms -> 0.1x of the step).  The flush happens at the end of the backward pass
# (autograd engine callback) — `loss.backward(); p.grad` keeps working — or, under data parallelism, group by group
# from dp.DataParallel.exchange(), which sends each group's bucket off while the next group computes.
DEFER_WGRAD = os.environ.get("SRK_DEFER_WGRAD", "1") != "0"
WGRAD_GROUP_MAX = int(os.environ.get("SRK_WGRAD_GROUP_MAX", "0"))   # > 0: split geometry groups into chunks of this many layers
# Deferral trades launches and partial-slab traffic for cache locality: a weight gradient launched right behind its
# layer's data gradient finds dy (and x) in the 256 MB Infinity Cache, one launched at the end of the pass reads them
# from HBM.  Records are therefore flushed (grouped) as soon as their x + dy bytes exceed DEFER_MAX_BYTES: a 16-patch
# EDSR shard (8 MB per layer) defers ~19 layers at a time, a 128-patch batch (67 MB per layer) falls back to
# one-or-two-layer groups, i.e. the per-layer behaviour.
DEFER_MAX_BYTES = int(float(os.environ.get("SRK_DEFER_MAX_MB", "160")) * (1 << 20))
_PENDING = []                      # [(key, desc, x, dy, y_mask, slope, wacc, bacc)] in backward order
_DEFER = {"queued": False, "manual": 0, "bytes": 0}


class manual_wgrad_flush(object):
    """Context: the end-of-backward callback leaves the recorded weight gradients pending; the caller flushes them
    (flush_wgrads) — used by the trainers to interleave the grouped launches with the gradient exchange."""

    def __enter__(self):
        _DEFER["manual"] += 1

    def __exit__(self, *a):
        _DEFER["manual"] -= 1


def _auto_flush():
    _DEFER["queued"] = False
    if not _DEFER["manual"]:
        flush_wgrads()


def pending_wgrad_groups(max_layers=None):
    """The recorded weight gradients as launch groups [[record, ...], ...]: same geometry (and bias / no bias), backward
    order, no two records of a group writing the same dw (shared weights), at most `max_layers` layers per group."""
    cap = max_layers or WGRAD_GROUP_MAX or 0
    groups, open_by_key = [], {}
    for rec in _PENDING:
        key, wptr = rec[0], rec[6].data_ptr()
        g = open_by_key.get(key)
        if g is None or wptr in g[1] or (cap and len(g[0]) >= cap):
            g = ([], set())
            groups.append(g)
            open_by_key[key] = g
        g[0].append(rec)
        g[1].add(wptr)
    return [g[0] for g in groups]


def launch_wgrad_group(recs):
    """One srk_conv2d_backward_weight_grouped call for records of one geometry (beta = 1: accumulate into the flat
    gradient views)."""
    lib = _lib.load()
    n = len(recs)
    d = recs[0][1]
    vp = ctypes.c_void_p
    xs = (vp * n)(*[r[2].data_ptr() for r in recs])
    dys = (vp * n)(*[r[3].data_ptr() for r in recs])
    masks = (BwdMask * n)(*[BwdMask(None if r[4] is None else r[4].data_ptr(), r[5]) for r in recs])
    dws = (vp * n)(*[r[6].data_ptr() for r in recs])
    has_bias = recs[0][7] is not None
    dbs = (vp * n)(*[r[7].data_ptr() for r in recs]) if has_bias else None
    dev = recs[0][3].device
    ws_bytes = int(lib.srk_conv2d_backward_weight_grouped_workspace_bytes(ctypes.byref(d), n))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    check(lib.srk_conv2d_backward_weight_grouped(ctypes.byref(d), n, xs, dys, masks, dws, dbs, 1.0, ptr(ws), ws.numel(),
                                                 stream_ptr()), "srk_conv2d_backward_weight_grouped")
    return ws   # (deferred reductions: the caller keeps the partial slabs alive until srk_wgrad_reduce_flush)


def flush_wgrads(on_group=None, max_layers=None):
    """Launch every recorded weight gradient (grouped by geometry).  `on_group(records)` runs after each group's
    launches — dp.DataParallel uses it to start that group's gradient bucket on its way."""
    if not _PENDING:
        return 0
    groups = pending_wgrad_groups(max_layers)
    del _PENDING[:]
    _DEFER["bytes"] = 0
    if on_group is not None or len(groups) < 2 or not MERGE_REDUCES:
        for recs in groups:
            launch_wgrad_group(recs)
            if on_group is not None:
                on_group(recs)
        return len(groups)
    # several launch groups and nobody waiting for one of them in particular: their slab reductions run as ONE launch
    # behind the last group (srk_wgrad_reduce_defer / _flush; the workspaces stay alive until then)
    lib = _lib.load()
    prev = lib.srk_wgrad_reduce_defer(1)
    keep = []   # the workspaces of the queued jobs: alive until the queue has run, on the error path as well
    try:
        for recs in groups:
            keep.append(launch_wgrad_group(recs))
        check(lib.srk_wgrad_reduce_flush(stream_ptr()), "srk_wgrad_reduce_flush")
    finally:
        lib.srk_wgrad_reduce_defer(prev)
        # a failure above may leave queued jobs behind: run them now (their slabs are still held by `keep`) rather than
        # inside somebody else's flush
        lib.srk_wgrad_reduce_flush(stream_ptr())
        del keep[:]
    return len(groups)


def drop_pending_wgrads():
    del _PENDING[:]
    _DEFER["bytes"] = 0


MERGE_REDUCES = os.environ.get("SRK_MERGE_REDUCES", "1") != "0"   # 0: every weight-gradient launch group reduces its slabs itself
FUSE_SKIP_GRAD = os.environ.get("SRK_FUSE_SKIP_GRAD", "1") != "0"  # 0: residual blocks sum their gradient fan-in with srk_axpby


class GradBox(object):
    """Gradient fan-in of a residual block without a separate add pass.  The block output gradient reaches the
    backward of the block's LAST conv as the gradient of its fused residual input; that backward parks it here
    (`res_box`) instead of returning it to autograd, and the backward of the block's FIRST conv -- which autograd
    runs later, and whose input is the same tensor as the skip -- hands it to srk_conv2d_backward_data as `add_to`
    (`add_box`), so dx = conv1^T(...) + dy_block comes out of the data-gradient kernel's epilogue."""
    __slots__ = ("g",)

    def __init__(self):
        self.g = None


class _Conv2d(torch.autograd.Function):
    """y = PS_r(act(conv(x, w) + b)) + residual, training-capable for act in {none, relu, lrelu}."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, cfg, packed, res_box=None, add_box=None):
        ctx.res_box, ctx.add_box = res_box, add_box
        require_cuda(x, weight, bias, residual)
        x = to_nhwc(x)
        if residual is not None:
            residual = to_nhwc(residual)
        ctx.wpb = None
        if packed is not None:
            wp, bp = packed[0], packed[1]
            if len(packed) > 2:
                ctx.wpb = packed[2]  # data-gradient layout already packed by the model's PackPlan
        else:
            wp = pack_weight_fwd(weight, cfg.transposed, cfg.ps_r)
            bp = pack_bias_ps(bias, cfg.ps_r)
        y = conv_forward_raw(x, wp, bp, weight, cfg, None, residual,
                             "train_fwd_tail" if (cfg.tail and LINEAR_TAIL_X3) else "train_fwd")
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.weight_ref = weight
        ctx.bias_ref = bias
        need_mask = cfg.act in (ACT_RELU, ACT_LRELU)
        ctx.x_relu_out = _is_relu_output(x)      # x = relu(conv(..)): this conv's dx may leave pre-masked (see PREMASK)
        if cfg.act == ACT_RELU and residual is None and cfg.ps_r <= 1:
            y._srk_relu_out = _ver(y)
        ctx.save_for_backward(x, weight, y if need_mask else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        cfg = ctx.cfg
        x, weight, y = ctx.saved_tensors
        premasked = y is not None and cfg.act == ACT_RELU and _premasked_for(dy, y)
        PREMASK_STATS["masks_skipped"] += 1 if premasked else 0
        dy = to_nhwc(dy if (_is_nchw_dense(dy) or _is_nhwc_dense(dy)) else dy.contiguous())
        d = _make_desc(x.shape, weight, cfg, "bwd")
        dres = dy if ctx.has_res else None
        if ctx.has_res and ctx.res_box is not None:
            ctx.res_box.g, dres = dy, None  # consumed by the first conv of the block (GradBox)
        dyc = dy
        if cfg.ps_r > 1:
            r = cfg.ps_r
            # the bf16x3 data- and weight-gradient kernels read the pixel-shuffled dy directly (they un-shuffle while
            # staging their tiles); everything else gets an explicit depth-from-space pass first
            fused_ps = (d.algo in (ALGO_AUTO, _lib.ALGO_MFMA_BF16X3) and not cfg.transposed and cfg.stride == 1
                        and d.KH <= 3 and d.KW <= 3 and d.Cin >= 8 and (d.Cout // (r * r)) % 8 == 0
                        and os.environ.get("SRK_FUSE_PS_BWD", "1") != "0")
            ctx_wpb = ctx.wpb
            if fused_ps:
                d.dy_ps_r = r
            else:
                ctx_wpb = None  # the PackPlan packs PS layers for the fused path ((i, j, c) contraction order)
                dyc = _empty_cl(d.N, d.Cout, d.OH, d.OW, dy)
                check(lib.srk_pixel_shuffle_backward(ptr(dy), ptr(dyc), d.N, d.OH, d.OW, d.Cout // (r * r), r,
                                                     stream_ptr()), "srk_pixel_shuffle_backward")
        mask = None
        if y is not None and not premasked:
            mask = BwdMask(ptr(y), cfg.slope if cfg.act == ACT_LRELU else 0.0)
        mref = ctypes.byref(mask) if mask is not None else None
        dx = dw = db = None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        wref, bref = ctx.weight_ref, ctx.bias_ref
        wacc = getattr(wref, "_srk_grad", None)
        bacc = getattr(bref, "_srk_grad", None) if ctx.has_bias else None
        flat_mode = wacc is not None and (not ctx.has_bias or bacc is not None)
        if need_w and flat_mode:
            # flat-buffer mode: accumulate straight into the (pre-zeroed) gradient views — on the side stream, forked
            # here (after dy / x are ready, before the data gradient is launched) so the two kernels overlap
            ws_bytes = 0 if (DEFER_WGRAD and not WGRAD_SIDE_STREAM) else \
                lib.srk_conv2d_backward_weight_workspace_bytes(ctypes.byref(d))
            if DEFER_WGRAD and not WGRAD_SIDE_STREAM:
                key = (d.N, d.H, d.W, d.Cin, d.OH, d.OW, d.Cout, d.KH, d.KW, d.stride, d.pad, d.transposed, d.out_pad,
                       d.algo, d.dy_ps_r, bacc is not None, str(dy.device))
                _PENDING.append((key, d, x, dyc, y if mask is not None else None,
                                 cfg.slope if cfg.act == ACT_LRELU else 0.0, wacc, bacc))
                _DEFER["bytes"] += 4 * (x.numel() + dyc.numel())
                if _DEFER["bytes"] > DEFER_MAX_BYTES:
                    flush_wgrads()   # (also under manual_wgrad_flush: these gradients are simply final before the exchange)
                elif not _DEFER["queued"]:   # first record of this backward pass: flush when the engine finishes it
                    _DEFER["queued"] = True
                    torch.autograd.Variable._execution_engine.queue_callback(_auto_flush)
            elif WGRAD_SIDE_STREAM:
                side, keep = _side(dy.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=dy.device)
                    check(lib.srk_conv2d_backward_weight(ctypes.byref(d), ptr(x), ptr(dyc), mref, ptr(wacc), ptr(bacc),
                                                         1.0, ptr(ws), ws.numel(), stream_ptr()),
                          "srk_conv2d_backward_weight")
                if not keep:  # first fork of this backward pass: join when the autograd engine finishes it
                    torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
                keep.append((x, dyc, y, ws))  # the caching allocator must not recycle these before the join
            else:
                ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=dy.device)
                check(lib.srk_conv2d_backward_weight(ctypes.byref(d), ptr(x), ptr(dyc), mref, ptr(wacc), ptr(bacc),
                                                     1.0, ptr(ws), ws.numel(), stream_ptr()),
                      "srk_conv2d_backward_weight")
        if ctx.needs_input_grad[0]:
            plan_wpb = ctx.wpb if cfg.ps_r <= 1 else ctx_wpb
            wpb = plan_wpb if plan_wpb is not None else pack_weight_bwd(weight, cfg.transposed, d.dy_ps_r)
            dx = _empty_cl(d.N, d.Cin, d.H, d.W, dy)
            add_to = None
            if ctx.add_box is not None and ctx.add_box.g is not None:
                add_to, ctx.add_box.g = ctx.add_box.g, None
                if tuple(add_to.shape) != tuple(dx.shape):
                    raise RuntimeError("conv backward: skip gradient %s does not match dx %s"
                                       % (tuple(add_to.shape), tuple(dx.shape)))
            if (ctx.x_relu_out and _premask_on() and add_to is None and d.dy_ps_r <= 1 and not x.retains_grad
                    and lib.srk_conv2d_backward_data_relu_supported(ctypes.byref(d), ptr(dyc), ptr(dx), mref)):
                rc = lib.srk_conv2d_backward_data_relu(ctypes.byref(d), ptr(dyc), ptr(wpb), ptr(dx), mref, ptr(x),
                                                       stream_ptr())
                if rc == _lib.ERR_UNSUPPORTED:
                    # the applicability test passed but no tile of the wave-specialised kernel fits this problem: the
                    # standard data gradient, dx unmarked (the layer below applies its own mask) -- as the forward does
                    check(lib.srk_conv2d_backward_data(ctypes.byref(d), ptr(dyc), ptr(wpb), ptr(dx), mref, None,
                                                       stream_ptr()), "srk_conv2d_backward_data")
                else:
                    check(rc, "srk_conv2d_backward_data_relu")
                    dx._srk_premasked = (x.data_ptr(), _ver(dx))
                    PREMASK_STATS["masked_dx"] += 1
            else:
                check(lib.srk_conv2d_backward_data(ctypes.byref(d), ptr(dyc), ptr(wpb), ptr(dx), mref, ptr(add_to),
                                                   stream_ptr()), "srk_conv2d_backward_data")
        if need_w and not flat_mode:
            ws_bytes = lib.srk_conv2d_backward_weight_workspace_bytes(ctypes.byref(d))
            ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=dy.device)
            dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            db = torch.empty(d.Cout, dtype=torch.float32, device=dy.device) if ctx.has_bias else None
            check(lib.srk_conv2d_backward_weight(ctypes.byref(d), ptr(x), ptr(dyc), mref, ptr(dw), ptr(db), 0.0,
                                                 ptr(ws), ws.numel(), stream_ptr()),
                  "srk_conv2d_backward_weight")
        return dx, dw, db, dres, None, None, None, None


def conv2d(x, weight, bias=None, residual=None, cfg=None, packed=None, res_box=None, add_box=None):
    """Fused conv for training (act limited to none/relu/lrelu; no act together with residual/ps).
    res_box / add_box: see GradBox (both convs of a residual block share one box)."""
    cfg = cfg or ConvCfg()
    if cfg.act not in (ACT_NONE, ACT_RELU, ACT_LRELU):
        raise RuntimeError("conv2d (autograd path) fuses only none/relu/lrelu; use conv2d_infer or an unfused act")
    if cfg.act != ACT_NONE and (residual is not None or cfg.ps_r > 1):
        raise RuntimeError("conv2d (autograd path): activation cannot be fused together with residual/pixel-shuffle")
    return _Conv2d.apply(x, weight, bias, residual, cfg, packed, res_box, add_box)


LINEAR_TAIL_X3 = os.environ.get("SRK_LINEAR_TAIL_X3", "1") != "0"


# ---- residual block with both convolutions in one launch (srk_resblock2_*) ----------------------------------------
RES2 = os.environ.get("SRK_RES2", "1") != "0"   # 0: residual blocks always run their two convs as separate launches


def _weight_grad(d, x, dyc, weight, bias, need_w):
    """Weight / bias gradient of one conv without an activation mask: recorded for the grouped deferred launch (flat
    gradient buffers) or computed now.  Returns (dw, db) for autograd (None, None when accumulated in place)."""
    if not need_w:
        return None, None
    lib = _lib.load()
    wacc = getattr(weight, "_srk_grad", None)
    bacc = getattr(bias, "_srk_grad", None) if bias is not None else None
    flat_mode = wacc is not None and (bias is None or bacc is not None)
    if flat_mode and DEFER_WGRAD and not WGRAD_SIDE_STREAM:
        key = (d.N, d.H, d.W, d.Cin, d.OH, d.OW, d.Cout, d.KH, d.KW, d.stride, d.pad, d.transposed, d.out_pad,
               d.algo, d.dy_ps_r, bacc is not None, str(dyc.device))
        _PENDING.append((key, d, x, dyc, None, 0.0, wacc, bacc))
        _DEFER["bytes"] += 4 * (x.numel() + dyc.numel())
        if _DEFER["bytes"] > DEFER_MAX_BYTES:
            flush_wgrads()
        elif not _DEFER["queued"]:
            _DEFER["queued"] = True
            torch.autograd.Variable._execution_engine.queue_callback(_auto_flush)
        return None, None
    ws = torch.empty(max(int(lib.srk_conv2d_backward_weight_workspace_bytes(ctypes.byref(d))), 16), dtype=torch.uint8,
                     device=dyc.device)
    if flat_mode:
        check(lib.srk_conv2d_backward_weight(ctypes.byref(d), ptr(x), ptr(dyc), None, ptr(wacc), ptr(bacc), 1.0,
                                             ptr(ws), ws.numel(), stream_ptr()), "srk_conv2d_backward_weight")
        return None, None
    dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
    db = torch.empty(d.Cout, dtype=torch.float32, device=dyc.device) if bias is not None else None
    check(lib.srk_conv2d_backward_weight(ctypes.byref(d), ptr(x), ptr(dyc), None, ptr(dw), ptr(db), 0.0, ptr(ws),
                                         ws.numel(), stream_ptr()), "srk_conv2d_backward_weight")
    return dw, db


def resblock2_applicable(x, w1, w2):
    """True when srk_resblock2_* can run this block: 3x3 C -> C -> C filters, a problem small enough that the fused
    tile pays (srk_resblock2_supported), and a precision mode whose kernels exist fused (not 'fp32')."""
    if not RES2 or x.dim() != 4 or os.environ.get("SRK_FORCE_ALGO"):
        return False
    n, c, h, w = x.shape
    if tuple(w1.shape) != (c, c, 3, 3) or tuple(w2.shape) != (c, c, 3, 3):
        return False
    mode = _MODES[_PRECISION["mode"]]
    if mode["train_fwd"] not in (ALGO_AUTO, _lib.ALGO_MFMA_BF16X3, _lib.ALGO_MFMA_BF16X6) or \
            mode["bwd"] not in (ALGO_AUTO, _lib.ALGO_MFMA_BF16X3, _lib.ALGO_MFMA_BF16X6):
        return False
    return bool(_lib.load().srk_resblock2_supported(n, h, w, c))


class _ResBlock2(torch.autograd.Function):
    """out = x + conv2(relu(conv1(x))) (base_networks.py:128-150, norm=None, activation='relu') as ONE forward and ONE
    data-gradient launch; the two weight gradients take the usual (grouped, deferred) path, without masks: the
    backward kernel hands out the already masked intermediate gradient."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, packed1, packed2):
        lib = _lib.load()
        require_cuda(x, w1, b1, w2, b2)
        x = to_nhwc(x)
        n, c, h, w = x.shape
        wp1 = packed1[0] if packed1 is not None else pack_weight_fwd(w1, False, 0)
        wp2 = packed2[0] if packed2 is not None else pack_weight_fwd(w2, False, 0)
        ctx.wpb1 = packed1[2] if packed1 is not None and len(packed1) > 2 else None
        ctx.wpb2 = packed2[2] if packed2 is not None and len(packed2) > 2 else None
        mid = _empty_cl(n, c, h, w, x)
        out = _empty_cl(n, c, h, w, x)
        algo = _MODES[_PRECISION["mode"]]["train_fwd"]
        xa = ya = None
        if F16X3:
            ya = _amax_alloc(x.device)
            if algo == _lib.ALGO_MFMA_BF16X6:   # fp32-faithful class: f16x3 (see F16X3 above)
                xa = amax_of(x, compute=F16X3_ALWAYS)
                if xa is not None:
                    algo = _lib.ALGO_MFMA_F16X3
        check(lib.srk_resblock2_forward(n, h, w, c, ptr(x), ptr(wp1), ptr(b1), ptr(wp2), ptr(b2), ptr(mid), ptr(out),
                                        algo, ptr(xa), ptr(ya), stream_ptr()), "srk_resblock2_forward")
        if ya is not None:
            _tag_amax(out, ya)
        ctx.refs = (w1, b1, w2, b2)
        ctx.save_for_backward(x, mid)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, mid = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.refs
        n, c, h, w = x.shape
        dy = to_nhwc(dy if (_is_nchw_dense(dy) or _is_nhwc_dense(dy)) else dy.contiguous())
        wpb1 = ctx.wpb1 if ctx.wpb1 is not None else pack_weight_bwd(w1, False, 0)
        wpb2 = ctx.wpb2 if ctx.wpb2 is not None else pack_weight_bwd(w2, False, 0)
        dmid = _empty_cl(n, c, h, w, dy)
        dx = _empty_cl(n, c, h, w, dy)
        check(lib.srk_resblock2_backward_data(n, h, w, c, ptr(dy), ptr(wpb2), ptr(wpb1), ptr(mid), ptr(dmid), ptr(dx),
                                              _MODES[_PRECISION["mode"]]["bwd"], stream_ptr()),
              "srk_resblock2_backward_data")
        cfg = ConvCfg(1, 1, False, 0, ACT_NONE, 0.0, 0)
        d = _make_desc(x.shape, w1, cfg, "bwd")
        need = ctx.needs_input_grad
        # backward order of the separate convs: conv2 first (x = mid, dy = dy), then conv1 (x = x, dy = dmid)
        dw2, db2 = _weight_grad(_make_desc(x.shape, w2, cfg, "bwd"), mid, dy, w2, b2, need[3] or (b2 is not None and need[4]))
        dw1, db1 = _weight_grad(d, x, dmid, w1, b1, need[1] or (b1 is not None and need[2]))
        return (dx if need[0] else None), dw1, db1, dw2, db2, None, None


def resblock2(x, w1, b1, w2, b2, packed1=None, packed2=None):
    return _ResBlock2.apply(x, w1, b1, w2, b2, packed1, packed2)


def conv2d_infer(x, weight, bias=None, residual=None, cfg=None, prelu_w=None, packed=None):
    """No-grad fully fused conv: any activation + residual + pixel shuffle in one kernel."""
    cfg = cfg or ConvCfg()
    require_cuda(x, weight, bias, residual, prelu_w)
    cout, cin, _, _ = _weight_dims(weight, cfg.transposed)
    # the Cin <= 4 bf16x3 first-layer kernel reads the caller's NCHW tensor in place (no layout copy)
    # (also the f16x3 form of the fp32-faithful class; conv_forward_raw converts the layout if that kernel does not apply)
    x_nchw = (x.dim() == 4 and cin <= 4 and cout >= 8 and not cfg.transposed and not _is_nhwc_dense(x)
              and _is_nchw_dense(x)
              and (_algo_for(cfg, "infer") == ALGO_AUTO or (F16X3 and _algo_for(cfg, "infer") == _lib.ALGO_MFMA_BF16X6))
              and not os.environ.get("SRK_FORCE_ALGO"))  # (the debugging override may pick a kernel without the in-place NCHW read)
    if not x_nchw:
        x = to_nhwc(x)
    if residual is not None:
        residual = to_nhwc(residual)
    if packed is not None:
        wp, bp = packed
    else:
        wp = pack_weight_fwd(weight, cfg.transposed, cfg.ps_r)
        bp = pack_bias_ps(bias, cfg.ps_r)
    return conv_forward_raw(x, wp, bp, weight, cfg, prelu_w, residual, "infer", x_nchw)


# ------------------------------------------------------------------------------------------------
# Pixel shuffle, activations, add / fork
# ------------------------------------------------------------------------------------------------
class _PixelShuffle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r):
        lib = _lib.load()
        require_cuda(x)
        x = to_nhwc(x)
        n, c, h, w = x.shape
        if c % (r * r):
            raise RuntimeError("pixel_shuffle: %d channels not divisible by r^2=%d" % (c, r * r))
        y = _empty_cl(n, c // (r * r), h * r, w * r, x)
        check(lib.srk_pixel_shuffle_forward(ptr(x), ptr(y), n, h, w, c // (r * r), r, stream_ptr()),
              "srk_pixel_shuffle_forward")
        ctx.r = r
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        r = ctx.r
        dy = to_nhwc(dy if (_is_nchw_dense(dy) or _is_nhwc_dense(dy)) else dy.contiguous())
        n, c, hr, wr = dy.shape
        dx = _empty_cl(n, c * r * r, hr // r, wr // r, dy)
        check(lib.srk_pixel_shuffle_backward(ptr(dy), ptr(dx), n, hr // r, wr // r, c, r, stream_ptr()),
              "srk_pixel_shuffle_backward")
        return dx, None


def pixel_shuffle(x, r):
    return _PixelShuffle.apply(x, int(r))


def _channels_inner(x):
    """Innermost (fastest) extent used for per-channel PReLU: C for NHWC 4-D, last dim for 2-D."""
    return x.shape[1] if x.dim() == 4 else x.shape[-1]


def _dense(x):
    if x.dim() == 4:
        return to_nhwc(x if (_is_nchw_dense(x) or _is_nhwc_dense(x)) else x.contiguous())
    return x.contiguous()


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind, slope, prelu_w):
        lib = _lib.load()
        require_cuda(x, prelu_w)
        x = _dense(x)
        y = torch.empty_like(x)
        check(lib.srk_act_forward(ptr(x), ptr(y), x.numel(), _channels_inner(x), kind, slope, ptr(prelu_w),
                                  0 if prelu_w is None else prelu_w.numel(), stream_ptr()), "srk_act_forward")
        ctx.kind, ctx.slope = kind, slope
        ctx.prelu_ref = prelu_w
        ctx.save_for_backward(x if kind == ACT_PRELU else y, prelu_w)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        saved, prelu_w = ctx.saved_tensors
        dy = _dense(dy)
        dx = torch.empty_like(dy)
        dpw = ret_dpw = None
        if prelu_w is not None:
            dpw = getattr(ctx.prelu_ref, "_srk_grad", None)
            if dpw is None:
                dpw = ret_dpw = torch.zeros_like(prelu_w)
        check(lib.srk_act_backward(ptr(dy), ptr(saved), ptr(dx), dy.numel(), _channels_inner(dy), ctx.kind,
                                   ctx.slope, ptr(prelu_w), 0 if prelu_w is None else prelu_w.numel(), ptr(dpw),
                                   stream_ptr()), "srk_act_backward")
        return dx, None, None, ret_dpw


def activation(x, kind, slope=0.0, prelu_w=None):
    if isinstance(kind, str) or kind is None:
        kind = ACT_BY_NAME[kind]
    if kind == ACT_NONE:
        return x
    return _Act.apply(x, int(kind), float(slope), prelu_w)


def _axpby(a, b, alpha, beta):
    lib = _lib.load()
    out = torch.empty_like(a)
    check(lib.srk_axpby(ptr(a), ptr(b), ptr(out), a.numel(), alpha, beta, stream_ptr()), "srk_axpby")
    return out


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        require_cuda(a, b)
        if a.shape != b.shape:
            raise RuntimeError("add: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        a, b = _dense(a), _dense(b)
        return _axpby(a, b, 1.0, 1.0)

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    """torch.add(out, residual) of the reference (base_networks.py:149, vdsr.py:31, edsr.py:42)."""
    return _Add.apply(a, b)


class _Fork(torch.autograd.Function):
    """Use a tensor twice (trunk + skip) with the gradient fan-in summed by srk_axpby instead of
    autograd's internal ATen add."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None:
            return g2
        if g2 is None:
            return g1
        return _axpby(_dense(g1), _dense(g2), 1.0, 1.0)


def fork(x):
    a, b = _Fork.apply(x)
    t = getattr(x, "_srk_amax", None)      # the views carry the running maximum of the tensor they alias
    if t is not None and t[1] == _ver(x) and t[2] == _AMAX_EPOCH[0]:
        _tag_amax(a, t[0])
        _tag_amax(b, t[0])
    return a, b


# ------------------------------------------------------------------------------------------------
# Losses (mean reduction), forward + gradient in one pass
# ------------------------------------------------------------------------------------------------
# Seeds of a backward pass.  `loss.backward()` makes autograd fill a ones tensor (one ATen launch) and the loss backward
# multiply its stored gradient by it (srk_scale_dev, a second launch); a data-parallel step seeds with 1/world instead.
# The train steps of this package know their seed when they compute the loss:
#   * backward(loss) seeds with a persistent per-device ones tensor, which _Loss.backward recognises (by address) and
#     answers with the gradient the forward already wrote -- no fill, no scale;
#   * inside `with loss_seed(value, tensor):` the loss forward folds `value` into that gradient, and a backward seeded with
#     `tensor` (dp.loss_seed) skips the multiply as well.
# Any other upstream gradient (a weighted sum of losses, user code) takes the general path.
_UNIT = {}
_LOSS_SEED = [None]


def unit_seed(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    t = _UNIT.get(idx)
    if t is None:
        t = _UNIT[idx] = torch.ones((), dtype=torch.float32, device=torch.device("cuda", idx))
    return t


def backward(losses, seed=None):
    """torch.autograd.backward(losses) seeded with the persistent ones tensor of the device (or `seed` for each loss)."""
    losses = list(losses) if isinstance(losses, (list, tuple)) else [losses]
    g = unit_seed(losses[0].device) if seed is None else seed
    torch.autograd.backward(losses, [g] * len(losses))


class loss_seed(object):
    """Context: losses computed inside store their gradient already multiplied by `value`; a backward pass seeded with
    `tensor` (which must hold `value`) then uses it as is."""

    def __init__(self, value, tensor):
        self.seed = (float(value), tensor)

    def __enter__(self):
        self.prev, _LOSS_SEED[0] = _LOSS_SEED[0], self.seed

    def __exit__(self, *a):
        _LOSS_SEED[0] = self.prev


class _Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, kind, eps):
        lib = _lib.load()
        require_cuda(pred, target)
        if pred.shape != target.shape:
            raise RuntimeError("loss: pred %s vs target %s" % (tuple(pred.shape), tuple(target.shape)))
        if pred.dim() == 4:
            pred = _dense(pred)
            n, c, h, w = pred.shape
            ts = target.stride()
        else:  # [B, F] (BCE on the discriminator output): treat as N=B, C=F, H=W=1
            pred = pred.contiguous()
            n, c = pred.shape[0], pred.numel() // pred.shape[0]
            h = w = 1
            target = target.contiguous()
            ts = (c, 1, 1, 1)
        strides = (ctypes.c_int64 * 4)(*[int(s) for s in ts])
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        need_grad = ctx.needs_input_grad[0]
        dpred = torch.empty_like(pred) if need_grad else None
        ws = torch.empty(int(lib.srk_loss_workspace_bytes()), dtype=torch.uint8, device=pred.device)
        seed = _LOSS_SEED[0]
        ctx.seed_ptr, ctx.seed_value = (seed[1].data_ptr(), seed[0]) if seed is not None else (0, 1.0)
        check(lib.srk_loss_forward_backward(kind, ptr(pred), ptr(target), strides, n, c, h, w, eps, ctx.seed_value, ptr(loss),
                                            ptr(dpred), ptr(ws), stream_ptr()), "srk_loss_forward_backward")
        ctx.dpred = dpred
        return loss

    @staticmethod
    def backward(ctx, g):
        dpred = ctx.dpred  # (kept on the ctx: a second backward over a retained graph must see it; freed with the graph)
        if dpred is None:
            return None, None, None, None
        gp = g.data_ptr()
        if ctx.seed_ptr:
            if gp == ctx.seed_ptr:      # the seed this loss was computed for: already folded into dpred
                return dpred, None, None, None
            # some other upstream gradient: undo the folded seed (dpred = seed * dL/dpred)
            g = g / ctx.seed_value
        elif gp == unit_seed(dpred.device).data_ptr():
            return dpred, None, None, None
        lib = _lib.load()
        out = torch.empty_like(dpred)
        check(lib.srk_scale_dev(ptr(dpred), ptr(g.contiguous()), ptr(out), dpred.numel(), stream_ptr()),
              "srk_scale_dev")
        return out, None, None, None


_CONSTS = {}


def _const_scalar(value, device):
    """A persistent 0-dim device tensor holding `value` (loss weights, label rows): no fill launch per step."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (float(value), idx)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.full((), float(value), dtype=torch.float32, device=torch.device("cuda", idx))
    return t


def const_rows(value, rows, device):
    """Persistent [rows, 1] label tensor (torch.ones(B, 1) / zeros(B, 1) of srgan.py:264-265 without a fill per step)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (float(value), int(rows), idx)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.full((int(rows), 1), float(value), dtype=torch.float32, device=torch.device("cuda", idx))
    return t


class _LossSum(torch.autograd.Function):
    """a * l1 + b * l2 of two scalar losses with srk_axpby (D_loss = real + fake, G_loss = mse + 1e-3 * GAN:
    srgan.py:283,306); seeded by the unit seed, the backward hands out persistent constants -- no ATen arithmetic in
    the captured step."""

    @staticmethod
    def forward(ctx, l1, l2, a, b):
        ctx.ab = (float(a), float(b))
        return _axpby(l1.reshape(1), l2.reshape(1), float(a), float(b)).reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.ab
        if g.data_ptr() == unit_seed(g.device).data_ptr():
            return (g if a == 1.0 else _const_scalar(a, g.device)), (g if b == 1.0 else _const_scalar(b, g.device)), None, None
        return (g if a == 1.0 else g * a), (g if b == 1.0 else g * b), None, None


def loss_sum(l1, l2, a=1.0, b=1.0):
    return _LossSum.apply(l1, l2, a, b)


def mse_loss(pred, target):
    """nn.MSELoss() (srcnn.py:84-86,129; vdsr.py:145; srgan.py:205,300)."""
    return _Loss.apply(pred, target, _lib.LOSS_MSE, 0.0)


def l1_loss(pred, target):
    """nn.L1Loss() (edsr.py:98-100,153)."""
    return _Loss.apply(pred, target, _lib.LOSS_L1, 0.0)


def charbonnier_loss(pred, target, eps=1e-6):
    """L1_Charbonnier_loss (lapsrn.py:75-85)."""
    return _Loss.apply(pred, target, _lib.LOSS_CHARBONNIER, float(eps))


def bce_loss(pred, target):
    """nn.BCELoss() (srgan.py:157,276-297)."""
    return _Loss.apply(pred, target, _lib.LOSS_BCE, 0.0)


# ------------------------------------------------------------------------------------------------
# BatchNorm2d / Linear (SRGAN)
# ------------------------------------------------------------------------------------------------
BN_FIN_APPLY = os.environ.get("SRK_BN_FIN_APPLY", "1") != "0"   # 0: split reductions as launches of their own (rounds 1 - 5)
BN_FUSE_ACT = os.environ.get("SRK_BN_FUSE_ACT", "1") != "0"   # 0: activations / residual adds after a BatchNorm stay passes of their own


# Collectives inside a captured step (SyncBN's [2C] sums): a capture cannot hold them, so the capturing side
# (trainers.GraphedSegments with a _Splitter) cuts its graph there -- end the graph, run the collective eagerly now and at
# every replay, begin the next graph.  Outside a capture the collective simply runs.
_SPLITTER = [None]


def _collective(fn):
    sp = _SPLITTER[0]
    if sp is not None and torch.cuda.is_current_stream_capturing():
        sp.collective(fn)
    else:
        fn()


class _BatchNorm(torch.autograd.Function):
    """y = act(bn(x)) [+ residual].  act in {none, relu, lrelu, prelu} and the residual ride in the BatchNorm kernels
    (srk_bn_*_act): the backward recomputes z = gamma * xhat + beta from x, so nothing of the activation is saved."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, sync_group, nbt=None,
                act=ACT_NONE, slope=0.0, prelu_w=None, residual=None, res_box=None):
        lib = _lib.load()
        ctx.res_box = res_box
        require_cuda(x, gamma, beta, prelu_w, residual)
        x = _dense(x)   # [N,C,H,W] stored NHWC, or [B,F] (BatchNorm1d of DenseBlock, base_networks.py:13): rows x C
        c = x.shape[1]
        rows = x.numel() // c
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        rstd = torch.empty(c, dtype=torch.float32, device=x.device)
        count = float(rows)
        if training:
            stats = torch.empty(2 * c, dtype=torch.float64, device=x.device)
            ws = torch.empty(int(lib.srk_bn_workspace_bytes(c)), dtype=torch.uint8, device=x.device)
            if nbt is not None and (nbt.dtype != torch.int64 or not nbt.is_cuda):
                raise RuntimeError("batch_norm: num_batches_tracked must be a CUDA int64 tensor")
            nbt_p = None if nbt is None else ctypes.c_void_p(nbt.data_ptr())
            part = getattr(x, "_srk_bn_partial", None)
            if part is not None and (part[2] != _ver(x) or part[0].shape[1] != 2 * c):
                part = None
            # finalize-in-apply (round 6): the apply kernel finishes the split reduction itself -- one launch less
            fin = (BN_FIN_APPLY and sync_group is None and lib.srk_bn_fused_supported(c)
                   and all(t is None or t.data_ptr() % 16 == 0 for t in (x, gamma, beta, residual)))
            if fin:
                if part is not None:      # column sums from the producing conv's epilogue (k_c64)
                    fin_part, fin_splits = part[0], int(part[1])
                else:
                    sp = ctypes.c_int(0)
                    check(lib.srk_bn_stats_partials(ptr(x), rows, c, ptr(ws), ctypes.byref(sp), stream_ptr()),
                          "srk_bn_stats_partials")
                    fin_part, fin_splits = ws, int(sp.value)
            elif sync_group is None and part is not None:
                # the conv that produced x left its column sums (k_c64's epilogue): reduce + finalize only
                check(lib.srk_bn_finalize_partials(ptr(part[0]), part[1], ptr(stats), rows, c, ptr(mean), ptr(rstd),
                                                   ptr(running_mean), ptr(running_var), momentum, eps, nbt_p,
                                                   stream_ptr()), "srk_bn_finalize_partials")
            elif sync_group is None:   # statistics + finalize: column sums, then reduce + mean / rstd / running stats
                check(lib.srk_bn_stats_finalize(ptr(x), ptr(stats), rows, c, ptr(mean), ptr(rstd), ptr(running_mean),
                                                ptr(running_var), momentum, eps, nbt_p, ptr(ws), stream_ptr()),
                      "srk_bn_stats_finalize")
            else:                    # SyncBN: the [2C] sums are all-reduced between the two phases
                import torch.distributed as dist
                check(lib.srk_bn_stats(ptr(x), ptr(stats), rows, c, ptr(ws), stream_ptr()), "srk_bn_stats")
                _collective(lambda: dist.all_reduce(stats, group=sync_group))
                count *= dist.get_world_size(sync_group)
                check(lib.srk_bn_finalize(ptr(stats), count, ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var),
                                          momentum, eps, c, nbt_p, stream_ptr()), "srk_bn_finalize")
        else:
            fin = False
            check(lib.srk_bn_eval_params(ptr(running_mean), ptr(running_var), eps, ptr(mean), ptr(rstd), c,
                                         stream_ptr()), "srk_bn_eval_params")
        y = torch.empty_like(x)
        fused = act != ACT_NONE or residual is not None
        # 4-D activations in front of a convolution of the fp32-faithful class: leave the output's running maximum
        # (only on the kernels' 16-byte path -- C % 4 == 0, aligned tensors; an unaligned view takes the scalar kernel,
        # which keeps no maximum: the convolution behind it then stays on the six-MFMA arithmetic)
        vec_ok = c % 4 == 0 and all(t is None or t.data_ptr() % 16 == 0 for t in (x, gamma, beta, residual))
        ya = _amax_alloc(x.device) if (F16X3 and x.dim() == 4 and vec_ok
                                       and (F16X3_ALWAYS or rows * ((c + 63) // 64) >= F16X3_MIN_PIXELS)
                                       and _MODES[_PRECISION["mode"]]["train_fwd" if training else "infer"]
                                       in (_lib.ALGO_MFMA_BF16X6, _lib.ALGO_MFMA_F16X3)) else None
        if residual is not None:
            residual = _dense(residual)
            if tuple(residual.shape) != tuple(x.shape) or residual.stride() != x.stride():
                raise RuntimeError("batch_norm: residual must have the shape and layout of x")
        if fin:
            check(lib.srk_bn_finalize_apply_act(ptr(fin_part), fin_splits, ptr(stats), rows, c, ptr(mean), ptr(rstd),
                                                ptr(running_mean), ptr(running_var), momentum, eps, nbt_p, ptr(x), ptr(y),
                                                ptr(gamma), ptr(beta), act, slope, ptr(prelu_w),
                                                0 if prelu_w is None else prelu_w.numel(), ptr(residual), ptr(ya),
                                                stream_ptr()), "srk_bn_finalize_apply_act")
        elif fused:
            check(lib.srk_bn_apply_act(ptr(x), ptr(y), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), rows, c, act, slope,
                                       ptr(prelu_w), 0 if prelu_w is None else prelu_w.numel(), ptr(residual),
                                       ptr(ya), stream_ptr()), "srk_bn_apply_act")
        else:
            check(lib.srk_bn_apply(ptr(x), ptr(y), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), rows, c, ACT_NONE, 0.0,
                                   ptr(ya), stream_ptr()), "srk_bn_apply")
        if ya is not None:
            _tag_amax(y, ya)    # the convolution behind a BatchNorm scales its fp16 planes by this (F16X3)
        ctx.training, ctx.count, ctx.sync_group = training, count, sync_group
        ctx.fin = fin
        ctx.gamma_ref, ctx.beta_ref, ctx.prelu_ref = gamma, beta, prelu_w
        ctx.act, ctx.slope, ctx.has_res = act, slope, residual is not None
        ctx.save_for_backward(x, gamma, mean, rstd, beta if act != ACT_NONE else None, prelu_w)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gamma, mean, rstd, beta, prelu_w = ctx.saved_tensors
        dy = _dense(dy)
        c = x.shape[1]
        rows = x.numel() // c
        dstats = torch.empty(2 * c, dtype=torch.float64, device=x.device)
        ws = torch.empty(int(lib.srk_bn_workspace_bytes(c)), dtype=torch.uint8, device=x.device)
        dgamma = getattr(ctx.gamma_ref, "_srk_grad", None)
        dbeta = getattr(ctx.beta_ref, "_srk_grad", None)
        ret_g = ret_b = ret_p = None
        if dgamma is None or dbeta is None:
            dgamma = ret_g = torch.zeros(c, dtype=torch.float32, device=x.device)
            dbeta = ret_b = torch.zeros(c, dtype=torch.float32, device=x.device)
        dres = dy if ctx.has_res else None   # the residual's gradient is dy itself
        if ctx.has_res and ctx.res_box is not None:
            ctx.res_box.g, dres = dy, None   # parked for the block's first conv, which adds it in its data-gradient kernel
        if ctx.fin and ctx.training and dy.data_ptr() % 16 == 0:
            # finalize-in-apply: column sums, then ONE kernel that reduces them (per 16-channel slab), adds the parameter
            # gradients and writes dx
            dprelu = None
            pn = 0 if prelu_w is None else prelu_w.numel()
            if ctx.act == ACT_PRELU:
                dprelu = getattr(ctx.prelu_ref, "_srk_grad", None)
                if dprelu is None:
                    dprelu = ret_p = torch.zeros_like(prelu_w)
            sp = ctypes.c_int(0)
            check(lib.srk_bn_backward_partials_act(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), rows, c,
                                                   ctx.act, ctx.slope, ptr(prelu_w), pn, ptr(ws), ctypes.byref(sp),
                                                   stream_ptr()), "srk_bn_backward_partials_act")
            dx = torch.empty_like(dy)
            check(lib.srk_bn_backward_finalize_apply_act(ptr(ws), int(sp.value), ptr(dstats), ctx.count, ptr(dy), ptr(x),
                                                         ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dx), rows, c,
                                                         ptr(dgamma), ptr(dbeta), ctx.act, ctx.slope, ptr(prelu_w), pn,
                                                         ptr(dprelu), stream_ptr()), "srk_bn_backward_finalize_apply_act")
            return dx, ret_g, ret_b, None, None, None, None, None, None, None, None, None, ret_p, dres, None
        if ctx.act != ACT_NONE:
            dprelu = None
            pn = 0 if prelu_w is None else prelu_w.numel()
            if ctx.act == ACT_PRELU:
                dprelu = getattr(ctx.prelu_ref, "_srk_grad", None)
                if dprelu is None:
                    dprelu = ret_p = torch.zeros_like(prelu_w)
            check(lib.srk_bn_backward_stats_grads_act(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
                                                      ptr(dstats), rows, c, ptr(dgamma), ptr(dbeta), ctx.act, ctx.slope,
                                                      ptr(prelu_w), pn, ptr(dprelu), ptr(ws), stream_ptr()),
                  "srk_bn_backward_stats_grads_act")
            if ctx.training and ctx.sync_group is not None:
                import torch.distributed as dist
                grp = ctx.sync_group
                _collective(lambda: dist.all_reduce(dstats, group=grp))
            dx = torch.empty_like(dy)
            use = dstats if ctx.training else torch.zeros_like(dstats)
            check(lib.srk_bn_backward_apply_act(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(use),
                                                ctx.count, ptr(dx), rows, c, ctx.act, ctx.slope, ptr(prelu_w), pn,
                                                stream_ptr()), "srk_bn_backward_apply_act")
            return dx, ret_g, ret_b, None, None, None, None, None, None, None, None, None, ret_p, dres, None
        # statistics of the backward + the parameter gradients (from the LOCAL sums: the DP gradient all-reduce happens
        # later) in two launches
        check(lib.srk_bn_backward_stats_grads(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(dstats), rows, c, ptr(dgamma),
                                              ptr(dbeta), ptr(ws), stream_ptr()), "srk_bn_backward_stats_grads")
        if ctx.training and ctx.sync_group is not None:
            import torch.distributed as dist
            grp = ctx.sync_group
            _collective(lambda: dist.all_reduce(dstats, group=grp))
        dx = torch.empty_like(dy)
        # eval-mode BN: statistics are constants -> no mean/projection terms in dx
        use = dstats if ctx.training else torch.zeros_like(dstats)
        check(lib.srk_bn_backward_apply(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(use), ctx.count,
                                        ptr(dx), rows, c, stream_ptr()), "srk_bn_backward_apply")
        return dx, ret_g, ret_b, None, None, None, None, None, None, None, None, None, None, dres, None


def batch_norm(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, sync_group=None,
               num_batches_tracked=None, act=ACT_NONE, slope=0.0, prelu_w=None, residual=None, res_box=None):
    """nn.BatchNorm2d (base_networks.py:46,117,161) on [N,C,H,W]; nn.BatchNorm1d (base_networks.py:13) on [B,F].
    act / prelu_w / residual: y = act(bn(x)) [+ residual] in the same launches (see bn_fusable).  res_box: GradBox the
    residual's gradient is parked in instead of being returned to autograd (the block's first conv adds it)."""
    return _BatchNorm.apply(x, gamma, beta, running_mean, running_var, bool(training), float(momentum), float(eps),
                            sync_group, num_batches_tracked, int(act), float(slope), prelu_w, residual, res_box)


def bn_fusable(x, act, prelu_w=None, bn=None):
    """Can `act` (and a residual add) ride in the BatchNorm kernels for this input?  ReLU / LeakyReLU / PReLU with one
    slope or one per channel, channel count a multiple of 4 (the fused kernels move float4s), per-shard statistics
    (SyncBN keeps the separate passes)."""
    if bn is not None and getattr(bn, "sync_group", None) is not None:
        return False
    if not BN_FUSE_ACT or act not in (ACT_NONE, ACT_RELU, ACT_LRELU, ACT_PRELU) or x.dim() < 2 or x.shape[1] % 4:
        return False
    return act != ACT_PRELU or (prelu_w is not None and prelu_w.numel() in (1, x.shape[1]))


class _InstanceNorm(torch.autograd.Function):
    """nn.InstanceNorm2d defaults (affine=False, no running statistics; base_networks.py:48,83,119,163): every
    (sample, channel) plane is normalised with its own biased statistics.  A sample of an NHWC tensor is a dense
    [H*W, C] matrix, so this is the BatchNorm kernels applied per sample (unused by every reference net: N small
    launches rather than a dedicated kernel)."""

    @staticmethod
    def forward(ctx, x, eps):
        lib = _lib.load()
        require_cuda(x)
        x = _dense(x)
        n, c, h, w = x.shape
        rows = h * w
        mean = torch.empty((n, c), dtype=torch.float32, device=x.device)
        rstd = torch.empty((n, c), dtype=torch.float32, device=x.device)
        stats = torch.empty(2 * c, dtype=torch.float64, device=x.device)
        ws = torch.empty(int(lib.srk_bn_workspace_bytes(c)), dtype=torch.uint8, device=x.device)
        y = torch.empty_like(x)
        xs, ys = x.permute(0, 2, 3, 1), y.permute(0, 2, 3, 1)   # [n][h][w][c] views of the NHWC storage
        for i in range(n):
            check(lib.srk_bn_stats(ptr(xs[i]), ptr(stats), rows, c, ptr(ws), stream_ptr()), "srk_bn_stats")
            check(lib.srk_bn_finalize(ptr(stats), float(rows), ptr(mean[i]), ptr(rstd[i]), None, None, 0.0, eps, c, None,
                                      stream_ptr()), "srk_bn_finalize")
            check(lib.srk_bn_apply(ptr(xs[i]), ptr(ys[i]), ptr(mean[i]), ptr(rstd[i]), None, None, rows, c, ACT_NONE, 0.0,
                                   None, stream_ptr()), "srk_bn_apply")
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, mean, rstd = ctx.saved_tensors
        dy = _dense(dy)
        n, c, h, w = x.shape
        rows = h * w
        dstats = torch.empty(2 * c, dtype=torch.float64, device=x.device)
        ws = torch.empty(int(lib.srk_bn_workspace_bytes(c)), dtype=torch.uint8, device=x.device)
        dx = torch.empty_like(dy)
        xs, dys, dxs = x.permute(0, 2, 3, 1), dy.permute(0, 2, 3, 1), dx.permute(0, 2, 3, 1)
        for i in range(n):
            check(lib.srk_bn_backward_stats(ptr(dys[i]), ptr(xs[i]), ptr(mean[i]), ptr(rstd[i]), ptr(dstats), rows, c,
                                            ptr(ws), stream_ptr()), "srk_bn_backward_stats")
            check(lib.srk_bn_backward_apply(ptr(dys[i]), ptr(xs[i]), ptr(mean[i]), ptr(rstd[i]), None, ptr(dstats),
                                            float(rows), ptr(dxs[i]), rows, c, stream_ptr()), "srk_bn_backward_apply")
        return dx, None


def instance_norm(x, eps=1e-5):
    """nn.InstanceNorm2d(C) with its defaults (base_networks.py:48)."""
    return _InstanceNorm.apply(x, float(eps))


class _RowNorm(torch.autograd.Function):
    """nn.InstanceNorm1d on a [B, F] activation (DenseBlock(norm='instance'), base_networks.py:12-13): every row normalised
    with its own biased statistics (srk_rownorm_*)."""

    @staticmethod
    def forward(ctx, x, eps):
        lib = _lib.load()
        require_cuda(x)
        x = x.contiguous()
        rows, cols = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        check(lib.srk_rownorm_forward(ptr(x), ptr(y), ptr(mean), ptr(rstd), rows, cols, eps, stream_ptr()),
              "srk_rownorm_forward")
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        check(lib.srk_rownorm_backward(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(dx), x.shape[0], x.shape[1], stream_ptr()),
              "srk_rownorm_backward")
        return dx, None


def row_norm(x, eps=1e-5):
    """nn.InstanceNorm1d(F) with its defaults on a [B, F] tensor."""
    if x.dim() != 2:
        raise RuntimeError("row_norm expects a [B, F] tensor, got shape %s" % (tuple(x.shape),))
    return _RowNorm.apply(x, float(eps))


class _UpsampleNearest(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r):
        lib = _lib.load()
        require_cuda(x)
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = _empty_cl(n, c, h * r, w * r, x)
        check(lib.srk_upsample_nearest_forward(ptr(x), ptr(y), n, h, w, c, r, stream_ptr()), "srk_upsample_nearest_forward")
        ctx.r = r
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        r = ctx.r
        dy = _dense(dy)
        n, c, hr, wr = dy.shape
        dx = _empty_cl(n, c, hr // r, wr // r, dy)
        check(lib.srk_upsample_nearest_backward(ptr(dy), ptr(dx), n, hr // r, wr // r, c, r, stream_ptr()),
              "srk_upsample_nearest_backward")
        return dx, None


def upsample_nearest(x, r):
    """torch.nn.Upsample(scale_factor=r, mode='nearest') (base_networks.py:206)."""
    return _UpsampleNearest.apply(x, int(r))


def max_pool2x2(x):
    """nn.MaxPool2d(2, 2) of vgg19.features[4] (srgan.py:84-90).  No-grad only: the reference feeds the VGG loss
    detached tensors (srgan.py:302-305)."""
    if grad_enabled_for(x):
        raise RuntimeError("max_pool2x2 has no backward (the reference's VGG loss carries no gradient, srgan.py:302-305); "
                           "call it under torch.no_grad() or on detached tensors")
    lib = _lib.load()
    require_cuda(x)
    x = to_nhwc(x)
    n, c, h, w = x.shape
    if h < 2 or w < 2:
        raise RuntimeError("max_pool2x2: input %s too small" % (tuple(x.shape),))
    y = _empty_cl(n, c, h // 2, w // 2, x)
    check(lib.srk_maxpool2x2_forward(ptr(x), ptr(y), n, h, w, c, stream_ptr()), "srk_maxpool2x2_forward")
    return y


def grad_enabled_for(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _strides4(t):
    """Element strides (n, c, h, w) of a [N,C,H,W] tensor in any dense-or-view layout, as a ctypes array."""
    return (ctypes.c_int64 * 4)(*[int(v) for v in t.stride()])


def psnr(pred, gt):
    """utils.PSNR (utils.py:208-216) on the device: returns (psnr, mse) as 0-dim device tensors, no host sync.
    pred / gt: [N,C,H,W] or [C,H,W] CUDA tensors of any strides (channels_last net outputs, NCHW targets, crops)."""
    lib = _lib.load()
    require_cuda(pred, gt)
    if pred.shape != gt.shape:
        raise RuntimeError("psnr: pred %s vs gt %s" % (tuple(pred.shape), tuple(gt.shape)))
    p = pred.detach()
    g = gt.detach()
    if p.dim() == 3:
        p, g = p.unsqueeze(0), g.unsqueeze(0)
    if p.dim() != 4:
        raise RuntimeError("psnr expects [N,C,H,W] or [C,H,W] tensors, got shape %s" % (tuple(pred.shape),))
    n, c, h, w = p.shape
    out = torch.empty(2, dtype=torch.float32, device=p.device)
    ws = torch.empty(int(lib.srk_psnr_workspace_bytes()), dtype=torch.uint8, device=p.device)
    check(lib.srk_psnr(ptr(p), _strides4(p), ptr(g), _strides4(g), n, c, h, w, ptr(out[0:1]), ptr(out[1:2]), ptr(ws),
                       stream_ptr()), "srk_psnr")
    return out[0], out[1]


def _channel_affine_raw(x, sub, div, clamp01=False):
    """y = (x - sub[c]) / div[c] per channel (utils.norm / utils.denorm, utils.py:219-239), same storage layout as x
    ([C,H,W] / [B,C,H,W] NCHW-contiguous or channels_last); bit-equal to torchvision's Normalize."""
    lib = _lib.load()
    require_cuda(x)
    if x.dim() not in (3, 4):
        raise RuntimeError("channel_affine expects [C,H,W] or [B,C,H,W], got shape %s" % (tuple(x.shape),))
    c = x.shape[-3]
    if len(sub) < c or len(div) < c:
        raise RuntimeError("channel_affine: %d channels but only %d constants" % (c, min(len(sub), len(div))))
    x4 = x if x.dim() == 4 else x.unsqueeze(0)
    if _is_nchw_dense(x4):
        xs = x4 if x4.is_contiguous() else x4.contiguous()
        inner = x.shape[-1] * x.shape[-2]
    elif _is_nhwc_dense(x4):
        xs, inner = x4, 1
    else:
        xs = x4.contiguous()
        inner = x.shape[-1] * x.shape[-2]
    y = torch.empty_like(xs)
    if y.stride() != xs.stride():
        raise RuntimeError("channel_affine: internal layout mismatch")
    fa = (ctypes.c_float * c)(*[float(v) for v in sub[:c]])
    fb = (ctypes.c_float * c)(*[float(v) for v in div[:c]])
    check(lib.srk_channel_affine(ptr(xs), ptr(y), xs.numel(), c, inner, fa, fb, int(bool(clamp01)), stream_ptr()),
          "srk_channel_affine")
    return y if x.dim() == 4 else y[0]


class _ChannelAffine(torch.autograd.Function):
    """(x - sub[c]) / div[c] with its gradient dy / div[c] (the same kernel with sub = 0)."""

    @staticmethod
    def forward(ctx, x, sub, div):
        ctx.div = tuple(float(v) for v in div)
        return _channel_affine_raw(x, sub, div)

    @staticmethod
    def backward(ctx, dy):
        return _channel_affine_raw(dy, (0.0,) * len(ctx.div), ctx.div), None, None


def channel_affine(x, sub, div, clamp01=False):
    """utils.norm / utils.denorm (utils.py:219-239).  Differentiable like the reference's tensor arithmetic when x
    requires grad; the clamped form ((img + 1) / 2).clamp(0, 1) is only available without a graph (the reference uses it
    on detached images when it saves results) and raises otherwise instead of dropping the gradient silently."""
    if torch.is_grad_enabled() and x.requires_grad:
        if clamp01:
            raise RuntimeError("channel_affine(clamp01=True) has no backward: call it on a detached tensor "
                               "(utils.denorm of an image that requires grad)")
        return _ChannelAffine.apply(x, tuple(sub), tuple(div))
    return _channel_affine_raw(x, sub, div, clamp01)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        lib = _lib.load()
        require_cuda(x, weight, bias)
        x = x.contiguous()
        b, fin = x.shape
        fout = weight.shape[0]
        y = torch.empty((b, fout), dtype=torch.float32, device=x.device)
        check(lib.srk_linear_forward(ptr(x), ptr(weight), ptr(bias), ptr(y), b, fin, fout, act, slope, stream_ptr()),
              "srk_linear_forward")
        ctx.act, ctx.slope = act, slope
        ctx.weight_ref, ctx.bias_ref = weight, bias
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, weight, y = ctx.saved_tensors
        dy = dy.contiguous()
        b, fin = x.shape
        fout = weight.shape[0]
        if y is not None:
            dz = torch.empty_like(dy)
            check(lib.srk_act_backward(ptr(dy), ptr(y), ptr(dz), dy.numel(), fout, ctx.act, ctx.slope, None, 0, None,
                                       stream_ptr()), "srk_act_backward")
            dy = dz
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        has_bias = ctx.bias_ref is not None
        wacc = getattr(ctx.weight_ref, "_srk_grad", None)
        bacc = getattr(ctx.bias_ref, "_srk_grad", None) if has_bias else None
        need_w = ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
        if not need_w:   # frozen parameters (trainers.srgan_step(prune_dead_grads=True)): data gradient only
            if dx is not None:
                check(lib.srk_linear_backward(ptr(x), ptr(weight), ptr(dy), ptr(dx), None, None, b, fin, fout, 0.0,
                                              stream_ptr()), "srk_linear_backward")
            return dx, None, None, None, None
        if wacc is not None and (not has_bias or bacc is not None):
            check(lib.srk_linear_backward(ptr(x), ptr(weight), ptr(dy), ptr(dx), ptr(wacc), ptr(bacc), b, fin, fout,
                                          1.0, stream_ptr()), "srk_linear_backward")
            return dx, None, None, None, None
        dw = torch.empty_like(weight)
        db = torch.empty(fout, dtype=torch.float32, device=x.device) if has_bias else None
        check(lib.srk_linear_backward(ptr(x), ptr(weight), ptr(dy), ptr(dx), ptr(dw), ptr(db), b, fin, fout, 0.0,
                                      stream_ptr()), "srk_linear_backward")
        return dx, dw, db, None, None


def linear(x, weight, bias=None, act=ACT_NONE, slope=0.0):
    """nn.Linear + activation of DenseBlock (base_networks.py:7,26-35). PReLU is applied unfused."""
    return _Linear.apply(x, weight, bias, int(act), float(slope))


_INTERP = {"nearest": 0, "bilinear": 2, "bicubic": 3}


def img_interp(imgs, scale_factor, interpolation="bicubic"):
    """utils.img_interp (utils.py:242-269) on the GPU: uint8 quantisation, Pillow's two-pass 8-bit resampling and
    the /255 of ToTensor in three small kernels, bit-exact with the reference's PIL round trip.  `imgs` is a
    [B,C,H,W] or [C,H,W] CUDA tensor; returns a dense NCHW tensor of size int(H*s) x int(W*s) like the reference."""
    require_cuda(imgs)
    if interpolation not in _INTERP:
        raise ValueError("interpolation must be one of %s" % sorted(_INTERP))
    squeeze = imgs.dim() == 3
    x = imgs.unsqueeze(0) if squeeze else imgs
    if x.dim() != 4:
        raise RuntimeError("img_interp expects a [B,C,H,W] or [C,H,W] tensor, got shape %s" % (tuple(imgs.shape),))
    x = x.detach().float().contiguous()
    n, c, h, w = x.shape
    oh, ow = int(h * scale_factor), int(w * scale_factor)
    if oh <= 0 or ow <= 0:
        raise RuntimeError("img_interp: empty output for scale %r" % (scale_factor,))
    lib = _lib.load()
    flt = _INTERP[interpolation]
    y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device)
    nbytes = int(lib.srk_img_interp_workspace_bytes(n, c, h, w, oh, ow, flt))
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=x.device)
    check(lib.srk_img_interp(ptr(x), ptr(y), n, c, h, w, oh, ow, flt, ptr(ws), ws.numel(), stream_ptr()), "srk_img_interp")
    return y[0] if squeeze else y

