"""Leaf layers: parameter containers with torch's names, shapes and default initialisation
(so state_dict files and utils.weights_init_* stay drop-in), whose forward runs the HIP kernels.

They subclass the torch.nn classes ONLY for parameter registration / state_dict / class-name
matching (the reference initialises by `classname.find('Conv2d')`, utils.py:76-113); the ATen
forward is never called.
"""
import torch

from . import ops
from ._lib import ACT_BY_NAME, ACT_LRELU, ACT_NONE, ACT_PRELU, ACT_RELU

# bumped by optim.step(); cached packed weights older than this are re-packed (inference cache)
_WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    _WEIGHT_EPOCH[0] += 1


def _single(v):
    """Stride / padding / output_padding: one value for both axes (the C ABI's srk_conv_desc carries one of each)."""
    if isinstance(v, (tuple, list)):
        if len(set(v)) != 1:
            raise NotImplementedError("only isotropic stride / padding / output_padding are supported, got %r" % (v,))
        return int(v[0])
    return int(v)


def grad_mode(*tensors):
    """True when autograd will need a backward for this call."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class _PackCache(object):
    """Packed-weight cache for no-grad execution, invalidated when the parameters change."""

    def __init__(self):
        self.key = None
        self.val = None

    def get(self, weight, bias, transposed, ps_r):
        key = (weight.data_ptr(), ops._ver(weight), None if bias is None else (bias.data_ptr(), ops._ver(bias)),
               transposed, ps_r, _WEIGHT_EPOCH[0], str(weight.device))
        if key != self.key:
            self.val = (ops.pack_weight_fwd(weight.detach(), transposed, ps_r),
                        ops.pack_bias_ps(None if bias is None else bias.detach(), ps_r))
            self.key = key
        return self.val


def _plan_views(m, ps_r):
    """(wp_fwd, bias_packed, wp_bwd) from the model's PackPlan when it is current for this step."""
    plan = getattr(m, "_plan", None)
    if plan is None:
        return None
    owner, plan_ps, wpf, bp, wpb, wver, bver = plan
    if (plan_ps if plan_ps > 1 else 0) != (ps_r if ps_r > 1 else 0):
        return None
    if not owner.current():
        # the parameters changed since the plan was packed (an optimizer step of THIS model in the middle of a train
        # step: SRGAN runs D again after d_opt.step(), and G's forward of the D step follows g_opt.step()): re-pack the
        # whole model with one launch instead of two small pack launches per layer and direction
        owner.pack()
        owner, plan_ps, wpf, bp, wpb, wver, bver = m._plan
    if ops._ver(m.weight) != wver or (m.bias is not None and ops._ver(m.bias) != bver):
        return None
    return (wpf, bp if (ps_r > 1 and m.bias is not None) else m.bias, wpb)


class Conv2d(torch.nn.Conv2d):
    """torch.nn.Conv2d surface (base_networks.py:42,112-113,156) on srk_conv2d_*."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super(Conv2d, self).__init__(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        self._s, self._p = _single(self.stride), _single(self.padding)   # (kernel_size may be (kh, kw): taken from the weight)
        self._cache = _PackCache()

    def run(self, x, act=ACT_NONE, slope=0.0, prelu_w=None, residual=None, ps_r=0, res_box=None, add_box=None):
        """conv (+ fused epilogue).  In grad mode only none/relu/lrelu are fused (see ops.conv2d);
        the caller applies other activations unfused.  res_box / add_box: ops.GradBox of a residual block."""
        cfg = ops.ConvCfg(self._s, self._p, False, 0, act, slope, ps_r)
        cfg.tail = act == ACT_NONE and getattr(self, "_linear_tail", False)
        if grad_mode(x, self.weight, self.bias, residual, prelu_w):
            return ops.conv2d(x, self.weight, self.bias, residual, cfg, _plan_views(self, ps_r), res_box, add_box)
        packed = self._cache.get(self.weight, self.bias, False, ps_r)
        return ops.conv2d_infer(x, self.weight, self.bias, residual, cfg, prelu_w, packed)

    def forward(self, x):
        return self.run(x)


class ConvTranspose2d(torch.nn.ConvTranspose2d):
    """torch.nn.ConvTranspose2d surface (base_networks.py:77; fsrcnn.py:33)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, bias=True):
        super(ConvTranspose2d, self).__init__(in_channels, out_channels, kernel_size, stride, padding,
                                              output_padding, bias=bias)
        self._s, self._p = _single(self.stride), _single(self.padding)   # (kernel_size may be (kh, kw): taken from the weight)
        self._op = _single(self.output_padding)
        self._cache = _PackCache()

    def run(self, x, act=ACT_NONE, slope=0.0, prelu_w=None):
        cfg = ops.ConvCfg(self._s, self._p, True, self._op, act, slope, 0)
        if grad_mode(x, self.weight, self.bias, prelu_w):
            return ops.conv2d(x, self.weight, self.bias, None, cfg, _plan_views(self, 0))
        packed = self._cache.get(self.weight, self.bias, True, 0)
        return ops.conv2d_infer(x, self.weight, self.bias, None, cfg, prelu_w, packed)

    def forward(self, x, output_size=None):
        if output_size is not None:
            raise NotImplementedError("output_size is not supported")
        return self.run(x)


class PixelShuffle(torch.nn.PixelShuffle):
    """torch.nn.PixelShuffle surface (base_networks.py:157)."""

    def forward(self, x):
        return ops.pixel_shuffle(x, self.upscale_factor)


class BatchNorm2d(torch.nn.BatchNorm2d):
    """torch.nn.BatchNorm2d surface (base_networks.py:46,117,161). `sync_group` (a
    torch.distributed group) turns the batch statistics into SyncBN sums over the DP ranks."""

    sync_group = None

    def run(self, x, act=ACT_NONE, slope=0.0, prelu_w=None, residual=None, res_box=None):
        """act(bn(x)) [+ residual] in the BatchNorm's own launches (callers check ops.bn_fusable first).  res_box: the
        ops.GradBox that takes the residual's gradient (see ResnetBlock)."""
        training = self.training or self.running_mean is None
        momentum = 0.1 if self.momentum is None else self.momentum
        # (num_batches_tracked is bumped by the finalize kernel: no launch of its own)
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, training, momentum,
                              self.eps, self.sync_group if training else None,
                              self.num_batches_tracked if training else None, act, slope, prelu_w, residual, res_box)

    def forward(self, x):
        return self.run(x)


class BatchNorm1d(torch.nn.BatchNorm1d):
    """torch.nn.BatchNorm1d surface on [B, F] inputs (DenseBlock's default norm, base_networks.py:10-11): the same
    two-phase statistics kernels as BatchNorm2d with rows = B, channels = F."""

    sync_group = None

    def forward(self, x):
        if x.dim() != 2:
            raise NotImplementedError("BatchNorm1d is implemented for [B, F] inputs (DenseBlock); got %s" % (tuple(x.shape),))
        training = self.training or self.running_mean is None
        momentum = 0.1 if self.momentum is None else self.momentum
        # (num_batches_tracked is bumped by the finalize kernel: no launch of its own)
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, training, momentum,
                              self.eps, self.sync_group if training else None,
                              self.num_batches_tracked if training else None)


class InstanceNorm2d(torch.nn.InstanceNorm2d):
    """torch.nn.InstanceNorm2d(C) with its defaults — no affine parameters, no running statistics, so no state_dict
    entries (base_networks.py:48,83,119,163) — on per-sample statistics kernels."""

    def __init__(self, num_features):
        super(InstanceNorm2d, self).__init__(num_features)

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.num_features:
            raise RuntimeError("InstanceNorm2d(%d): bad input shape %s" % (self.num_features, tuple(x.shape)))
        return ops.instance_norm(x, self.eps)


class InstanceNorm1d(torch.nn.InstanceNorm1d):
    """torch.nn.InstanceNorm1d(F) with its defaults on the [B, F] output of a Linear (DenseBlock(norm='instance'),
    base_networks.py:12-13).  torch reads a 2-D input as ONE unbatched sample of B channels x F positions: every row is
    normalised with its own biased statistics; no parameters, no state_dict entries."""

    def __init__(self, num_features):
        super(InstanceNorm1d, self).__init__(num_features)

    def forward(self, x):
        if x.dim() != 2:
            raise NotImplementedError("InstanceNorm1d is implemented for [B, F] inputs (DenseBlock); got %s" % (tuple(x.shape),))
        return ops.row_norm(x, self.eps)


class Linear(torch.nn.Linear):
    """torch.nn.Linear surface (base_networks.py:7)."""

    def run(self, x, act=ACT_NONE, slope=0.0):
        return ops.linear(x, self.weight, self.bias, act, slope)

    def forward(self, x):
        return self.run(x)


class ReLU(torch.nn.ReLU):
    kind, slope = ACT_RELU, 0.0

    def forward(self, x):
        return ops.activation(x, ACT_RELU)


class LeakyReLU(torch.nn.LeakyReLU):
    kind = ACT_LRELU

    @property
    def slope(self):
        return float(self.negative_slope)

    def forward(self, x):
        return ops.activation(x, ACT_LRELU, self.negative_slope)


class PReLU(torch.nn.PReLU):
    kind, slope = ACT_PRELU, 0.0

    def forward(self, x):
        return ops.activation(x, ACT_PRELU, 0.0, self.weight)


class Tanh(torch.nn.Tanh):
    kind, slope = ACT_BY_NAME["tanh"], 0.0

    def forward(self, x):
        return ops.activation(x, self.kind)


class Sigmoid(torch.nn.Sigmoid):
    kind, slope = ACT_BY_NAME["sigmoid"], 0.0

    def forward(self, x):
        return ops.activation(x, self.kind)


def make_activation(name):
    """The if-chain of base_networks.py:49-60 as a table."""
    if name is None:
        return None
    table = {"relu": lambda: ReLU(True), "prelu": PReLU, "lrelu": lambda: LeakyReLU(0.2, True), "tanh": Tanh,
             "sigmoid": Sigmoid}
    if name not in table:
        return None  # the reference silently builds no activation for unknown names
    return table[name]()


def make_norm2d(norm, channels):
    if norm is None:
        return None
    if norm == "batch":
        return BatchNorm2d(channels)
    if norm == "instance":
        return InstanceNorm2d(channels)
    return None


def make_norm1d(norm, features):
    """DenseBlock's norm (base_networks.py:9-13)."""
    if norm is None:
        return None
    if norm == "batch":
        return BatchNorm1d(features)
    if norm == "instance":
        return InstanceNorm1d(features)
    return None
