"""The shared block library of the reference (base_networks.py:4-214) on the MI355X kernels.

Same six class names, constructor signatures, sub-module attribute names (`conv`, `deconv`, `fc`,
`bn`, `act`, `ps`, `upsample`) and therefore the same state_dict keys.  Each block's forward maps
to as few kernel launches as the arithmetic allows:

    ConvBlock / DeconvBlock   conv + bias + activation                     -> 1 launch
    ResnetBlock (no norm)     conv+act, conv + residual add                -> 2 launches
    PSBlock (no norm)         conv + bias + pixel-shuffle store (+ act)    -> 1 launch (+1 in training w/ PReLU)

With autograd enabled only ReLU / LeakyReLU are fused into the conv (their gradient mask can be
recovered from the saved output); PReLU / tanh / sigmoid then run as a separate pointwise kernel
that saves what its backward needs.
"""
import torch

from . import ops
from .layers import (ACT_LRELU, ACT_NONE, ACT_PRELU, ACT_RELU, BatchNorm2d, Conv2d, ConvTranspose2d, Linear,
                     PixelShuffle, _plan_views, grad_mode, make_activation, make_norm1d, make_norm2d)

_FUSABLE_IN_TRAINING = (ACT_NONE, ACT_RELU, ACT_LRELU)


class _Block(torch.nn.Module):
    """Common activation / norm plumbing of the reference blocks."""

    def _setup(self, channels, activation, norm, norm1d=False):
        self.norm = norm
        bn = make_norm1d(norm, channels) if norm1d else make_norm2d(norm, channels)
        if bn is not None:
            self.bn = bn
        self.activation = activation
        act = make_activation(activation)
        if act is not None:
            self.act = act

    def _act_args(self):
        """(kind, slope, prelu_weight) of this block's activation."""
        act = getattr(self, "act", None)
        if self.activation is None or act is None:
            return ACT_NONE, 0.0, None
        return act.kind, act.slope, (act.weight if act.kind == ACT_PRELU else None)

    def _post(self, out, fused_act):
        """norm -> activation after the main op (reference order: act(bn(op(x))))."""
        if self.norm is not None:
            kind, slope, pw = self._act_args()
            if (not fused_act and kind != ACT_NONE and isinstance(self.bn, BatchNorm2d)
                    and ops.bn_fusable(out, kind, pw, self.bn)):
                return self.bn.run(out, kind, slope, pw)   # act(bn(x)) in the BatchNorm's launches
            out = self.bn(out)
        if self.activation is not None and not fused_act:
            out = self.act(out)
        return out

    def _fuse_act(self, *tensors):
        """Can the activation ride in the conv epilogue for this call?"""
        if self.norm is not None:
            return False
        kind, _, pw = self._act_args()
        if kind == ACT_NONE:
            return True
        if grad_mode(*tensors, pw):
            return kind in _FUSABLE_IN_TRAINING
        return True


class DenseBlock(_Block):
    """base_networks.py:4-36"""

    def __init__(self, input_size, output_size, bias=True, activation='relu', norm='batch'):
        super(DenseBlock, self).__init__()
        self.fc = Linear(input_size, output_size, bias=bias)
        self._setup(output_size, activation, norm, norm1d=True)

    def forward(self, x):
        kind, slope, pw = self._act_args()
        fuse = kind != ACT_PRELU and self.norm is None   # act(bn(fc(x))): the activation follows the norm
        out = self.fc.run(x, kind if fuse else ACT_NONE, slope)
        return self._post(out, fuse)


class ConvBlock(_Block):
    """base_networks.py:39-71"""

    def __init__(self, input_size, output_size, kernel_size=4, stride=2, padding=1, bias=True, activation='relu',
                 norm='batch'):
        super(ConvBlock, self).__init__()
        self.conv = Conv2d(input_size, output_size, kernel_size, stride, padding, bias=bias)
        self._setup(output_size, activation, norm)

    def forward(self, x, residual=None):
        """`residual` (not in the reference signature) lets a caller fuse `torch.add(out, residual)`
        (vdsr.py:31, edsr.py:42) into this block's kernel when the block has no activation."""
        kind, slope, pw = self._act_args()
        fuse = self._fuse_act(x, self.conv.weight, self.conv.bias, residual)
        fuse_res = residual is not None and self.norm is None and (kind == ACT_NONE or not grad_mode(
            x, self.conv.weight, self.conv.bias, residual, pw)) and fuse
        out = self.conv.run(x, kind if fuse else ACT_NONE, slope, pw if fuse else None,
                            residual if fuse_res else None)
        out = self._post(out, fuse)
        if residual is not None and not fuse_res:
            out = ops.add(out, residual)
        return out


class DeconvBlock(_Block):
    """base_networks.py:74-106"""

    def __init__(self, input_size, output_size, kernel_size=4, stride=2, padding=1, bias=True, activation='relu',
                 norm='batch'):
        super(DeconvBlock, self).__init__()
        self.deconv = ConvTranspose2d(input_size, output_size, kernel_size, stride, padding, bias=bias)
        self._setup(output_size, activation, norm)

    def forward(self, x):
        kind, slope, pw = self._act_args()
        fuse = self._fuse_act(x, self.deconv.weight, self.deconv.bias)
        out = self.deconv.run(x, kind if fuse else ACT_NONE, slope, pw if fuse else None)
        return self._post(out, fuse)


class ResnetBlock(_Block):
    """base_networks.py:109-150 — note ONE `bn` (and one `act`) shared by both convs."""

    def __init__(self, num_filter, kernel_size=3, stride=1, padding=1, bias=True, activation='relu', norm='batch'):
        super(ResnetBlock, self).__init__()
        self.conv1 = Conv2d(num_filter, num_filter, kernel_size, stride, padding, bias=bias)
        self.conv2 = Conv2d(num_filter, num_filter, kernel_size, stride, padding, bias=bias)
        self._setup(num_filter, activation, norm)

    def forward(self, x):
        kind, slope, pw = self._act_args()
        training = grad_mode(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias, pw)
        if self.norm is None:
            fuse = self._fuse_act(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias)
            if (training and fuse and kind == ACT_RELU and self.conv1._s == 1 and self.conv1._p == 1
                    and self.conv2._s == 1 and self.conv2._p == 1
                    and ops.resblock2_applicable(x, self.conv1.weight, self.conv2.weight)):
                # small problems (a strong-scaled shard): both convs, the ReLU and the skip in one launch per direction
                return ops.resblock2(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                     _plan_views(self.conv1, 0), _plan_views(self.conv2, 0))
            # training: the skip gradient (= the block output gradient) is added by conv1's data-gradient kernel
            # (ops.GradBox); it needs x itself to require grad, else there is no fan-in to sum
            box = ops.GradBox() if (training and x.requires_grad and ops.FUSE_SKIP_GRAD) else None
            residual = x
            if training and box is None:
                x, residual = ops.fork(x)  # unfused: gradient fan-in summed by srk_axpby
            out = self.conv1.run(x, kind if fuse else ACT_NONE, slope, pw if fuse else None, add_box=box)
            if not fuse:
                out = self.act(out)
            return self.conv2.run(out, ACT_NONE, 0.0, None, residual, res_box=box)  # conv2 + residual add, one kernel
        fused_bn = isinstance(self.bn, BatchNorm2d) and ops.bn_fusable(x, kind, pw, self.bn)
        if fused_bn and training and x.requires_grad and ops.FUSE_SKIP_GRAD:
            # as in the no-norm block: the skip gradient (= the block output gradient, handed over by the second
            # BatchNorm's backward) is added by conv1's data-gradient kernel -- no fan-in pass of its own
            box = ops.GradBox()
            with ops.bn_partial_request():   # the convs leave the BatchNorm's column sums where their kernel can
                c1 = self.conv1.run(x, add_box=box)
            out = self.bn.run(c1, kind, slope, pw)
            with ops.bn_partial_request():
                c2 = self.conv2.run(out)
            return self.bn.run(c2, residual=x, res_box=box)
        if training:
            x, residual = ops.fork(x)  # gradient fan-in summed by srk_axpby
        else:
            residual = x
        if fused_bn:
            # act(bn(conv1)) and bn(conv2) + x each in the BatchNorm's own launches (ONE shared bn: base_networks.py:117)
            with ops.bn_partial_request(training):
                c1 = self.conv1.run(x)
            out = self.bn.run(c1, kind, slope, pw)
            with ops.bn_partial_request(training):
                c2 = self.conv2.run(out)
            return self.bn.run(c2, residual=residual)
        out = self.bn(self.conv1.run(x))
        if self.activation is not None:
            out = self.act(out)
        out = self.bn(self.conv2.run(out))
        return ops.add(out, residual)


class PSBlock(_Block):
    """base_networks.py:153-185 — conv -> PixelShuffle -> [bn] -> [act]; the shuffle is fused into
    the conv's store (the BN, when present, has `output_size` channels and runs after it)."""

    def __init__(self, input_size, output_size, scale_factor, kernel_size=3, stride=1, padding=1, bias=True,
                 activation='relu', norm='batch'):
        super(PSBlock, self).__init__()
        self.conv = Conv2d(input_size, output_size * scale_factor ** 2, kernel_size, stride, padding, bias=bias)
        self.ps = PixelShuffle(scale_factor)
        self._r = int(scale_factor)
        self.conv._ps_r = self._r  # PackPlan packs this conv's filter in pixel-shuffle channel order
        self._setup(output_size, activation, norm)

    def forward(self, x):
        kind, slope, pw = self._act_args()
        training = grad_mode(x, self.conv.weight, self.conv.bias, pw)
        fuse = self.norm is None and (kind == ACT_NONE or not training)
        out = self.conv.run(x, kind if fuse else ACT_NONE, slope, pw if fuse else None, None, self._r)
        return self._post(out, fuse)


class UpsampleNearest(torch.nn.Upsample):
    """torch.nn.Upsample(scale_factor=r, mode='nearest') surface (base_networks.py:206) on srk_upsample_nearest_*."""

    def __init__(self, scale_factor):
        super(UpsampleNearest, self).__init__(scale_factor=scale_factor, mode='nearest')
        self._r = int(scale_factor)

    def forward(self, x):
        return ops.upsample_nearest(x, self._r)


class Upsample2xBlock(torch.nn.Module):
    """base_networks.py:188-214"""

    def __init__(self, input_size, output_size, bias=True, upsample='deconv', activation='relu', norm='batch'):
        super(Upsample2xBlock, self).__init__()
        scale_factor = 2
        if upsample == 'deconv':
            self.upsample = DeconvBlock(input_size, output_size, kernel_size=4, stride=2, padding=1, bias=bias,
                                        activation=activation, norm=norm)
        elif upsample == 'ps':
            self.upsample = PSBlock(input_size, output_size, scale_factor=scale_factor, bias=bias,
                                    activation=activation, norm=norm)
        elif upsample == 'rnc':
            # 3. Resize and Convolution (base_networks.py:204-210): nn.Sequential(Upsample(x2, nearest), ConvBlock 3x3)
            # — same container and indices, so the state_dict keys are `upsample.1.conv.weight` etc. like the reference
            self.upsample = torch.nn.Sequential(
                UpsampleNearest(scale_factor),
                ConvBlock(input_size, output_size, kernel_size=3, stride=1, padding=1, bias=bias,
                          activation=activation, norm=norm))
        else:
            raise ValueError("unknown upsample mode %r" % (upsample,))

    def forward(self, x):
        return self.upsample(x)
