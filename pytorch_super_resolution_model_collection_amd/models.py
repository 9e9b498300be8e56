"""Network topologies of the reference (srcnn.py:13-29, espcn.py:13-29, fsrcnn.py:13-55,
vdsr.py:13-36, edsr.py:13-45, lapsrn.py:14-85, srgan.py:14-81) built from the MI355X blocks.

Same constructor signatures, attribute names (=> state_dict keys), forward semantics and
`weight_init` distributions.  Differences are purely about launch count: residual adds are handed
to the producing conv's epilogue and tensors that fan out in training go through `ops.fork` so the
gradient fan-in is one srk_axpby launch.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops, utils
from ._lib import ACT_RELU
from .base_networks import ConvBlock, DeconvBlock, DenseBlock, PSBlock, ResnetBlock, Upsample2xBlock
from .layers import Conv2d, ConvTranspose2d, PReLU, grad_mode


def _training_graph(module, x):
    return grad_mode(x, *[p for p in module.parameters()])


class SRCNNNet(nn.Module):
    """srcnn.py:13-29 — 9-5-5, valid padding."""

    def __init__(self, num_channels, base_filter):
        super(SRCNNNet, self).__init__()
        self.layers = nn.Sequential(
            ConvBlock(num_channels, base_filter, 9, 1, 0, norm=None),
            ConvBlock(base_filter, base_filter // 2, 5, 1, 0, norm=None),
            ConvBlock(base_filter // 2, num_channels, 5, 1, 0, activation=None, norm=None))

    def forward(self, x):
        return self.layers(x)

    def weight_init(self, mean=0.0, std=0.001):
        for m in self.modules():
            utils.weights_init_normal(m, mean=mean, std=std)


class ESPCNNet(nn.Module):
    """espcn.py:13-29 — 5-3-3 + PixelShuffle(r)."""

    def __init__(self, num_channels, base_filter, scale_factor):
        super(ESPCNNet, self).__init__()
        self.layers = nn.Sequential(
            ConvBlock(num_channels, base_filter, 5, 1, 0, activation='relu', norm=None),
            ConvBlock(base_filter, base_filter // 2, 3, 1, 0, activation='relu', norm=None),
            PSBlock(base_filter // 2, num_channels, scale_factor, 3, 1, 0, activation=None, norm=None))

    def forward(self, x):
        return self.layers(x)

    def weight_init(self):
        for m in self.modules():
            utils.weights_init_normal(m)


class FSRCNNNet(nn.Module):
    """fsrcnn.py:13-55 — note the four 3x3 mapping convs have NO activation between them and are
    followed by one bare PReLU (`mid_part.5`)."""

    def __init__(self, num_channels, scale_factor, d, s, m):
        super(FSRCNNNet, self).__init__()
        self.first_part = ConvBlock(num_channels, d, 5, 1, 0, activation='prelu', norm=None)
        self.layers = [ConvBlock(d, s, 1, 1, 0, activation='prelu', norm=None)]
        for _ in range(m):
            self.layers.append(ConvBlock(s, s, 3, 1, 1, activation=None, norm=None))
        self.layers.append(PReLU())
        self.layers.append(ConvBlock(s, d, 1, 1, 0, activation='prelu', norm=None))
        self.mid_part = nn.Sequential(*self.layers)
        self.last_part = ConvTranspose2d(d, num_channels, 9, scale_factor, 3, output_padding=1)

    def forward(self, x):
        return self.last_part(self.mid_part(self.first_part(x)))

    def weight_init(self, mean=0.0, std=0.02):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(mean, std)
                if m.bias is not None:
                    m.bias.data.zero_()
            if isinstance(m, nn.ConvTranspose2d):
                m.weight.data.normal_(0.0, 0.0001)
                if m.bias is not None:
                    m.bias.data.zero_()


class VDSRNet(nn.Module):
    """vdsr.py:13-36 — bias-free 3x3 stack + global residual (fused into output_conv's store)."""

    def __init__(self, num_channels, base_filter, num_residuals):
        super(VDSRNet, self).__init__()
        self.input_conv = ConvBlock(num_channels, base_filter, 3, 1, 1, norm=None, bias=False)
        self.residual_layers = nn.Sequential(*[ConvBlock(base_filter, base_filter, 3, 1, 1, norm=None, bias=False)
                                               for _ in range(num_residuals)])
        self.output_conv = ConvBlock(base_filter, num_channels, 3, 1, 1, activation=None, norm=None, bias=False)
        self.output_conv.conv._linear_tail = True   # only the global residual add lies between it and the loss (ops.py)

    def forward(self, x):
        out = self.residual_layers(self.input_conv(x))
        return self.output_conv(out, residual=x)

    def weight_init(self):
        for m in self.modules():
            utils.weights_init_kaming(m)


def _trunk_with_skip(head_out, trunk, mid_conv, training):
    """out = mid_conv(trunk(h)) + h   (edsr.py:37-42, srgan.py:34-39)"""
    if training:
        h, skip = ops.fork(head_out)
    else:
        h = skip = head_out
    return mid_conv(trunk(h), residual=skip)


class EDSRNet(nn.Module):
    """edsr.py:13-45 — 16 BN-free ResnetBlocks, 2x pixel-shuffle upsamplers, L1-trained."""

    def __init__(self, num_channels, base_filter, num_residuals):
        super(EDSRNet, self).__init__()
        self.input_conv = ConvBlock(num_channels, base_filter, 3, 1, 1, activation=None, norm=None)
        self.residual_layers = nn.Sequential(*[ResnetBlock(base_filter, norm=None) for _ in range(num_residuals)])
        self.mid_conv = ConvBlock(base_filter, base_filter, 3, 1, 1, activation=None, norm=None)
        self.upscale4x = nn.Sequential(
            Upsample2xBlock(base_filter, base_filter, upsample='ps', activation=None, norm=None),
            Upsample2xBlock(base_filter, base_filter, upsample='ps', activation=None, norm=None))
        self.output_conv = ConvBlock(base_filter, num_channels, 3, 1, 1, activation=None, norm=None)
        # nothing but additions, pixel shuffles and convolutions lies between these layers' outputs and the loss: no
        # activation mask depends on their rounding, so their training forward takes the bf16x3 products of the
        # backward pass (the ReLU-carrying body keeps the fp32-faithful bf16x6; ops.set_precision("bf16x6") keeps it
        # everywhere)
        for m in (self.mid_conv.conv, self.upscale4x[0].upsample.conv, self.upscale4x[1].upsample.conv,
                  self.output_conv.conv):
            m._linear_tail = True

    def weight_init(self, mean=0.0, std=0.02):
        for m in self.modules():
            utils.weights_init_normal(m, mean=mean, std=std)

    def forward(self, x):
        out = _trunk_with_skip(self.input_conv(x), self.residual_layers, self.mid_conv, _training_graph(self, x))
        return self.output_conv(self.upscale4x(out))


def get_upsample_filter(size):
    """lapsrn.py:14-24 — 2-D bilinear kernel."""
    factor = (size + 1) // 2
    center = factor - 1 if size % 2 == 1 else factor - 0.5
    og = np.ogrid[:size, :size]
    filt = (1 - abs(og[0] - center) / factor) * (1 - abs(og[1] - center) / factor)
    return torch.from_numpy(filt).float()


class LapSRNNet(nn.Module):
    """lapsrn.py:27-72 — two pyramid levels that SHARE the feature branch modules
    (`convt_F1` and `convt_F2` are Sequentials over the same block objects)."""

    def __init__(self, num_channels, base_filter, num_convs):
        super(LapSRNNet, self).__init__()
        self.input_conv = ConvBlock(num_channels, base_filter, 3, 1, 1, activation='lrelu', norm=None, bias=False)
        conv_blocks = [ConvBlock(base_filter, base_filter, 3, 1, 1, activation='lrelu', norm=None, bias=False)
                       for _ in range(num_convs)]
        conv_blocks.append(DeconvBlock(base_filter, base_filter, 4, 2, 1, activation='lrelu', norm=None, bias=False))
        self.convt_I1 = DeconvBlock(num_channels, num_channels, 4, 2, 1, activation=None, norm=None, bias=False)
        self.convt_R1 = ConvBlock(base_filter, num_channels, 3, 1, 1, activation=None, norm=None, bias=False)
        self.convt_F1 = nn.Sequential(*conv_blocks)
        self.convt_I2 = DeconvBlock(num_channels, num_channels, 4, 2, 1, activation=None, norm=None, bias=False)
        self.convt_R2 = ConvBlock(base_filter, num_channels, 3, 1, 1, activation=None, norm=None, bias=False)
        self.convt_F2 = nn.Sequential(*conv_blocks)

    def weight_init(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
                if m.bias is not None:
                    m.bias.data.zero_()
            if isinstance(m, nn.ConvTranspose2d):
                c1, c2, h, w = m.weight.data.size()
                m.weight.data.copy_(get_upsample_filter(h).view(1, 1, h, w).repeat(c1, c2, 1, 1))
                if m.bias is not None:
                    m.bias.data.zero_()

    def forward(self, x):
        training = _training_graph(self, x)
        out = self.input_conv(x)
        f1 = self.convt_F1(out)
        if training:
            f1a, f1b = ops.fork(f1)
        else:
            f1a = f1b = f1
        i1 = self.convt_I1(x)
        x_coarse = self.convt_R1(f1a, residual=i1)
        if training:
            xc_out, xc_in = ops.fork(x_coarse)
        else:
            xc_out = xc_in = x_coarse
        f2 = self.convt_F2(f1b)
        i2 = self.convt_I2(xc_in)
        x_finer = self.convt_R2(f2, residual=i2)
        return xc_out, x_finer


class SRGANGenerator(nn.Module):
    """srgan.py:14-46 — SRResNet: 9x9 head, 16 ResnetBlocks (shared BN + PReLU each), BN'd mid conv,
    2x PS+PReLU upsamplers, 9x9 tail."""

    def __init__(self, num_channels, base_filter, num_residuals):
        super(SRGANGenerator, self).__init__()
        self.input_conv = ConvBlock(num_channels, base_filter, 9, 1, 4, activation='prelu', norm=None)
        self.residual_layers = nn.Sequential(*[ResnetBlock(base_filter, activation='prelu')
                                               for _ in range(num_residuals)])
        self.mid_conv = ConvBlock(base_filter, base_filter, 3, 1, 1, activation=None)
        self.upscale4x = nn.Sequential(
            Upsample2xBlock(base_filter, base_filter, upsample='ps', activation='prelu', norm=None),
            Upsample2xBlock(base_filter, base_filter, upsample='ps', activation='prelu', norm=None))
        self.output_conv = ConvBlock(base_filter, num_channels, 9, 1, 4, activation=None, norm=None)

    def forward(self, x):
        out = _trunk_with_skip(self.input_conv(x), self.residual_layers, self.mid_conv, _training_graph(self, x))
        return self.output_conv(self.upscale4x(out))

    def weight_init(self, mean=0.0, std=0.02):
        for m in self.modules():
            utils.weights_init_normal(m, mean=mean, std=std)


class SRGANDiscriminator(nn.Module):
    """srgan.py:49-81 — 8 convs (stride 1/2 alternating, BN + LeakyReLU) + 2 dense layers."""

    def __init__(self, num_channels, base_filter, image_size):
        super(SRGANDiscriminator, self).__init__()
        self.image_size = image_size
        self.input_conv = ConvBlock(num_channels, base_filter, 3, 1, 1, activation='lrelu', norm=None)
        self.conv_blocks = nn.Sequential(
            ConvBlock(base_filter, base_filter, 3, 2, 1, activation='lrelu'),
            ConvBlock(base_filter, base_filter * 2, 3, 1, 1, activation='lrelu'),
            ConvBlock(base_filter * 2, base_filter * 2, 3, 2, 1, activation='lrelu'),
            ConvBlock(base_filter * 2, base_filter * 4, 3, 1, 1, activation='lrelu'),
            ConvBlock(base_filter * 4, base_filter * 4, 3, 2, 1, activation='lrelu'),
            ConvBlock(base_filter * 4, base_filter * 8, 3, 1, 1, activation='lrelu'),
            ConvBlock(base_filter * 8, base_filter * 8, 3, 2, 1, activation='lrelu'))
        self.dense_layers = nn.Sequential(
            DenseBlock(base_filter * 8 * image_size // 16 * image_size // 16, base_filter * 16, activation='lrelu',
                       norm=None),
            DenseBlock(base_filter * 16, 1, activation='sigmoid', norm=None))

    def forward(self, x):
        out = self.conv_blocks(self.input_conv(x))
        out = ops.flatten_nchw(out)  # out.view(B, -1) in (C,H,W) order, srgan.py:75
        return self.dense_layers(out)

    def weight_init(self, mean=0.0, std=0.02):
        for m in self.modules():
            utils.weights_init_normal(m, mean=mean, std=std)


class _MaxPool2x2(nn.MaxPool2d):
    """nn.MaxPool2d(kernel_size=2, stride=2) surface (vgg19.features[4]) on srk_maxpool2x2_forward."""

    def __init__(self):
        super(_MaxPool2x2, self).__init__(kernel_size=2, stride=2)

    def forward(self, x):
        return ops.max_pool2x2(x)


class FeatureExtractor(nn.Module):
    """srgan.py:84-90 — `nn.Sequential(*list(vgg19.features.children())[:feature_layer + 1])`: for the reference's
    feature_layer = 8 that is conv3-64, ReLU, conv64-64, ReLU, MaxPool 2x2, conv64-128, ReLU, conv128-128, ReLU, kept
    under the same indices so a torchvision `vgg19` checkpoint's `features.{0,2,5,7}.{weight,bias}` load directly
    (`load_vgg19`).  The reference downloads pretrained weights (srgan.py:144); without network access the weights
    are whatever the caller loads (kaiming-normal until then, like torchvision's own init).  The module is evaluated
    without gradients only (srgan.py:302-305): conv + ReLU run as one fused kernel each."""

    # torchvision.models.vgg19 `features` layout up to index 36: numbers = conv3x3 output channels, 'M' = MaxPool2d(2,2)
    VGG19_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M')

    def __init__(self, netVGG=None, feature_layer=8):
        super(FeatureExtractor, self).__init__()
        from .layers import Conv2d, ReLU
        mods, cin = [], 3
        for v in self.VGG19_CFG:
            if v == 'M':
                mods.append(_MaxPool2x2())
            else:
                mods += [Conv2d(cin, v, 3, 1, 1), ReLU(True)]
                cin = v
        self.features = nn.Sequential(*mods[:feature_layer + 1])
        for m in self.features:
            if isinstance(m, nn.Conv2d):   # torchvision's VGG initialisation
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                m.bias.data.zero_()
        for p in self.parameters():
            p.requires_grad_(False)
        if netVGG is not None:
            self.load_vgg19(netVGG.state_dict() if hasattr(netVGG, "state_dict") else netVGG)

    def load_vgg19(self, state_dict):
        """Copy `features.N.weight/bias` of a torchvision vgg19 state_dict (or a path to its .pth file) for the
        layers this extractor keeps; classifier / deeper feature entries are ignored."""
        if isinstance(state_dict, str):
            state_dict = torch.load(state_dict, map_location="cpu")
        own = self.state_dict()
        picked = {k: v for k, v in state_dict.items() if k in own}
        missing = [k for k in own if k not in picked]
        if missing:
            raise KeyError("vgg19 state_dict lacks %s" % missing)
        self.load_state_dict(picked)
        from .layers import bump_weight_epoch
        bump_weight_epoch()
        return self

    def forward(self, x):
        with torch.no_grad():
            out = x.detach()
            mods = list(self.features)
            i = 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, nn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                    out = m.run(out, ACT_RELU)   # conv + bias + ReLU in one kernel
                    i += 2
                else:
                    out = m(out)
                    i += 1
            return out
