"""Tensor-side helpers of the reference's utils.py that sit on the training loop: weight
initialisers (utils.py:76-113), border crop (`shave`, :197-205), PSNR (:208-216) and the
mean/std normalisation constants (:219-239).  Plotting / GIF / PNG helpers are out of scope
(SURVEY.md §2 rows 11-12)."""
import math

import torch


def _init_by_classname(m, conv_linear_init, norm_init):
    name = m.__class__.__name__
    if any(name.find(k) != -1 for k in ("Linear", "Conv2d", "ConvTranspose2d")):
        conv_linear_init(m.weight)
        if getattr(m, "bias", None) is not None:
            m.bias.data.zero_()
    elif name.find("Norm") != -1:
        norm_init(m.weight)
        if m.bias is not None:
            m.bias.data.zero_()


def weights_init_normal(m, mean=0.0, std=0.02):
    """utils.py:76-93: N(mean, std) for Linear/Conv2d/ConvTranspose2d weights, zero bias,
    N(1, 0.02) for *Norm* weights — matched by class NAME exactly like the reference."""
    _init_by_classname(m, lambda w: w.data.normal_(mean, std), lambda w: w.data.normal_(1.0, 0.02))


def weights_init_kaming(m):
    """utils.py:96-113 (kaiming-normal, fan_in, gain sqrt(2))."""
    _init_by_classname(m, lambda w: torch.nn.init.kaiming_normal_(w), lambda w: w.data.normal_(1.0, 0.02))


def shave(imgs, border_size=0):
    """utils.py:197-205: crop `border_size` pixels from every side (a view; no copy needed)."""
    if border_size == 0:
        return imgs
    return imgs[..., border_size:-border_size, border_size:-border_size]


def PSNR(pred, gt):
    """utils.py:208-216 (prediction clamped to [0,1], peak 1.0)."""
    pred = pred.detach().float().cpu().clamp(0, 1)
    mse = torch.mean((pred - gt.detach().float().cpu()) ** 2).item()
    if mse == 0:
        return 100
    return 10 * math.log10(1.0 / mse)


VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)


def norm(img, vgg=False):
    """utils.py:219-229 applied per channel on a [C,H,W] or [B,C,H,W] tensor (the reference's 4-D use
    is broken on its own stack; this is the intended arithmetic)."""
    mean, std = (VGG_MEAN, VGG_STD) if vgg else ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    c = img.shape[-3]
    m = torch.tensor(mean[:c], dtype=img.dtype, device=img.device).view(-1, 1, 1)
    s = torch.tensor(std[:c], dtype=img.dtype, device=img.device).view(-1, 1, 1)
    return (img - m) / s


def denorm(img, vgg=False):
    """utils.py:232-239"""
    if vgg:
        c = img.shape[-3]
        m = torch.tensor((-2.118, -2.036, -1.804)[:c], dtype=img.dtype, device=img.device).view(-1, 1, 1)
        s = torch.tensor((4.367, 4.464, 4.444)[:c], dtype=img.dtype, device=img.device).view(-1, 1, 1)
        return (img - m) / s
    return ((img + 1) / 2).clamp(0, 1)


def img_interp(imgs, scale_factor, interpolation='bicubic'):
    """utils.py:242-269 — the per-image PIL round trip (ToPILImage -> resize -> ToTensor) as GPU kernels,
    bit-exact with it (ops.img_interp)."""
    from . import ops
    return ops.img_interp(imgs, scale_factor, interpolation)


def print_network(net):
    """utils.py:14-20"""
    num_params = sum(p.numel() for p in net.parameters())
    print(net)
    print('Total number of parameters: %d' % num_params)
