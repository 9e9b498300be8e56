"""Tensor-side helpers of the reference's utils.py that sit on the training loop: weight
initialisers (utils.py:76-113), border crop (`shave`, :197-205), PSNR (:208-216) and the
mean/std normalisation constants (:219-239).  Plotting / GIF / PNG helpers are out of scope
(SURVEY.md §2 rows 11-12)."""
import torch


def _init_by_classname(m, conv_linear_init, norm_init):
    name = m.__class__.__name__
    if any(name.find(k) != -1 for k in ("Linear", "Conv2d", "ConvTranspose2d")):
        conv_linear_init(m.weight)
        if getattr(m, "bias", None) is not None:
            m.bias.data.zero_()
    elif name.find("Norm") != -1 and getattr(m, "weight", None) is not None:
        # (parameter-free norms — InstanceNorm2d's default — have nothing to initialise; the reference's
        #  `m.weight.data.normal_` would raise on them, utils.py:89)
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
    """utils.py:208-216 (prediction clamped to [0,1], peak 1.0; 100 for identical images).  GPU tensors: computed on
    the device by srk_psnr and returned as a 0-dim DEVICE tensor — no host copy and no sync per image (float() it, or
    torch.stack a list of them, when the numbers are needed).  CPU tensors (what the reference passes) are moved to the
    current device first; there is no host arithmetic."""
    from . import ops
    if not pred.is_cuda:
        pred = pred.to("cuda")
    return ops.psnr(pred.float(), gt.to(pred.device).float())[0]


VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)
VGG_DENORM_MEAN = (-2.118, -2.036, -1.804)
VGG_DENORM_STD = (4.367, 4.464, 4.444)


def norm(img, vgg=False):
    """utils.py:219-229 applied per channel on a [C,H,W] or [B,C,H,W] tensor (the reference's 4-D use is broken on
    its own stack; this is the intended arithmetic).  One kernel (srk_channel_affine), bit-equal
    to torchvision's Normalize (sub_(mean).div_(std))."""
    from . import ops
    mean, std = (VGG_MEAN, VGG_STD) if vgg else ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    return ops.channel_affine(img, mean, std)


def denorm(img, vgg=False):
    """utils.py:232-239: vgg -> Normalize(mean=[-2.118,-2.036,-1.804], std=[4.367,4.464,4.444]); else
    ((img + 1) / 2).clamp(0, 1) — written as (img - (-1)) / 2, the same two IEEE operations."""
    from . import ops
    if vgg:
        return ops.channel_affine(img, VGG_DENORM_MEAN, VGG_DENORM_STD)
    return ops.channel_affine(img, (-1.0,) * 8, (2.0,) * 8, clamp01=True)


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
