"""GPU parity of the residual block with both convolutions in one launch (srk_resblock2_*, conv_res2.hip):
against the oracle's ResnetBlock (reference: base_networks.py:109-150, norm=None) in float64 on the CPU, and against
the same block run as two separate conv launches (what the fused kernels replace)."""
import numpy as np
import pytest
import torch

from conftest import assert_close_elementwise, rel_err
from oracle import fill, ref_modules as R

pytestmark = pytest.mark.gpu


def _pkg():
    import pytorch_super_resolution_model_collection_amd as pkg
    return pkg


def _run(pkg, blk, x, dy, fused):
    old = pkg.ops.RES2
    pkg.ops.RES2 = fused
    try:
        for p in blk.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        if pkg.ops.F16X3:   # as inside a net, where the block's input comes from a conv kernel: worth the srk_absmax pass
            pkg.ops._tag_amax(xi, None)
        y = blk(xi)
        y.backward(dy)
        torch.cuda.synchronize()
        return y.detach(), xi.grad.detach(), {k: p.grad.detach().clone() for k, p in blk.named_parameters()}
    finally:
        pkg.ops.RES2 = old


# (N, H, W): the EDSR shard, ragged edges in both directions, a single tile, a tile row, one pixel
SHAPES = [(16, 32, 32), (3, 13, 21), (1, 8, 8), (2, 5, 40), (1, 1, 1), (2, 17, 9), (2, 6, 11)]


def _frac_outside(a, b, rtol):
    a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
    atol = rtol * float(np.sqrt(np.mean(b * b))) if b.size else 0.0
    return float(np.mean(np.abs(a - b) > atol + rtol * np.abs(b)))


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("bias", [True, False])
def test_resblock2_matches_oracle_and_separate_launches(gpu, shape, bias):
    """The ReLU between the convs makes the gradients discontinuous in the pre-activation: an element within rounding
    of zero may legitimately land on either side (the reference's own fp32 sum order decides there, too).  So the mask is
    checked against the float64 oracle wherever it is decidable, and the gradients against the oracle's float64 backward
    evaluated WITH the mask the kernel used -- exact arithmetic parity, element-wise, no luck involved."""
    import torch.nn.functional as F
    pkg = _pkg()
    lib = pkg._lib.load()
    n, h, w = shape
    ref = R.ResnetBlock(64, bias=bias, activation='relu', norm=None)
    fill.fill_module(ref, seed=900 + h)
    ref = ref.double()
    blk = pkg.base_networks.ResnetBlock(64, bias=bias, activation='relu', norm=None).to(gpu)
    blk.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    x = fill.randn((n, 64, h, w), 901 + w)
    dy = fill.randn((n, 64, h, w), 902 + w)
    assert pkg.ops.resblock2_applicable(x.to(gpu), blk.conv1.weight, blk.conv2.weight)
    w1, w2 = ref.conv1.weight.detach(), ref.conv2.weight.detach()
    b1 = ref.conv1.bias.detach() if bias else None
    b2 = ref.conv2.bias.detach() if bias else None
    # ---- forward through the C ABI: intermediate and output against the float64 oracle
    CL = torch.channels_last
    xg = x.to(gpu).contiguous(memory_format=CL)
    mid_g, out_g = torch.empty_like(xg), torch.empty_like(xg)
    P = pkg._lib.ptr
    wp1, wp2 = pkg.ops.pack_weight_fwd(blk.conv1.weight, False, 0), pkg.ops.pack_weight_fwd(blk.conv2.weight, False, 0)
    # the fp32-faithful class the module path uses: f16x3 (with the input's running maximum) or bf16x6
    algo = pkg._lib.ALGO_MFMA_F16X3 if pkg.ops.F16X3 else pkg._lib.ALGO_MFMA_BF16X6
    xa = pkg.ops.amax_of(xg) if pkg.ops.F16X3 else None
    ya = torch.zeros(pkg._lib.AMAX_FLOATS, device=gpu)
    rc = lib.srk_resblock2_forward(n, h, w, 64, P(xg), P(wp1), P(blk.conv1.bias), P(wp2), P(blk.conv2.bias), P(mid_g),
                                   P(out_g), algo, P(xa), P(ya), pkg._lib.stream_ptr())
    assert rc == 0, lib.srk_last_error_string()
    torch.cuda.synchronize()
    z = F.conv2d(x.double(), w1, b1, padding=1)
    assert_close_elementwise(mid_g, z.clamp(min=0), 1e-5, what="fused mid vs oracle")
    assert_close_elementwise(out_g, ref(x.double()).detach(), 1e-5, what="fused forward vs oracle")
    mask = (mid_g > 0).cpu()
    decidable = z.abs() > 1e-5 * float(z.pow(2).mean().sqrt())
    assert bool(((z > 0) == mask)[decidable].all())
    # ---- float64 backward with the kernel's mask
    dmid_ref = F.conv_transpose2d(dy.double(), w2, padding=1) * mask
    dx_ref = F.conv_transpose2d(dmid_ref, w1, padding=1) + dy.double()
    mid64 = z.clamp(min=0) * mask
    want = {"conv2.weight": torch.nn.grad.conv2d_weight(mid64, w2.shape, dy.double(), padding=1),
            "conv1.weight": torch.nn.grad.conv2d_weight(x.double(), w1.shape, dmid_ref, padding=1)}
    if bias:
        want["conv2.bias"] = dy.double().sum((0, 2, 3))
        want["conv1.bias"] = dmid_ref.sum((0, 2, 3))
    # ---- backward through the C ABI (bf16x3 products, ~5e-6 relative per product)
    dyg = dy.to(gpu).contiguous(memory_format=CL)
    dmid_g, dx_g = torch.empty_like(xg), torch.empty_like(xg)
    wb1, wb2 = pkg.ops.pack_weight_bwd(blk.conv1.weight, False, 0), pkg.ops.pack_weight_bwd(blk.conv2.weight, False, 0)
    rc = lib.srk_resblock2_backward_data(n, h, w, 64, P(dyg), P(wb2), P(wb1), P(mid_g), P(dmid_g), P(dx_g), 0,
                                         pkg._lib.stream_ptr())
    assert rc == 0, lib.srk_last_error_string()
    torch.cuda.synchronize()
    assert_close_elementwise(dmid_g, dmid_ref, 1e-4, what="fused d_mid vs oracle")
    assert_close_elementwise(dx_g, dx_ref, 1e-4, what="fused dx vs oracle")
    # ---- the module path (autograd): fused vs oracle, and vs the two separate launches it replaces
    y1, dx1, g1 = _run(pkg, blk, x.to(gpu), dy.to(gpu), True)
    assert lib.srk_last_kernel_name().decode().startswith("k_res2<")   # the fused kernels did run
    y0, dx0, g0 = _run(pkg, blk, x.to(gpu), dy.to(gpu), False)
    assert not lib.srk_last_kernel_name().decode().startswith("k_res2<")
    assert torch.equal(y1, out_g) and torch.equal(dx1, dx_g)
    assert float(ya.max()) == float(out_g.abs().max())      # the running maximum the next layer scales by
    # (fused: f16x3; the two separate small-problem launches keep bf16x6 -- ops._f16x3_pays: two fp32-faithful arithmetics,
    #  each within ~1e-6 of max|y| of float64)
    assert_close_elementwise(y1, y0, 5e-6 if pkg.ops.F16X3 else 2e-6, what="fused forward vs separate launches")
    for k, g in want.items():
        assert_close_elementwise(g1[k], g, 1e-4, what="fused path grad " + k)
    # (a mask element the two paths decide differently moves 9 x 64 gradient elements: allow a handful)
    assert _frac_outside(dx1, dx0, 2e-5) < 2e-3
    for k in want:
        assert rel_err(g1[k], g0[k]) < 1e-3, k


def test_resblock2_only_for_small_problems(gpu):
    """Large batches keep the separate kernels (the fused tile does 1.56x the first conv's matrix work); other channel
    counts, other activations and the exact-fp32 mode never take the fused path."""
    pkg = _pkg()
    lib = pkg._lib.load()
    assert lib.srk_resblock2_supported(16, 32, 32, 64) == 1
    assert lib.srk_resblock2_supported(127, 32, 32, 64) == 1
    assert lib.srk_resblock2_supported(128, 32, 32, 64) == 0      # 8 tiles per CU: two launches of the canvas ring kernel (round 6)
    assert lib.srk_resblock2_supported(256, 32, 32, 64) == 0
    assert lib.srk_resblock2_supported(16, 32, 32, 32) == 0
    blk = pkg.base_networks.ResnetBlock(64, activation='relu', norm=None).to(gpu)
    x = torch.randn(2, 64, 16, 16, device=gpu)
    assert pkg.ops.resblock2_applicable(x, blk.conv1.weight, blk.conv2.weight)
    old = pkg.ops.get_precision()
    try:
        pkg.ops.set_precision("fp32")
        assert not pkg.ops.resblock2_applicable(x, blk.conv1.weight, blk.conv2.weight)
    finally:
        pkg.ops.set_precision(old)
    lre = pkg.base_networks.ResnetBlock(64, activation='lrelu', norm=None).to(gpu)
    y = lre(x.clone().requires_grad_(True))
    y.sum().backward()
    assert not lib.srk_last_kernel_name().decode().startswith("k_res2<")
    # error behaviour of the C entry points: unsupported sizes are refused, not mis-computed
    wp = pkg.ops.pack_weight_fwd(blk.conv1.weight, False, 0)
    big = torch.empty(1, device=gpu)
    rc = lib.srk_resblock2_forward(256, 32, 32, 64, pkg._lib.ptr(big), pkg._lib.ptr(wp), None, pkg._lib.ptr(wp), None,
                                   pkg._lib.ptr(big), pkg._lib.ptr(big), 0, None, None, pkg._lib.stream_ptr())
    assert rc != 0 and b"unsupported" in lib.srk_last_error_string()


def test_resblock2_in_graphed_edsr_step(gpu):
    """The 16-patch EDSR shard step (flat gradient buffers, deferred grouped weight gradients, one hipGraph) with and
    without the fused blocks: same loss trajectory over 3 steps and the same gradients in the first one.  (Parameters
    are not compared: Adam turns a gradient element that rounds to the other side of zero into a +-lr difference.)"""
    pkg = _pkg()
    res = {}
    for fused in (True, False):
        old = pkg.ops.RES2
        pkg.ops.RES2 = fused
        try:
            net = pkg.EDSRNet(3, 64, 4)
            fill.fill_module(net, 3, 0.5)
            net.to(gpu).train()
            flat, opt, dp, _ = pkg.trainers.build("edsr", net, 1e-4)
            lr = fill.rand((16, 3, 32, 32), 41).to(gpu)
            hr = fill.rand((16, 3, 128, 128), 42).to(gpu)
            step = pkg.trainers.GraphedStep(net, opt, pkg.ops.l1_loss, (lr, hr), warmup=1)
            losses = [float(step(lr, hr).detach())]
            torch.cuda.synchronize()
            grads = opt.flat.grad.detach().clone()
            losses += [float(step(lr, hr).detach()) for _ in range(2)]
            torch.cuda.synchronize()
            res[fused] = (losses, grads)
            step.close()
        finally:
            pkg.ops.RES2 = old
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) <= 2e-5 * abs(b), (res[True][0], res[False][0])
    assert float(res[True][1].abs().max()) > 0
    assert rel_err(res[True][1], res[False][1]) < 1e-3
