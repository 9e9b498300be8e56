"""Parity at BASELINE.json's full sizes (c2 ESPCN x4 inference 64 x 256x256, c3 VDSR x4 training 256 x 41x41,
c4 EDSR x4 training 128 x 32x32 -> 128x128).  The small golden fixtures pin the arithmetic; these tests pin the
things that only exist at full size: the tile pickers, the large-problem kernel selection (256-pixel tiles,
resident-filter kernel, persistent weight-gradient blocks with many tiles each, split-K slab reduction over hundreds of
blocks), batch independence, and the data-parallel shard algebra.  The checker is the CPU oracle on the same seeded
inputs (the oracle finishes these sizes in seconds) plus size-independent properties."""
import numpy as np
import pytest
import torch

from conftest import assert_close_elementwise, rel_err
from oracle import fill, ref_modules as R

pytestmark = pytest.mark.gpu


def _pkg():
    import pytorch_super_resolution_model_collection_amd as pkg
    return pkg


def _write_record(name, obj):
    """Numbers a reviewer should see, not only pass / fail: gpurun_out/ travels back from the GPU box (the builder copies
    them to profiles/ under the round's tag)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), "w") as fh:
            json.dump(obj, fh, indent=1)
    except OSError:
        pass


@pytest.fixture(autouse=True)
def _restore_precision():
    pkg = _pkg()
    yield
    pkg.ops.set_precision("mixed")


def test_c2_espcn_full_batch_inference(gpu):
    """ESPCN x4 on the full c2 batch: images 0, 31, 63 of the batch-64 output equal the oracle on those images
    (1e-4; contract 1e-3), and permuting the batch permutes the output bit-exactly (batch independence)."""
    pkg = _pkg()
    net = pkg.ESPCNNet(3, 64, 4)
    fill.fill_module(net, 1234, 1.0)
    ora = fill.fill_module(R.ESPCN(3, 64, 4), 1234, 1.0).eval()
    x = fill.rand((64, 3, 256, 256), 2024)
    net.to(gpu).eval()
    with torch.no_grad():
        y = net(x.to(gpu))
        assert tuple(y.shape) == (64, 3, 992, 992)
        pick = [0, 31, 63]
        ref = ora(x[pick])
        assert rel_err(y[pick], ref) < 1e-4
        perm = torch.arange(63, -1, -1)
        y2 = net(x[perm].to(gpu))
        assert torch.equal(y2, y[perm.to(gpu)])


def test_c2_first_layer_linearity_full_size(gpu):
    """conv5(3->64) without activation is linear: f(a x1 + b x2) - bias = a (f(x1) - bias) + b (f(x2) - bias)
    at 64 x 256x256 (row-packed kernel reading NCHW in place), to fp32/bf16x3 accuracy."""
    pkg = _pkg()
    conv = pkg.layers.Conv2d(3, 64, 5, 1, 0, bias=False)
    fill.fill_module(conv, 5, 1.0)
    conv.to(gpu)
    x1, x2 = fill.rand((64, 3, 256, 256), 11).to(gpu), fill.rand((64, 3, 256, 256), 12).to(gpu)
    with torch.no_grad():
        lhs = conv(0.75 * x1 - 0.5 * x2)
        rhs = 0.75 * conv(x1) - 0.5 * conv(x2)
    assert rel_err(lhs, rhs) < 5e-5


def _one_step_case(pkg, gpu, kind, prod_net, ora_net, x, t, step_kind, clip, lr, tol_grad):
    """One full-size training step in the product and in the oracle: loss, every parameter gradient
    (checked through the parameter update of plain SGD-free arithmetic: we compare the gradients themselves)."""
    prod_net.to(gpu).train()
    flat = pkg.optim.FlatParams(prod_net)
    opt = pkg.optim.make_optimizer(kind, flat, lr)
    opt.zero_grad()
    loss_fn = pkg.ops.mse_loss if step_kind == "mse" else pkg.ops.l1_loss
    loss = loss_fn(prod_net(x.to(gpu)), t.to(gpu))
    loss.backward()
    ora_net.train()
    ora_net.zero_grad()
    oloss = (torch.nn.functional.mse_loss if step_kind == "mse" else torch.nn.functional.l1_loss)(ora_net(x), t)
    oloss.backward()
    assert abs(float(loss) - float(oloss)) <= 2e-5 * abs(float(oloss)) + 1e-9
    ogr = dict((n, p.grad) for n, p in ora_net.named_parameters())
    gmax = max(float(g.abs().max()) for g in ogr.values())
    worst = 0.0
    for n, p in prod_net.named_parameters():
        g, og = p.grad.detach().cpu().double(), ogr[n].double()
        err = float((g - og).abs().max()) / max(float(og.abs().max()), 1e-3 * gmax)
        worst = max(worst, err)
        assert err < tol_grad, (n, err)
        # ... and element by element (|a - b| <= atol + rtol |b|, atol = rtol * rms(b)): small elements must be right too
        assert_close_elementwise(p.grad, ogr[n], tol_grad, what="%s full-size grad %s" % (kind, n))
    return worst, flat, opt


def test_c3_vdsr_full_size_step(gpu):
    """VDSR x4, 256 patches of 41x41 (c3): loss and all 20 weight gradients of one step vs the oracle."""
    pkg = _pkg()
    net = pkg.VDSRNet(3, 64, 18)
    fill.fill_module(net, 7, 1.0)
    ora = fill.fill_module(R.VDSR(3, 64, 18), 7, 1.0)
    x, t = fill.rand((256, 3, 41, 41), 101), fill.rand((256, 3, 41, 41), 102)
    _one_step_case(pkg, gpu, "vdsr", net, ora, x, t, "mse", 0.4, 1e-2, 1e-3)


def test_c4_edsr_full_size_step_and_shard_algebra(gpu):
    """EDSR-baseline x4, 128 patches 32x32 -> 128x128 (c4): loss and all 74 parameter gradients vs the oracle;
    then the data-parallel identity at full size: the SUM over 8 contiguous shards of 16 of the gradients seeded
    with 1/8 (what 8 ranks all-reduce) equals the full-batch gradient."""
    pkg = _pkg()
    net = pkg.EDSRNet(3, 64, 16)
    fill.fill_module(net, 9, 0.5)
    ora = fill.fill_module(R.EDSR(3, 64, 16), 9, 0.5)
    x, t = fill.rand((128, 3, 32, 32), 201), fill.rand((128, 3, 128, 128), 202)
    _, flat, opt = _one_step_case(pkg, gpu, "edsr", net, ora, x, t, "l1", None, 1e-4, 1e-3)
    full = flat.grad.clone()
    acc = torch.zeros_like(full)
    seed = torch.tensor(1.0 / 8, device=gpu)
    for r in range(8):
        lo, hi = pkg.dp.shard_range(128, r, 8)
        assert (lo, hi) == (16 * r, 16 * r + 16)
        opt.zero_grad()
        loss = pkg.ops.l1_loss(net(x[lo:hi].to(gpu)), t[lo:hi].to(gpu))
        loss.backward(seed)
        acc += flat.grad
    assert rel_err(acc, full) < 2e-4


def test_c4_shard_grouped_weight_gradients(gpu):
    """The strong-scaled shard of c4 (16 patches per GPU): one EDSR step with the deferred, grouped weight gradients
    (33 body layers in one launch) against the oracle at the contract tolerance, and against the same step with one
    weight-gradient launch per layer (ops.DEFER_WGRAD = False) to summation-order accuracy."""
    pkg = _pkg()
    x, t = fill.rand((16, 3, 32, 32), 211), fill.rand((16, 3, 128, 128), 212)
    ora = fill.fill_module(R.EDSR(3, 64, 16), 9, 0.5)
    grads = []
    for defer in (True, False):
        pkg.ops.DEFER_WGRAD = defer
        try:
            net = pkg.EDSRNet(3, 64, 16)
            fill.fill_module(net, 9, 0.5)
            if defer:
                _, flat, opt = _one_step_case(pkg, gpu, "edsr", net, ora, x, t, "l1", None, 1e-4, 1e-3)
            else:
                net.to(gpu).train()
                flat = pkg.optim.FlatParams(net)
                flat.zero_grad()
                flat.plan.pack()
                pkg.ops.l1_loss(net(x.to(gpu)), t.to(gpu)).backward()
            grads.append(flat.grad.clone())
        finally:
            pkg.ops.DEFER_WGRAD = True
    assert rel_err(grads[0], grads[1]) < 2e-5


def test_c4_shard_step_uses_small_problem_kernels_and_matches(gpu):
    """The per-GPU shard of c4 (16 patches) selects the channel-split 64-pixel blocks (small-problem
    configuration); its forward must equal the same images inside the full batch (large-problem kernels)."""
    pkg = _pkg()
    net = pkg.EDSRNet(3, 64, 16)
    fill.fill_module(net, 9, 0.5)
    net.to(gpu).eval()
    x = fill.rand((128, 3, 32, 32), 201).to(gpu)
    with torch.no_grad():
        full = net(x)
        part = net(x[48:64])
    assert rel_err(part, full[48:64]) < 5e-5


def test_c5_srgan_full_size_adversarial_step(gpu):
    """SRGAN x4 (c5) at the reference's default batch: 16 LR crops 32x32 -> 128x128, one full D step + G step
    (srgan.py:249-310) against the oracle: both losses, and the updated parameters of G and D (the 9x9 convs at
    128x128, the stride-2 discriminator convs, the 32768->1024 Linear and every BatchNorm run at full size here)."""
    pkg = _pkg()
    G, D = pkg.SRGANGenerator(3, 64, 16), pkg.SRGANDiscriminator(3, 64, 128)
    fill.fill_module(G, 5, 0.7)
    fill.fill_module(D, 6, 1.0)
    oG = fill.fill_module(R.Generator(3, 64, 16), 5, 0.7)
    oD = fill.fill_module(R.Discriminator(3, 64, 128), 6, 1.0)
    G.to(gpu).train()
    D.to(gpu).train()
    g_opt = pkg.optim.make_optimizer("srgan_g", pkg.optim.FlatParams(G), 1e-4)
    d_opt = pkg.optim.make_optimizer("srgan_d", pkg.optim.FlatParams(D), 1e-2)
    step = pkg.trainers.srgan_step(G, D, g_opt, d_opt)
    og_opt = R.make_optimizer("srgan_g", oG.parameters(), 1e-4)
    od_opt = R.make_optimizer("srgan_d", oD.parameters(), 1e-2)
    lr_img, hr_img = fill.rand((16, 3, 32, 32), 501), fill.rand((16, 3, 128, 128), 502)
    d_loss, g_loss = step(lr_img.to(gpu), hr_img.to(gpu))
    od_loss, og_loss = R.step_srgan(oG, oD, og_opt, od_opt, lr_img, hr_img)
    assert abs(float(d_loss) - od_loss) <= 1e-4 * abs(od_loss)
    assert abs(float(g_loss) - og_loss) <= 1e-4 * abs(og_loss)
    # Gradients left in the buffers by the step (G: the G step's; D: D step + the G step's accumulation, a reference
    # quirk both sides share), per tensor in the L2 norm, against an fp64 run of the oracle.  At this size the problem
    # is ill-conditioned in fp32: the backward sums of a BatchNorm that feeds another BatchNorm cancel to ~1e-4 of their
    # mass, so the ~3e-7 rounding of a conv output comes back as ~3e-3 in a gradient (tools/srgan_bisect.py: with the
    # convs alone evaluated in fp64 the same step is within 1e-4, with only the BatchNorm in fp64 nothing changes).
    # The reference's own CPU path is not reproducible below that either: stock torch fp32 deviates from its fp64 run by
    # 2.0e-3 with oneDNN convolutions and by 5.5e-3 with ATen's native ones (tools/c5_oracle_spread.py) — two equally
    # valid evaluation orders of the same torch.nn modules.  The bar is therefore the contract 1e-3 or 1.5x the larger
    # of those two deviations, both measured here on the same inputs.
    import copy
    oG64 = fill.fill_module(R.Generator(3, 64, 16), 5, 0.7).double()
    oD64 = fill.fill_module(R.Discriminator(3, 64, 128), 6, 1.0).double()
    R.step_srgan(oG64, oD64, R.make_optimizer("srgan_g", oG64.parameters(), 1e-4),
                 R.make_optimizer("srgan_d", oD64.parameters(), 1e-2), lr_img.double(), hr_img.double())
    # second fp32 evaluation of the oracle: ATen's native convolutions instead of oneDNN's
    oG_n = fill.fill_module(R.Generator(3, 64, 16), 5, 0.7)
    oD_n = fill.fill_module(R.Discriminator(3, 64, 128), 6, 1.0)
    prev = torch.backends.mkldnn.enabled
    torch.backends.mkldnn.enabled = False
    try:
        R.step_srgan(oG_n, oD_n, R.make_optimizer("srgan_g", oG_n.parameters(), 1e-4),
                     R.make_optimizer("srgan_d", oD_n.parameters(), 1e-2), lr_img, hr_img)
    finally:
        torch.backends.mkldnn.enabled = prev
    record = {"what": "tests/test_fullsize_gpu.py::test_c5_srgan_full_size_adversarial_step: worst per-tensor L2 error of "
                      "the gradients a full-size SRGAN step leaves behind, against an fp64 run of the oracle"}
    for tag, net, ora, ora_n, ora64 in (("G", G, oG, oG_n, oG64), ("D", D, oD, oD_n, oD64)):
        g32 = dict((n, p.grad.double()) for n, p in ora.named_parameters())
        g32n = dict((n, p.grad.double()) for n, p in ora_n.named_parameters())
        g64 = dict((n, p.grad) for n, p in ora64.named_parameters())
        gmax = max(float(g.abs().max()) for g in g64.values())
        worst_p, worst_o = 0.0, 0.0
        for n, p in net.named_parameters():
            den = max(float(g64[n].norm()), 1e-3 * gmax * g64[n].numel() ** 0.5)
            worst_p = max(worst_p, float((p.grad.detach().cpu().double() - g64[n]).norm()) / den)
            worst_o = max(worst_o, float((g32[n] - g64[n]).norm()) / den, float((g32n[n] - g64[n]).norm()) / den)
        # FIXED regression bars a little above what rounds 2 - 5 measured (G 2.31e-3, D 2.60e-3 vs float64; DESIGN 10.4
        # explains why no fp32 evaluation of these modules gets below ~1.4e-3): a change that moves a gradient error from
        # 2.6e-3 to 3.5e-3 fails here, whatever torch's own spread is on that day ...
        bar = {"G": 3.0e-3, "D": 3.2e-3}[tag]
        record[tag] = {"product_vs_fp64": worst_p, "torch_fp32_vs_fp64_worse_of_onednn_and_native": worst_o, "fixed_bar": bar}
        assert worst_p <= bar, (tag, worst_p, bar)
        # ... and, as before, never worse than 1.5x (G: 1.0x) the spread of the reference's own two fp32 evaluations
        assert worst_p <= max(1e-3, 1.5 * worst_o), (worst_p, worst_o)
        if tag == "G":   # the better-conditioned net: no slack over the reference's own fp32 spread
            assert worst_p <= max(1e-3, 1.0 * worst_o), (worst_p, worst_o)
    # the UPDATED parameters (what the docstring promises): Adam's first step moves every G parameter by ~lr whatever the
    # gradient's size, so G is compared on the update direction where the gradient is decidable; D (SGD + Nesterov,
    # lr 1e-2 / 100) on the update itself
    p0G = dict((n, p.detach().clone()) for n, p in fill.fill_module(R.Generator(3, 64, 16), 5, 0.7).named_parameters())
    upd = {}
    for tag, net, ora in (("G", G, oG), ("D", D, oD)):
        num = den = 0.0
        for (n, p), (_, q) in zip(net.named_parameters(), ora.named_parameters()):
            num += float((p.detach().cpu().double() - q.detach().double()).pow(2).sum())
            den += float(q.detach().double().pow(2).sum())
        upd[tag] = (num / den) ** 0.5
        assert upd[tag] < 1e-3, (tag, upd[tag])          # parameters after the step, relative L2 over the whole net (contract)
    agree = total = 0
    for n, p in G.named_parameters():
        q = dict(oG.named_parameters())[n].detach()
        d_ref, d_got = (q - p0G[n]), (p.detach().cpu() - p0G[n])
        strong = d_ref.abs() > 0.5e-4      # |Adam step| ~ lr = 1e-4 where the gradient is far from zero
        agree += int(((d_ref > 0) == (d_got > 0))[strong].sum())
        total += int(strong.sum())
    assert total > 0 and agree >= 0.995 * total, (agree, total)   # (a sign can differ where |gradient| is below its ~2e-3 error)
    record["updated_parameters_rel_l2"] = upd
    record["G_adam_step_sign_agreement"] = [agree, total]
    _write_record("c5_parity.json", record)
