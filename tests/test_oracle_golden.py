"""CPU: the oracle (oracle/ref_modules.py) against the committed golden vectors that were produced
by the REFERENCE's classes (tests/golden/make_golden.py asserted bit-equality at generation time).
On a different host CPU oneDNN may pick other kernels, hence a small tolerance here."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import fill, ref_modules as R

TOL = 2e-5

NETS = {
    "srcnn": (R.SRCNN, (3, 64), (2, 3, 20, 20), 1.0),
    "espcn": (R.ESPCN, (3, 64, 4), (2, 3, 16, 16), 1.0),
    "fsrcnn": (R.FSRCNN, (3, 4, 56, 12, 4), (2, 3, 12, 12), 1.0),
    "vdsr": (R.VDSR, (3, 64, 18), (2, 3, 13, 13), 1.0),
    "edsr": (R.EDSR, (3, 64, 16), (2, 3, 8, 8), 0.5),
    "lapsrn": (R.LapSRN, (3, 64, 10), (1, 3, 8, 8), 1.0),
    "srgan_g": (R.Generator, (3, 64, 16), (2, 3, 8, 8), 0.7),
    "srgan_d": (R.Discriminator, (3, 64, 32), (2, 3, 32, 32), 1.0),
}


@pytest.mark.parametrize("name", list(NETS))
def test_oracle_net_matches_reference_vectors(nets_golden, name):
    cls, args, ishape, gain = NETS[name]
    net = fill.fill_module(cls(*args), 1234, gain)
    net.train()
    x = fill.rand(ishape, 4321).requires_grad_(True)
    out = net(x)
    outs = out if isinstance(out, (tuple, list)) else (out,)
    loss = 0
    for i, o in enumerate(outs):
        assert rel_err(o, nets_golden["%s.out%d" % (name, i)]) < TOL
        loss = loss + (o * fill.randn(tuple(o.shape), 77 + i)).sum() / o.numel()
    loss.backward()
    assert rel_err(x.grad, nets_golden[name + ".dx"]) < 10 * TOL
    names = [str(n) for n in nets_golden[name + ".grad_names"]]
    params = dict(net.named_parameters())
    assert rel_err(params[names[0]].grad, nets_golden[name + ".grad_first"]) < 10 * TOL
    assert rel_err(params[names[-1]].grad, nets_golden[name + ".grad_last"]) < 10 * TOL


def test_oracle_srcnn_c1_trajectory(train_golden):
    """BASELINE config c1 (SRCNN x2, 32x32 LR -> 64x64 bicubic stand-in, batch 16, CPU)."""
    net = fill.fill_module(R.SRCNN(3, 64), 99)
    opt = R.make_optimizer("srcnn", net.parameters(), 1e-2)
    losses = [R.step_mse(net, opt, fill.rand((16, 3, 64, 64), 10 + i), fill.rand((16, 3, 48, 48), 20 + i))
              for i in range(3)]
    assert rel_err(np.array(losses), train_golden["srcnn_c1_lr1e-2.losses"]) < TOL
    for k, v in net.state_dict().items():
        assert rel_err(v, train_golden["srcnn_c1_lr1e-2.final.%s" % k]) < 10 * TOL


def test_oracle_vdsr_and_edsr_trajectories(train_golden):
    net = fill.fill_module(R.VDSR(3, 64, 18), 99)
    opt = R.make_optimizer("vdsr", net.parameters(), 1e-2)
    losses = [R.step_mse(net, opt, fill.rand((4, 3, 17, 17), 50 + i), fill.rand((4, 3, 17, 17), 60 + i), clip=0.4)
              for i in range(3)]
    assert rel_err(np.array(losses), train_golden["vdsr.losses"]) < 10 * TOL
    net = fill.fill_module(R.EDSR(3, 64, 16), 99, 0.5)
    opt = R.make_optimizer("edsr", net.parameters(), 1e-4)
    losses = [R.step_l1(net, opt, fill.rand((4, 3, 8, 8), 70 + i), fill.rand((4, 3, 32, 32), 80 + i))
              for i in range(3)]
    assert rel_err(np.array(losses), train_golden["edsr.losses"]) < 10 * TOL


def test_reference_quirks_are_preserved():
    """shared BN in ResnetBlock, shared LapSRN branch, FSRCNN deconv geometry."""
    rb = R.ResnetBlock(8, activation='prelu')
    assert sum(1 for _ in rb.modules() if isinstance(_, torch.nn.BatchNorm2d)) == 1
    lap = R.LapSRN(3, 8, 2)
    assert lap.convt_F1[0] is lap.convt_F2[0]
    assert len(lap.state_dict()) == 2 * (2 + 1) + 5  # aliased keys are listed twice
    fs = R.FSRCNN(3, 4, 56, 12, 4)
    assert tuple(fs(torch.zeros(1, 3, 12, 12)).shape) == (1, 3, 4 * (12 - 5) + 4, 4 * (12 - 5) + 4)


R2_BLOCKS = {
    "rnc": (lambda: R.Upsample2xBlock(8, 12, upsample='rnc', activation='relu', norm=None), ("rand", (2, 8, 5, 6), 601), 602, 1.0),
    "rnc_prelu": (lambda: R.Upsample2xBlock(16, 16, upsample='rnc', activation='prelu', norm=None), ("rand", (1, 16, 7, 4), 603), 604, 1.0),
    "inst": (lambda: R.ConvBlock(8, 16, 3, 1, 1, activation='lrelu', norm='instance'), ("randn", (3, 8, 9, 7), 611), 612, 1.0),
    "resinst": (lambda: R.ResnetBlock(16, activation='relu', norm='instance'), ("randn", (2, 16, 6, 8), 613), 614, 0.7),
    "dense_bn": (lambda: R.DenseBlock(24, 10, activation='lrelu', norm='batch'), ("randn", (6, 24), 621), 622, 1.0),
}


@pytest.mark.parametrize("tag", list(R2_BLOCKS))
def test_oracle_block_variants_match_reference_vectors(blocks_r2, tag):
    """Block variants no reference net instantiates ('rnc' upsampler, instance norm, DenseBlock + BatchNorm1d) against
    tests/golden/blocks_r2.npz (reference classes, make_golden_r2.py)."""
    make, (kind, shape, xs), gs, gain = R2_BLOCKS[tag]
    mod = fill.fill_module(make(), 4242, gain)
    mod.train()
    x = getattr(fill, kind)(shape, xs).requires_grad_(True)
    y = mod(x)
    (y * fill.randn(tuple(y.shape), gs)).sum().backward()
    assert rel_err(y, blocks_r2[tag + ".y"]) < TOL
    assert rel_err(x.grad, blocks_r2[tag + ".dx"]) < 10 * TOL
    for n, p in mod.named_parameters():
        assert rel_err(p.grad, blocks_r2["%s.grad.%s" % (tag, n)]) < 10 * TOL, n
