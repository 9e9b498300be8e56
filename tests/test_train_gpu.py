"""Train-trajectory parity on the MI355X: the reference's train-step bodies (transcribed in the
oracle and pinned bit-for-bit against the reference when the fixtures were generated) vs the HIP
path, 3 steps each: losses + parameter fingerprints (full final parameters for config c1)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import fill

pytestmark = pytest.mark.gpu
TOL = 2e-4  # contract 1e-3


def _pkg():
    import pytorch_super_resolution_model_collection_amd as pkg
    return pkg


def _check_params(net, golden, tag, tol=TOL):
    names = [str(n) for n in golden[tag + ".param_names"]]
    sums = golden[tag + ".param_sums"]
    sd = net.state_dict()
    for n, (s, l2) in zip(names, sums):
        t = sd[n].detach().double().cpu()
        assert abs(float(t.pow(2).sum().sqrt()) - l2) <= tol * max(l2, 1e-8), "param L2 " + n
        assert abs(float(t.sum()) - s) <= 10 * tol * max(l2, 1e-8), "param sum " + n


def _run(gpu, kind, net, lr, batches, gain):
    pkg = _pkg()
    fill.fill_module(net, 99, gain)
    net.to(gpu).train()
    flat, opt, dp, step = pkg.trainers.build(kind, net, lr)
    losses = []
    for b in batches:
        out = step(*[t.to(gpu) for t in b])
        outs = out if isinstance(out, tuple) else (out,)
        losses.append(sum(float(o) for o in outs))
    return losses


B = fill.rand


@pytest.mark.parametrize("lr,tag", [(1e-5, "srcnn_c1_lr1e-5"), (1e-2, "srcnn_c1_lr1e-2")])
def test_srcnn_c1(gpu, train_golden, lr, tag):
    """BASELINE config c1 exactly: SRCNN x2, B=16, 3x64x64 -> 3x48x48, MSE, SGD."""
    pkg = _pkg()
    net = pkg.SRCNNNet(3, 64)
    batches = [(B((16, 3, 64, 64), 10 + i), B((16, 3, 48, 48), 20 + i)) for i in range(3)]
    losses = _run(gpu, "srcnn", net, lr, batches, 1.0)
    assert rel_err(np.array(losses), train_golden[tag + ".losses"]) < TOL
    _check_params(net, train_golden, tag)
    if lr == 1e-2:
        for k, v in net.state_dict().items():
            assert rel_err(v, train_golden["%s.final.%s" % (tag, k)]) < TOL, k


def test_fsrcnn(gpu, train_golden):
    pkg = _pkg()
    net = pkg.FSRCNNNet(3, 4, 56, 12, 4)
    batches = [(B((4, 3, 12, 12), 30 + i), B((4, 3, 32, 32), 40 + i)) for i in range(3)]
    losses = _run(gpu, "fsrcnn", net, 1e-3, batches, 1.0)
    assert rel_err(np.array(losses), train_golden["fsrcnn.losses"]) < TOL
    _check_params(net, train_golden, "fsrcnn")


def test_vdsr_with_clip(gpu, train_golden):
    pkg = _pkg()
    net = pkg.VDSRNet(3, 64, 18)
    batches = [(B((4, 3, 17, 17), 50 + i), B((4, 3, 17, 17), 60 + i)) for i in range(3)]
    losses = _run(gpu, "vdsr", net, 1e-2, batches, 1.0)
    assert rel_err(np.array(losses), train_golden["vdsr.losses"]) < TOL
    _check_params(net, train_golden, "vdsr")


def test_edsr_l1_adam(gpu, train_golden):
    pkg = _pkg()
    net = pkg.EDSRNet(3, 64, 16)
    batches = [(B((4, 3, 8, 8), 70 + i), B((4, 3, 32, 32), 80 + i)) for i in range(3)]
    losses = _run(gpu, "edsr", net, 1e-4, batches, 0.5)
    assert rel_err(np.array(losses), train_golden["edsr.losses"]) < TOL
    _check_params(net, train_golden, "edsr")


def test_lapsrn_shared_weights(gpu, train_golden):
    pkg = _pkg()
    net = pkg.LapSRNNet(3, 64, 10)
    batches = [(B((2, 3, 8, 8), 90 + i), B((2, 3, 16, 16), 100 + i), B((2, 3, 32, 32), 110 + i)) for i in range(3)]
    losses = _run(gpu, "lapsrn", net, 1e-4, batches, 1.0)
    assert rel_err(np.array(losses), train_golden["lapsrn.losses"]) < TOL
    _check_params(net, train_golden, "lapsrn")


def test_srgan_adversarial_step(gpu, train_golden):
    pkg = _pkg()
    G, D = pkg.SRGANGenerator(3, 64, 16), pkg.SRGANDiscriminator(3, 64, 32)
    fill.fill_module(G, 5, 0.7)
    fill.fill_module(D, 6, 1.0)
    G.to(gpu).train()
    D.to(gpu).train()
    g_opt = pkg.optim.make_optimizer("srgan_g", pkg.optim.FlatParams(G), 1e-4)
    d_opt = pkg.optim.make_optimizer("srgan_d", pkg.optim.FlatParams(D), 1e-2)
    step = pkg.trainers.srgan_step(G, D, g_opt, d_opt)
    losses = []
    for i in range(2):
        d_loss, g_loss = step(B((4, 3, 8, 8), 120 + i).to(gpu), B((4, 3, 32, 32), 130 + i).to(gpu))
        losses.append((float(d_loss), float(g_loss)))
    assert rel_err(np.array(losses), train_golden["srgan.losses"]) < 5e-4
    _check_params(G, train_golden, "srgan.G", 5e-4)
    _check_params(D, train_golden, "srgan.D", 5e-4)


def test_graphed_step_equals_eager(gpu):
    """hipGraph-captured step (zero_grad+fwd+loss+bwd+clip+SGD) reproduces the eager trajectory."""
    pkg = _pkg()
    batches = [(B((4, 3, 17, 17), 50 + i).to(gpu), B((4, 3, 17, 17), 60 + i).to(gpu)) for i in range(4)]

    def make():
        net = pkg.VDSRNet(3, 64, 4)
        fill.fill_module(net, 3, 1.0)
        net.to(gpu).train()
        flat = pkg.optim.FlatParams(net)
        return net, pkg.optim.make_optimizer("vdsr", flat, 1e-2)

    net_a, opt_a = make()
    step = pkg.trainers.mse_step(net_a, opt_a, None, clip=0.4)
    ref_losses = [float(step(*b)) for b in batches]
    net_b, opt_b = make()
    snapshot = opt_b.flat.data.clone()
    g = pkg.trainers.GraphedStep(net_b, opt_b, pkg.ops.mse_loss, batches[0], clip=0.4, warmup=2)
    # warm-up steps moved the parameters: restore the initial state before the measured replay
    opt_b.flat.data.copy_(snapshot)
    opt_b.buf.zero_()
    losses = [float(g(*b)) for b in batches]
    assert rel_err(np.array(losses), np.array(ref_losses)) < 1e-5
    assert rel_err(opt_b.flat.data, opt_a.flat.data) < 1e-5


def test_flat_params_keep_state_dict_and_grads(gpu):
    pkg = _pkg()
    net = pkg.ESPCNNet(3, 64, 4)
    fill.fill_module(net, 1)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    net.to(gpu)
    flat = pkg.optim.FlatParams(net)
    for k, v in net.state_dict().items():
        assert torch.equal(v.cpu(), before[k])
    for p in net.parameters():
        assert p.grad is not None and p.grad.data_ptr() == p._srk_grad.data_ptr()
        assert p.data_ptr() % 16 == 0
    assert flat.numel >= sum(p.numel() for p in net.parameters())


def test_trainer_objects_and_cli(gpu, tmp_path):
    """The reference's trainer surface: MODEL(args).train()/test()/save_model()/load_model() driven by
    main.py's flags, on synthetic patches; checkpoints reload into the CPU oracle (same state_dict)."""
    import main as cli
    from oracle import ref_modules as R
    from pytorch_super_resolution_model_collection_amd.sr_trainers import TRAINERS
    for name, extra in (("EDSR", ["--crop_size", "32"]), ("VDSR", ["--crop_size", "17"]),
                        ("ESPCN", ["--crop_size", "48"]), ("SRGAN", ["--crop_size", "32", "--batch_size", "2"])):
        args = cli.parse_args(["--model_name", name, "--num_epochs", "2", "--save_epochs", "1", "--batch_size", "2",
                               "--steps_per_epoch", "2", "--lr", "1e-4", "--save_dir", str(tmp_path)] + extra)
        t = TRAINERS[name](args)
        hist = t.train()
        assert len(hist) == 2
        assert all(np.isfinite(np.array(hist)).ravel())
        assert t.load_model()
        psnr = t.test()
        assert isinstance(psnr, list)
    # EDSR checkpoint written with the reference's file-name pattern loads into the oracle
    import glob
    files = glob.glob(str(tmp_path / "EDSR" / "model" / "EDSR_param_ch3_batch2_epoch2_lr0.0001.pkl"))
    assert files, "EDSR checkpoint name does not follow edsr.py:329-335"
    R.EDSR(3, 64, 16).load_state_dict(torch.load(files[0]))
