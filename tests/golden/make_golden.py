#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ by running the REFERENCE's own
classes (imported from /root/reference, which exists only in the build container).

It (1) imports the reference modules behind stubs for their non-arithmetic imports (SURVEY.md
App. C), (2) asserts that oracle/ref_modules.py is BIT-EQUAL to the reference for every net and
train step below (same torch build => same ATen/oneDNN kernels) — this is what pins the oracle —
and (3) freezes inputs + expected outputs as .npz fixtures.  Nothing from the reference's
source is stored: fixtures hold arrays only.

Run:  python tests/golden/make_golden.py       (needs /root/reference; CPU only)
"""
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from oracle import fill, ref_modules as R  # noqa: E402
from oracle.kat_table import CONV_KATS, conv_case_inputs  # noqa: E402

REF = "/root/reference"


def import_reference():
    import math

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms", math=math, np=np)
    tv.models = stub("torchvision.models")
    stub("imageio")
    stub("logger", Logger=object)
    import scipy.misc
    scipy.misc.imsave = lambda *a, **k: None
    sys.path.insert(0, REF)
    mods = {}
    for n in ("srcnn", "espcn", "fsrcnn", "vdsr", "edsr", "lapsrn", "srgan"):
        mods[n] = __import__(n)
    return mods


def same(a, b, what):
    if isinstance(a, (tuple, list)):
        for i, (x, y) in enumerate(zip(a, b)):
            same(x, y, "%s[%d]" % (what, i))
        return
    if not torch.equal(a, b):
        raise SystemExit("oracle != reference for %s (max diff %g)" % (what, (a - b).abs().max().item()))


def grads_of(module):
    out = {}
    seen = set()
    for n, p in module.named_parameters():
        if p.grad is not None and p.data_ptr() not in seen:
            seen.add(p.data_ptr())
            out[n] = p.grad.detach().clone()
    return out


NETS = {
    # name: (reference ctor getter, oracle ctor, ctor args, input shape, fill gain)
    "srcnn": (lambda m: m["srcnn"].Net, R.SRCNN, (3, 64), (2, 3, 20, 20), 1.0),
    "espcn": (lambda m: m["espcn"].Net, R.ESPCN, (3, 64, 4), (2, 3, 16, 16), 1.0),
    "fsrcnn": (lambda m: m["fsrcnn"].Net, R.FSRCNN, (3, 4, 56, 12, 4), (2, 3, 12, 12), 1.0),
    "vdsr": (lambda m: m["vdsr"].Net, R.VDSR, (3, 64, 18), (2, 3, 13, 13), 1.0),
    "edsr": (lambda m: m["edsr"].Net, R.EDSR, (3, 64, 16), (2, 3, 8, 8), 0.5),
    "lapsrn": (lambda m: m["lapsrn"].Net, R.LapSRN, (3, 64, 10), (1, 3, 8, 8), 1.0),
    "srgan_g": (lambda m: m["srgan"].Generator, R.Generator, (3, 64, 16), (2, 3, 8, 8), 0.7),
    "srgan_d": (lambda m: m["srgan"].Discriminator, R.Discriminator, (3, 64, 32), (2, 3, 32, 32), 1.0),
}


def net_forward_backward(net, x):
    """Forward + a deterministic scalar objective (weighted sum of outputs) backward."""
    net.zero_grad()
    x = x.clone().requires_grad_(True)
    out = net(x)
    outs = out if isinstance(out, (tuple, list)) else (out,)
    loss = 0
    for i, o in enumerate(outs):
        g = fill.randn(tuple(o.shape), 77 + i)
        loss = loss + (o * g).sum() / o.numel()
    loss.backward()
    return outs, x.grad.detach().clone(), grads_of(net)


def gen_nets(mods, out):
    for name, (ref_get, ora_cls, args, ishape, gain) in NETS.items():
        ref, ora = ref_get(mods)(*args), ora_cls(*args)
        # exercise the reference's own initialiser once (distribution check lives in the tests)
        ref.weight_init()
        assert list(ref.state_dict().keys()) == list(ora.state_dict().keys()), name
        fill.fill_module(ref, 1234, gain)
        fill.fill_module(ora, 1234, gain)
        for (k1, v1), (k2, v2) in zip(ref.state_dict().items(), ora.state_dict().items()):
            same(v1, v2, "%s weights %s" % (name, k1))
        x = fill.rand(ishape, 4321)
        if name.startswith("srgan"):
            ref.train(); ora.train()
        o1, dx1, g1 = net_forward_backward(ref, x)
        o2, dx2, g2 = net_forward_backward(ora, x)
        same(list(o1), list(o2), name + " forward")
        same(dx1, dx2, name + " dx")
        for k in g1:
            same(g1[k], g2[k], "%s grad %s" % (name, k))
        for i, o in enumerate(o2):
            out["%s.out%d" % (name, i)] = o.detach().numpy()
        out[name + ".dx"] = dx2.numpy()
        names, arr = fill.tensor_checksums(g2)
        out[name + ".grad_names"] = np.array(names)
        out[name + ".grad_sums"] = arr
        # a couple of full gradients (first and last parameter) for element-wise checks
        out[name + ".grad_first"] = g2[names[0]].numpy()
        out[name + ".grad_last"] = g2[names[-1]].numpy()
        if name.startswith("srgan"):  # eval-mode forward too (running statistics path)
            ref.eval(); ora.eval()
            with torch.no_grad():
                e1, e2 = ref(x), ora(x)
            same(e1, e2, name + " eval forward")
            out[name + ".eval_out"] = e2.numpy()
            bnk = [k for k in ora.state_dict() if k.endswith("running_mean") or k.endswith("running_var")]
            rn, ra = fill.tensor_checksums({k: ora.state_dict()[k] for k in bnk})
            out[name + ".bn_names"] = np.array(rn)
            out[name + ".bn_sums"] = ra
        print("net", name, "ok:", [tuple(o.shape) for o in o2])


def gen_train(mods, out):
    """Train trajectories: 3 steps each, losses + final parameter fingerprints."""
    def run(tag, ref, ora, kind, lr, stepper, batches, full_params=False):
        fill.fill_module(ref, 99, 0.5 if kind in ("edsr", "srgan_g") else 1.0)
        fill.fill_module(ora, 99, 0.5 if kind in ("edsr", "srgan_g") else 1.0)
        o1, o2 = R.make_optimizer(kind, ref.parameters(), lr), R.make_optimizer(kind, ora.parameters(), lr)
        l1 = [stepper(ref, o1, *b) for b in batches]
        l2 = [stepper(ora, o2, *b) for b in batches]
        assert np.array_equal(np.array(l1), np.array(l2)), (tag, l1, l2)
        for (k, a), (_, b) in zip(ref.state_dict().items(), ora.state_dict().items()):
            same(a, b, "%s final %s" % (tag, k))
        out[tag + ".losses"] = np.array(l2, dtype=np.float64)
        names, arr = fill.tensor_checksums({k: v for k, v in ora.state_dict().items() if torch.is_floating_point(v)})
        out[tag + ".param_names"] = np.array(names)
        out[tag + ".param_sums"] = arr
        if full_params:
            for k, v in ora.state_dict().items():
                out["%s.final.%s" % (tag, k)] = v.numpy()
        print("train", tag, "ok: losses", l2)

    B = lambda shape, seed: fill.rand(shape, seed)
    # c1 exactly: SRCNN x2, B=16, 3x64x64 -> 3x48x48, SGD (srcnn.py:79,127-131); reference lr and a visible lr
    for lr, tag in ((1e-5, "srcnn_c1_lr1e-5"), (1e-2, "srcnn_c1_lr1e-2")):
        batches = [(B((16, 3, 64, 64), 10 + i), B((16, 3, 48, 48), 20 + i)) for i in range(3)]
        run(tag, mods["srcnn"].Net(3, 64), R.SRCNN(3, 64), "srcnn", lr, R.step_mse, batches, full_params=(lr == 1e-2))
    batches = [(B((4, 3, 12, 12), 30 + i), B((4, 3, 32, 32), 40 + i)) for i in range(3)]
    run("fsrcnn", mods["fsrcnn"].Net(3, 4, 56, 12, 4), R.FSRCNN(3, 4, 56, 12, 4), "fsrcnn", 1e-3, R.step_mse, batches)
    batches = [(B((4, 3, 17, 17), 50 + i), B((4, 3, 17, 17), 60 + i)) for i in range(3)]
    run("vdsr", mods["vdsr"].Net(3, 64, 18), R.VDSR(3, 64, 18), "vdsr", 1e-2,
        lambda m, o, a, b: R.step_mse(m, o, a, b, clip=0.4), batches)
    batches = [(B((4, 3, 8, 8), 70 + i), B((4, 3, 32, 32), 80 + i)) for i in range(3)]
    run("edsr", mods["edsr"].Net(3, 64, 16), R.EDSR(3, 64, 16), "edsr", 1e-4, R.step_l1, batches)
    batches = [(B((2, 3, 8, 8), 90 + i), B((2, 3, 16, 16), 100 + i), B((2, 3, 32, 32), 110 + i)) for i in range(3)]
    run("lapsrn", mods["lapsrn"].Net(3, 64, 10), R.LapSRN(3, 64, 10), "lapsrn", 1e-4,
        lambda m, o, a, b, c: sum(R.step_lapsrn(m, o, a, b, c)), batches)

    # SRGAN adversarial step x2 (D sized for 32x32 crops to keep the fixture generator light)
    def gan(G, D):
        fill.fill_module(G, 5, 0.7); fill.fill_module(D, 6, 1.0)
        G.train(); D.train()
        go, do = R.make_optimizer("srgan_g", G.parameters(), 1e-4), R.make_optimizer("srgan_d", D.parameters(), 1e-2)
        ls = []
        for i in range(2):
            ls.append(R.step_srgan(G, D, go, do, B((4, 3, 8, 8), 120 + i), B((4, 3, 32, 32), 130 + i)))
        return ls
    G1, D1 = mods["srgan"].Generator(3, 64, 16), mods["srgan"].Discriminator(3, 64, 32)
    G2, D2 = R.Generator(3, 64, 16), R.Discriminator(3, 64, 32)
    l1, l2 = gan(G1, D1), gan(G2, D2)
    assert np.array_equal(np.array(l1), np.array(l2)), (l1, l2)
    for tag, a, b in (("G", G1, G2), ("D", D1, D2)):
        for (k, u), (_, v) in zip(a.state_dict().items(), b.state_dict().items()):
            same(u, v, "srgan %s final %s" % (tag, k))
        names, arr = fill.tensor_checksums({k: v for k, v in b.state_dict().items() if torch.is_floating_point(v)})
        out["srgan.%s.param_names" % tag] = np.array(names)
        out["srgan.%s.param_sums" % tag] = arr
    out["srgan.losses"] = np.array(l2, dtype=np.float64)
    print("train srgan ok: losses", l2)


def gen_ops(out):
    """Op-level known answers computed with the torch.nn classes the reference instantiates."""
    for i, (tag, cin, cout, k, s, p, tr, op, H, W, N, act) in enumerate(CONV_KATS):
        if tr:
            m = nn.ConvTranspose2d(cin, cout, k, s, p, output_padding=op)
        else:
            m = nn.Conv2d(cin, cout, k, s, p)
        x, w, b, g = conv_case_inputs(i)
        m.weight.data.copy_(w)
        m.bias.data.copy_(b)
        a = {None: None, "relu": nn.ReLU(), "lrelu": nn.LeakyReLU(0.2)}[act]
        x = x.requires_grad_(True)
        y = m(x)
        if a is not None:
            y = a(y)
        assert tuple(y.shape) == tuple(g.shape), (tag, y.shape, g.shape)
        (y * g).sum().backward()
        for key, val in (("y", y), ("dx", x.grad), ("dw", m.weight.grad), ("db", m.bias.grad)):
            out["conv.%s.%s" % (tag, key)] = val.detach().numpy()
    out["conv.tags"] = np.array([c[0] for c in CONV_KATS])
    # pixel shuffle (base_networks.py:157)
    for r, C in ((2, 64), (4, 3), (3, 2)):
        x = fill.randn((2, C * r * r, 5, 6), 5000 + r).requires_grad_(True)
        y = nn.PixelShuffle(r)(x)
        g = fill.randn(tuple(y.shape), 5100 + r)
        (y * g).sum().backward()
        out["ps.r%d.x" % r], out["ps.r%d.y" % r] = x.detach().numpy(), y.detach().numpy()
        out["ps.r%d.g" % r], out["ps.r%d.dx" % r] = g.numpy(), x.grad.numpy()
    # activations (base_networks.py:50-60)
    for name, mod in (("relu", nn.ReLU()), ("prelu", nn.PReLU()), ("prelu_c", nn.PReLU(8)),
                      ("lrelu", nn.LeakyReLU(0.2)), ("tanh", nn.Tanh()), ("sigmoid", nn.Sigmoid())):
        if name == "prelu_c":
            mod.weight.data.copy_(fill.rand((8,), 61, -0.3, 0.5))
        x = fill.randn((2, 8, 5, 7), 6000).requires_grad_(True)
        y = mod(x)
        g = fill.randn(tuple(y.shape), 6100)
        (y * g).sum().backward()
        out["act.%s.x" % name], out["act.%s.y" % name] = x.detach().numpy(), y.detach().numpy()
        out["act.%s.g" % name], out["act.%s.dx" % name] = g.numpy(), x.grad.numpy()
        if name.startswith("prelu"):
            out["act.%s.w" % name] = mod.weight.detach().numpy()
            out["act.%s.dw" % name] = mod.weight.grad.numpy()
    # losses (srcnn.py:84-86; edsr.py:98-100; lapsrn.py:75-85; srgan.py:157)
    p = fill.rand((2, 3, 9, 11), 7000, 0.02, 0.98).requires_grad_(True)
    t = fill.rand((2, 3, 9, 11), 7001)
    for name, fn in (("mse", nn.MSELoss()), ("l1", nn.L1Loss()), ("charbonnier", R.L1_Charbonnier_loss()),
                     ("bce", nn.BCELoss())):
        p.grad = None
        l = fn(p, t)
        l.backward()
        out["loss.%s.value" % name], out["loss.%s.dpred" % name] = l.detach().numpy(), p.grad.numpy().copy()
    out["loss.pred"], out["loss.target"] = p.detach().numpy(), t.numpy()
    # BatchNorm2d train x2 calls + eval (base_networks.py:46; shared-BN double update, App. B-5)
    bn = nn.BatchNorm2d(16)
    bn.weight.data.copy_(fill.rand((16,), 8000, 0.5, 1.5)); bn.bias.data.copy_(fill.randn((16,), 8001, 0.1))
    x = fill.randn((4, 16, 5, 6), 8002).requires_grad_(True)
    y1 = bn(x); y2 = bn(y1 * 0.5 + 0.1)
    g = fill.randn(tuple(y2.shape), 8003)
    (y2 * g).sum().backward()
    for key, val in (("x", x), ("gamma", bn.weight), ("beta", bn.bias), ("y1", y1), ("y2", y2), ("g", g),
                     ("dx", x.grad), ("dgamma", bn.weight.grad), ("dbeta", bn.bias.grad),
                     ("running_mean", bn.running_mean), ("running_var", bn.running_var)):
        out["bn." + key] = val.detach().numpy().copy()
    bn.eval()
    out["bn.eval_y"] = bn(x).detach().numpy()
    # Linear (base_networks.py:7)
    fc = nn.Linear(96, 20)
    fc.weight.data.copy_(fill.randn((20, 96), 9000, 0.1)); fc.bias.data.copy_(fill.randn((20,), 9001, 0.1))
    x = fill.randn((5, 96), 9002).requires_grad_(True)
    y = nn.LeakyReLU(0.2)(fc(x))
    g = fill.randn((5, 20), 9003)
    (y * g).sum().backward()
    for key, val in (("x", x), ("w", fc.weight), ("b", fc.bias), ("y", y), ("g", g), ("dx", x.grad),
                     ("dw", fc.weight.grad), ("db", fc.bias.grad)):
        out["fc." + key] = val.detach().numpy()
    # optimizers + clip (a15): 3 steps on a flat vector with given gradients
    p0 = fill.randn((1000,), 9100)
    grads = [fill.randn((1000,), 9101 + i, 0.5) for i in range(3)]
    out["opt.p0"] = p0.numpy()
    out["opt.grads"] = np.stack([g.numpy() for g in grads])
    for kind in ("srcnn", "fsrcnn", "vdsr", "edsr", "srgan_d"):
        prm = nn.Parameter(p0.clone())
        o = R.make_optimizer(kind, [prm], 1e-2)
        for g in grads:
            prm.grad = g.clone()
            o.step()
        out["opt.%s.final" % kind] = prm.detach().numpy()
    prm = nn.Parameter(p0.clone()); prm.grad = grads[0].clone() * 3
    tot = nn.utils.clip_grad_norm_([prm], 0.4)
    out["opt.clip.norm"] = np.array(float(tot)); out["opt.clip.grad"] = prm.grad.numpy()


def main():
    if not os.path.isdir(REF):
        raise SystemExit("this generator needs the reference checkout at %s" % REF)
    torch.manual_seed(1234)
    torch.set_num_threads(8)
    mods = import_reference()
    ops, nets, train = {}, {}, {}
    gen_ops(ops)
    gen_nets(mods, nets)
    gen_train(mods, train)
    for name, d in (("ops_kat", ops), ("nets", nets), ("train_traj", train)):
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **d)
        print("wrote %s (%.2f MB, %d arrays)" % (path, os.path.getsize(path) / 1e6, len(d)))


if __name__ == "__main__":
    main()
