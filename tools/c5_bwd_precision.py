#!/usr/bin/env python3
"""c5 (SRGAN adversarial step, 16 x 32x32 -> 128x128): worst per-tensor L2 error of the gradients the step leaves behind,
against an fp64 run of the oracle, per precision mode -- is the 3-MFMA backward what keeps the product at ~2.5e-3?
   python tools/c5_bwd_precision.py [mode ...]        (default: mixed faithful fp32)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from oracle import fill, ref_modules as R
dev = torch.device("cuda:0")
modes = sys.argv[1:] or ["mixed", "faithful", "fp32"]
lr_img, hr_img = fill.rand((16, 3, 32, 32), 501), fill.rand((16, 3, 128, 128), 502)
torch.set_num_threads(min(64, os.cpu_count() or 8))
oG64 = fill.fill_module(R.Generator(3, 64, 16), 5, 0.7).double()
oD64 = fill.fill_module(R.Discriminator(3, 64, 128), 6, 1.0).double()
R.step_srgan(oG64, oD64, R.make_optimizer("srgan_g", oG64.parameters(), 1e-4), R.make_optimizer("srgan_d", oD64.parameters(), 1e-2),
             lr_img.double(), hr_img.double())
oG = fill.fill_module(R.Generator(3, 64, 16), 5, 0.7)
oD = fill.fill_module(R.Discriminator(3, 64, 128), 6, 1.0)
R.step_srgan(oG, oD, R.make_optimizer("srgan_g", oG.parameters(), 1e-4), R.make_optimizer("srgan_d", oD.parameters(), 1e-2), lr_img, hr_img)


def worst(named, ora64):
    g64 = dict((n, p.grad) for n, p in ora64.named_parameters())
    gmax = max(float(g.abs().max()) for g in g64.values())
    w, wn = 0.0, ""
    for n, g in named:
        den = max(float(g64[n].norm()), 1e-3 * gmax * g64[n].numel() ** 0.5)
        e = float((g.detach().cpu().double() - g64[n]).norm()) / den
        if e > w:
            w, wn = e, n
    return w, wn


print("torch fp32 (oneDNN) vs fp64: G %.3e (%s)  D %.3e (%s)" % (worst([(n, p.grad) for n, p in oG.named_parameters()], oG64)
                                                                 + worst([(n, p.grad) for n, p in oD.named_parameters()], oD64)))
for mode in modes:
    pkg.ops.set_precision(mode)
    G, D = pkg.SRGANGenerator(3, 64, 16), pkg.SRGANDiscriminator(3, 64, 128)
    fill.fill_module(G, 5, 0.7)
    fill.fill_module(D, 6, 1.0)
    G.to(dev).train(); D.to(dev).train()
    g_opt = pkg.optim.make_optimizer("srgan_g", pkg.optim.FlatParams(G), 1e-4)
    d_opt = pkg.optim.make_optimizer("srgan_d", pkg.optim.FlatParams(D), 1e-2)
    step = pkg.trainers.srgan_step(G, D, g_opt, d_opt)
    step(lr_img.to(dev), hr_img.to(dev))
    torch.cuda.synchronize()
    print("%-9s vs fp64: G %.3e (%s)  D %.3e (%s)" % ((mode,) + worst([(n, p.grad) for n, p in G.named_parameters()], oG64)
                                                      + worst([(n, p.grad) for n, p in D.named_parameters()], oD64)))
