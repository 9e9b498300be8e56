import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pytorch_super_resolution_model_collection_amd as pkg
from oracle import fill, ref_modules as R
gpu = torch.device("cuda:0")
lr_img, hr_img = fill.rand((16, 3, 32, 32), 501), fill.rand((16, 3, 128, 128), 502)
oG = fill.fill_module(R.Generator(3, 64, 16), 5, 0.7); oD = fill.fill_module(R.Discriminator(3, 64, 128), 6, 1.0)
og_opt = R.make_optimizer("srgan_g", oG.parameters(), 1e-4); od_opt = R.make_optimizer("srgan_d", oD.parameters(), 1e-2)
R.step_srgan(oG, oD, og_opt, od_opt, lr_img, hr_img)
for mode in ("mixed", "fp32", "bf16x6bwd"):
    if mode == "bf16x6bwd":
        pkg.ops._MODES["x"] = {"infer": 0, "train_fwd": 5, "bwd": 5}; pkg.ops._PRECISION["mode"] = "x"
    else:
        pkg.ops.set_precision(mode)
    G, D = pkg.SRGANGenerator(3, 64, 16), pkg.SRGANDiscriminator(3, 64, 128)
    fill.fill_module(G, 5, 0.7); fill.fill_module(D, 6, 1.0); G.to(gpu).train(); D.to(gpu).train()
    g_opt = pkg.optim.make_optimizer("srgan_g", pkg.optim.FlatParams(G), 1e-4)
    d_opt = pkg.optim.make_optimizer("srgan_d", pkg.optim.FlatParams(D), 1e-2)
    step = pkg.trainers.srgan_step(G, D, g_opt, d_opt)
    step(lr_img.to(gpu), hr_img.to(gpu))
    for name, net, ora in (("G", G, oG), ("D", D, oD)):
        ogr = dict((n, p.grad) for n, p in ora.named_parameters())
        gmax = max(float(g.abs().max()) for g in ogr.values())
        errs = []
        for n, p in net.named_parameters():
            g, og = p.grad.detach().cpu().double(), ogr[n].double()
            errs.append((float((g - og).norm()) / max(float(og.norm()), 1e-3 * gmax * og.numel() ** 0.5), n))
        errs.sort(reverse=True)
        print(mode, name, " ".join("%s=%.2e" % (n.replace("residual_layers", "rl"), e) for e, n in errs[:4]), flush=True)
