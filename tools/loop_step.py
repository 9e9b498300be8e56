#!/usr/bin/env python3
"""Loop one training step (hipGraph replay where the trainer captures one) for N seconds -- power / clock probes.
   python tools/loop_step.py vdsr|edsr128|edsr16|srgan <seconds>"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0"); what = sys.argv[1]; secs = float(sys.argv[2])
if what == "vdsr":
    net = pkg.VDSRNet(3, 64, 18); net.weight_init(); net.to(dev).train()
    flat = pkg.optim.FlatParams(net); opt = pkg.optim.make_optimizer("vdsr", flat, 1e-5)
    x = torch.rand(256, 3, 41, 41, device=dev); t = torch.rand(256, 3, 41, 41, device=dev)
    step = pkg.trainers.mse_step(net, opt, None, clip=0.4)
elif what.startswith("edsr"):
    B = int(what[4:])
    net = pkg.EDSRNet(3, 64, 16); net.weight_init(); net.to(dev).train()
    flat = pkg.optim.FlatParams(net); opt = pkg.optim.make_optimizer("edsr", flat, 1e-5)
    x = torch.rand(B, 3, 32, 32, device=dev); t = torch.rand(B, 3, 128, 128, device=dev)
    step = pkg.trainers.l1_step(net, opt, None)
else:
    G, D = pkg.SRGANGenerator(3, 64, 16), pkg.SRGANDiscriminator(3, 64, 128)
    G.weight_init(); D.weight_init(); G.to(dev).train(); D.to(dev).train()
    gflat, dflat = pkg.optim.FlatParams(G), pkg.optim.FlatParams(D)
    g_opt, d_opt = pkg.optim.make_optimizer("srgan_g", gflat, 1e-4), pkg.optim.make_optimizer("srgan_d", dflat, 1e-4)
    step = pkg.trainers.srgan_step(G, D, g_opt, d_opt)
    x = torch.rand(16, 3, 32, 32, device=dev); t = torch.rand(16, 3, 128, 128, device=dev)
for _ in range(5): step(x, t)
torch.cuda.synchronize(); t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(20): step(x, t)
    torch.cuda.synchronize(); n += 20
print("%s: %.3f ms per step over %.1f s" % (what, (time.time() - t0) / n * 1e3, time.time() - t0))
