#!/usr/bin/env python3
"""A few eager SRGAN adversarial steps (batch 16, 32->128) for rocprofv3 kernel traces."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
G, D = pkg.SRGANGenerator(3, 64, 16), pkg.SRGANDiscriminator(3, 64, 128)
G.weight_init(); D.weight_init(); G.to(dev).train(); D.to(dev).train()
gflat, dflat = pkg.optim.FlatParams(G), pkg.optim.FlatParams(D)
g_opt, d_opt = pkg.optim.make_optimizer("srgan_g", gflat, 1e-4), pkg.optim.make_optimizer("srgan_d", dflat, 1e-4)
step = pkg.trainers.srgan_step(G, D, g_opt, d_opt)
x = torch.rand(B, 3, 32, 32, device=dev); t = torch.rand(B, 3, 128, 128, device=dev)
for _ in range(3): step(x, t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step(x, t)
torch.cuda.synchronize(); print("srgan step B=%d: %.3f ms" % (B, (time.perf_counter() - t0) / 5 * 1e3))
