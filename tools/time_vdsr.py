#!/usr/bin/env python3
"""VDSR x4 training step (c3: batch 256, 41x41), eager, ms per step and the kernels of one body layer.
   python tools/time_vdsr.py [batch]      (SRK_BFW=0: per-tile kernels instead of the wave-specialised slices)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = pkg._lib.load()
torch.manual_seed(1)
net = pkg.VDSRNet(3, 64, 18); net.weight_init(); net.to(dev).train()
flat = pkg.optim.FlatParams(net); opt = pkg.optim.make_optimizer("vdsr", flat, 1e-5)
x = torch.rand(B, 3, 41, 41, device=dev); t = torch.rand(B, 3, 41, 41, device=dev)
step = pkg.trainers.mse_step(net, opt, None, clip=0.4)
for _ in range(5): loss = step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): loss = step(x, t)
torch.cuda.synchronize()
print("SRK_BFW=%s  %.3f ms/step  loss %.9g  last kernel %s" % (os.environ.get("SRK_BFW", ""), (time.perf_counter() - t0) / 20 * 1e3, float(loss), lib.srk_last_kernel_name().decode()))
