#!/usr/bin/env python3
"""Time the VDSR x4 train step (batch 256, 41x41; SGD + clip) as one hipGraph, the way bench.py's c3 runs it."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = pkg.VDSRNet(3, 64, 18); net.weight_init(); net.to(dev).train()
flat = pkg.optim.FlatParams(net); opt = pkg.optim.make_optimizer("vdsr", flat, 1e-5)
x = torch.rand(B, 3, 41, 41, device=dev); t = torch.rand(B, 3, 41, 41, device=dev)
g = pkg.trainers.GraphedStep(net, opt, pkg.ops.mse_loss, (x, t), clip=0.4)
for _ in range(5): g(x, t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    e0.record()
    for _ in range(20): g(x, t)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print("vdsr x4 train step B=%d: %.3f ms/step (graph), %.1f patches/s" % (B, best, B / best * 1e3))
