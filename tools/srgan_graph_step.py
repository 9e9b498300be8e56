#!/usr/bin/env python3
"""Time the SRGAN x4 adversarial step (batch 16, 32 -> 128) as ONE hipGraph, the way bench.py's c5 runs it.
   python tools/srgan_graph_step.py [iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G, D = pkg.SRGANGenerator(3, 64, 16), pkg.SRGANDiscriminator(3, 64, 128)
torch.manual_seed(1234)
G.weight_init(); D.weight_init(); G.to(dev).train(); D.to(dev).train()
gflat, dflat = pkg.optim.FlatParams(G), pkg.optim.FlatParams(D)
g_opt, d_opt = pkg.optim.make_optimizer("srgan_g", gflat, 1e-4), pkg.optim.make_optimizer("srgan_d", dflat, 1e-4)
x = torch.rand(16, 3, 32, 32, device=dev); t = torch.rand(16, 3, 128, 128, device=dev)
step = pkg.trainers.GraphedFn(pkg.trainers.srgan_step(G, D, g_opt, d_opt, lazy_pack=True), (x, t), flats=[gflat, dflat])
for _ in range(3): step(x, t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    e0.record()
    for _ in range(iters): step(x, t)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / iters)
print("srgan x4 adversarial step B=16 (one hipGraph): %.3f ms" % best)
