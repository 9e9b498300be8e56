#!/usr/bin/env python3
"""EDSR x4 train-step time vs per-GPU batch (what each rank does under strong scaling of the global batch 128)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
for B in (128, 64, 32, 16):
    net = pkg.EDSRNet(3, 64, 16); net.weight_init(); net.to(dev).train()
    flat = pkg.optim.FlatParams(net); opt = pkg.optim.make_optimizer("edsr", flat, 1e-5)
    x = torch.rand(B, 3, 32, 32, device=dev); t = torch.rand(B, 3, 128, 128, device=dev)
    step = pkg.trainers.GraphedStep(net, opt, pkg.ops.l1_loss, (x, t), warmup=2)
    for _ in range(3): step(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step(x, t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("B=%3d  %.3f ms/step  %.0f patches/s  (ideal from B=128: x%.1f)" % (B, dt * 1e3, B / dt, 128 / B))
