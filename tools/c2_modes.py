#!/usr/bin/env python3
"""ESPCN x4 inference (c2: 256x256 LR, batch 64) in the bf16x3 and the fp32-faithful (f16x3) arithmetic: ms per batch,
kernels, error against the exact-fp32 kernels.   python tools/c2_modes.py [mode ...]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
lib = pkg._lib.load()
torch.manual_seed(1234)
net = pkg.ESPCNNet(3, 64, 4); net.weight_init(); net.to(dev).eval()
x = torch.rand(64, 3, 256, 256, device=dev)


def run(mode, n=20):
    pkg.ops.set_precision(mode)
    with torch.no_grad():
        for _ in range(3):
            y = net(x)
        names, h = [], x
        for l in net.layers:
            h = l(h)
            names.append(lib.srk_last_kernel_name().decode())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            y = net(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, y, names


modes = sys.argv[1:] or ["mixed", "bf16x6"]
pkg.ops.set_precision("fp32")
with torch.no_grad():
    yf = net(x)
for m in modes:
    ms, y, names = run(m)
    print("%-8s %.3f ms/batch = %.1f k img/s  %s  max err vs exact fp32 kernels %.2e" % (m, ms, 64 / ms, names, float((y - yf).abs().max() / yf.abs().max())))
