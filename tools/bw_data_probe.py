#!/usr/bin/env python3
"""Does the achievable HBM rate depend on the DATA?  Fill / copy / add on 1 GiB buffers holding zeros, a constant, uniform
random floats, and on the c2 first layer itself with a zero and a random input.   python tools/bw_data_probe.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = torch.device("cuda:0")
n = 1024 * 1024 * 1024 // 4
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
y = torch.empty(n, device=dev)
GB = n * 4 / 1e9
for name, x in (("zeros", torch.zeros(n, device=dev)), ("ones", torch.ones(n, device=dev)), ("rand", torch.rand(n, device=dev)),
                ("randn*1e3", torch.randn(n, device=dev) * 1e3)):
    tc = t(lambda: y.copy_(x)); ts = t(lambda: x.sum()); ta = t(lambda: torch.add(x, 1.0, out=y))
    print("%-10s copy %.1f us (%.2f TB/s r+w)   sum %.1f us (%.2f TB/s read)   add %.1f us (%.2f TB/s r+w)" % (name, tc * 1e3, 2 * GB / tc, ts * 1e3, GB / ts, ta * 1e3, 2 * GB / ta))
tf = t(lambda: y.fill_(1.0)); print("fill(1.0) %.1f us (%.2f TB/s write)" % (tf * 1e3, GB / tf))
import pytorch_super_resolution_model_collection_amd as pkg
lib = pkg._lib.load()
net = pkg.ESPCNNet(3, 64, 4); net.weight_init(); net.to(dev).eval()
with torch.no_grad():
    for name, x in (("zeros", torch.zeros(64, 3, 256, 256, device=dev)), ("rand", torch.rand(64, 3, 256, 256, device=dev)),
                    ("const", torch.full((64, 3, 256, 256), 0.5, device=dev))):
        hs, h = [], x
        for l in net.layers:
            hs.append(h); h = l(h)
        for i in range(3):
            tl = t(lambda: net.layers[i](hs[i]))
            print("input %-6s layer %d  %.1f us  %s" % (name, i, tl * 1e3, lib.srk_last_kernel_name().decode()))
