#!/usr/bin/env python3
"""c2: eager 3-launch forward vs the same forward replayed from a hipGraph."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
net = pkg.ESPCNNet(3, 64, 4); net.weight_init(); net.to(dev).eval()
x = torch.rand(64, 3, 256, 256, device=dev)


def timeit(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    print("eager : %.4f ms" % timeit(lambda: net(x)))
    side = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        net(x)
        with torch.cuda.graph(g, stream=side):
            y = net(x)
    print("graph : %.4f ms" % timeit(g.replay))
