#!/usr/bin/env python3
"""c2 experiment: ESPCN x4 batch 64 as one stream vs split over S streams (kernels of different layers co-resident)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
net = pkg.ESPCNNet(3, 64, 4); net.weight_init(); net.to(dev).eval()
x = torch.rand(64, 3, 256, 256, device=dev)


def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    print("1 stream: %.3f ms" % timeit(lambda: net(x)))
    for S in (2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(S)]
        parts = list(torch.chunk(x, S, 0))

        def run():
            cur = torch.cuda.current_stream()
            for s in streams: s.wait_stream(cur)
            for s, p in zip(streams, parts):
                with torch.cuda.stream(s):
                    net(p)
            for s in streams: cur.wait_stream(s)
        print("%d streams: %.3f ms" % (S, timeit(run)))
