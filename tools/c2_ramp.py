#!/usr/bin/env python3
"""Per-step time of the c2 forward behind an idle gap: how long does the clock governor take to settle?
   python tools/c2_ramp.py [idle seconds]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
net = pkg.ESPCNNet(3, 64, 4); net.weight_init(); net.to(dev).eval()
x = torch.rand(64, 3, 256, 256, device=dev)
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
with torch.no_grad():
    for _ in range(3): net(x)
    torch.cuda.synchronize()
    for trial in range(2):
        time.sleep(idle)
        n = 300
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            net(x); ev[i + 1].record()
        torch.cuda.synchronize()
        t = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
        def avg(a, b): return sum(t[a:b]) / (b - a)
        print("after %.1f s idle: steps 0-4 %.3f  5-24 %.3f  25-49 %.3f  50-99 %.3f  100-199 %.3f  200-299 %.3f ms" % (
            idle, avg(0, 5), avg(5, 25), avg(25, 50), avg(50, 100), avg(100, 200), avg(200, 300)))
