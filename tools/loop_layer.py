#!/usr/bin/env python3
"""Loop one c2 layer for N seconds (power / clock probes):  python tools/loop_layer.py <layer> <seconds>"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
net = pkg.ESPCNNet(3, 64, 4); net.weight_init(); net.to(dev).eval()
x = torch.rand(64, 3, 256, 256, device=dev)
i = int(sys.argv[1]); secs = float(sys.argv[2])
with torch.no_grad():
    hs, h = [], x
    for l in net.layers:
        hs.append(h); h = l(h)
    l = net.layers[i]
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(50): l(hs[i])
        torch.cuda.synchronize(); n += 50
    print("layer %d: %.1f us per launch over %.1f s" % (i, (time.time() - t0) / n * 1e6, time.time() - t0))
