#!/usr/bin/env python3
"""Inference forward time of every model family at a few sizes (hipGraph-free, CUDA events, 10 iterations)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
CASES = [("espcn", lambda: pkg.ESPCNNet(3, 64, 4), (64, 3, 256, 256)),
         ("srcnn", lambda: pkg.SRCNNNet(3, 64), (64, 3, 256, 256)),
         ("fsrcnn", lambda: pkg.FSRCNNNet(3, 4, 56, 12, 4), (64, 3, 128, 128)),
         ("vdsr", lambda: pkg.VDSRNet(3, 64, 18), (16, 3, 256, 256)),
         ("edsr", lambda: pkg.EDSRNet(3, 64, 16), (16, 3, 128, 128)),
         ("lapsrn", lambda: pkg.LapSRNNet(3, 64, 10), (16, 3, 128, 128)),
         ("srgan_g", lambda: pkg.SRGANGenerator(3, 64, 16), (16, 3, 128, 128))]
for name, mk, shape in CASES:
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    net = mk(); net.weight_init() if hasattr(net, "weight_init") else None
    net.to(dev).eval()
    x = torch.rand(*shape, device=dev)
    with torch.no_grad():
        for _ in range(3): y = net(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): y = net(x)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-8s in %-18s %8.3f ms  %9.1f images/s" % (name, "x".join(map(str, shape)), ms, shape[0] / ms * 1e3))
