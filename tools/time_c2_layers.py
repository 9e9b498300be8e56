#!/usr/bin/env python3
"""us per layer of c2 (ESPCN x4, 64 x 256x256, default precision), calls queued back to back.
   python tools/time_c2_layers.py [layer ...]      (SRK_DBG / SRK_ROWSW_DBG / SRK_ROWS_DBG ablations apply)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
lib = pkg._lib.load()
torch.manual_seed(1234)
net = pkg.ESPCNNet(3, 64, 4); net.weight_init(); net.to(dev).eval()
x = torch.rand(64, 3, 256, 256, device=dev)
if os.environ.get("NHWC"): x = x.contiguous(memory_format=torch.channels_last)   # first layer reads NHWC in place
pkg.ops.set_precision(os.environ.get("PRECISION", "mixed"))
which = [int(a) for a in sys.argv[1:]] or [0, 1, 2]
with torch.no_grad():
    hs, h = [], x
    for l in net.layers:
        hs.append(h); h = l(h)
    for i in which:
        l = net.layers[i]
        for _ in range(5): l(hs[i])
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): l(hs[i])
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        print("layer %d  %8.1f us  %s  SRK_DBG=%s" % (i, best * 1e3, lib.srk_last_kernel_name().decode(), os.environ.get("SRK_DBG", "")))
