#!/usr/bin/env python3
"""Per-layer cost of the running-maximum bookkeeping (ep.y_amax) in the c2 (ESPCN x4, 64 x 256x256) f16x3 kernels:
each layer timed alone on a fixed, tagged input with and without SRK_NO_YAMAX=1; then the first layer on the per-tile
(SRK_ROWSW=0) and the persistent (SRK_ROWSW=1) row-packed kernel.   python tools/time_yamax.py"""
import os, sys, torch
os.environ["SRK_ENV_LIVE"] = "1"   # this tool flips SRK_* switches in-process: the library must re-read them (csrc/api.hip env_str)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
lib = pkg._lib.load()
torch.manual_seed(1234)
net = pkg.ESPCNNet(3, 64, 4); net.weight_init(); net.to(dev).eval()
x = torch.rand(64, 3, 256, 256, device=dev)
pkg.ops.set_precision("mixed")


def t(fn, n=60):
    """us per call, calls queued back to back (one call bracketed by events would include the host's launch path)"""
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n // 3): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (n // 3))
    return best * 1e3


with torch.no_grad():
    os.environ.pop("SRK_NO_YAMAX", None)
    hs, h = [], x
    for l in net.layers:
        hs.append(h); h = l(h)
    for i, l in enumerate(net.layers):
        t(lambda: l(hs[i]), 40)   # (the first measurement of a process reads ~15 % high: clocks still ramping)
        for v in ("", "1", "", "1"):
            if v: os.environ["SRK_NO_YAMAX"] = v
            else: os.environ.pop("SRK_NO_YAMAX", None)
            us = t(lambda: l(hs[i]))
            print("layer %d  NO_YAMAX=%-1s  %8.1f us  %s" % (i, v, us, lib.srk_last_kernel_name().decode()))
    for sw in ("0", "1", "0", "1"):
        os.environ["SRK_ROWSW"] = sw
        us = t(lambda: net.layers[0](hs[0]), 60)
        print("layer 0 SRK_ROWSW=%s  %8.1f us  %s" % (sw, us, lib.srk_last_kernel_name().decode()))
