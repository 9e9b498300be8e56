#!/usr/bin/env python3
"""Is a small conv in a chain slower because its filter is cold in the L2s?  32 conv3x3 64->64 launches (16 x 32 x 32, the
SRGAN / EDSR-shard body shape) in one hipGraph, chained (each reads the previous output), with ONE filter for all of
them vs 32 different filters (442 KB of prepared planes each).   python tools/cold_filters.py"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd import _lib, ops
from pytorch_super_resolution_model_collection_amd._lib import ptr, stream_ptr, check
lib = _lib.load()
dev = torch.device("cuda:0")
B, H, NL = 16, 32, 32
x0 = torch.randn(B, 64, H, H, device=dev).contiguous(memory_format=torch.channels_last) * 0.1
ws = [torch.randn(64, 64, 3, 3, device=dev) * 0.04 for _ in range(NL)]
bs = torch.zeros(64, device=dev)
wps = [ops.pack_weight_fwd(w, False, 0) for w in ws]
bufs = [torch.empty_like(x0) for _ in range(2)]
for algo_name, algo in (("bf16x3", _lib.ALGO_MFMA_BF16X3), ("bf16x6", _lib.ALGO_MFMA_BF16X6)):
    for mode in ("one filter", "32 filters"):
        cfg = ops.ConvCfg(1, 1, False, 0, 1, 0.0, 0, algo)
        d = ops._make_desc(x0.shape, ws[0], cfg, "infer")

        def chain():
            src = x0
            for i in range(NL):
                dst = bufs[i & 1]
                ep = _lib.Epilogue(ptr(bs), None, None, 0.0, 1, 0, 0, None, None)
                wp = wps[0] if mode == "one filter" else wps[i]
                check(lib.srk_conv2d_forward(ctypes.byref(d), ptr(src), ptr(wp), ptr(dst), ctypes.byref(ep), stream_ptr()), "fwd")
                src = dst
        for _ in range(2):
            chain()
        g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                chain()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        print("%-7s %-11s %.2f us per conv  (%s)" % (algo_name, mode, e0.elapsed_time(e1) / (10 * NL) * 1e3, lib.srk_last_kernel_name().decode()))
