#!/usr/bin/env python3
"""Time the fused residual-block launches (srk_resblock2_forward / _backward_data) alone: 20 launches in a hipGraph.
   python tools/time_res2.py [B] [H]     (SRK_DBG: 1 skip the halo staging, 4 skip the matrix work)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd._lib import ALGO_MFMA_BF16X6, load, ptr, stream_ptr, check
lib = load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
CL = torch.channels_last
x = torch.randn(B, 64, H, H, device=dev).contiguous(memory_format=CL)
dy = torch.randn(B, 64, H, H, device=dev).contiguous(memory_format=CL)
w1, w2 = torch.randn(64, 64, 3, 3, device=dev) * 0.05, torch.randn(64, 64, 3, 3, device=dev) * 0.05
b1, b2 = torch.randn(64, device=dev), torch.randn(64, device=dev)
wf1, wf2 = pkg.ops.pack_weight_fwd(w1, False, 0), pkg.ops.pack_weight_fwd(w2, False, 0)
wb1, wb2 = pkg.ops.pack_weight_bwd(w1, False, 0), pkg.ops.pack_weight_bwd(w2, False, 0)
mid, out, dmid, dx = (torch.empty_like(x) for _ in range(4))


def fwd():
    check(lib.srk_resblock2_forward(B, H, H, 64, ptr(x), ptr(wf1), ptr(b1), ptr(wf2), ptr(b2), ptr(mid), ptr(out),
                                    ALGO_MFMA_BF16X6, None, None, stream_ptr()), "fwd")


xa = pkg.ops.amax_of(x)
ya = torch.zeros(256, device=dev)


def fwd16():
    check(lib.srk_resblock2_forward(B, H, H, 64, ptr(x), ptr(wf1), ptr(b1), ptr(wf2), ptr(b2), ptr(mid), ptr(out),
                                    pkg._lib.ALGO_MFMA_F16X3, ptr(xa), ptr(ya), stream_ptr()), "fwd16")


def bwd():
    check(lib.srk_resblock2_backward_data(B, H, H, 64, ptr(dy), ptr(wb2), ptr(wb1), ptr(mid), ptr(dmid), ptr(dx), 0,
                                          stream_ptr()), "bwd")


for name, fn in (("forward bf16x6", fwd), ("forward f16x3", fwd16), ("backward bf16x3", bwd)):
    for _ in range(3):
        fn()
    side, g = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(20):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    flop = 2.0 * B * H * H * 64 * 64 * 9 * 2
    print("res2 %-16s B=%d %dx%d  %.2f us  %.1f TF (useful)  [SRK_DBG=%s]" % (name, B, H, H, us, flop / us / 1e6,
                                                                              os.environ.get("SRK_DBG", "0")))
