#!/usr/bin/env python3
"""Role-time sums inside the wave-specialised weight-gradient kernel (first stager wave: LDS commit / load issue / barrier
wait; worker wave 0: K loop / barrier wait) for a VDSR body layer (256 x 41 x 41, 64 -> 64) and an EDSR body layer
(128 x 32 x 32).  Needs a library built with SRK_BUILD_EXPERIMENTS=1.   python tools/wgrad_prof.py"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd import _lib
ops = pkg.ops
lib = _lib.load()
P = _lib.ptr
dev = torch.device("cuda:0")
lib.srk_debug_wgrad_prof.argtypes = [ctypes.c_void_p]
lib.srk_debug_wgrad_prof.restype = None
prof = torch.zeros(8192 * 16, dtype=torch.int64, device=dev)
CL = torch.channels_last
for name, n, h in (("VDSR body layer 256 x 41 x 41", 256, 41), ("EDSR body layer 128 x 32 x 32", 128, 32)):
    x = torch.randn(n, 64, h, h, device=dev).contiguous(memory_format=CL)
    dy = torch.randn(n, 64, h, h, device=dev).contiguous(memory_format=CL)
    w = torch.zeros(64, 64, 3, 3, device=dev)
    d = ops._make_desc(x.shape, w, ops.ConvCfg(1, 1, False, 0, 0, 0.0, 0), "bwd")
    ws = torch.empty(int(lib.srk_conv2d_backward_weight_workspace_bytes(ctypes.byref(d))) + 16, dtype=torch.uint8, device=dev)
    dw, db = torch.zeros_like(w), torch.zeros(64, device=dev)

    def run():
        rc = lib.srk_conv2d_backward_weight(ctypes.byref(d), P(x), P(dy), None, P(dw), P(db), 0.0, P(ws), ws.numel(), _lib.stream_ptr())
        assert rc == 0, lib.srk_last_error_string()
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    prof.zero_()
    lib.srk_debug_wgrad_prof(P(prof))
    run()
    torch.cuda.synchronize()
    lib.srk_debug_wgrad_prof(None)
    t = prof.view(-1, 16).cpu().double()
    t = t[t[:, 10] > 0]
    tot_s = t[:, 0] + t[:, 1] + t[:, 2]
    tot_w = t[:, 8] + t[:, 9]
    print("%s: %.1f us per call (kernel + reduce), %d blocks x %.1f tiles x %.1f K steps" % (name, us, t.shape[0], float(t[:, 10].mean()),
                                                                                              float((t[:, 11] / t[:, 10]).mean())))
    print("   stager wave: commit %4.1f %%  issue %4.1f %%  barrier wait %4.1f %%   (per tile: %.0f / %.0f / %.0f clocks)" % (
        100 * float((t[:, 0] / tot_s).mean()), 100 * float((t[:, 1] / tot_s).mean()), 100 * float((t[:, 2] / tot_s).mean()),
        float((t[:, 0] / t[:, 3]).mean()), float((t[:, 1] / t[:, 3]).mean()), float((t[:, 2] / t[:, 3]).mean())))
    print("   worker wave: K loop %4.1f %%  barrier wait %4.1f %%   (per tile: %.0f / %.0f clocks; per K step %.0f clocks for 864 of MFMA)" % (
        100 * float((t[:, 8] / tot_w).mean()), 100 * float((t[:, 9] / tot_w).mean()), float((t[:, 8] / t[:, 10]).mean()),
        float((t[:, 9] / t[:, 10]).mean()), float((t[:, 8] / t[:, 11]).mean())))
    if float(t[:, 12].sum()) > 0:   # k_wgrad_tr: the part of the K loop in front of a tile's first K step (ring bookkeeping, addresses, first reads issued)
        print("   worker wave: per tile %.0f clocks in front of the first K step; (K loop - that) per K step %.0f" % (
            float((t[:, 12] / t[:, 10]).mean()), float(((t[:, 8] - t[:, 12]) / t[:, 11]).mean())))
        print("   worker wave: block start -> first tile staged %.0f clocks, tile loops %.0f, last tile -> slab stored %.0f, block life %.0f "
              "(longest block %.0f)" % (float(t[:, 13].mean()), float((t[:, 8] + t[:, 9]).mean()), float(t[:, 14].mean()),
                                        float(t[:, 15].mean()), float(t[:, 15].max())))
