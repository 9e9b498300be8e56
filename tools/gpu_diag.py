#!/usr/bin/env python3
"""GPU diagnostic: runs every conv KAT with every algo and prints the error table (no asserts),
then times the hot conv shapes.  Usage on the GPU box: python tools/gpu_diag.py [--time]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__  # noqa: E402

__graft_entry__.build()
import pytorch_super_resolution_model_collection_amd as pkg  # noqa: E402
from oracle import fill  # noqa: E402
from oracle.kat_table import CONV_KATS, conv_case_inputs  # noqa: E402

ops = pkg.ops
dev = torch.device("cuda:0")
kat = np.load(os.path.join(ROOT, "tests", "golden", "ops_kat.npz"))
ACTS = {None: 0, "relu": 1, "lrelu": 3}


def rel(a, b):
    a = a.detach().float().cpu().numpy().astype(np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def kat_table():
    print("%-14s %-8s %10s %10s %10s %10s" % ("case", "algo", "y", "dx", "dw", "db"))
    for idx, c in enumerate(CONV_KATS):
        tag, cin, cout, k, s, p, tr, op, H, W, N, act = c
        for algo_name, algo in (("auto", 0), ("mfma32", 2), ("generic", 1)):
            x, w, b, g = conv_case_inputs(idx)
            xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
            try:
                cfg = ops.ConvCfg(s, p, bool(tr), op, ACTS[act], 0.2 if act == "lrelu" else 0.0, 0, algo)
                y = ops.conv2d(xg, wg, bg, None, cfg)
                ey = rel(y, kat["conv.%s.y" % tag])
                y.backward(g.to(dev))
                torch.cuda.synchronize()
                print("%-14s %-8s %10.2e %10.2e %10.2e %10.2e" % (
                    tag, algo_name, ey, rel(xg.grad, kat["conv.%s.dx" % tag]), rel(wg.grad, kat["conv.%s.dw" % tag]),
                    rel(bg.grad, kat["conv.%s.db" % tag])))
            except Exception as e:  # noqa: BLE001
                print("%-14s %-8s FAILED: %s" % (tag, algo_name, str(e)[:150]))


def time_fn(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def timing():
    shapes = [
        # (tag, N, Cin, H, W, Cout, k, pad, act, ps)
        ("espcn.l1 3->64 k5", 64, 3, 256, 256, 64, 5, 0, 1, 0),
        ("espcn.l2 64->32 k3", 64, 64, 252, 252, 32, 3, 0, 1, 0),
        ("espcn.l3 32->48 k3+ps4", 64, 32, 250, 250, 48, 3, 0, 0, 4),
        ("vdsr body 64->64 41x41 B256", 256, 64, 41, 41, 64, 3, 1, 1, 0),
        ("edsr body 64->64 32x32 B128", 128, 64, 32, 32, 64, 3, 1, 1, 0),
        ("edsr up2 64->256 64x64 B128", 128, 64, 64, 64, 256, 3, 1, 0, 2),
        ("edsr tail 64->3 128x128 B128", 128, 64, 128, 128, 3, 3, 1, 0, 0),
    ]
    for tag, N, cin, H, W, cout, k, pad, act, ps in shapes:
        x = torch.randn(N, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        cfg = ops.ConvCfg(1, pad, False, 0, act, 0.0, ps)
        wp, bp = ops.pack_weight_fwd(w, False, ps), ops.pack_bias_ps(b, ps)
        oh, ow = H + 2 * pad - k + 1, W + 2 * pad - k + 1
        flop = 2.0 * N * oh * ow * cout * cin * k * k
        byts = 4.0 * (N * H * W * cin + N * oh * ow * cout)
        for aname, algo in (("auto", 0), ("fp32", 2)):
            cfg.algo = algo
            with torch.no_grad():
                ms = time_fn(lambda: ops.conv2d_infer(x, w, b, None, cfg, None, (wp, bp)))
            print("%-32s %-5s %8.3f ms  %7.1f TFLOP/s  %7.1f GB/s (compulsory)" % (tag, aname, ms, flop / ms / 1e9,
                                                                                 byts / ms / 1e6))
    # training-direction kernels on the VDSR body shape
    N, C, H, W = 256, 64, 41, 41
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).requires_grad_(True)
    cfg = ops.ConvCfg(1, 1, False, 0, 1, 0.0, 0)
    y = ops.conv2d(x, w, None, None, cfg)
    g = torch.randn_like(y)

    def fb():
        y = ops.conv2d(x, w, None, None, cfg)
        y.backward(g)
    ms = time_fn(fb, iters=5, warm=2)
    flop = 3 * 2.0 * N * H * W * C * C * 9
    print("%-32s %8.3f ms  %7.1f TFLOP/s (fwd+dgrad+wgrad)" % ("vdsr body fwd+bwd", ms, flop / ms / 1e9))


if __name__ == "__main__":
    print("device:", torch.cuda.get_device_name(0), "| torch", torch.__version__, "| hip", torch.version.hip)
    t0 = time.time()
    kat_table()
    print("kat table: %.1fs" % (time.time() - t0))
    if "--time" in sys.argv:
        timing()
