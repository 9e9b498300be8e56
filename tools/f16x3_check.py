#!/usr/bin/env python3
"""SRK_ALGO_MFMA_F16X3 against the other conv arithmetics: error vs float64 and time per launch, on the body-layer shapes
of c3 (VDSR 256 x 41 x 41) and c4 (EDSR 128 x 32 x 32, and the 16-patch shard).   python tools/f16x3_check.py"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd import _lib, ops
from pytorch_super_resolution_model_collection_amd._lib import ptr, stream_ptr, check
lib = _lib.load()
dev = torch.device("cuda:0")
CL = torch.channels_last
torch.manual_seed(0)


def run(x, wp, b, w, algo, xa=None, ya=None, act=1):
    cfg = ops.ConvCfg(1, w.shape[-1] // 2, False, 0, act, 0.0, 0, algo)
    d = ops._make_desc(x.shape, w, cfg, "infer")
    y = ops._empty_cl(d.N, d.Cout, d.OH, d.OW, x)
    ep = _lib.Epilogue(ptr(b), None, None, 0.0, act, 0, 0, ptr(xa), ptr(ya))
    check(lib.srk_conv2d_forward(ctypes.byref(d), ptr(x), ptr(wp), ptr(y), ctypes.byref(ep), stream_ptr()), "fwd")
    return y


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


CASES = [(256, 41, 1.0, 0.05, 64, 64, 3), (128, 32, 1.0, 0.05, 64, 64, 3), (16, 32, 1.0, 0.05, 64, 64, 3),
         (4, 24, 1e-4, 3.0, 64, 64, 3), (4, 24, 300.0, 1e-3, 64, 64, 3), (16, 56, 30.0, 0.2, 64, 32, 5), (16, 56, 1.0, 0.05, 64, 32, 5),
         (64, 250, 1.0, 0.05, 32, 48, 3)]
if len(sys.argv) > 1:
    CASES = [CASES[int(a)] for a in sys.argv[1:]]
for (B, H, scale_x, scale_w, CI, CO, KS) in CASES:
    x = (torch.randn(B, CI, H, H, device=dev).clamp_min(0) * scale_x).contiguous(memory_format=CL)
    w = torch.randn(CO, CI, KS, KS, device=dev) * scale_w
    b = torch.randn(CO, device=dev) * 0.1 * scale_x * scale_w * 24
    wp = ops.pack_weight_fwd(w, False, 0)
    xa = ops.amax_of(x)
    ya = torch.zeros(256, device=dev)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=KS // 2))
    print("--- conv%dx%d %d->%d + ReLU, B=%d %dx%d, |x|~%g |w|~%g" % (KS, KS, CI, CO, B, H, H, scale_x, scale_w))
    for name, algo, a in (("bf16x3", _lib.ALGO_MFMA_BF16X3, None), ("bf16x6", _lib.ALGO_MFMA_BF16X6, None),
                          ("f16x3", _lib.ALGO_MFMA_F16X3, xa), ("fp32 mfma", _lib.ALGO_MFMA, None)):
        y = run(x, wp, b, w, algo, a, ya if a is not None else None)
        e = (y.double() - ref)
        rms = float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        mx = float(e.abs().max() / ref.abs().max())
        flips = int(((y > 0) != (ref > 0)).sum())
        us = timeit(lambda: run(x, wp, b, w, algo, a, None)) if B >= 16 else float("nan")
        print("  %-10s rms %.2e  max %.2e  relu-sign flips %d / %d   %.1f us  (%s)" % (name, rms, mx, flips, ref.numel(), us,
              lib.srk_last_kernel_name().decode()))
    torch.cuda.synchronize()
    got = float(ya.max())
    want = float(run(x, wp, b, w, _lib.ALGO_MFMA, None, None).abs().max())
    print("  y_amax slots max %.6g vs max|y| %.6g" % (got, want))
