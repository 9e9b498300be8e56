#!/usr/bin/env python3
"""Error of the conv arithmetic variants against an fp64 CPU convolution (64->64 3x3, N(0,1) data):
max-norm relative error and the number of ReLU sign decisions that differ from the fp64 result."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
ops = pkg.ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
N, C, H = 8, 64, 48
x = torch.randn(N, C, H, H, generator=g)
w = torch.randn(C, C, 3, 3, generator=g) * 0.05
b = torch.randn(C, generator=g) * 0.1
ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, 1)
cpu32 = torch.nn.functional.conv2d(x, w, b, 1, 1)
scale = float(ref.abs().max())


def report(name, y):
    y = y.double().cpu()
    err = float((y - ref).abs().max()) / scale
    rms = float((y - ref).pow(2).mean().sqrt()) / float(ref.pow(2).mean().sqrt())
    flips = int(((y > 0) != (ref > 0)).sum())
    print("%-22s max-rel %.3e  rms-rel %.3e  relu-sign flips %d / %d" % (name, err, rms, flips, ref.numel()))


report("torch CPU fp32", cpu32)
for name, algo, env in (("fp32 MFMA", 2, {}), ("generic fp32", 1, {}), ("bf16x3 (lds weights)", 4, {"SRK_BFD_SMALL": "0"}),
                        ("bf16x3 (bfd small)", 4, {}), ("bf16x6 (small)", 5, {}), ("bf16x6 (big)", 5, {"SRK_BFD_SMALL": "0"})):
    for k in ("SRK_BFD_SMALL", "SRK_BF3_DIRECT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    cfg = ops.ConvCfg(1, 1, False, 0, 0, 0.0, 0, algo)
    with torch.no_grad():
        y = ops.conv2d_infer(x.to(dev), w.to(dev), b.to(dev), None, cfg)
    report(name, y)
