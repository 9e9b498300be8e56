#!/usr/bin/env python3
"""Runs one conv shape repeatedly (for rocprofv3 --pmc runs). Usage: prof_conv.py <shape> [algo] [iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
ops = pkg.ops
SHAPES = {  # N, Cin, H, W, Cout, k, pad, act, ps
    "vdsr": (256, 64, 41, 41, 64, 3, 1, 1, 0),
    "espcn2": (64, 64, 252, 252, 32, 3, 0, 1, 0),
    "espcn3": (64, 32, 250, 250, 48, 3, 0, 0, 4),
    "espcn1": (64, 3, 256, 256, 64, 5, 0, 1, 0),
    "edsrup": (128, 64, 64, 64, 256, 3, 1, 0, 2),
}
name = sys.argv[1] if len(sys.argv) > 1 else "vdsr"
algo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N, cin, H, W, cout, k, pad, act, ps = SHAPES[name]
dev = torch.device("cuda:0")
x = torch.randn(N, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(cout, cin, k, k, device=dev) * 0.05
b = torch.randn(cout, device=dev)
cfg = ops.ConvCfg(1, pad, False, 0, act, 0.0, ps, algo)
wp, bp = ops.pack_weight_fwd(w, False, ps), ops.pack_bias_ps(b, ps)
with torch.no_grad():
    for _ in range(iters):
        y = ops.conv2d_infer(x, w, b, None, cfg, None, (wp, bp))
torch.cuda.synchronize()
print("done", name, tuple(y.shape))
