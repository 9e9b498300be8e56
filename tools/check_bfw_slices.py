#!/usr/bin/env python3
"""Output-channel slices of k_conv_bfw (64 -> 64 / 128 / 256, plain and pixel-shuffled stores) against the per-tile
bf16x3 kernels (SRK_BFW=0): same arithmetic and accumulation order, so the outputs must be equal."""
import os, sys, torch
os.environ["SRK_ENV_LIVE"] = "1"   # this tool flips SRK_* switches in-process: the library must re-read them (csrc/api.hip env_str)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
ops = pkg.ops
dev = torch.device("cuda:0"); lib = pkg._lib.load()
ops.set_precision("bf16x3")
torch.manual_seed(3)
for (n, hw, cin, cout, r, act) in [(128, 32, 64, 64, 0, ops.ACT_RELU), (128, 32, 64, 128, 0, ops.ACT_NONE),
                                   (128, 32, 64, 256, 2, ops.ACT_NONE), (32, 64, 64, 256, 2, ops.ACT_NONE),
                                   (128, 32, 64, 256, 0, ops.ACT_LRELU)]:
    x = torch.randn(n, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    cfg = ops.ConvCfg(1, 1, False, 0, act, 0.2, r)
    outs, names = [], []
    for mode in ("0", None):
        if mode is None: os.environ.pop("SRK_BFW", None)
        else: os.environ["SRK_BFW"] = mode
        with torch.no_grad():
            y = ops.conv2d_infer(x, w, b, None, cfg)
        names.append(lib.srk_last_kernel_name().decode()); outs.append(y.float().contiguous())
    d = (outs[0] - outs[1]).abs()
    print((n, hw, cin, cout, r), names, "max diff %.3e" % float(d.max()), "mismatching channels:",
          sorted(set((d.amax(dim=(0, 2, 3)) > 0).nonzero().flatten().tolist()))[:12])
