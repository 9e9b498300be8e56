#!/usr/bin/env python3
"""Where does k_conv_bfr differ from k_conv_bfw?  python tools/ring_dbg.py cin cout H W N pad [mode]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SRK_ENV_LIVE"] = "1"
os.environ["SRK_BFW"] = "1"
os.environ.setdefault("SRK_BF3_DIRECT", "0")
import pytorch_super_resolution_model_collection_amd as pkg
ops = pkg.ops
lib = pkg._lib.load()
cin, cout, H, W, N, pad = (int(a) for a in sys.argv[1:7])
mode = sys.argv[7] if len(sys.argv) > 7 else "bf16x3"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
x = torch.randn(N, cin, H, W, generator=g).to(dev)
w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(dev)
b = (torch.randn(cout, generator=g) * 0.1).to(dev)
ops.F16X3_ALWAYS = mode == "mixed"
ops.set_precision(mode)
cfg = ops.ConvCfg(1, pad, False, 0, 1, 0.0, 0, 0)
outs = {}
for ring in ("0", "1"):
    os.environ["SRK_BFR"] = ring
    with torch.no_grad():
        outs[ring] = ops.conv2d_infer(x, w, b, None, cfg).clone()
    torch.cuda.synchronize()
    print("SRK_BFR=%s kernel %s timeouts %d" % (ring, lib.srk_last_kernel_name().decode(), lib.srk_ring_timeouts(1)))
d = (outs["0"] != outs["1"])
print("differing elements: %d of %d" % (int(d.sum()), d.numel()))
if d.any():
    print("per image:", d.flatten(1).sum(1).tolist())
    print("per channel:", d.sum((0, 2, 3)).tolist())
    rows = d.sum((0, 1, 3)).tolist(); cols = d.sum((0, 1, 2)).tolist()
    print("per row:", rows); print("per col:", cols)
    i = d.nonzero()[0].tolist(); print("first:", i, float(outs["0"][tuple(i)]), float(outs["1"][tuple(i)]))
