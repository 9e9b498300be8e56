#!/usr/bin/env python3
"""The many-tap few-output-channel conv (k_conv_rown) against k_conv_tapn on one box: us per launch (HIP events over 30 queued
launches) for every tile height, both arithmetics.   python tools/time_rown.py [N H W Cin Cout K pad]"""
import os, sys, torch
os.environ["SRK_ENV_LIVE"] = "1"   # this tool flips SRK_ROWN / SRK_ROWN_TH between calls of one process
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
ops = pkg.ops
a = [int(v) for v in sys.argv[1:8]] or [16, 128, 128, 64, 3, 9, 4]
N, H, W, cin, cout, k, pad = a
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(7)
x = torch.randn(N, cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev)
b = torch.randn(cout, generator=g).to(dev)
flop = 2.0 * N * (H + 2 * pad - k + 1) * (W + 2 * pad - k + 1) * cin * cout * k * k
QUICK = bool(os.environ.get("ROWN_QUICK"))
for algo, aname in (((4, "bf16x3"),) if QUICK else ((4, "bf16x3"), (5, "bf16x6"))):
    cfg = ops.ConvCfg(1, pad, False, 0, 0, 0.0, 0, algo)
    wp, bp = ops.pack_weight_fwd(w, False, 0), ops.pack_bias_ps(b, 0)
    ref = None
    for env in (({"SRK_ROWN_TH": "8"}, {"SRK_ROWN_TH": "64"}) if QUICK else
                ({"SRK_ROWN": "0"}, {"SRK_ROWN_TH": "8"}, {"SRK_ROWN_TH": "16"}, {"SRK_ROWN_TH": "32"}, {"SRK_ROWN_TH": "64"}, {})):
        for kk in ("SRK_ROWN", "SRK_ROWN_TH"):
            os.environ.pop(kk, None)
        os.environ.update(env)
        with torch.no_grad():
            for _ in range(5):
                y = ops.conv2d_infer(x, w, b, None, cfg, None, (wp, bp))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                y = ops.conv2d_infer(x, w, b, None, cfg, None, (wp, bp))
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        name = pkg._lib.load().srk_last_kernel_name().decode()
        if ref is None:
            ref = y
        err = float((y - ref).abs().max() / ref.abs().max())
        print("%-7s %-22s %-28s %8.1f us  %6.1f TFLOP/s  vs first %.2e" % (aname, str(env or "default"), name, us, flop / us / 1e6, err))
