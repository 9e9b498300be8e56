#!/usr/bin/env python3
"""Time single fused-conv launches for the workhorse shapes (HIP events, 20 reps) under the current env
(SRK_DBG / SRK_BF3_WAVES / SRK_FORCE_ALGO) and print a checksum so variants can be compared for equality.
Usage: time_shapes.py [shape ...]"""
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
    "edsr128": (128, 64, 32, 32, 64, 3, 1, 1, 0),
    "edsr16": (16, 64, 32, 32, 64, 3, 1, 1, 0),
    "edsrup16": (16, 64, 64, 64, 256, 3, 1, 0, 2),
    "srgan9": (16, 64, 128, 128, 3, 9, 4, 0, 0),
    "edsrtail128": (128, 64, 128, 128, 3, 3, 1, 0, 0),
    "edsrtail16": (16, 64, 128, 128, 3, 3, 1, 0, 0),
    "vdsrtail": (256, 64, 41, 41, 3, 3, 1, 0, 0),
}
names = sys.argv[1:] or ["espcn2", "espcn3", "vdsr", "edsr128", "edsr16", "edsrup16"]
dev = torch.device("cuda:0")
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in ("SRK_DBG", "SRK_BF3_WAVES", "SRK_FORCE_ALGO") if k in os.environ)
for name in names:
    N, cin, H, W, cout, k, pad, act, ps = SHAPES[name]
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(N, cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    cfg = ops.ConvCfg(1, pad, False, 0, act, 0.0, ps, 0)
    wp, bp = ops.pack_weight_fwd(w, False, ps), ops.pack_bias_ps(b, ps)
    with torch.no_grad():
        for _ in range(3):
            y = ops.conv2d_infer(x, w, b, None, cfg, None, (wp, bp))
        torch.cuda.synchronize()
        # 20 launches captured in a hipGraph: back-to-back on the GPU, no host launch cost in the timing
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for _ in range(20):
                    y = ops.conv2d_infer(x, w, b, None, cfg, None, (wp, bp))
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    r = ps if ps > 1 else 1
    flop = 2.0 * N * (y.shape[2] // r) * (y.shape[3] // r) * cout * cin * k * k
    print("%-9s [%s] %8.4f ms  %6.1f TF  sum=%.6e abs=%.6e" % (name, tag, ms, flop / ms / 1e9, float(y.double().sum()),
                                                            float(y.double().abs().sum())))
