#!/usr/bin/env python3
"""Time the data-gradient call for the reconstruction-conv shapes (and others), 20 launches in a hipGraph."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd._lib import ConvDesc, check, load, ptr, stream_ptr
lib = load()
SHAPES = {"edsrtail128": (128, 64, 128, 128, 3, 3, 1), "edsrtail16": (16, 64, 128, 128, 3, 3, 1),
          "vdsrtail": (256, 64, 41, 41, 3, 3, 1), "edsr128": (128, 64, 32, 32, 64, 3, 1)}
dev = torch.device("cuda:0")
for name in (sys.argv[1:] or list(SHAPES)):
    N, cin, H, W, cout, k, pad = SHAPES[name]
    torch.manual_seed(3)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    dy = torch.randn(N, cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    dx = torch.empty(N, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    wpb = pkg.ops.pack_weight_bwd(w, False)
    d = ConvDesc(N, H, W, cin, H, W, cout, k, k, 1, pad, 0, 0, 0)
    def run():
        check(lib.srk_conv2d_backward_data(ctypes.byref(d), ptr(dy), ptr(wpb), ptr(dx), None, None, stream_ptr()), "dgrad")
    for _ in range(3): run()
    side = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(20): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    ref = torch.nn.grad.conv2d_input((1, cin, H, W), w.double().cpu(), dy[:1].double().cpu().contiguous(), 1, pad)
    err = float((dx[:1].double().cpu() - ref).abs().max() / ref.abs().max())
    assert bool(torch.isfinite(dx).all())
    print("%-12s dgrad %.4f ms  %.1f TF  img0 rel err %.2e" % (name, ms, 2.0 * N * H * W * cin * cout * k * k / ms / 1e9, err))
