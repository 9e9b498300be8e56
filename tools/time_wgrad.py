#!/usr/bin/env python3
"""Time the weight-gradient call (kernel + slab reduce) for workhorse shapes, 20 launches in a hipGraph."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd._lib import ConvDesc, check, load, ptr, stream_ptr
lib = load()
SHAPES = {"edsr128": (128, 64, 32, 32, 64, 3, 1), "vdsr": (256, 64, 41, 41, 64, 3, 1), "edsr16": (16, 64, 32, 32, 64, 3, 1),
          "edsrtail128": (128, 64, 128, 128, 3, 3, 1), "edsrtail16": (16, 64, 128, 128, 3, 3, 1), "vdsrtail": (256, 64, 41, 41, 3, 3, 1),
          # SRGAN discriminator class: few pixels, many channels (the slab reduce and its scattered dw stores dominate)
          # first layers (Cin = 3: k_wgrad_mfma_smallcin): VDSR 3x3, SRGAN-G 9x9 at 32x32, SRGAN-D 3x3 at 128x128
          "vdsrhead": (256, 3, 41, 41, 64, 3, 1), "ghead9": (16, 3, 32, 32, 64, 9, 4), "dhead": (16, 3, 128, 128, 64, 3, 1),
          "d256": (16, 128, 32, 32, 256, 3, 1), "d512": (16, 256, 16, 16, 512, 3, 1), "d512b": (16, 512, 8, 8, 512, 3, 1)}
dev = torch.device("cuda:0")
for name in (sys.argv[1:] or list(SHAPES)):
    N, cin, H, W, cout, k, pad = SHAPES[name]
    torch.manual_seed(3)
    x = torch.randn(N, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    dw = torch.zeros(cout, cin, k, k, device=dev); db = torch.zeros(cout, device=dev)
    from pytorch_super_resolution_model_collection_amd._lib import BwdMask
    ym = torch.randn(N, cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)   # ReLU mask (VDSR / EDSR conv1)
    mask = BwdMask(ym.data_ptr(), 0.0) if os.environ.get("MASK", "1") != "0" else None
    d = ConvDesc(N, H, W, cin, H, W, cout, k, k, 1, pad, 0, 0, 0)
    ws = torch.empty(int(lib.srk_conv2d_backward_weight_workspace_bytes(ctypes.byref(d))), dtype=torch.uint8, device=dev)
    def run():
        check(lib.srk_conv2d_backward_weight(ctypes.byref(d), ptr(x), ptr(dy), ctypes.byref(mask) if mask else None, ptr(dw), ptr(db), 0.0, ptr(ws),
                                             ws.numel(), stream_ptr()), "wgrad")
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
    if os.environ.get("LOOP_SECS"):   # keep the layer looping for the power / clock probe (tools/power_probe.sh)
        import time
        t0 = time.time(); n = 0
        while time.time() - t0 < float(os.environ["LOOP_SECS"]):
            for _ in range(20): g.replay()
            torch.cuda.synchronize(); n += 400
        ms = (time.time() - t0) / n * 1e3
    print("%-8s wgrad+reduce %.4f ms  %.1f TF  checksum %.6e" % (name, ms, 2.0 * N * H * W * cin * cout * k * k / ms / 1e9,
                                                             float(dw.double().abs().sum())))
