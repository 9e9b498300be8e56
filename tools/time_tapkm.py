#!/usr/bin/env python3
"""Data gradient of the many-tap few-output-channel conv (SRGAN-G output conv: dx[64] from dy[3], 9x9) -- k_conv_tapkm against
the kernel it replaces (SRK_TAPKM=0: k_conv_mfma_tg), us per launch over 30 queued launches.  python tools/time_tapkm.py [N H W Cin Cout K pad]"""
import ctypes, os, sys, torch
os.environ["SRK_ENV_LIVE"] = "1"   # this tool flips SRK_TAPKM between calls of one process
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
ops, L = pkg.ops, pkg._lib
lib = L.load()
a = [int(v) for v in sys.argv[1:8]] or [16, 128, 128, 64, 3, 9, 4]
N, H, W, cin, cout, k, pad = a
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(7)
w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev)
cfg = ops.ConvCfg(1, pad, False, 0, 0, 0.0, 0, 0)
d = ops._make_desc((N, cin, H, W), w, cfg, "bwd")
dy = torch.randn(N, cout, d.OH, d.OW, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
wpb = ops.pack_weight_bwd(w, False, 0)
dx = torch.empty((N, cin, H, W), device=dev).contiguous(memory_format=torch.channels_last)
flop = 2.0 * N * d.OH * d.OW * cin * cout * k * k
ref = None
for env in ({"SRK_TAPKM": "0"}, {}):
    os.environ.pop("SRK_TAPKM", None)
    os.environ.update(env)
    def run():
        rc = lib.srk_conv2d_backward_data(ctypes.byref(d), L.ptr(dy), L.ptr(wpb), L.ptr(dx), None, None, L.stream_ptr())
        assert rc == 0, lib.srk_last_error_string()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    if ref is None:
        ref = dx.clone()
    err = float((dx - ref).abs().max() / ref.abs().max())
    print("%-20s %-24s %8.1f us  %6.1f TFLOP/s  %6.1f GB/s written  vs first %.2e" % (
        str(env or "default"), lib.srk_last_kernel_name().decode(), us, flop / us / 1e6, dx.numel() * 4 / us / 1e3, err))
