#!/usr/bin/env python3
"""The fused first-layer + 3x3 pair (srk_conv2d_pair_forward) against the two-call form and torch fp64, and its time.
   python tools/pair_check.py [N H W]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd import base_networks as bn
ops = pkg.ops; lib = pkg._lib.load(); dev = torch.device("cuda:0")
N, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 256, 256)
torch.manual_seed(3)
net = pkg.ESPCNNet(3, 64, 4); net.weight_init()
for m in net.modules():
    if isinstance(m, pkg.layers.Conv2d) and m.bias is not None: m.bias.data.uniform_(-0.1, 0.1)
net.to(dev).eval()
x = torch.rand(N, 3, H, W, device=dev)
ops.PAIR = "1"
with torch.no_grad():
    lib.srk_ring_timeouts(1)
    hf = bn.fused_conv_pair(net.layers[0], net.layers[1], x)
    print("fused:", None if hf is None else (tuple(hf.shape), lib.srk_last_kernel_name().decode()), "timeouts", lib.srk_ring_timeouts(1))
    h2 = net.layers[1](net.layers[0](x))
    print("two calls:", lib.srk_last_kernel_name().decode())
    if hf is not None:
        ref = torch.relu(torch.nn.functional.conv2d(torch.relu(torch.nn.functional.conv2d(x.double().cpu(), net.layers[0].conv.weight.double().cpu(), net.layers[0].conv.bias.double().cpu())),
                                                    net.layers[1].conv.weight.double().cpu(), net.layers[1].conv.bias.double().cpu())) if N * H * W <= 64 * 64 * 64 else None
        d = (hf - h2).abs().max().item() / h2.abs().max().item()
        print("fused vs two calls: max rel %.3e" % d, " amax tags:", float(hf._srk_amax[0].max()) if getattr(hf, "_srk_amax", None) else None, float(hf.abs().max()))
        if ref is not None:
            print("fused vs fp64: %.3e   two calls vs fp64: %.3e" % ((hf.cpu().double() - ref).abs().max().item() / ref.abs().max().item(),
                                                                    (h2.cpu().double() - ref).abs().max().item() / ref.abs().max().item()))
        bad = (hf - h2).abs() > 1e-4 * h2.abs().max()
        if bad.any():
            print("mismatches:", int(bad.sum()), "of", bad.numel(), "per channel", bad.sum((0, 2, 3)).tolist()[:8], "rows", bad.sum((0, 1, 3)).tolist()[:20])
    def timeit(fn, n=20):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    if hf is not None:
        print("fused pair: %.1f us    two calls: %.1f us" % (timeit(lambda: bn.fused_conv_pair(net.layers[0], net.layers[1], x)),
                                                             timeit(lambda: net.layers[1](net.layers[0](x)))))
        ops.PAIR = "1"; t_f = timeit(lambda: net(x)); ops.PAIR = None; t_2 = timeit(lambda: net(x))
        print("whole net: fused %.1f us (%.0f img/s)   three launches %.1f us (%.0f img/s)" % (t_f, N / t_f * 1e6, t_2, N / t_2 * 1e6))
