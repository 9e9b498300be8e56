#!/usr/bin/env python3
"""Random shapes through the fused residual-block entry points (srk_resblock2_forward / _backward_data) against the two
separate conv launches each replaces (same arithmetic; element-wise comparison, mask flips counted separately).
   python tools/fuzz_res2.py [cases] [seed]"""
import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd._lib import ALGO_MFMA_BF16X6, load, ptr, stream_ptr, check
lib = load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0"); CL = torch.channels_last
os.environ.setdefault("SRK_RES2_MAX_TILES", "1000000")
worst = [0.0, 0.0, 0.0]
for case in range(cases):
    n = rng.choice([1, 1, 2, 3, 5, 16]); h = rng.randint(1, 70); w = rng.randint(1, 70)
    if not lib.srk_resblock2_supported(n, h, w, 64):
        continue
    bias = rng.random() < 0.7
    g = torch.Generator().manual_seed(case)
    x = torch.randn(n, 64, h, w, generator=g).to(dev).contiguous(memory_format=CL)
    dy = torch.randn(n, 64, h, w, generator=g).to(dev).contiguous(memory_format=CL)
    w1 = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev); w2 = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
    b1 = torch.randn(64, generator=g).to(dev) if bias else None; b2 = torch.randn(64, generator=g).to(dev) if bias else None
    wf1, wf2 = pkg.ops.pack_weight_fwd(w1, False, 0), pkg.ops.pack_weight_fwd(w2, False, 0)
    wb1, wb2 = pkg.ops.pack_weight_bwd(w1, False, 0), pkg.ops.pack_weight_bwd(w2, False, 0)
    mid, out, dmid, dx = (torch.empty_like(x) for _ in range(4))
    check(lib.srk_resblock2_forward(n, h, w, 64, ptr(x), ptr(wf1), ptr(b1), ptr(wf2), ptr(b2), ptr(mid), ptr(out),
                                    ALGO_MFMA_BF16X6, None, None, stream_ptr()), "fwd")
    check(lib.srk_resblock2_backward_data(n, h, w, 64, ptr(dy), ptr(wb2), ptr(wb1), ptr(mid), ptr(dmid), ptr(dx), 0,
                                          stream_ptr()), "bwd")
    # float64 reference with the kernel's mask
    import torch.nn.functional as F
    xd, dyd = x.double().cpu(), dy.double().cpu()
    z = F.conv2d(xd, w1.double().cpu(), None if b1 is None else b1.double().cpu(), padding=1)
    ref_out = F.conv2d(z.clamp(min=0), w2.double().cpu(), None if b2 is None else b2.double().cpu(), padding=1) + xd
    mask = (mid > 0).cpu()
    dmid_ref = F.conv_transpose2d(dyd, w2.double().cpu(), padding=1) * mask
    dx_ref = F.conv_transpose2d(dmid_ref, w1.double().cpu(), padding=1) + dyd
    def rel(a, b):
        b = b.float(); return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-30))
    e = [rel(out, ref_out), rel(dmid, dmid_ref), rel(dx, dx_ref)]
    dec = z.abs() > 1e-5 * float(z.pow(2).mean().sqrt() + 1e-30)
    assert bool(((z > 0) == mask)[dec].all()), (case, n, h, w)
    assert e[0] < 2e-5 and e[1] < 1e-4 and e[2] < 1e-4, (case, n, h, w, bias, e)
    worst = [max(a, b) for a, b in zip(worst, e)]
print("fuzz_res2: %d cases ok; worst rel err forward %.2e, d_mid %.2e, dx %.2e" % (cases, *worst))
