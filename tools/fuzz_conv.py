#!/usr/bin/env python3
"""Randomised shape fuzz of the conv entry points against torch fp64 on CPU (run on the GPU box):
   python tools/fuzz_conv.py [cases] [seed] [big]     (env switches such as SRK_BFW=1 select forced kernel families;
   "big": fewer, benchmark-class shapes that take the large-problem kernels)"""
import os, random, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
ops = pkg.ops
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
big = len(sys.argv) > 3 and sys.argv[3] == "big"
ACT = {None: 0, "relu": 1, "lrelu": 3}
worst = 0.0
bad = 0
for i in range(cases):
    cin = rng.choice([3, 8, 16, 24, 32, 40, 64, 72, 96, 128])
    cout = rng.choice([1, 2, 3, 8, 16, 32, 48, 64, 96, 128])
    k = rng.choice([1, 3, 3, 3, 5])
    p = rng.choice([0, k // 2])
    s = rng.choice([1, 1, 1, 2])
    H, W, N = rng.randint(k + 2, 44), rng.randint(k + 2, 44), rng.randint(1, 4)
    if big:
        cin, cout, k, s = rng.choice([32, 64]), rng.choice([3, 32, 48, 64]), 3, 1
        p = rng.choice([0, 1])
        H, W, N = rng.randint(48, 140), rng.randint(48, 140), rng.randint(8, 32)
    act = rng.choice([None, "relu", "lrelu"])
    g = torch.Generator().manual_seed(1000 + i)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    pre = torch.nn.functional.conv2d(xr, wr, br, s, p)
    ref = pre if act is None else (torch.relu(pre) if act == "relu" else torch.nn.functional.leaky_relu(pre, 0.2))
    gr = torch.randn(ref.shape, generator=g)
    cfg = ops.ConvCfg(s, p, False, 0, ACT[act], 0.2 if act == "lrelu" else 0.0, 0, 0)
    def rel(a, r):
        return float((a.detach().cpu().double() - r).abs().max() / max(float(r.abs().max()), 1e-30))
    with torch.no_grad():
        yi = ops.conv2d_infer(x.to(dev), w.to(dev), b.to(dev), None, cfg)
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    yt = ops.conv2d(xg, wg, bg, None, cfg)
    yt.backward(gr.to(dev))
    # reference gradients with the activation derivative decided by the PRODUCT's forward output: a pre-activation within
    # fp32 rounding of zero may legitimately fall on either side (torch fp32 itself differs from fp64 there), and one
    # flipped unit moves max-norm gradients by ~1e-2 at benchmark sizes
    slope = 0.0 if act == "relu" else 0.2
    gmask = torch.ones_like(pre) if act is None else torch.where(yt.detach().cpu().double() > 0, 1.0, slope)
    pre.backward(gr.double() * gmask)
    errs = (rel(yi, ref.detach()), rel(yt, ref.detach()), rel(xg.grad, xr.grad), rel(wg.grad, wr.grad), rel(bg.grad, br.grad))
    worst = max(worst, max(errs))
    if max(errs) > 2e-4 or any(e != e for e in errs):
        bad += 1
        print("BAD case %d: cin %d cout %d k %d p %d s %d %dx%d N %d act %s -> %s" % (i, cin, cout, k, p, s, H, W, N, act,
                                                                                     " ".join("%.2e" % e for e in errs)))
print("cases %d, failures %d, worst relative error %.2e" % (cases, bad, worst))
sys.exit(1 if bad else 0)
