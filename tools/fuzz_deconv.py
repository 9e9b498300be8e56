#!/usr/bin/env python3
"""Randomised fuzz of ConvTranspose2d (forward + all gradients) and of conv + fused pixel shuffle + residual (inference)
against torch fp64 on CPU.  python tools/fuzz_deconv.py [cases] [seed]"""
import os, random, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
ops = pkg.ops
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
F = torch.nn.functional
bad, worst = 0, 0.0
def rel(a, r):
    return float((a.detach().cpu().double() - r).abs().max() / max(float(r.abs().max()), 1e-30))
for i in range(cases):
    g = torch.Generator().manual_seed(5000 + i)
    if i % 2 == 0:  # deconv
        cin, cout = rng.choice([3, 8, 32, 56, 64]), rng.choice([3, 8, 32, 64])
        s = rng.choice([1, 2, 4])
        k = rng.choice([s, s + 1, 2 * s, 2 * s + 1, 9])
        p = rng.randint(0, (k - 1) // 2)
        op = rng.randint(0, s - 1)
        H, W, N = rng.randint(3, 20), rng.randint(3, 20), rng.randint(1, 3)
        x = torch.randn(N, cin, H, W, generator=g)
        w = torch.randn(cin, cout, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        ref = F.conv_transpose2d(xr, wr, br, s, p, op)
        if min(ref.shape[2:]) < 1:
            continue
        gr = torch.randn(ref.shape, generator=g)
        ref.backward(gr.double())
        cfg = ops.ConvCfg(s, p, True, op, 0, 0.0, 0, 0)
        xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
        y = ops.conv2d(xg, wg, bg, None, cfg)
        y.backward(gr.to(dev))
        errs = (rel(y, ref.detach()), rel(xg.grad, xr.grad), rel(wg.grad, wr.grad), rel(bg.grad, br.grad))
        tag = "deconv cin %d cout %d k %d s %d p %d op %d %dx%d N %d" % (cin, cout, k, s, p, op, H, W, N)
    else:  # conv + pixel shuffle (+ residual), inference
        r = rng.choice([2, 3, 4])
        C = rng.choice([1, 3, 4, 16])
        cin, k = rng.choice([8, 32, 64]), rng.choice([1, 3, 5])
        p = rng.choice([0, k // 2])
        H, W, N = rng.randint(k + 1, 30), rng.randint(k + 1, 30), rng.randint(1, 3)
        x = torch.randn(N, cin, H, W, generator=g)
        w = torch.randn(C * r * r, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        b = torch.randn(C * r * r, generator=g) * 0.1
        ref = F.pixel_shuffle(F.conv2d(x.double(), w.double(), b.double(), 1, p), r)
        res = torch.randn(ref.shape, generator=g) if rng.random() < 0.5 else None
        if res is not None:
            ref = ref + res.double()
        cfg = ops.ConvCfg(1, p, False, 0, 0, 0.0, r, 0)
        with torch.no_grad():
            y = ops.conv2d_infer(x.to(dev), w.to(dev), b.to(dev), None if res is None else res.to(dev), cfg)
        errs = (rel(y, ref),)
        tag = "conv+ps cin %d C %d r %d k %d p %d %dx%d N %d res %s" % (cin, C, r, k, p, H, W, N, res is not None)
    if os.environ.get("FUZZ_VERBOSE"): print(tag, flush=True)
    worst = max(worst, max(errs))
    if max(errs) > 2e-4 or any(e != e for e in errs):
        bad += 1
        print("BAD", tag, " ".join("%.2e" % e for e in errs))
print("cases %d, failures %d, worst relative error %.2e" % (cases, bad, worst))
sys.exit(1 if bad else 0)
