#!/usr/bin/env python3
"""Random 3x3 stride-1 weight-gradient problems through srk_conv2d_backward_weight with SRK_WGRAD_TR=1 and =0: wherever the
transpose-read kernel takes the problem its dw must equal k_wgrad_bf's bit for bit (same products, same order), db to fp32
summation noise, both against torch float64 at 1e-4.   python tools/fuzz_wgrad_tr.py [cases] [seed]"""
import ctypes, os, random, sys, torch
os.environ["SRK_ENV_LIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
L = pkg._lib
lib = L.load()
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
hit = bad = 0
for case in range(cases):
    cin = rng.choice([32, 64, 64, 96, 128]); cout = rng.choice([64, 64, 128, 192, 256])
    H = rng.randint(3, 70); W = rng.randint(3, 70); p = rng.choice([0, 1, 1, 2])
    if H + 2 * p < 3 or W + 2 * p < 3:
        continue
    OH, OW = H + 2 * p - 2, W + 2 * p - 2
    px = OH * OW
    N = max(1, min(64, rng.choice([1, 2, 4]) * (40000 // max(px, 1) + 1)))
    ps_r = 2 if (cout == 256 and rng.random() < 0.5) else 0
    use_mask = ps_r == 0 and rng.random() < 0.5
    bias = rng.random() < 0.7
    slope = rng.choice([0.0, 0.2])
    g = torch.Generator().manual_seed(1000 + case)
    x = torch.randn(N, cin, H, W, generator=g); dy = torch.randn(N, cout, OH, OW, generator=g)
    y = torch.randn(N, cout, OH, OW, generator=g) if use_mask else None
    dym = dy.double()
    if y is not None:
        dym = torch.where(y > 0, dym, dym * slope)
    wr = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    br = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.double(), wr, br, 1, p).backward(dym)
    if ps_r:
        dy = torch.nn.functional.pixel_shuffle(dy, ps_r)
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    xg, dyg, yg = cl(x), cl(dy), (None if y is None else cl(y))
    d = L.ConvDesc(N, H, W, cin, OH, OW, cout, 3, 3, 1, p, 0, 0, 0, 0, ps_r)
    out = {}
    for tag in ("1", "0"):
        os.environ["SRK_WGRAD_TR"] = tag
        ws = torch.empty(int(lib.srk_conv2d_backward_weight_workspace_bytes(ctypes.byref(d))) + 16, dtype=torch.uint8, device=dev)
        dw = torch.zeros(cout, cin, 3, 3, device=dev); db = torch.zeros(cout, device=dev) if bias else None
        m = L.BwdMask(None if yg is None else yg.data_ptr(), slope)
        L.check(lib.srk_conv2d_backward_weight(ctypes.byref(d), L.ptr(xg), L.ptr(dyg), ctypes.byref(m) if yg is not None else None,
                                               L.ptr(dw), L.ptr(db), 0.0, L.ptr(ws), ws.numel(), L.stream_ptr()), "wgrad")
        torch.cuda.synchronize()
        out[tag] = (dw, db, lib.srk_last_kernel_name().decode())
    tr = out["1"][2].startswith("k_wgrad_tr")
    hit += tr
    ew = float((out["1"][0].double().cpu() - wr.grad).abs().max() / wr.grad.abs().max())
    ok = ew < 1e-4 and (not tr or torch.equal(out["1"][0], out["0"][0]))
    if bias:
        eb = float((out["1"][1].double().cpu() - br.grad).abs().max() / br.grad.abs().max())
        ok = ok and eb < 2e-5
    if not ok:
        bad += 1
    print("%3d N %2d %3d->%3d %2dx%2d pad %d ps %d mask %d bias %d  %-22s err %.1e %s" % (case, N, cin, cout, H, W, p, ps_r, use_mask, bias,
                                                                                         out["1"][2], ew, "ok" if ok else "MISMATCH"))
print("cases on k_wgrad_tr: %d, mismatches: %d" % (hit, bad))
sys.exit(1 if bad else 0)
