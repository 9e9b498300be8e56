#!/usr/bin/env python3
"""Randomised one-step training fuzz: loss and every parameter gradient of one training step of each trainable family at
random widths / depths / image sizes / batch sizes against the CPU oracle (stock torch.nn fp32, same parameters).
Gradients are compared per tensor in the L2 norm (a pre-activation within fp32 rounding of zero may fall on either side
in two fp32 implementations; the max norm would flag those single units).   python tools/fuzz_train.py [cases] [seed]"""
import os, random, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from oracle import fill, ref_modules as R
dev = torch.device("cuda:0")
if os.environ.get("FUZZ_PRECISION"):
    pkg.ops.set_precision(os.environ["FUZZ_PRECISION"])  # mixed | bf16x3 | fp32
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
F = torch.nn.functional
bad, worst = 0, 0.0
for i in range(cases):
    fam = rng.choice(["srcnn", "espcn", "fsrcnn", "vdsr", "edsr"])
    N, H, W = rng.randint(1, 6), rng.randint(20, 48), rng.randint(20, 48)
    loss_p, loss_o = pkg.ops.mse_loss, F.mse_loss
    if fam == "srcnn":
        a = (3, rng.choice([16, 32, 64])); prod, ora = pkg.SRCNNNet(*a), R.SRCNN(*a); oh, ow = H - 16, W - 16
    elif fam == "espcn":
        r = rng.choice([2, 3, 4]); a = (3, rng.choice([32, 64]), r); prod, ora = pkg.ESPCNNet(*a), R.ESPCN(*a); oh, ow = (H - 8) * r, (W - 8) * r
    elif fam == "fsrcnn":
        a = (3, 4, rng.choice([32, 56]), rng.choice([8, 12]), rng.choice([2, 4])); prod, ora = pkg.FSRCNNNet(*a), R.FSRCNN(*a); oh, ow = 4 * (H - 5) + 4, 4 * (W - 5) + 4
    elif fam == "vdsr":
        a = (3, rng.choice([32, 64]), rng.choice([2, 6])); prod, ora = pkg.VDSRNet(*a), R.VDSR(*a); oh, ow = H, W
    else:
        a = (3, rng.choice([32, 64]), rng.choice([2, 4])); prod, ora = pkg.EDSRNet(*a), R.EDSR(*a); oh, ow = 4 * H, 4 * W
        loss_p, loss_o = pkg.ops.l1_loss, F.l1_loss
    gain = 0.5 if fam == "edsr" else 1.0
    fill.fill_module(prod, 300 + i, gain)
    fill.fill_module(ora, 300 + i, gain)
    prod.to(dev).train(); ora.train()
    x, t = fill.rand((N, 3, H, W), 700 + i), fill.rand((N, 3, oh, ow), 800 + i)
    flat = pkg.optim.FlatParams(prod)
    flat.zero_grad()
    lp = loss_p(prod(x.to(dev)), t.to(dev)); lp.backward()
    ora.zero_grad(); lo = loss_o(ora(x), t); lo.backward()
    errs = [abs(float(lp.detach()) - float(lo.detach())) / max(abs(float(lo)), 1e-30)]
    og = dict((n, p.grad) for n, p in ora.named_parameters())
    gmax = max(float(g.norm()) / g.numel() ** 0.5 for g in og.values())
    for n, p in prod.named_parameters():
        g, o = p.grad.detach().cpu().double(), og[n].double()
        den = max(float(o.norm()), 1e-3 * gmax * o.numel() ** 0.5)
        errs.append(float((g - o).norm()) / den)
    e = max(errs)
    worst = max(worst, e)
    if e > 1e-3 or e != e:
        bad += 1
        print("BAD %s%s N %d %dx%d -> loss %.2e worst grad %.2e" % (fam, a, N, H, W, errs[0], max(errs[1:])))
print("cases %d, failures %d, worst relative error %.2e" % (cases, bad, worst))
sys.exit(1 if bad else 0)
