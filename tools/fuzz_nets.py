#!/usr/bin/env python3
"""Randomised whole-network fuzz: inference output of every model family at random widths / depths / scales / image
sizes / batch sizes against the CPU oracle modules (stock torch.nn, fp32) with identical parameters.
   python tools/fuzz_nets.py [cases] [seed]"""
import os, random, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from oracle import fill, ref_modules as R
dev = torch.device("cuda:0")
if os.environ.get("FUZZ_PRECISION"):
    pkg.ops.set_precision(os.environ["FUZZ_PRECISION"])  # mixed | bf16x3 | fp32
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad, worst = 0, 0.0
for i in range(cases):
    fam = rng.choice(["srcnn", "espcn", "fsrcnn", "vdsr", "edsr", "lapsrn", "srgan_g"])
    N = rng.randint(1, 5)
    H, W = rng.randint(20, 72), rng.randint(20, 72)
    if fam == "srcnn":
        bf = rng.choice([16, 32, 64]); a = (3, bf); prod, ora = pkg.SRCNNNet(*a), R.SRCNN(*a)
    elif fam == "espcn":
        a = (3, rng.choice([32, 64]), rng.choice([2, 3, 4])); prod, ora = pkg.ESPCNNet(*a), R.ESPCN(*a)
    elif fam == "fsrcnn":
        a = (3, rng.choice([2, 3, 4]), rng.choice([32, 56]), rng.choice([8, 12]), rng.choice([2, 4])); prod, ora = pkg.FSRCNNNet(*a), R.FSRCNN(*a)
    elif fam == "vdsr":
        a = (3, rng.choice([32, 64]), rng.choice([2, 6, 18])); prod, ora = pkg.VDSRNet(*a), R.VDSR(*a)
    elif fam == "edsr":
        a = (3, rng.choice([32, 64]), rng.choice([2, 8, 16])); prod, ora = pkg.EDSRNet(*a), R.EDSR(*a); H, W = min(H, 48), min(W, 48)
    elif fam == "lapsrn":
        a = (3, rng.choice([32, 64]), rng.choice([2, 5, 10])); prod, ora = pkg.LapSRNNet(*a), R.LapSRN(*a); H, W = min(H, 40), min(W, 40)
    else:
        a = (3, rng.choice([32, 64]), rng.choice([2, 8])); prod, ora = pkg.SRGANGenerator(*a), R.Generator(*a); H, W = min(H, 40), min(W, 40)
    gain = 0.5 if fam in ("edsr", "srgan_g") else 1.0
    fill.fill_module(prod, 100 + i, gain)
    fill.fill_module(ora, 100 + i, gain)
    prod.to(dev).eval(); ora.eval()
    x = fill.rand((N, 3, H, W), 900 + i)
    with torch.no_grad():
        y = prod(x.to(dev)); r = ora(x)
    ys = y if isinstance(y, (tuple, list)) else (y,)
    rs = r if isinstance(r, (tuple, list)) else (r,)
    err = max(float((a_.cpu().double() - b_.double()).abs().max() / max(float(b_.abs().max()), 1e-30)) for a_, b_ in zip(ys, rs))
    worst = max(worst, err)
    if err > 5e-4 or err != err:
        bad += 1
        print("BAD %s%s N %d %dx%d -> %.2e" % (fam, a, N, H, W, err))
print("cases %d, failures %d, worst relative error %.2e" % (cases, bad, worst))
sys.exit(1 if bad else 0)
