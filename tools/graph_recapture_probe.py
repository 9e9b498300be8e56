"""Probe: a train step captured as a hipGraph (trainers.GraphedFn), replayed, then captured AGAIN as a second graph.
Observed on MI355X / ROCm 7.0.2 / torch 2.10 with the LapSRN step: if the FIRST graph object is destroyed after the
second one was captured, the second graph's replays produce slightly different (run-to-run varying) parameter updates;
with the first graph kept alive, or destroyed BEFORE the second capture, replays are bit-equal to the eager steps
(`recap` vs `recap_keep` below; EDSR does not show it).  The package therefore never re-captures a live step
(trainers.AutoGraph: learning rates are device scalars) and destroys a graph before the next capture (GraphedStep.close).
   python tools/graph_recapture_probe.py          KIND=edsr python tools/graph_recapture_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_super_resolution_model_collection_amd as pkg
from oracle import fill
dev = torch.device("cuda:0")
KIND = os.environ.get("KIND", "lapsrn")
def make():
    if KIND == "lapsrn":
        net = pkg.LapSRNNet(3, 64, 3); fill.fill_module(net, 3, 0.5); net.to(dev).train()
        flat = pkg.optim.FlatParams(net); opt = pkg.optim.make_optimizer("lapsrn", flat, 1e-3)
        return net, flat, opt, pkg.trainers.lapsrn_step(net, opt, None)
    net = pkg.EDSRNet(3, 64, 2); fill.fill_module(net, 3, 0.5); net.to(dev).train()
    flat = pkg.optim.FlatParams(net); opt = pkg.optim.make_optimizer("edsr", flat, 1e-3)
    return net, flat, opt, pkg.trainers.l1_step(net, opt, None)
if KIND == "lapsrn":
    xs = [(fill.rand((2, 3, 8, 8), 10 + i).to(dev), fill.rand((2, 3, 16, 16), 20 + i).to(dev), fill.rand((2, 3, 32, 32), 30 + i).to(dev)) for i in range(8)]
else:
    xs = [(fill.rand((2, 3, 8, 8), 10 + i).to(dev), fill.rand((2, 3, 32, 32), 30 + i).to(dev)) for i in range(8)]
def run(mode):
    net, flat, opt, step = make(); out = []; g = None; keep = []
    for i, b in enumerate(xs):
        if mode == "eager" or i == 0:
            o = step(*b)
        else:
            if g is None or (mode.startswith("recap") and i == 4):
                if mode == "recap_keep": keep.append(g)
                if mode == "recap_eagerfirst" and i == 4:
                    o = step(*b); out.append(float(sum(o).detach()) if isinstance(o, tuple) else float(o.detach())); g = pkg.trainers.GraphedFn(step, xs[i], warmup=0, flats=[flat]); continue
                g = pkg.trainers.GraphedFn(step, b, warmup=0, flats=[flat])
            o = g(*b)
        out.append(float(sum(o).detach()) if isinstance(o, tuple) else float(o.detach()))
    return out
for m in ("eager", "graph", "recap", "recap_keep", "recap_eagerfirst"):
    print("%-17s" % m, ["%.6f" % v for v in run(m)])
