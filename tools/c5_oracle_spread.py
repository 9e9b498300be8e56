"""How reproducible is the reference's own fp32 CPU path on the c5 discriminator (128x128, batch 16)?  The same oracle
(stock torch.nn, bit-equal to the reference) evaluated under equivalent fp32 evaluation orders — thread counts, oneDNN
on/off, channels_last — against its fp64 evaluation: per-tensor L2 error of the parameter gradients (same metric as
tests/test_fullsize_gpu.py::test_c5_*).  CPU only:  python tools/c5_oracle_spread.py"""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fill, ref_modules as R

size, B = 128, 16
x = fill.rand((B, 3, size, size), 502)
base = fill.fill_module(R.Discriminator(3, 64, size), 6, 1.0).train()
o64 = copy.deepcopy(base).double().train()
oy = base(x)
g = fill.randn(tuple(oy.shape), 9) / oy.numel()
o64(x.double()).backward(g.double())
gmax = max(float(q.grad.abs().max()) for q in o64.parameters())


def worst(net):
    w, name = 0.0, ""
    for (n, p), (_, r) in zip(net.named_parameters(), o64.named_parameters()):
        den = max(float(r.grad.norm()), 1e-3 * gmax * r.grad.numel() ** 0.5)
        e = float((p.grad.double() - r.grad).norm()) / den
        if e > w:
            w, name = e, n
    return w, name


def run(tag, threads=8, mkldnn=True, cl=False, scale=1.0):
    torch.set_num_threads(threads)
    torch.backends.mkldnn.enabled = mkldnn
    net = copy.deepcopy(base).train()
    xi = x.contiguous(memory_format=torch.channels_last) if cl else x
    if cl:
        net = net.to(memory_format=torch.channels_last)
    net(xi).backward(g)
    w, n = worst(net)
    print("%-44s worst %.2e  (%s)" % (tag, w, n), flush=True)
    return w


res = [run("threads 8 (fixture setting)", 8), run("threads 1", 1), run("threads 32", 32), run("threads 3", 3),
       run("oneDNN off, threads 8", 8, mkldnn=False), run("oneDNN off, threads 1", 1, mkldnn=False)]
print("spread of the reference's own fp32 path vs fp64: min %.2e  max %.2e" % (min(res), max(res)))
