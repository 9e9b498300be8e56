import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pytorch_super_resolution_model_collection_amd as pkg
from oracle import fill, ref_modules as R
gpu = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "mixed"
pkg.ops.set_precision(mode)


def cmp(tag, net, ora, x, B):
    import copy
    net.to(gpu).train(); ora.train()
    ora64 = copy.deepcopy(ora).double().train()
    y = net(x.to(gpu)); oy = ora(x)
    g = fill.randn(tuple(oy.shape), 9) / oy.numel()
    y.backward(g.to(gpu)); oy.backward(g)
    oy64 = ora64(x.double()); oy64.backward(g.double())
    gmax64 = max(float(q.grad.abs().max()) for _, q in ora64.named_parameters())
    worst_p, worst_o = 0.0, 0.0
    for (n, p), (_, q), (_, r) in zip(net.named_parameters(), ora.named_parameters(), ora64.named_parameters()):
        den = max(float(r.grad.norm()), 1e-3 * gmax64 * r.grad.numel() ** 0.5)
        ep = float((p.grad.detach().cpu().double() - r.grad).norm()) / den
        eo = float((q.grad.double() - r.grad).norm()) / den
        if os.environ.get("ALL"):
            print("   vs fp64  %-32s product %.2e   torch-fp32 %.2e" % (n, ep, eo), flush=True)
        worst_p, worst_o = max(worst_p, ep), max(worst_o, eo)
    print(tag, "worst L2-rel error vs fp64: product %.2e, torch fp32 oracle %.2e" % (worst_p, worst_o), flush=True)
    print(tag, "fwd err %.2e" % (float((y.detach().cpu() - oy.detach()).abs().max()) / float(oy.abs().max())), flush=True)
    errs = []
    gmax = max(float(q.grad.abs().max()) for _, q in ora.named_parameters())
    for (n, p), (_, q) in zip(net.named_parameters(), ora.named_parameters()):
        a, b = p.grad.detach().cpu().double(), q.grad.double()
        errs.append((float((a - b).norm()) / max(float(b.norm()), 1e-3 * gmax * b.numel() ** 0.5), n))
    if os.environ.get("ALL"):
        for e, n in errs:
            print("   %-32s %.2e" % (n, e), flush=True)
    errs.sort(reverse=True)
    print(tag, " ".join("%s=%.2e" % (n, e) for e, n in errs[:5]), flush=True)


for size, B in [tuple(int(v) for v in os.environ.get('CASE', '128,16').split(','))]:
    D = pkg.SRGANDiscriminator(3, 64, size); oD = R.Discriminator(3, 64, size)
    fill.fill_module(D, 6, 1.0); fill.fill_module(oD, 6, 1.0)
    cmp("D %dx%d B=%d" % (size, size, B), D, oD, fill.rand((B, 3, size, size), 502), B)
