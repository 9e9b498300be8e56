"""Where does the c5 discriminator's gradient error come from?  Replace one op family at a time by a torch fp64
evaluation (fp32 storage in between, exactly like the product) and print the worst per-tensor L2 error against the
fp64 oracle.  Debugging tool (GPU box): python tools/srgan_bisect.py"""
import os, sys, copy, torch
import torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pytorch_super_resolution_model_collection_amd as pkg
from oracle import fill, ref_modules as R
gpu = torch.device("cuda:0")
ops = pkg.ops
orig = {k: getattr(ops, k) for k in ("conv2d", "batch_norm", "linear", "activation")}


def conv2d_ref(x, weight, bias=None, residual=None, cfg=None, packed=None, res_box=None, add_box=None):
    cfg = cfg or ops.ConvCfg()
    xd, wd = x.double(), weight.double()
    bd = None if bias is None else bias.double()
    if cfg.transposed:
        y = F.conv_transpose2d(xd, wd, bd, cfg.stride, cfg.pad, cfg.out_pad)
    else:
        y = F.conv2d(xd, wd, bd, cfg.stride, cfg.pad)
    if cfg.act == 1:
        y = F.relu(y)
    elif cfg.act == 3:
        y = F.leaky_relu(y, cfg.slope)
    if cfg.ps_r > 1:
        y = F.pixel_shuffle(y, cfg.ps_r)
    if residual is not None:
        y = y + residual.double()
    return y.float().contiguous(memory_format=torch.channels_last)


def bn_ref(x, gamma, beta, rm, rv, training, momentum=0.1, eps=1e-5, sync_group=None, num_batches_tracked=None):
    xd = x.double()
    dims = [0, 2, 3] if x.dim() == 4 else [0]
    shape = [1, -1, 1, 1] if x.dim() == 4 else [1, -1]
    mean = xd.mean(dims)
    var = xd.var(dims, unbiased=False)
    y = (xd - mean.view(shape)) / torch.sqrt(var.view(shape) + eps) * gamma.double().view(shape) + beta.double().view(shape)
    return y.float()


def linear_ref(x, weight, bias=None, act=0, slope=0.0):
    y = F.linear(x.double(), weight.double(), None if bias is None else bias.double())
    if act == 3:
        y = F.leaky_relu(y, slope)
    elif act == 5:
        y = torch.sigmoid(y)
    elif act == 1:
        y = F.relu(y)
    return y.float()


def act_ref(x, kind, slope=0.0, prelu_w=None):
    if isinstance(kind, str) or kind is None:
        kind = ops.ACT_BY_NAME[kind]
    if kind == 0:
        return x
    if kind == 3:
        return F.leaky_relu(x.double(), slope).float()
    if kind == 1:
        return F.relu(x)
    if kind == 5:
        return torch.sigmoid(x.double()).float()
    return orig["activation"](x, kind, slope, prelu_w)


REPL = {"conv2d": conv2d_ref, "batch_norm": bn_ref, "linear": linear_ref, "activation": act_ref}
size, B = 128, 16
x = fill.rand((B, 3, size, size), 502)
oD = fill.fill_module(R.Discriminator(3, 64, size), 6, 1.0).train()
o64 = copy.deepcopy(oD).double().train()
oy = oD(x); g = fill.randn(tuple(oy.shape), 9) / oy.numel(); oy.backward(g)
o64(x.double()).backward(g.double())
gmax64 = max(float(q.grad.abs().max()) for q in o64.parameters())


def run(tag, repl):
    for k in orig:
        setattr(ops, k, REPL[k] if k in repl else orig[k])
    D = pkg.SRGANDiscriminator(3, 64, size)
    fill.fill_module(D, 6, 1.0)
    D.to(gpu).train()
    y = D(x.to(gpu))
    y.backward(g.to(gpu))
    worst, wo, name = 0.0, 0.0, ""
    rows = []
    for (n, p), (_, q), (_, r) in zip(D.named_parameters(), oD.named_parameters(), o64.named_parameters()):
        den = max(float(r.grad.norm()), 1e-3 * gmax64 * r.grad.numel() ** 0.5)
        ep = float((p.grad.detach().cpu().double() - r.grad).norm()) / den
        eo = float((q.grad.double() - r.grad).norm()) / den
        rows.append((n, ep, eo))
        if ep > worst:
            worst, name = ep, n
        wo = max(wo, eo)
    print("%-34s worst %.2e (%s)   torch-fp32 %.2e" % (tag, worst, name, wo), flush=True)
    if os.environ.get("ALL"):
        for n, ep, eo in rows:
            print("     %-32s %.2e  %.2e" % (n, ep, eo))
    for k in orig:
        setattr(ops, k, orig[k])


run("product", ())
run("bn -> fp64", ("batch_norm",))
run("conv -> fp64", ("conv2d",))
run("linear+act -> fp64", ("linear", "activation"))
run("bn+conv -> fp64", ("batch_norm", "conv2d"))
run("everything -> fp64", ("batch_norm", "conv2d", "linear", "activation"))
pkg.ops.set_precision("fp32")
run("product fp32 mode", ())
run("fp32 mode, bn -> fp64", ("batch_norm",))
