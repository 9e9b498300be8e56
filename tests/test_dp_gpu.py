"""Data-parallel equivalence on the GPU (SURVEY.md §4 tier 4): two ranks, each on its shard of the
global batch with the 1/world loss seed + SUM all-reduce of the flat gradient buffer, must reproduce
the single-process full-batch gradients and the same optimizer step.  The test box has one GPU, so
both ranks share device 0 and use the gloo backend (SRK_DIST_BACKEND / SRK_SINGLE_GPU test switches
of dp.init_from_env); the RCCL path differs only in the transport."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SRK_DIST_BACKEND="gloo", SRK_SINGLE_GPU="1")
    sys.path.insert(0, ROOT)
    import pytorch_super_resolution_model_collection_amd as pkg
    from oracle import fill
    r, w, local = pkg.dp.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    gb = 6
    x, t = fill.rand((gb, 3, 8, 8), 1), fill.rand((gb, 3, 32, 32), 2)

    def make():
        net = pkg.EDSRNet(3, 64, 4)
        fill.fill_module(net, 7, 0.5)
        net.to(dev).train()
        flat = pkg.optim.FlatParams(net)
        return net, flat, pkg.optim.make_optimizer("edsr", flat, 1e-3)

    # data-parallel step on this rank's shard
    net, flat, opt = make()
    dp = pkg.dp.DataParallel(flat, bucket_bytes=1 << 20)
    dp.broadcast_params()
    step = pkg.trainers.l1_step(net, opt, dp)
    lo, hi = pkg.dp.shard_range(gb, r, w)
    loss = step(x[lo:hi].to(dev), t[lo:hi].to(dev))
    g_dp, p_dp = flat.grad.clone().cpu(), flat.data.clone().cpu()
    # graph-captured DP step from the same start (second model)
    net2, flat2, opt2 = make()
    dp2 = pkg.dp.DataParallel(flat2)
    snap = flat2.data.clone()
    gstep = pkg.trainers.GraphedStep(net2, opt2, pkg.ops.l1_loss, (x[lo:hi].to(dev), t[lo:hi].to(dev)), dp=dp2,
                                     warmup=1)
    flat2.data.copy_(snap)
    opt2.exp_avg.zero_(); opt2.exp_avg_sq.zero_(); opt2.step_dev.zero_()
    gstep(x[lo:hi].to(dev), t[lo:hi].to(dev))
    p_graph = flat2.data.clone().cpu()
    res = {"g": g_dp, "p": p_dp, "p_graph": p_graph, "loss": float(dp.allreduce_scalar(loss.detach().clone()))}
    if r == 0:  # single-process reference on the full batch
        net1, flat1, opt1 = make()
        step1 = pkg.trainers.l1_step(net1, opt1, None)
        l1 = step1(x.to(dev), t.to(dev))
        res.update(g1=flat1.grad.clone().cpu(), p1=flat1.data.clone().cpu(), loss1=float(l1))
    torch.save(res, out % r)
    torch.distributed.destroy_process_group()


def test_dp_two_ranks_match_single_process(gpu, tmp_path):
    world, port = 2, 29700 + os.getpid() % 200
    out = str(tmp_path / "dp%d.pt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["p"], r1["p"])          # replicas stay identical
    scale = r0["g1"].abs().max()
    assert (r0["g"] - r0["g1"]).abs().max() <= 2e-5 * scale                          # mean of shard grads == full-batch grad
    assert (r0["p"] - r0["p1"]).abs().max() <= 1e-5 * r0["p1"].abs().max() + 2e-6   # same Adam step
    assert abs(r0["loss"] - r0["loss1"]) <= 1e-6 * abs(r0["loss1"]) + 1e-7
    assert (r0["p_graph"] - r0["p"]).abs().max() <= 1e-6 * r0["p"].abs().max() + 2e-6  # hipGraph DP step == eager DP step
