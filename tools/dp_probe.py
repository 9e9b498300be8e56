import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pytorch_super_resolution_model_collection_amd as pkg
rank, world, local = pkg.dp.init_from_env()
torch.cuda.set_device(local); dev = torch.device("cuda", local)
for tag, B in (("first", 64), ("second", 64)):
    net = pkg.EDSRNet(3, 64, 16); torch.manual_seed(1234); net.weight_init(); net.to(dev).train()
    flat = pkg.optim.FlatParams(net); opt = pkg.optim.make_optimizer("edsr", flat, 1e-5)
    dp = pkg.dp.DataParallel(flat); dp.broadcast_params()
    x = torch.rand(B, 3, 32, 32, device=dev); t = torch.rand(B, 3, 128, 128, device=dev)
    step = pkg.trainers.GraphedStep(net, opt, pkg.ops.l1_loss, (x, t), dp=dp, warmup=2)
    ts = []
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(x, t); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    if rank == 0: print(tag, " ".join("%.1f" % v for v in ts), flush=True)
    # time the all-reduce alone
    torch.cuda.synchronize(); t0 = time.perf_counter(); dp.allreduce_grads(); torch.cuda.synchronize()
    if rank == 0: print(tag, "allreduce alone %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
torch.distributed.barrier(); torch.distributed.destroy_process_group()
