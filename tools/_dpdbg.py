import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_super_resolution_model_collection_amd as pkg
rank, world, local = pkg.dp.init_from_env()
torch.cuda.set_device(local); dev = torch.device("cuda", local)
B = 64
net = pkg.EDSRNet(3, 64, 16); torch.manual_seed(1234); net.weight_init(); net.to(dev).train()
flat = pkg.optim.FlatParams(net); opt = pkg.optim.make_optimizer("edsr", flat, 1e-5)
dp = pkg.dp.DataParallel(flat); dp.broadcast_params()
x = torch.rand(B, 3, 32, 32, device=dev); t = torch.rand(B, 3, 128, 128, device=dev)
step = pkg.trainers.GraphedStep(net, opt, pkg.ops.l1_loss, (x, t), dp=dp, warmup=2)
seg = step.seg
mode = os.environ.get("DBG", "phases")
for it in range(14):
    torch.distributed.barrier(); torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for g, d, wgraphs, sends, _ in seg.plan:
        g.replay()
        if mode == "phases": torch.cuda.synchronize()
        ts.append(time.perf_counter())
        if d is not None and d.world > 1:
            works = []
            if not wgraphs:
                d.send(sends[0] if sends else [(0, d.flat.grad.numel())], works)
            for wg, ranges in zip(wgraphs, sends):
                wg.replay()
                if mode == "phases": torch.cuda.synchronize()
                ts.append(time.perf_counter())
                d.send(ranges, works)
                ts.append(time.perf_counter())
            for w in works:
                w.wait()
            ts.append(time.perf_counter())
            if mode == "phases": torch.cuda.synchronize()
            ts.append(time.perf_counter())
    torch.cuda.synchronize(); ts.append(time.perf_counter())
    if rank == 0:
        print("it %2d total %8.2f ms  phases(ms): %s" % (it, (ts[-1] - ts[0]) * 1e3, " ".join("%.2f" % ((b - a) * 1e3) for a, b in zip(ts, ts[1:]))), flush=True)
torch.distributed.barrier(); torch.distributed.destroy_process_group()
