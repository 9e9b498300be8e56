"""Two ranks (gloo, one shared GPU) through the data-parallel hipGraph paths of bench.py at bench sizes; prints the first
Python exception instead of dying in a graph destructor.
SRK_SINGLE_GPU=1 SRK_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dp_graph_probe.py"""
import os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_super_resolution_model_collection_amd as pkg
rank, world, local = pkg.dp.init_from_env()
torch.cuda.set_device(local); dev = torch.device("cuda", local)
B = int(os.environ.get("B", "64"))
try:
    if os.environ.get("MODE", "edsr") == "srgan":
        G, D = pkg.SRGANGenerator(3, 64, 16), pkg.SRGANDiscriminator(3, 64, 128)
        torch.manual_seed(1234); G.weight_init(); D.weight_init(); G.to(dev).train(); D.to(dev).train()
        gflat, dflat = pkg.optim.FlatParams(G), pkg.optim.FlatParams(D)
        g_opt = pkg.optim.make_optimizer("srgan_g", gflat, 1e-4); d_opt = pkg.optim.make_optimizer("srgan_d", dflat, 1e-4)
        g_dp, d_dp = pkg.dp.DataParallel(gflat), pkg.dp.DataParallel(dflat)
        g_dp.broadcast_params(); d_dp.broadcast_params()
        lr_img = torch.rand(16, 3, 32, 32, device=dev); hr_img = torch.rand(16, 3, 128, 128, device=dev)
        sstep = pkg.trainers.GraphedSegments(pkg.trainers.srgan_segments(G, D, g_opt, d_opt, g_dp, d_dp), (lr_img, hr_img))
        for _ in range(3): out = sstep(lr_img, hr_img)
        torch.cuda.synchronize()
        print("rank", rank, "srgan graph-split DP step ok", [float(v) for v in out], flush=True)
        torch.distributed.barrier(); torch.distributed.destroy_process_group(); os._exit(0)
    net = pkg.EDSRNet(3, 64, 16); torch.manual_seed(1234); net.weight_init(); net.to(dev).train()
    flat = pkg.optim.FlatParams(net); opt = pkg.optim.make_optimizer("edsr", flat, 1e-5)
    dp = pkg.dp.DataParallel(flat); dp.broadcast_params()
    x = torch.rand(B, 3, 32, 32, device=dev); t = torch.rand(B, 3, 128, 128, device=dev)
    if os.environ.get("EAGER"):   # the same DP step without graphs (overlapped exchange, eager launches)
        estep = pkg.trainers.l1_step(net, opt, dp)
        class _S(object):
            loss = None
            def __call__(self, a, b):
                self.loss = estep(a, b)
                return self.loss
        step = _S()
    else:
        step = pkg.trainers.GraphedStep(net, opt, pkg.ops.l1_loss, (x, t), dp=dp, warmup=2)
    import time
    for _ in range(3): step(x, t)
    torch.cuda.synchronize()
    for i in range(int(os.environ.get("NSTEPS", "4"))):
        torch.distributed.barrier(); t0 = time.perf_counter(); step(x, t); torch.cuda.synchronize()
        print("rank", rank, "step %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
    if not os.environ.get("EAGER"): print("rank", rank, "sends per group:", [[(lo, hi - lo) for lo, hi in r] for _, _, _, r_all, _ in step.seg.plan for r in (r_all or [])], flush=True)
    print("rank", rank, "edsr graph-split DP step ok, loss", float(step.loss), flush=True)
except Exception:
    traceback.print_exc(); sys.stdout.flush(); sys.stderr.flush(); os._exit(1)
torch.distributed.barrier(); torch.distributed.destroy_process_group()
