"""Time the EDSR x4 train step one DP rank runs under strong scaling (global batch 128 / N ranks): hipGraph-captured
zero_grad + pack + forward + L1 + backward + Adam, B patches of 32x32 -> 128x128.   python tools/shard_step.py [B] [iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
net = pkg.EDSRNet(3, 64, 16); net.weight_init(); net.to(dev).train()
flat, opt, dp, step = pkg.trainers.build("edsr", net, 1e-5)
x = torch.rand(B, 3, 32, 32, device=dev); t = torch.rand(B, 3, 128, 128, device=dev)
g = pkg.trainers.GraphedStep(net, opt, pkg.ops.l1_loss, (x, t))
for _ in range(10): g(x, t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    e0.record()
    for _ in range(iters): g(x, t)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / iters)
print("edsr x4 train step B=%d: %.3f ms/step (graph), %.1f patches/s, loss %.5f" % (B, best, B / best * 1e3, float(g.loss)))
