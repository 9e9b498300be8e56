#!/usr/bin/env python3
"""A few eager VDSR x4 training steps (c3: batch 256, 41x41) for rocprofv3 kernel traces."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = pkg.VDSRNet(3, 64, 18); net.weight_init(); net.to(dev).train()
flat = pkg.optim.FlatParams(net); opt = pkg.optim.make_optimizer("vdsr", flat, 1e-5)
x = torch.rand(B, 3, 41, 41, device=dev); t = torch.rand(B, 3, 41, 41, device=dev)
step = pkg.trainers.mse_step(net, opt, None, clip=0.4)
for _ in range(5): step(x, t)
torch.cuda.synchronize()
