#!/usr/bin/env python3
"""Role-time sums inside k_conv_bfw (producer wave 0: LDS commit / load issue / barrier wait; consumer wave 0: tap loop /
tile parking / barrier wait) for the two 3x3 layers of c2 (ESPCN x4, 64 x 256 x 256) and a VDSR body layer.
Needs a library built with SRK_BUILD_EXPERIMENTS=1.   python tools/bfw_prof.py"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd import _lib
ops = pkg.ops
lib = _lib.load()
P = _lib.ptr
dev = torch.device("cuda:0")
lib.srk_debug_bfw_prof.argtypes = [ctypes.c_void_p]
lib.srk_debug_bfw_prof.restype = None
prof = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
CL = torch.channels_last


def run(name, x, w, b, cfg):
    with torch.no_grad():
        for _ in range(3):
            y = ops.conv2d_infer(x, w, b, None, cfg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d_infer(x, w, b, None, cfg)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        prof.zero_()
        lib.srk_debug_bfw_prof(P(prof))
        ops.conv2d_infer(x, w, b, None, cfg)
        torch.cuda.synchronize()
        lib.srk_debug_bfw_prof(None)
    t = prof.view(-1, 16).cpu().double()
    t = t[t[:, 3] > 0]
    kern = lib.srk_last_kernel_name().decode()
    tot_p, tot_c = t[:, 4].mean(), t[:, 12].mean()
    print("%s: %s  %.3f ms per launch, %d blocks x %.0f stages; loop ticks: producer %.0f, consumer %.0f" % (
        name, kern, ms, t.shape[0], float(t[:, 3].mean()), float(tot_p), float(tot_c)))
    print("   producer wave: commit %4.1f %%  issue %4.1f %%  barrier wait %4.1f %%   (per stage: %.0f / %.0f / %.0f ticks)" % (
        100 * t[:, 0].mean() / tot_p, 100 * t[:, 1].mean() / tot_p, 100 * t[:, 2].mean() / tot_p,
        float((t[:, 0] / t[:, 3]).mean()), float((t[:, 1] / t[:, 3]).mean()), float((t[:, 2] / t[:, 3]).mean())))
    print("   consumer wave: taps   %4.1f %%  park  %4.1f %%  barrier wait %4.1f %%   (per stage: %.0f / %.0f / %.0f ticks)" % (
        100 * t[:, 8].mean() / tot_c, 100 * t[:, 9].mean() / tot_c, 100 * t[:, 10].mean() / tot_c,
        float((t[:, 8] / t[:, 11]).mean()), float((t[:, 9] / t[:, 11]).mean()), float((t[:, 10] / t[:, 11]).mean())))


RELU = 1
x1 = torch.rand(64, 64, 252, 252, device=dev).contiguous(memory_format=CL)
pkg.ops._tag_amax(x1, None)
run("c2 conv3 64->32 + ReLU", x1, torch.randn(32, 64, 3, 3, device=dev) * 0.02, torch.zeros(32, device=dev), ops.ConvCfg(1, 0, False, 0, RELU, 0.0, 0))
x2 = torch.rand(64, 32, 250, 250, device=dev).contiguous(memory_format=CL)
run("c2 conv3 32->48 + PS4", x2, torch.randn(48, 32, 3, 3, device=dev) * 0.02, torch.zeros(48, device=dev), ops.ConvCfg(1, 0, False, 0, 0, 0.0, 4))
x3 = torch.rand(256, 64, 41, 41, device=dev).contiguous(memory_format=CL)
run("VDSR body 64->64 + ReLU", x3, torch.randn(64, 64, 3, 3, device=dev) * 0.02, None, ops.ConvCfg(1, 1, False, 0, RELU, 0.0, 0))
