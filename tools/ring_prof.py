#!/usr/bin/env python3
"""Role-time sums inside k_conv_bfr (first producer wave: wait for a free slot / wait for its loads / split + commit / signal /
issue; consumer wave 0: wait for a full slot / the 18 steps / signal / parking) on the two 3x3 layers of c2.
Needs a -DBFR_PROF build:  EXTRA=-DBFR_PROF TAG=prof tools/ring_ablate.sh 0;  SRK_LIB_PATH=variants/ring_0prof.so python tools/ring_prof.py"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd import _lib
ops = pkg.ops
lib = _lib.load()
P = _lib.ptr
dev = torch.device("cuda:0")
lib.srk_debug_ring_prof.argtypes = [ctypes.c_void_p]
lib.srk_debug_ring_prof.restype = None
prof = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
net = pkg.ESPCNNet(3, 64, 4); net.weight_init(); net.to(dev).eval()
x = torch.rand(64, 3, 256, 256, device=dev)
vdsr = len(sys.argv) > 1 and sys.argv[1] == "vdsr"   # a VDSR body layer (64 -> 64 on 256 patches of 41 x 41: the canvas variant)
with torch.no_grad():
    hs, h = [], x
    for l in net.layers:
        hs.append(h); h = l(h)
    if vdsr:
        vnet = pkg.VDSRNet(3, 64, 18); vnet.weight_init(); vnet.to(dev).train()   # (the training forward: all 18 body layers;
        hs = {1: torch.rand(256, 3, 41, 41, device=dev)}                           #  the sums below are over their launches)
        net = type("N", (), {"layers": {1: vnet}})
    for i in ((1,) if vdsr else (1, 2)):
        l = net.layers[i]
        for _ in range(3): l(hs[i])
        torch.cuda.synchronize()
        prof.zero_()
        lib.srk_debug_ring_prof(P(prof))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10): l(hs[i])     # back to back: a lone launch behind an idle gap runs at a lower clock (DVFS ramp)
        e0.record(); l(hs[i]); e1.record(); torch.cuda.synchronize()
        lib.srk_debug_ring_prof(None)
        t = prof.view(-1, 16).cpu().double()
        t = t[t[:, 13] > 0]
        st = t[:, 13].mean()      # own stages of consumer wave 0 (half of the block's)
        print("layer %d %s: %.1f us with stamps, %d blocks, %.0f stages per group" % (i, lib.srk_last_kernel_name().decode(), e0.elapsed_time(e1) * 1e3, t.shape[0], st))
        ent, ext = t[:, 7], t[:, 15]      # block entry / consumer wave 0 exit, 100 MHz wall clock
        print("   wall clock (us): first block in -> last out %.1f; block start spread %.1f; entry -> loop %.1f (max %.1f); loop %.1f (max %.1f); loop end -> exit %.1f; last out - mean out %.1f"
              % ((ext.max() - ent.min()) / 100, (ent.max() - ent.min()) / 100, t[:, 6].mean() / 100, t[:, 6].max() / 100, t[:, 14].mean() / 100, t[:, 14].max() / 100,
                 (ext - ent - t[:, 6] - t[:, 14]).mean() / 100, (ext.max() - ext.mean()) / 100))
        full = prof.view(-1, 16)[:256].cpu().double()
        if full[:, 13].min() > 0:   # tile loop (wall us) by XCD (block & 7) and by position inside the XCD (block >> 3)
            lp = full[:, 14] / 100
            print("   tile loop by XCD: " + " ".join("%.1f" % lp[x::8].mean() for x in range(8)) +
                  "; by quarter of the XCD's blocks: " + " ".join("%.1f" % lp.view(32, 8)[i * 8:(i + 1) * 8].mean() for i in range(4)) +
                  "; sd inside an XCD %.1f" % lp.view(32, 8).std(0).mean())
        q = prof.view(-1, 16)[2048:2048 + 256].cpu().double()
        q = q[q[:, 1] > 0]
        print("   prologue split (us): arguments + input maximum %.2f, filter copy %.2f, consumer set-up %.2f, barrier %.2f"
              % (q[:, 0].mean() / 100, q[:, 1].mean() / 100, q[:, 2].mean() / 100, t[:, 6].mean() / 100 - q[:, :3].sum(1).mean() / 100))
        tp, tc = t[:, 5].mean(), t[:, 12].mean()
        names = ["wait free", "wait loads", "split+commit", "signal", "issue"]
        print("   producer (ticks per stage, %% of its loop): " + "  ".join("%s %.0f (%.0f %%)" % (n, t[:, j].mean() / (2 * st), 100 * t[:, j].mean() / tp) for j, n in enumerate(names)) + "   loop %.0f" % tp)
        print("   consumer loop: %.0f ticks in %.1f us of wall clock -> %.3f GHz" % (tc, t[:, 14].mean() / 100.0, tc / (t[:, 14].mean() / 100.0) / 1e3))
        names = ["wait full", "steps", "signal", "park"]
        print("   consumer (ticks per own stage, %% of its loop): " + "  ".join("%s %.0f (%.0f %%)" % (n, t[:, 8 + j].mean() / st, 100 * t[:, 8 + j].mean() / tc) for j, n in enumerate(names)) + "   loop %.0f" % tc)
