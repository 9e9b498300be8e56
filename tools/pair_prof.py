#!/usr/bin/env python3
"""Role-time sums of the fused pair kernel (k_conv_bfr<..,fused>): first producer wave {wait free, rendezvous waits, K steps,
epilogue + split + write, staging}, consumer wave 0 {steps, park}.  Needs a -DBFR_PROF build (tools/ring_ablate.sh)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd import _lib, base_networks as bn
ops = pkg.ops; lib = _lib.load(); dev = torch.device("cuda:0")
lib.srk_debug_ring_prof.argtypes = [ctypes.c_void_p]; lib.srk_debug_ring_prof.restype = None
prof = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
net = pkg.ESPCNNet(3, 64, 4); net.weight_init(); net.to(dev).eval()
x = torch.rand(64, 3, 256, 256, device=dev)
ops.PAIR = "1"
with torch.no_grad():
    for _ in range(5): bn.fused_conv_pair(net.layers[0], net.layers[1], x)
    torch.cuda.synchronize(); prof.zero_(); lib.srk_debug_ring_prof(_lib.ptr(prof))
    for _ in range(5): bn.fused_conv_pair(net.layers[0], net.layers[1], x)
    torch.cuda.synchronize(); lib.srk_debug_ring_prof(None)
t = prof.view(-1, 16).cpu().double(); t = t[t[:, 13] > 0]
tiles = t[:, 13].mean()   # own stages of consumer wave 0 = tiles of the block (2 chunks x half the tiles)
tp, tc = t[:, 5].mean(), t[:, 12].mean()
print("%s: %d blocks, %.0f tiles per block; consumer loop %.0f ticks in %.1f us -> %.3f GHz" % (lib.srk_last_kernel_name().decode(), t.shape[0], tiles, tc, t[:, 14].mean() / 100, tc / (t[:, 14].mean() / 100) / 1e3))
names = ["wait free", "rendezvous", "K steps", "epilogue+split+write", "staging"]
print("   producer (ticks per tile, % of loop): " + "  ".join("%s %.0f (%.0f %%)" % (n, t[:, j].mean() / tiles, 100 * t[:, j].mean() / tp) for j, n in enumerate(names)) + "   loop %.0f" % tp)
print("   consumer (ticks per own stage): steps %.0f (%.0f %%)  park %.0f per stage (%.0f %%)" % (t[:, 9].mean() / tiles, 100 * t[:, 9].mean() / tc, t[:, 11].mean() / tiles, 100 * t[:, 11].mean() / tc))
