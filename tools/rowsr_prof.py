#!/usr/bin/env python3
"""Phase clocks of k_conv_rowsr on the c2 first layer (a library built with -DSRK_ROWSR_PROF: conv_rowsw.hip only):
shader-clock sums per wave over its stages -- 0 matrix loop (+ deferred stores), 1 commit (incl. the wait for the staged
pixels), 2 issue of the next loads, 3 parking arithmetic, 4 barrier.   SRK_LIB_PATH=variants/rowsw_prof.so python tools/rowsr_prof.py"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
lib = pkg._lib.load()
net = pkg.ESPCNNet(3, 64, 4); net.weight_init(); net.to(dev).eval()
x = torch.rand(64, 3, 256, 256, device=dev)
prof = torch.zeros(2 * 512 * 4 * 8, dtype=torch.int32, device=dev)
with torch.no_grad():
    for _ in range(3): net.layers[0](x)
    torch.cuda.synchronize()
    lib.srk_debug_rowsr_prof.argtypes = [ctypes.c_void_p]
    lib.srk_debug_rowsr_prof(ctypes.c_void_p(prof.data_ptr()))
    net.layers[0](x)
    torch.cuda.synchronize()
    lib.srk_debug_rowsr_prof(ctypes.c_void_p(0))
print(lib.srk_last_kernel_name().decode())
p = prof[:512 * 32].view(512, 4, 8).cpu().double()
g = prof[512 * 32:].view(512, 4, 8).cpu().double()
S = p[:, :, 5]
names = ["matrix loop + stores", "commit (wait for pixels)", "issue loads", "park", "barrier"]
tot = p[:, :, :5].sum(-1)
print("stages per block: min %d max %d; clocks per stage (mean over waves): %.0f" % (S.min(), S.max(), (tot / S.clamp(min=1)).mean()))
for w in range(4):
    print("wave %d: " % w + "  ".join("%s %.0f (%.0f %%)" % (names[i], (p[:, w, i] / S[:, w].clamp(min=1)).mean(), 100 * p[:, w, i].sum() / tot[:, w].sum()) for i in range(5)))
if g.sum() > 0:
    print("matrix loop by row group R (clocks per stage, wave 0): " + "  ".join("R%d %.0f" % (i, (g[:, 0, i] / S[:, 0].clamp(min=1)).mean()) for i in range(8)))
