#!/usr/bin/env python3
"""In-kernel phase times of the fused residual-block kernels (clock64 stamps of thread 0 of every block; needs a library
built with SRK_BUILD_EXPERIMENTS=1).   python tools/res2_prof.py [B] [TH]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2:
    os.environ["SRK_RES2_TH"] = sys.argv[2]  # (the half-tile variant was removed with the round-4 LDS layout)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd import _lib
ops = pkg.ops
lib = _lib.load()
P = _lib.ptr
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CL = torch.channels_last
lib.srk_debug_res2_prof.argtypes = [ctypes.c_void_p]
lib.srk_debug_res2_prof.restype = None
w1, w2 = torch.randn(64, 64, 3, 3, device=dev) * 0.02, torch.randn(64, 64, 3, 3, device=dev) * 0.02
b1, b2 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
wpf = [ops.pack_weight_fwd(w1, False, 0), ops.pack_weight_fwd(w2, False, 0)]
wpb = [ops.pack_weight_bwd(w1, False, 0), ops.pack_weight_bwd(w2, False, 0)]
x = torch.randn(B, 64, 32, 32, device=dev).contiguous(memory_format=CL)
mid, out, dmid, dx = [torch.empty_like(x) for _ in range(4)]
xa = ops.amax_of(x)
ya = torch.zeros(_lib.AMAX_FLOATS, device=dev)
nblk = 4096
prof = torch.zeros(nblk * 16, dtype=torch.int64, device=dev)
NAMES = ["filter prefetch + halo loads + split + LDS commit", "barrier", "conv1 taps", "gate/bias loads issued .. mid planes written (2 barriers)",
         "barrier", "conv2 taps", "exchange + barrier", "stores issued", "amax commit"]


def fwd():
    return lib.srk_resblock2_forward(B, 32, 32, 64, P(x), P(wpf[0]), P(b1), P(wpf[1]), P(b2), P(mid), P(out), _lib.ALGO_MFMA_F16X3,
                                     P(xa), P(ya), _lib.stream_ptr())


def bwd():
    return lib.srk_resblock2_backward_data(B, 32, 32, 64, P(x), P(wpb[1]), P(wpb[0]), P(mid), P(dmid), P(dx), 0, _lib.stream_ptr())


for name, fn in (("forward f16x3", fwd), ("backward bf16x3", bwd)):
    for _ in range(5):
        assert fn() == 0, lib.srk_last_error_string()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 20 * 1e3
    prof.zero_()
    lib.srk_debug_res2_prof(P(prof))
    fn()
    torch.cuda.synchronize()
    lib.srk_debug_res2_prof(None)
    t = prof.view(nblk, 16).cpu()
    used = t[:, 0] != 0
    t = t[used].double()
    n = t.shape[0]
    span = float(t[:, 9].max() - t[:, 0].min())
    print("%s (%s): %d blocks, %.2f us per launch back to back; stamps span %.0f ticks; block start spread %.0f ticks, "
          "mean block lifetime %.0f ticks" % (name, lib.srk_last_kernel_name().decode(), n, per, span,
                                                float(t[:, 0].max() - t[:, 0].min()), float((t[:, 9] - t[:, 0]).mean())))
    for i, nm in enumerate(NAMES):
        d = t[:, i + 1] - t[:, i]
        print("   %-70s mean %7.0f  min %7.0f  max %7.0f ticks" % (nm, float(d.mean()), float(d.min()), float(d.max())))
    sub = [("conv1 done -> barrier 1 passed", 3, 10), ("red written + barrier 2", 10, 11), ("red read, bias/gate, ReLU, mid stored", 11, 12),
           ("tile maximum (f16x3: wave max, barrier 3)", 12, 13), ("split + mid planes written, residual loads issued", 13, 4)]
    for nm, a, b in sub:
        d = t[:, b] - t[:, a]
        print("      . %-66s mean %7.0f  min %7.0f  max %7.0f ticks" % (nm, float(d.mean()), float(d.min()), float(d.max())))

