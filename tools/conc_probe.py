#!/usr/bin/env python3
"""Does a latency-bound kernel chain (the fused residual blocks' data gradients of the 16-patch EDSR shard) overlap with a
throughput kernel (the grouped weight gradient of the same layers) when the two are replayed as separate hipGraphs on
two streams?   python tools/conc_probe.py [B]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd import _lib
ops = pkg.ops
lib = _lib.load()
P = _lib.ptr
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CL = torch.channels_last
NB = 16
w = [(torch.randn(64, 64, 3, 3, device=dev) * 0.02) for _ in range(2 * NB)]
wpb = [ops.pack_weight_bwd(t, False, 0) for t in w]
acts = [torch.randn(B, 64, 32, 32, device=dev).contiguous(memory_format=CL) for _ in range(NB + 1)]
mids = [torch.randn(B, 64, 32, 32, device=dev).contiguous(memory_format=CL) for _ in range(NB)]
dmids = [torch.empty_like(acts[0]) for _ in range(NB)]
dws = [torch.zeros(64, 64, 3, 3, device=dev) for _ in range(2 * NB)]
dbs = [torch.zeros(64, device=dev) for _ in range(2 * NB)]


def chain():
    for i in range(NB):
        rc = lib.srk_resblock2_backward_data(B, 32, 32, 64, P(acts[i]), P(wpb[2 * i + 1]), P(wpb[2 * i]), P(mids[i]), P(dmids[i]),
                                             P(acts[i + 1]), 0, _lib.stream_ptr())
        assert rc == 0, lib.srk_last_error_string()


cfg = ops.ConvCfg(1, 1, False, 0, 0, 0.0, 0)
d = ops._make_desc(acts[0].shape, w[0], cfg, "bwd")
recs = []
for i in range(NB):
    recs.append((None, d, mids[i], acts[i], None, 0.0, dws[2 * i + 1], dbs[2 * i + 1]))
    recs.append((None, d, acts[i + 1], dmids[i], None, 0.0, dws[2 * i], dbs[2 * i]))


def wgrads():
    ops.launch_wgrad_group(recs[:17])
    ops.launch_wgrad_group(recs[17:])


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    chain(); wgrads()
    torch.cuda.synchronize()
    gA, gB, gAB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gA, stream=side):
        chain()
    with torch.cuda.graph(gB, stream=side):
        wgrads()
    with torch.cuda.graph(gAB, stream=side):
        chain(); wgrads()
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def par():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): gA.replay()
    with torch.cuda.stream(s2): gB.replay()
    cur.wait_stream(s1); cur.wait_stream(s2)


print("B=%d  chain alone %.1f us, wgrads alone %.1f us, one graph (sequential) %.1f us, two graphs back to back %.1f us, "
      "two graphs on two streams %.1f us" % (B, timeit(gA.replay), timeit(gB.replay), timeit(gAB.replay),
                                              timeit(lambda: (gA.replay(), gB.replay())), timeit(par)))
