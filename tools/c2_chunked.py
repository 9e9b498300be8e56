"""Does running the c2 batch in sub-batches (so that the 64-channel intermediate of a sub-batch stays in the 256 MB
Infinity Cache between the layer that writes it and the layer that reads it) beat one launch per layer over all 64 images?
python tools/c2_chunked.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_super_resolution_model_collection_amd as pkg
dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = pkg.ESPCNNet(3, 64, 4); net.weight_init(); net.to(dev).eval()
x = torch.rand(64, 3, 256, 256, device=dev)

def run(chunk):
    outs = []
    with torch.no_grad():
        for i in range(0, 64, chunk):
            outs.append(net(x[i:i + chunk]))
    return outs

def timeit(fn, reps=10):
    for _ in range(3): fn()
    side = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            out = fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out

ref = None
for chunk in (64, 32, 16, 8, 4, 2):
    ms, out = timeit(lambda: run(chunk))
    y = torch.cat(out)
    if ref is None: ref = y
    print("chunk %2d: %.3f ms per 64 images  (%.0f images/s)  equal=%s" % (chunk, ms, 64 / ms * 1e3, torch.equal(y, ref)), flush=True)
