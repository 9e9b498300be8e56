import sys, torch, ctypes
sys.path.insert(0, "/root/repo")
import pytorch_super_resolution_model_collection_amd as pkg
from pytorch_super_resolution_model_collection_amd._lib import load, ptr, stream_ptr, check
lib = load(); dev = torch.device("cuda:0")
def timeit(fn, reps=40):
    for _ in range(3): fn()
    side = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3
for n in (1024, 65536, 4 << 20):
    a = torch.randn(n, device=dev); b = torch.randn(n, device=dev); o = torch.empty(n, device=dev)
    print("axpby n=%8d: %.2f us per launch" % (n, timeit(lambda: check(lib.srk_axpby(ptr(a), ptr(b), ptr(o), n, 1.0, 1.0, stream_ptr()), "axpby"))))
    print("torch add n=%8d: %.2f us per launch" % (n, timeit(lambda: torch.add(a, b, out=o))))
