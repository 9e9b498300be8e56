import torch, time
dev = torch.device("cuda:0")
# a kernel that occupies few CUs for a while: elementwise chain on a small tensor, repeated
a = torch.rand(64, 1024, device=dev); b = torch.rand(64, 1024, device=dev)
def work(t):
    for _ in range(50): t = torch.sin(t) * 1.0001
    return t
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
s1 = torch.cuda.Stream(); cap = torch.cuda.Stream()
def seq():
    work(a); work(b)
def par():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    with torch.cuda.stream(s1): work(b)
    work(a)
    cur.wait_stream(s1)
g1 = torch.cuda.CUDAGraph(); g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    seq(); par()
    with torch.cuda.graph(g1, stream=cap): seq()
    with torch.cuda.graph(g2, stream=cap): par()
print("eager seq %.3f ms, eager par %.3f ms" % (timeit(seq), timeit(par)))
print("graph seq %.3f ms, graph par %.3f ms" % (timeit(g1.replay), timeit(g2.replay)))
