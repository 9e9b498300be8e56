"""Where does an occasional ~80 ms data-parallel step come from (VERDICT r05 task 1d)?  One rank over backend "nccl"
(SRK_DP_FORCE_COMM=1), the EDSR step of tests/test_dp_gpu.py::_rccl_one_rank_worker, per-step wall times for
plain / dp_eager / dp_graph, with every candidate timed beside it: cyclic-GC pauses (gc.callbacks), caching-allocator
hipMallocs (torch.cuda.memory_stats num_device_alloc), and RCCL's own lazy work (NCCL_DEBUG=INFO lines, time-stamped by
arrival on stderr).  Usage: python tools/dp_step_times.py [steps]"""
import gc
import os
import sys
import time

os.environ.setdefault("SRK_DP_FORCE_COMM", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import socket  # noqa: E402
with socket.socket() as _s:
    _s.bind(("127.0.0.1", 0))
    os.environ["MASTER_PORT"] = str(_s.getsockname()[1])

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import pytorch_super_resolution_model_collection_amd as pkg  # noqa: E402
from oracle import fill  # noqa: E402  (tool, not product)

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x, t = fill.rand((16, 3, 32, 32), 1).to(dev), fill.rand((16, 3, 128, 128), 2).to(dev)

gc_log = []
_gc_t0 = [0.0]


def _gc_cb(phase, info):
    if phase == "start":
        _gc_t0[0] = time.perf_counter()
    else:
        gc_log.append((info["generation"], time.perf_counter() - _gc_t0[0], info["collected"]))


gc.callbacks.append(_gc_cb)


def make(use_dp):
    net = pkg.EDSRNet(3, 64, 4)
    fill.fill_module(net, 7, 0.5)
    net.to(dev).train()
    flat = pkg.optim.FlatParams(net)
    opt = pkg.optim.make_optimizer("edsr", flat, 1e-4)
    dp = pkg.dp.DataParallel(flat, bucket_bytes=256 << 10) if use_dp else None
    if dp is not None:
        dp.broadcast_params()
    return net, flat, opt, dp


for name, use_dp, graphed in (("plain", False, False), ("dp_eager", True, False), ("dp_graph", True, True)):
    net, flat, opt, dp = make(use_dp)
    if graphed:
        step = pkg.trainers.GraphedStep(net, opt, pkg.ops.l1_loss, (x, t), dp=dp, warmup=0)
    else:
        step = pkg.trainers.l1_step(net, opt, dp)
    rows = []
    for i in range(steps):
        torch.cuda.synchronize()
        n_gc = len(gc_log)
        a0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
        t0 = time.perf_counter()
        loss = step(x, t)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        float(loss.detach())
        rows.append((t2 - t0, t1 - t0, gc_log[n_gc:], torch.cuda.memory_stats().get("num_device_alloc", 0) - a0))
    print("== %s: step ms (host-issue ms) [gc gen:ms] {hipMallocs}" % name)
    for i, (tot, host, gcs, na) in enumerate(rows):
        flag = "  <-- slow" if i >= 2 and tot > 5 * sorted(r[0] for r in rows[2:])[len(rows[2:]) // 2] else ""
        print("  %2d  %8.2f (%7.2f) %s {%d}%s" % (i, 1e3 * tot, 1e3 * host,
                                                " ".join("[g%d:%.1f]" % (g, 1e3 * d) for g, d, _ in gcs), na, flag))
    sys.stdout.flush()
    if graphed:
        step.close()
dist.barrier()
dist.destroy_process_group()

# hypothesis check: what ONE full (generation-2) collection of this process's heap costs -- an automatic one lands inside
# whichever step crosses the threshold
t0 = time.perf_counter()
n = gc.collect()
print("full gc.collect(): %.1f ms, %d unreachable, %d tracked objects" % (1e3 * (time.perf_counter() - t0), n, len(gc.get_objects())))
t0 = time.perf_counter()
gc.collect()
print("second full gc.collect(): %.1f ms" % (1e3 * (time.perf_counter() - t0)))
gc.freeze()
t0 = time.perf_counter()
gc.collect()
print("after gc.freeze(): %.1f ms" % (1e3 * (time.perf_counter() - t0)))
