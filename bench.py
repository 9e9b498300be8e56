#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): LR->HR images/s of ESPCN x4 inference (config c2:
256x256 LR, batch 64 per GPU, fp32) on N MI355X GPUs, printed as ONE JSON line by rank 0.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one forward pass of the hot path over one synthetic batch that is already resident in
HBM.  Inference replicas share nothing, so N GPUs = N independent shards of the image stream
(weak scaling, no data-path collective).  The same JSON line also carries
  * roofline      — the dominant kernel (conv 64->32, 3x3) timed live with HIP events on the launch
                    stream, against the fp32 MFMA peak, plus the whole-net HBM-roofline fraction
                    the north star quotes (61.11 MB compulsory traffic per image);
  * cpu_baseline  — the oracle (stock torch.nn = the arithmetic the reference invokes) timed on this
                    box's host cores on a bounded sample of the same workload;
  * extra         — training throughput of configs c3 (VDSR, 1 GPU) and c4 (EDSR, global batch 128
                    sharded over the N ranks with the RCCL gradient all-reduce; strong scaling).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL (see dp.py); before HIP initialises

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2 peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
BF16X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0  # 3 bf16 MFMAs per fp32-equivalent product
ESPCN_BYTES_PER_IMG = 61.11e6  # per-layer compulsory activation traffic, 256x256 LR
ESPCN_FLOP_PER_IMG = 4.614e9


DTYPE_NOTE = {
    "mixed": "f32 storage/accumulate; fp32-faithful products on the fp16 matrix cores: operands scaled by exact powers of "
             "two, split into two fp16 planes, 3 MFMAs per product (f16x3, ~3e-7 rms vs fp64, like fp32 MFMA)",
    "bf16x6": "f32 storage/accumulate; fp32-faithful products (f16x3 / bf16x6 operand splits, ~3e-7 rms)",
    "bf16x3": "f32 storage/accumulate; products by 3-term bf16 split on bf16 MFMA (bf16x3, ~5e-6 rel)",
    "fp32": "f32"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="LR images per GPU per step (c2: 64)")
    ap.add_argument("--lr-size", type=int, default=256)
    ap.add_argument("--no-extra", action="store_true", help="skip the c3/c4 training side metrics")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra-steps", type=int, default=20, help="timed steps of every side metric (after 5 warm-ups)")
    ap.add_argument("--precision", default="mixed", choices=["mixed", "bf16x3", "bf16x6", "fp32"],
                    help="conv arithmetic (ops.set_precision); inference uses bf16x3 unless 'fp32'")
    return ap.parse_args()


_JSON_FD = [None]


def claim_stdout():
    """stdout carries ONE JSON line and nothing else: RCCL prints a version banner through C stdio (which, with stdout a
    pipe, would surface after the line, at exit), libraries may print more.  File descriptor 1 is pointed at stderr for
    the whole run and the line is written to a private duplicate of the original stdout."""
    if _JSON_FD[0] is None:
        sys.stdout.flush()
        _JSON_FD[0] = os.dup(1)
        os.dup2(2, 1)


def emit_json(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD[0] is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD[0], line)


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1 or torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize()


def max_over_ranks(seconds, world, dev):
    if world == 1 and not torch.distributed.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def span_over_ranks(seconds, world, dev):
    """(min, max) over ranks of a per-rank time -- how far apart the ranks finish the same timed region."""
    if world == 1 and not torch.distributed.is_initialized():
        return seconds, seconds
    t = torch.tensor([seconds, -seconds], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(-t[1].item()), float(t[0].item())


def rank_local_steps(fn, steps, warmup, dev):
    """Per-rank time of `steps` calls with NO barrier on either side (device-synchronised only): what this rank's GPU
    needs on its own, to be compared over ranks."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def rccl_ranks_seen(dev):
    """SUM all-reduce of a ones tensor: the number of ranks the collective library really connected (WORLD_SIZE is only
    what the launcher claimed).  None without a process group."""
    if not torch.distributed.is_initialized():
        return None
    one = torch.ones(1, dtype=torch.float32, device=dev)
    torch.distributed.all_reduce(one, op=torch.distributed.ReduceOp.SUM)
    return int(round(float(one.item())))


def time_steps(fn, steps, warmup, world, dev):
    for _ in range(warmup):
        fn()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier_sync(world)
    return max_over_ranks(time.perf_counter() - t0, world, dev)


def espcn_layer_events(net, x, steps, discard=0):
    """HIP events on the launch stream around each of the three fused kernels (conv5+ReLU,
    conv3+ReLU, conv3+pixel-shuffle store). Returns average ms per layer over the last `steps` of
    `discard + steps` forwards (the first ones run while the clock governor ramps up from idle)."""
    import pytorch_super_resolution_model_collection_amd as pkg
    lib = pkg._lib.load()
    batches = x if isinstance(x, (list, tuple)) else [x]   # the first layer reads the NCHW batch in place, exactly as net(x) does
    evs, names = [], ["", "", ""]
    with torch.no_grad():
        for k in range(discard + steps):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            h = batches[k % len(batches)]
            h = h.view(h.shape)   # (a new tensor object, as in the timed loop: the first bracket holds the |x| maximum pass too)
            e[0].record()
            for i, layer in enumerate(net.layers):
                h = layer(h)
                names[i] = lib.srk_last_kernel_name().decode()   # the kernel the dispatcher actually launched
                e[i + 1].record()
            evs.append(e)
    torch.cuda.synchronize()
    evs = evs[discard:]
    return [sum(e[i].elapsed_time(e[i + 1]) for e in evs) / len(evs) for i in range(3)], names


def pmc_traffic(kernel_label, batch, lr_size):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_c2_pmc_traffic.json,
    written by tools/profile_round.sh + tools/summarize_prof.py: rocprofv3 counters cannot be collected from
    inside this process).  Only valid for the default c2 shape the passes were taken on."""
    import glob
    if batch != 64 or lr_size != 256:
        return None, None
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_c2_pmc_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        kernels = json.load(fh).get("kernels", {})
    def targs(name):
        """(kernel, leading integer template arguments, f16 flag).  The dispatcher's label carries the integers and an
        "f16" marker; rocprofv3 prints every template argument (the f16 flag is the last one of these kernels)."""
        name = name.replace(" ", "")
        i = name.find("<")
        if i < 0:
            return name.split("(")[0].split("::")[-1], [], False, False
        j = name.find(">", i)
        base = name[:i].split("::")[-1].replace("void", "")
        args = name[i + 1:j].split(",")
        ints = []
        for a in args:
            if not a.lstrip("-").isdigit():
                break
            ints.append(a)
        rest = args[len(ints):]
        if "f16" in rest or "mask" in rest or not rest:          # the dispatcher's label
            return base, ints, "f16" in rest, "mask" in rest
        if base == "k_conv_bfw":                                  # <NTW, TT, MTW, F16, MASK, OMASK>
            return base, ints, rest[0] == "true", len(rest) > 1 and rest[1] == "true"
        if base == "k_conv_bf3_rows":                             # <NT, VEC_ONLY, F16>
            return base, ints, rest[-1] == "true", False
        return base, ints, rest[0] == "true", False               # k_conv_rowsw<NTW, QT, F16>, ...

    want = targs(kernel_label.split(" ")[0])
    for name, rec in kernels.items():
        if rec.get("hbm_bytes") and targs(name) == want:
            return int(rec["hbm_bytes"]), os.path.join("profiles", os.path.basename(files[-1]))
    return None, None


def training_roofline(tag, px, layer_px_note, layers_per_step=1, steps=5):
    """extra.<tag>_roofline: the three kernels that carry a 64 -> 64 3x3 body layer of a training config (forward, data
    gradient, weight gradient) from the COMMITTED rocprofv3 passes of this round (profiles/*_<tag>_kernel_stats.csv and
    *_<tag>_pmc_traffic.json, tools/profile_round.sh: the counters cannot be collected from inside this process):
    algorithmic bytes per launch (read each operand once, write the result once), HBM bytes the counters saw
    (FETCH_SIZE x 2 + WRITE_SIZE per the gfx950 note of MI355X_MICROARCH.md), their ratio, and the achieved fraction of the
    matrix peak (bf16x3 / f16x3: 2.5 PF / 3 MFMAs per product) from the average kernel time of the same run."""
    import csv
    import glob
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    stats = sorted(glob.glob(os.path.join(here, "*_%s_kernel_stats.csv" % tag)))
    traffic = sorted(glob.glob(os.path.join(here, "*_%s_pmc_traffic.json" % tag)))
    if not stats or not traffic:
        return None
    with open(traffic[-1]) as fh:
        hbm = json.load(fh).get("kernels", {})
    with open(stats[-1]) as fh:
        rows = list(csv.DictReader(fh))
    t = 4.0 * 64 * px                     # one 64-channel fp32 tensor of the layer
    flop = 2.0 * px * 64 * 64 * 9
    # (role, kernel-name keys, algorithmic bytes per LAYER, operands, layers' worth of FLOPs per launch or None = from calls)
    # (k_conv_bfr<2, 2, F16, true, OMASK>: the ring kernel's canvas variant, round 6 -- VDSR's body layers)
    roles = [("forward", ("k_conv_bfr<2, 2, true, true, false", "k_conv_bfw<2, 9, 2, true, false", "k_conv_bfd<2, 2, 2, 2", "k_conv_bfd<2, 2, 2, 3",
                          "k_conv_bfd<4, 4, 1, 2"),
              2 * t, "x, y", 1.0),
             ("data_gradient", ("k_conv_bfr<2, 2, false, true, ", "k_conv_bfw<2, 9, 2, false, ", "k_conv_bf3<4, 4, true>"), 3 * t,
              "dy, one activation (mask of dy, or of dx for the layer below), dx", 1.0),
             ("weight_gradient", ("k_wgrad_tr<", "k_wgrad_bf<2, 2, 2, true"), 3 * t, "x, dy, activation mask", None),
             # residual blocks fused per tile (conv -> ReLU -> conv -> + skip in one launch): useful work of two layers
             ("fused_block_forward", ("k_res2<2, false", "k_res2<3, false"), 3 * t, "x, intermediate, y", 2.0),
             ("fused_block_data_gradient", ("k_res2<2, true",), 4 * t, "dy, saved intermediate, its gradient, dx", 2.0)]
    if any("k_conv_bfw<2, 9, 2, false, false, true" in r["Name"] or "k_conv_bfr<2, 2, false, true, true" in r["Name"] for r in rows):
        # pre-masked gradients (ops.PREMASK): dy arrives already multiplied by this layer's ReLU gradient
        roles[2] = ("weight_gradient", ("k_wgrad_tr<", "k_wgrad_bf<2, 2, 2, true"), 2 * t, "x, dy (pre-masked by the data gradient above)", None)
    if any("k_res2<" in r["Name"] for r in rows):   # body layers run fused per block: no stand-alone forward / data gradient
        roles = [r for r in roles if r[0] not in ("forward", "data_gradient")]
    out = {"from_profiles": True, "live": False,
           "source": [os.path.basename(stats[-1]), os.path.basename(traffic[-1])], "layer": layer_px_note, "kernels": {}}
    for role, keys, alg, what, nlay in roles:
        best = None
        for r in rows:
            if any(k in r["Name"] for k in keys) and (best is None or float(r["TotalDurationNs"]) > float(best["TotalDurationNs"])):
                best = r
        if best is None:
            continue
        name = best["Name"]
        counter = next((v.get("hbm_bytes") for k, v in hbm.items() if k == name), None)
        us = float(best["AverageNs"]) / 1e3
        if nlay is None:   # weight gradients are launched per layer or grouped over several layers of one geometry
            nlay = max(1.0, round(layers_per_step * steps / float(best["Calls"]))) if ("true, true" in name or "k_wgrad_tr<true" in name) else 1.0
        lay_alg, lay_flop = alg * (nlay if role == "weight_gradient" else 1.0), flop * nlay
        six = "k_conv_bfd<2, 2, 2, 3" in name      # the bf16x6 forward of earlier rounds: six MFMAs per product
        peak = BF16_MFMA_PEAK_TFLOPS / (6.0 if six else 3.0)
        out["kernels"][role] = {"kernel": name.replace("void srk::", "").split("(")[0], "avg_us": round(us, 1),
                                "layers_per_launch": nlay,
                                "algorithmic_bytes": int(lay_alg), "algorithmic_operands": what, "counter_hbm_bytes": counter,
                                "traffic_ratio": round(counter / lay_alg, 3) if counter else None,
                                "achieved_TFLOPs": round(lay_flop / us / 1e6, 1),
                                "mfmas_per_product": 6 if six else 3,
                                "frac_of_mfma_peak": round(lay_flop / us / 1e6 / peak, 4),
                                "hbm_GBps": round(counter / us / 1e3, 1) if counter else None}
    return out


def pixel_shuffle_rates(pkg, dev, batch, lr_size):
    """The standalone pixel-shuffle kernel (base_networks.py:157,181; SURVEY 8d: 23.62 MB per c2 image) at the c2 shape:
    [B, 48, 248, 248] -> [B, 3, 992, 992], forward and backward, HIP events on the launch stream, 8 bytes per element
    algorithmic.  The product path fuses the shuffle into the last conv's store; this is the unfused kernel."""
    hw = lr_size - 8
    x = torch.rand(batch, 48, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    out = {}
    with torch.no_grad():
        for key, fn in (("fwd", lambda: pkg.ops.pixel_shuffle(x, 4)),):
            y = fn()
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out["ps_kernel_GBps"] = round(2 * 4 * x.numel() * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        g = y
        lib = pkg._lib.load()
        dx = torch.empty_like(x)

        def bwd():
            pkg._lib.check(lib.srk_pixel_shuffle_backward(pkg._lib.ptr(g), pkg._lib.ptr(dx), batch, hw, hw, 3, 4,
                                                          pkg._lib.stream_ptr()), "srk_pixel_shuffle_backward")
        for _ in range(3):
            bwd()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            bwd()
        e1.record()
        torch.cuda.synchronize()
        out["ps_kernel_backward_GBps"] = round(2 * 4 * x.numel() * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    out["ps_kernel_shape"] = "[%d, 48, %d, %d] fp32 NHWC -> [%d, 3, %d, %d], r = 4" % (batch, hw, hw, batch, 4 * hw, 4 * hw)
    return out


def cpu_baseline(batch_cap=8, lr_size=256):
    """Oracle ESPCN x4 forward on the host cores (bounded sample: B=8; per thread count 2 warm-ups + best of 3; the
    thread count that gives the best rate is the one reported — all hardware threads oversubscribe oneDNN here)."""
    from oracle import fill, ref_modules as R
    net = fill.fill_module(R.ESPCN(3, 64, 4)).eval()
    x = fill.rand((batch_cap, 3, lr_size, lr_size), 1234)
    best, best_threads = None, 0
    ncpu = os.cpu_count() or 1
    with torch.no_grad():
        for nthr in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), 32, 16, 8}):
            if nthr > ncpu:
                continue
            torch.set_num_threads(nthr)
            for i in range(5):
                t0 = time.perf_counter()
                net(x)
                dt = time.perf_counter() - t0
                if i >= 2 and (best is None or dt < best):
                    best, best_threads = dt, nthr
    try:
        import psutil
        physical = psutil.cpu_count(logical=False)
    except Exception:  # noqa: BLE001
        physical = None
    # "cores" = the torch threads the reported (best) run used -- the bench contract's definition of the field; what the
    # host has is listed as host_* beside it
    return {"value": round(batch_cap / best, 2), "unit": "images/s", "cores": best_threads,
            "host_physical_cores": physical, "host_threads": ncpu, "kind": "port",
            "sample": "oracle ESPCN x4 forward (stock torch.nn CPU fp32), batch %d of %dx%d LR, best of 3; `cores` = the torch "
                      "thread count of the best run (%d) out of the tried %s on a host with %d hardware threads -- more threads "
                      "were slower, not unavailable" % (batch_cap, lr_size, lr_size, best_threads,
                                                        sorted(t for t in {ncpu, ncpu // 2, ncpu // 4, 32, 16, 8} if 0 < t <= ncpu), ncpu)}


# Per-sample algorithmic conv FLOPs of the training configs and the bf16 MFMA work the default
# "mixed" precision issues for them: forward = 6 bf16 MFMAs per product (bf16x6), data + weight gradient = 3 each (bf16x3).
# Upper estimates: the few Cin<=4 / 9x9 / 64->3 layers run on other kernels (fp32 MFMA, taps-as-N).
C3_FWD, C4_FWD = 2.2425e9, 4.0615e9
# of which: layers whose training forward runs bf16x3 in "mixed" (no activation between them and the loss, ops.py):
# VDSR's reconstruction conv; EDSR's body-end, 2 upsampler and reconstruction convs (75.5 + 302 + 1208 + 56.6 MFLOP)
C3_TAIL_FWD, C4_TAIL_FWD = 5.81e6, 1.6421e9
C5_G_FWD, C5_D_FWD = 4.543e9, 3.1437e9
C5_STEP = 55.6e9     # SRGAN adversarial step as the reference executes it, per sample


def mfma3_peak_frac(train_flop_per_sample, seconds_per_sample):
    """Algorithmic conv FLOPs of a training sample (forward + data gradient + weight gradient = 3 x forward; c5: SURVEY.md's
    55.6 GFLOP) / time / (2.5 PF / 3): the fraction of the matrix peak at THREE 16-bit MFMAs per fp32-equivalent product,
    which is what every large kernel of the training path issues (f16x3 forward, bf16x3 / f16x3 backward).  Reproducible by
    hand from ms_per_step; layers that still run six MFMAs (small problems: bf16x6) or the fp32 pipe count at their
    algorithmic FLOPs, i.e. they lower this number, they do not inflate it."""
    return train_flop_per_sample / seconds_per_sample / (BF16X3_PEAK_TFLOPS * 1e12)


ARITHMETIC = {
    "c3": "fp32 storage / accumulation; forward f16x3 (fp32-faithful, 3 MFMAs; reconstruction conv bf16x3), data and weight "
          "gradients %s, SGD + clip in fp32",
    "c4": "fp32 storage / accumulation; forward f16x3 in the residual trunk (fused blocks), bf16x3 in the activation-free tail "
          "(body-end, up-sampler, reconstruction convs); data and weight gradients %s; Adam in fp32",
    "c4_shard16": "as c4; at 16 patches the small-problem forward convs outside the fused blocks run bf16x6 (6 MFMAs)",
    "c5": "fp32 storage / accumulation; forward bf16x6 (small 3x3 problems) / f16x3 (large discriminator layers), data and "
          "weight gradients %s (incl. the stride-2 convs: k_wgrad_s2), BatchNorm statistics in double"}


PROGRESS = {"section": "start", "since": 0.0}   # which side metric is running (read by the watchdog)


def extras_watchdog(result, extra, rank, seconds):
    """N > 1 only.  A side metric that dies on ONE rank (say a refused graph capture) leaves the others waiting in a
    collective for ever, and the headline line -- measured before any of them -- would never be printed.  After `seconds`
    every rank leaves on its own clock (they start it behind the same barrier); rank 0 prints the line first, with what
    the side metrics had produced so far and the reason."""
    import threading

    def fire():
        if rank == 0 and result is not None:
            extra["extras_error"] = ("side metrics did not finish within %d s (SRK_BENCH_EXTRA_TIMEOUT); headline unaffected; "
                                     "section running when the watchdog fired: %s (for %.0f s)"
                                     % (seconds, PROGRESS["section"], time.perf_counter() - PROGRESS["since"]))
            result["extra"] = dict(extra)
            emit_json(result)
        sys.stdout.flush()
        os._exit(0)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def train_extra(pkg, dev, rank, world, nsteps=20, out=None, cpu_baselines=True):
    """Side metrics: c1 (SRCNN step incl. the bicubic pre-step) and c3 (VDSR x4, 41x41, batch 256) on one GPU; c4 EDSR x4
    training with global batch 128 sharded over the ranks (strong scaling; grouped weight gradients + overlapped bucketed
    RCCL exchange) and with 128 per GPU (weak); the 16-patch shard step that bounds 8-GPU strong scaling; c5 SRGAN."""
    out = {} if out is None else out
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    WARM = 5
    out["train_untimed_warmup_steps"] = WARM   # (per section; the shard step states its own)
    # the N > 1 code path: more than one rank, or a one-rank process group forced by SRK_DP_FORCE_COMM=1 (dry run of the
    # data-parallel sections with real RCCL collectives on a single GPU; the numbers then mean nothing)
    multi = world > 1 or torch.distributed.is_initialized()

    def run(kind, net, inp, tgt, loss_fn, clip, use_dp, steps=nsteps, warmup=WARM):
        net.to(dev).train()
        flat = pkg.optim.FlatParams(net)
        opt = pkg.optim.make_optimizer(kind, flat, 1e-5)
        dp = None
        if use_dp and multi:
            dp = pkg.dp.DataParallel(flat)
            dp.broadcast_params()
        step = pkg.trainers.GraphedStep(net, opt, loss_fn, (inp, tgt), dp=dp, clip=clip, warmup=2)
        # the batch sits in the step's static input buffers when the timed region starts (where an input pipeline's H2D
        # copies land: data.py writes a batch to any device tensor); a call with other tensors would copy them in first
        for sbuf, b in zip(step.static, (inp, tgt)):
            sbuf.copy_(b)
        inp0, tgt0 = inp, tgt
        inp, tgt = step.static[0], step.static[1]
        sec = time_steps(lambda: step(inp, tgt), steps, warmup, world if use_dp else 1, dev)
        # ... and the rounds-1..4 protocol beside it: the batch arrives in OTHER resident tensors, every step copies it into
        # the static buffers first (two device-to-device copies inside the timed region)
        run.sec_with_copy = time_steps(lambda: step(inp0, tgt0), steps, 2, world if use_dp else 1, dev)
        run.warmup = warmup
        sec_nocomm = None
        run.rank_span = None
        if dp is not None:   # the same step with the all-reduces skipped: the difference is the exposed communication
            dp.comm_enabled = False
            sec_nocomm = time_steps(lambda: step(inp, tgt), steps, 2, world, dev)
            # ... and what every rank's GPU needs for its own shard with nobody to wait for (min / max over ranks)
            local = rank_local_steps(lambda: step(inp, tgt), steps, 2, dev)
            run.rank_span = span_over_ranks(local / steps, world, dev)
            dp.comm_enabled = True
        torch.cuda.synchronize()
        step.close()          # graphs are destroyed here, never by the garbage collector in the middle of a later capture
        return sec, steps, sec_nocomm

    def c1():
        # BASELINE c1: SRCNN x2, 16 LR patches 32x32 -> bicubic x2 (utils.img_interp, bit-exact PIL) -> 3x64x64 ->
        # 3x48x48, MSE vs shave(target, 8), SGD (srcnn.py:116-131).  The pre-steps run inside the timed step.
        net = pkg.SRCNNNet(3, 64)
        net.weight_init()
        net.to(dev).train()
        flat = pkg.optim.FlatParams(net)
        opt = pkg.optim.make_optimizer("srcnn", flat, 1e-5)
        step = pkg.trainers.mse_step(net, opt, None)
        inp = torch.rand(16, 3, 32, 32, generator=g).to(dev)
        tgt = torch.rand(16, 3, 64, 64, generator=g).to(dev)

        def one(a, b):
            return step(pkg.utils.img_interp(a, 2), pkg.utils.shave(b, 8).contiguous())

        try:  # the whole iteration (bicubic pre-step + crop + train step) as one hipGraph; eager if capture is refused
            graphed = pkg.trainers.GraphedFn(one, (inp, tgt), flats=[flat])
            out["c1_mode"] = "hipGraph"
        except Exception:  # noqa: BLE001
            graphed = one
            out["c1_mode"] = "eager"
        sec = time_steps(lambda: graphed(inp, tgt), 50, 10, 1, dev)
        out["c1_srcnn_x2_train_patches_per_s_batch_16"] = round(16 * 50 / sec, 1)
        out["c1_srcnn_ms_per_step"] = round(1e3 * sec / 50, 3)
        # the reference's CPU path for the same step on this box's host cores (oracle = stock torch + Pillow)
        from oracle import ref_modules as R, img_interp as OI
        onet = R.SRCNN(3, 64)
        oopt = R.make_optimizer("srcnn", onet.parameters(), 1e-5)
        ci, ct = inp.cpu(), tgt.cpu()
        best, best_threads = None, 0
        for nthr in sorted({os.cpu_count() or 1, 32, 8}):   # tiny tensors: all 256 host threads oversubscribe
            torch.set_num_threads(nthr)
            for i in range(5):
                t0 = time.perf_counter()
                R.step_mse(onet, oopt, OI.img_interp(ci, 2), ct[..., 8:-8, 8:-8])
                dt = time.perf_counter() - t0
                if i >= 2 and (best is None or dt < best):
                    best, best_threads = dt, nthr
        out["c1_cpu_oracle_patches_per_s"] = round(16 / best, 1)
        out["c1_cpu_oracle_threads"] = best_threads

    def c3():
        net = pkg.VDSRNet(3, 64, 18)
        net.weight_init()
        x = torch.rand(256, 3, 41, 41, generator=g).to(dev)
        t = torch.rand(256, 3, 41, 41, generator=g).to(dev)
        sec, k, _ = run("vdsr", net, x, t, pkg.ops.mse_loss, 0.4, False)
        out["c3_vdsr_x4_train_patches_per_s"] = round(256 * k / sec, 1)
        out["c3_vdsr_ms_per_step"] = round(1e3 * sec / k, 3)
        out["c3_vdsr_ms_per_step_with_input_copy"] = round(1e3 * run.sec_with_copy / k, 3)
        out["c3_vdsr_mfma3_peak_frac"] = round(mfma3_peak_frac(3 * C3_FWD, sec / k / 256), 4)
        out["c3_arithmetic"] = ARITHMETIC["c3"] % pkg.ops.backward_arithmetic()
        rl = training_roofline("c3", 256 * 41 * 41, "VDSR body layer: conv3x3 64 -> 64 on 256 x 41 x 41 pixels, 31.7 GFLOP", 18)
        if rl:
            out["c3_roofline"] = rl

    gb = 128

    def edsr():
        net = pkg.EDSRNet(3, 64, 16)
        torch.manual_seed(1234)
        net.weight_init()
        return net

    def c4_strong():
        lo, hi = pkg.dp.shard_range(gb, rank, world)
        x = torch.rand(gb, 3, 32, 32, generator=torch.Generator().manual_seed(99))[lo:hi].to(dev)
        t = torch.rand(gb, 3, 128, 128, generator=torch.Generator().manual_seed(98))[lo:hi].to(dev)
        sec, k, nocomm = run("edsr", edsr(), x, t, pkg.ops.l1_loss, None, True)
        out["c4_edsr_x4_train_patches_per_s_global_batch_128"] = round(gb * k / sec, 1)
        out["c4_edsr_ms_per_step"] = round(1e3 * sec / k, 3)
        out["c4_edsr_ms_per_step_with_input_copy"] = round(1e3 * run.sec_with_copy / k, 3)
        out["c4_edsr_mfma3_peak_frac"] = round(mfma3_peak_frac(3 * C4_FWD, sec / k / gb) / world, 4)
        out["c4_arithmetic"] = ARITHMETIC["c4"] % pkg.ops.backward_arithmetic()
        if not multi:
            rl = training_roofline("c4", 128 * 32 * 32, "EDSR body layer: conv3x3 64 -> 64 on 128 x 32 x 32 pixels, 9.66 GFLOP", 33)
            if rl:
                out["c4_roofline"] = rl
        out["c4_scaling"] = ("strong (global batch 128 sharded over %d rank(s); 6.07 MB of gradients per step as bucketed RCCL "
                             "all-reduces issued behind the grouped weight-gradient launches)" % world)
        if nocomm is not None:
            out["c4_exposed_comm_ms"] = round(1e3 * (sec - nocomm) / k, 3)
            out["c4_ms_per_step_without_exchange"] = round(1e3 * nocomm / k, 3)
        if run.rank_span is not None:
            out["c4_rank_local_ms_per_step_min_max"] = [round(1e3 * v, 3) for v in run.rank_span]
            out["c4_gradient_bytes_per_step"] = 4 * sum(p.numel() for p in edsr().parameters())
        if not multi:
            # the same step with fp32-faithful (bf16x6) products in EVERY forward conv: by default the activation-free
            # tail of the net (body-end, upsampler and reconstruction convs) runs its training forward on bf16x3
            pkg.ops.LINEAR_TAIL_X3 = False
            try:
                sec6, k6, _ = run("edsr", edsr(), x, t, pkg.ops.l1_loss, None, False)
            finally:
                pkg.ops.LINEAR_TAIL_X3 = True
            out["c4_edsr_ms_per_step_bf16x6_forward_everywhere"] = round(1e3 * sec6 / k6, 3)

    def c4_shard16():
        # what ONE of 8 ranks computes per step under strong scaling (16 of the 128 patches), timed on this GPU alone:
        # full-batch step / shard step is the ceiling of the 8-GPU speed-up before any communication
        x = torch.rand(16, 3, 32, 32, generator=g).to(dev)
        t = torch.rand(16, 3, 128, 128, generator=g).to(dev)
        # (60 warm-up steps: a 1 ms step timed 5 steps after its capture sits on the clock governor's ramp from idle -- DESIGN
        #  11.2: ~50 launches -- and reads 3 - 5 % slow; the step draws 730 W, nothing about it is power-limited)
        sec, k, _ = run("edsr", edsr(), x, t, pkg.ops.l1_loss, None, False, steps=50, warmup=60)
        out["c4_shard16_ms_per_step"] = round(1e3 * sec / k, 3)
        out["c4_shard16_ms_per_step_with_input_copy"] = round(1e3 * run.sec_with_copy / k, 3)
        out["c4_shard16_untimed_warmup_steps"] = run.warmup
        out["c4_shard16_mfma3_peak_frac"] = round(mfma3_peak_frac(3 * C4_FWD, sec / k / 16), 4)
        out["c4_shard16_arithmetic"] = ARITHMETIC["c4_shard16"]
        if "c4_edsr_ms_per_step" in out and world == 1:
            out["c4_strong_scaling_ceiling_at_8_gpus"] = round(out["c4_edsr_ms_per_step"] / out["c4_shard16_ms_per_step"], 2)

    def c4_weak():
        # the same step with the per-GPU batch held at 128 (weak scaling: global batch 128 * world)
        x = torch.rand(gb, 3, 32, 32, generator=g).to(dev)
        t = torch.rand(gb, 3, 128, 128, generator=g).to(dev)
        sec, k, nocomm = run("edsr", edsr(), x, t, pkg.ops.l1_loss, None, True)
        out["c4_weak_edsr_x4_train_patches_per_s_batch_128_per_gpu"] = round(world * gb * k / sec, 1)
        out["c4_weak_ms_per_step"] = round(1e3 * sec / k, 3)
        if nocomm is not None:
            out["c4_weak_exposed_comm_ms"] = round(1e3 * (sec - nocomm) / k, 3)
        if run.rank_span is not None:
            out["c4_weak_rank_local_ms_per_step_min_max"] = [round(1e3 * v, 3) for v in run.rank_span]

    def c5():
        # SRGAN x4 generator + discriminator adversarial step (srgan.py:249-310), reference default batch 16 per GPU,
        # 32x32 LR -> 128x128 HR crops; two models, two optimizers.  One GPU: the whole step as one hipGraph.  DP: graphs
        # split at the two gradient exchanges (trainers.GraphedSegments), D's 153 MB of gradients in 32 MB buckets.
        G, D = pkg.SRGANGenerator(3, 64, 16), pkg.SRGANDiscriminator(3, 64, 128)
        torch.manual_seed(1234)
        G.weight_init()
        D.weight_init()
        G.to(dev).train()
        D.to(dev).train()
        gflat, dflat = pkg.optim.FlatParams(G), pkg.optim.FlatParams(D)
        g_opt = pkg.optim.make_optimizer("srgan_g", gflat, 1e-4)
        d_opt = pkg.optim.make_optimizer("srgan_d", dflat, 1e-4)
        lr_img = torch.rand(16, 3, 32, 32, generator=g).to(dev)
        hr_img = torch.rand(16, 3, 128, 128, generator=g).to(dev)
        if multi:
            g_dp, d_dp = pkg.dp.DataParallel(gflat), pkg.dp.DataParallel(dflat)
            g_dp.broadcast_params()
            d_dp.broadcast_params()
            sstep = pkg.trainers.GraphedSegments(pkg.trainers.srgan_segments(G, D, g_opt, d_opt, g_dp, d_dp, lazy_pack=True),
                                                 (lr_img, hr_img))
        else:
            sstep = pkg.trainers.GraphedFn(pkg.trainers.srgan_step(G, D, g_opt, d_opt, lazy_pack=True), (lr_img, hr_img),
                                           flats=[gflat, dflat])
        k = max(20, nsteps)
        sec = time_steps(lambda: sstep(lr_img, hr_img), k, 5, world, dev)
        out["c5_srgan_x4_adv_step_patches_per_s_batch_16_per_gpu"] = round(world * 16 * k / sec, 1)
        out["c5_srgan_ms_per_step"] = round(1e3 * sec / k, 3)
        out["c5_srgan_mfma3_peak_frac"] = round(mfma3_peak_frac(C5_STEP, sec / k / 16), 4)
        out["c5_arithmetic"] = ARITHMETIC["c5"] % pkg.ops.backward_arithmetic()
        if not multi:
            # NOT the c5 metric: the same iteration without the two gradient computations whose results the reference
            # discards (G's in the D step, D's parameter gradients in the G step; `main.py --prune_dead_grads`) -- same
            # parameters after every step, listed beside the faithful number for what the option is worth
            del sstep          # (the first graph goes before the second one is captured)
            import gc
            gc.collect()
            pstep = pkg.trainers.GraphedFn(pkg.trainers.srgan_step(G, D, g_opt, d_opt, lazy_pack=True, prune_dead_grads=True),
                                           (lr_img, hr_img), flats=[gflat, dflat])
            sec_p = time_steps(lambda: pstep(lr_img, hr_img), k, 5, 1, dev)
            out["c5_srgan_ms_per_step_dead_gradients_pruned_not_the_metric"] = round(1e3 * sec_p / k, 3)

    def cpu_train():
        # the reference's CPU path beside c3 / c4 / c5: the oracle's train steps (stock torch.nn, the
        # reference's loss / optimizer lines) on bounded batches -- VDSR 32 of 256, EDSR 16 (one rank's shard of 128), SRGAN 4
        # of 16 -- one warm-up + best of 2 per thread count; the best thread count is reported
        from oracle import fill, ref_modules as R
        ncpu = os.cpu_count() or 1
        sweep = sorted({t for t in (16, 32, 64) if t <= ncpu} or {ncpu})

        def best_of(fn, batch):
            best, best_thr = None, 0
            for nthr in sweep:
                torch.set_num_threads(nthr)
                for i in range(3):
                    t0 = time.perf_counter()
                    fn()
                    dt = time.perf_counter() - t0
                    if i >= 1 and (best is None or dt < best):
                        best, best_thr = dt, nthr
            return round(batch / best, 1), best_thr

        net = fill.fill_module(R.VDSR(3, 64, 18))
        opt = R.make_optimizer("vdsr", net.parameters(), 1e-5)
        x, t = fill.rand((32, 3, 41, 41), 1234), fill.rand((32, 3, 41, 41), 1235)
        out["c3_cpu_oracle_patches_per_s"], out["c3_cpu_oracle_threads"] = best_of(lambda: R.step_mse(net, opt, x, t, 0.4), 32)
        out["c3_cpu_oracle_sample"] = "oracle VDSR step (MSE, SGD momentum + weight decay, clip 0.4), batch 32 of 256, 41x41"
        net = fill.fill_module(R.EDSR(3, 64, 16), gain=0.5)
        opt = R.make_optimizer("edsr", net.parameters(), 1e-5)
        x, t = fill.rand((16, 3, 32, 32), 1234), fill.rand((16, 3, 128, 128), 1235)
        out["c4_cpu_oracle_patches_per_s"], out["c4_cpu_oracle_threads"] = best_of(lambda: R.step_l1(net, opt, x, t), 16)
        out["c4_cpu_oracle_sample"] = "oracle EDSR x4 step (L1, Adam), batch 16 = one rank's shard of 128, 32x32 -> 128x128"
        G, D = fill.fill_module(R.Generator(3, 64, 16), gain=0.5), fill.fill_module(R.Discriminator(3, 64, 128))
        g_opt = R.make_optimizer("srgan_g", G.parameters(), 1e-4)
        d_opt = R.make_optimizer("srgan_d", D.parameters(), 1e-4)
        x, t = fill.rand((4, 3, 32, 32), 1234), fill.rand((4, 3, 128, 128), 1235)
        out["c5_cpu_oracle_patches_per_s"], out["c5_cpu_oracle_threads"] = best_of(lambda: R.step_srgan(G, D, g_opt, d_opt, x, t), 4)
        out["c5_cpu_oracle_sample"] = "oracle SRGAN adversarial step (D then G, srgan.py:249-310), batch 4 of 16, 32x32 -> 128x128"
        out["cpu_oracle_host_threads"] = ncpu

    # every side metric is isolated: a failure is reported in the JSON instead of losing the headline line
    sections = ([("c1", c1), ("c3", c3)] if not multi else []) + [("c4_strong", c4_strong)] + \
               ([("c4_shard16", c4_shard16)] if not multi else [("c4_weak", c4_weak)]) + [("c5", c5)] + \
               ([("cpu_train", cpu_train)] if not multi and rank == 0 and cpu_baselines else [])
    for name, fn in sections:
        PROGRESS["section"], PROGRESS["since"] = name, time.perf_counter()
        try:
            fn()
        except Exception as e:  # noqa: BLE001 - reported, not swallowed
            out[name + "_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    PROGRESS["section"], PROGRESS["since"] = "done", time.perf_counter()
    return out


def c2_graph_replay_ms(net, x, steps, dev):
    """ms per c2 forward replayed from a hipGraph (the three launches back to back, no host work between them)."""
    side = torch.cuda.Stream(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.stream(side):
        net(x)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            y = net(x)
    for _ in range(40):   # (capture leaves the GPU idle: let the clock governor settle again, as in front of the timed window)
        graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    del y
    return e0.elapsed_time(e1) / steps


def board_power_probe(fn, seconds=2.5):
    """Board power and shader clock (rocm-smi, median of the samples) while `fn` loops for `seconds`: the c2 layers run at
    the board's power cap, where the clock -- not a pipe -- is what a faster kernel buys back.  None when
    rocm-smi is not there."""
    import re
    import shutil
    import statistics
    import subprocess
    import threading
    if not shutil.which("rocm-smi"):
        return None
    stop = threading.Event()
    samples = []

    def sample():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            except Exception:  # noqa: BLE001
                return
            pw = re.findall(r"Power \(W\): ([0-9.]+)", out)
            ck = re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
            if pw and ck:
                samples.append((float(pw[0]), float(ck[0])))

    th = threading.Thread(target=sample, daemon=True)
    t0 = time.time()
    n = 0
    th.start()
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = (time.time() - t0) / max(n, 1)
    stop.set()
    th.join(timeout=6)
    samples = samples[1:] if len(samples) > 2 else samples   # (the first sample may predate the load)
    if not samples:
        return None
    return {"board_power_w": round(statistics.median(p for p, _ in samples), 1),
            "sclk_mhz": round(statistics.median(c for _, c in samples), 1), "samples": len(samples),
            "ms_per_step_while_sampling": round(dt * 1e3, 4)}


def c2_other_precisions(pkg, net, x, steps, warmup, dev):
    """The same c2 forward in the other arithmetics, next to the headline's fp32-faithful f16x3 products: bf16x3 (3-term
    bf16 split, ~5e-6 rel: the fast option, the headline of rounds 1 and 2), bf16x6 (exact 3-way bf16 split, 6 MFMAs: the
    fp32-faithful class before f16x3) and exact fp32 MFMA."""
    res = {}
    prev = pkg.ops.get_precision()
    prev_f16 = pkg.ops.F16X3
    for key, mode, f16 in (("bf16x3", "bf16x3", prev_f16), ("bf16x6", "bf16x6", False), ("fp32", "fp32", prev_f16)):
        try:
            pkg.ops.set_precision(mode)
            pkg.ops.F16X3 = f16
            with torch.no_grad():
                sec = time_steps(lambda: net(x), steps, warmup, 1, dev)
            res["c2_%s_images_per_s" % key] = round(x.shape[0] * steps / sec, 1)
            if key == "bf16x3":
                scale = (x.shape[-1] / 256.0) ** 2
                res["c2_bf16x3_hbm_roofline_frac"] = round(
                    ESPCN_BYTES_PER_IMG * scale * x.shape[0] * steps / sec / 1e9 / HBM_PEAK_GBS, 4)
        except Exception as e:  # noqa: BLE001
            res["c2_%s_error" % key] = "%s: %s" % (type(e).__name__, str(e)[:200])
    pkg.ops.F16X3 = prev_f16
    pkg.ops.set_precision(prev)
    return res


def main():
    args = parse()
    claim_stdout()
    import __graft_entry__
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        __graft_entry__.build()
    import pytorch_super_resolution_model_collection_amd as pkg
    rank, world, local = pkg.dp.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    pkg._lib.load()
    pkg.ops.set_precision(args.precision)

    # ---- c2: ESPCN x4 inference, random-init weights of the reference's distribution, synthetic LR batch
    torch.manual_seed(1234)
    net = pkg.ESPCNNet(3, 64, 4)
    net.weight_init()
    net.to(dev).eval()
    # Three resident LR batches taken in turn, each handed over as a NEW tensor object (what a loader delivers): the
    # absolute-maximum pass of the first layer (ops.amax_of caches it on the tensor object) therefore runs in EVERY timed
    # forward, as it would on a stream of new batches -- profiles/r06_c2_kernel_stats.csv: k_absmax calls == forwards.
    gen = torch.Generator().manual_seed(1234 + rank)
    xs = [torch.rand(args.batch, 3, args.lr_size, args.lr_size, generator=gen).to(dev) for _ in range(3)]
    x = xs[0]
    turn = [0]

    def step():
        turn[0] += 1
        xb = xs[turn[0] % 3]
        with torch.no_grad():
            return net(xb.view(xb.shape))

    y = step()
    assert tuple(y.shape) == (args.batch, 3, 4 * (args.lr_size - 8), 4 * (args.lr_size - 8))
    # The per-layer HIP events (40 forwards, the last 10 averaged) run BEFORE the timed window: behind an idle gap the clock
    # governor takes ~50 steps of this workload to settle (tools/c2_ramp.py: steps 5 - 24 behind 2 s of idle 1.11 ms, steps
    # 50+ 1.01 ms), so W = 5 warm-up steps alone would leave the whole timed window -- and the layer times -- inside that ramp.
    # Disclosed in the line: extra.c2_cold_window_ms_per_step is the same W + K window behind 2 s of idle,
    # extra.c2_power_probe.ms_per_step_while_sampling the rate sustained over seconds.
    pkg.trainers.quiesce_gc()   # set-up is done: no full cyclic collection (~80 ms with torch loaded) inside a timed window
    layer_ms, layer_kernels = espcn_layer_events(net, xs, max(3, min(args.steps, 10)), discard=30)
    sec = time_steps(step, args.steps, args.warmup, world, dev)
    imgs_per_s = world * args.batch * args.steps / sec
    ranks_seen = rccl_ranks_seen(dev)
    local_span = span_over_ranks(rank_local_steps(step, args.steps, 1, dev) / args.steps, world, dev)

    # measured device-to-device copy bandwidth on this box (read + write bytes / time), beside the vendor peak
    copy_gbps = None
    try:
        a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
        b = torch.empty_like(a)
        for _ in range(2):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = round(2 * a.numel() * 4 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del a, b
        torch.cuda.empty_cache()
    except Exception:  # noqa: BLE001
        pass

    result = None
    if rank == 0:
        H = args.lr_size
        flop_l2 = 2.0 * args.batch * (H - 6) ** 2 * 32 * 64 * 9
        dom = max(range(3), key=lambda i: layer_ms[i])
        bf3 = pkg.ops.get_precision() != "fp32"
        what = ["conv5x5 3->64 + ReLU", "conv3x3 64->32 + ReLU", "conv3x3 32->48 + pixel-shuffle store"]
        names = ["%s %s" % (k, w) for k, w in zip(layer_kernels, what)]   # kernel ids come from the dispatcher
        peak = BF16X3_PEAK_TFLOPS if bf3 else FP32_MFMA_PEAK_TFLOPS
        flops = [2.0 * args.batch * (H - 4) ** 2 * 64 * 3 * 25, flop_l2, 2.0 * args.batch * (H - 8) ** 2 * 48 * 32 * 9]
        achieved = flops[dom] / (layer_ms[dom] * 1e-3) / 1e12
        scale = (H / 256.0) ** 2
        traffic, traffic_source = pmc_traffic(names[dom], args.batch, H)   # (a committed rocprofv3 pass, not this run)
        result = {
            "metric": "ESPCN x4 LR->HR images/sec (infer)", "value": round(imgs_per_s, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * sec / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE_NOTE.get(pkg.ops.get_precision() if pkg.ops.F16X3 or pkg.ops.get_precision() != "mixed"
                                    else "bf16x6", "f32"), "data": "synthetic",
            "config": {"workload": "c2: ESPCN x4 inference, %dx%d LR, batch %d per GPU, fp32, random-init N(0,0.02)"
                                   % (H, H, args.batch),
                       "timing": "W warm-up + K timed forwards, barrier + synchronize on both sides; 40 forwards of per-layer "
                                 "event timing run in front of the window (the clock governor needs ~50 forwards from idle: "
                                 "extra.c2_cold_window_* is the same window behind 2 s of idle)",
                       "untimed_forwards_before_window": 1 + 40 + args.warmup,
                       "input": "3 resident batches in rotation, a new tensor object per forward: the first layer's |x| "
                                "maximum pass runs inside every timed forward",
                       "parallelism": "replicas x%d (no collective)" % world},
            "roofline": {"bound": "mfma", "kernel": names[dom], "achieved": round(achieved, 2),
                         "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "note": "achieved = algorithmic conv FLOPs / live HIP-event kernel time; peak = dense 16-bit "
                                 "MFMA peak / 3 (three fp16 / bf16 MFMAs per fp32-equivalent product)" if bf3 else
                                 "achieved = algorithmic conv FLOPs / live HIP-event kernel time; peak = fp32 MFMA",
                         "kernel_ms": round(layer_ms[dom], 4),
                         "layer_ms": [round(m, 4) for m in layer_ms], "layer_kernels": names,
                         "whole_net_hbm": {
                             "achieved_GBps": round(ESPCN_BYTES_PER_IMG * scale * imgs_per_s / world / 1e9, 1),
                             "peak_GBps": HBM_PEAK_GBS, "measured_copy_GBps": copy_gbps,
                             "frac": round(ESPCN_BYTES_PER_IMG * scale * imgs_per_s / world / 1e9 / HBM_PEAK_GBS, 4),
                             "note": "61.11 MB/img per-layer compulsory traffic (SURVEY.md 8d); north-star target 0.30"}},
        }
    extra = {}
    if torch.distributed.is_initialized():
        # what the collective library saw, so that an N > 1 record explains itself
        extra["rccl_ranks_seen"] = ranks_seen
        extra["dist_backend"] = torch.distributed.get_backend()
        extra["c2_rank_local_ms_per_step_min_max"] = [round(1e3 * v, 4) for v in local_span]
        extra["c2_images_per_s_per_gpu"] = round(imgs_per_s / world, 1)   # compare with the N = 1 record's `value`
        if ranks_seen != world and rank == 0 and result is not None:
            extra["rccl_error"] = "all-reduce of ones returned %s for WORLD_SIZE %d" % (ranks_seen, world)
    if not args.no_extra:
        dog = None
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
            dog = extras_watchdog(result, extra, rank, int(os.environ.get("SRK_BENCH_EXTRA_TIMEOUT", "420")))
        if world == 1 and rank == 0:
            # what is outside the three kernels of a step, and what a hipGraph of the forward does about it
            extra["c2_step_minus_kernels_ms"] = round(1e3 * sec / args.steps - sum(layer_ms), 4)
            try:
                extra["c2_graph_ms_per_step"] = round(c2_graph_replay_ms(net, x, max(10, args.steps), dev), 4)
                extra["c2_graph_images_per_s"] = round(args.batch / (extra["c2_graph_ms_per_step"] * 1e-3), 1)
            except Exception as e:  # noqa: BLE001
                extra["c2_graph_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
            # NOT the metric: the same forward when the caller DECLARES the input range (ops.declare_absmax(x, 1.0): a ToTensor
            # batch is in [0, 1]) -- the f16x3 first layer then needs no pass over x for its scale
            try:
                def step_declared():
                    turn[0] += 1
                    xb = xs[turn[0] % 3]
                    with torch.no_grad():
                        return net(pkg.ops.declare_absmax(xb.view(xb.shape), 1.0))
                dsec = time_steps(step_declared, args.steps, args.warmup, 1, dev)
                extra["c2_declared_input_bound_images_per_s_not_the_metric"] = round(args.batch * args.steps / dsec, 1)
            except Exception as e:  # noqa: BLE001
                extra["c2_declared_input_bound_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
            # the timed window again, cold: behind 2 s of idle (what `value` would be without the layer events in front of it)
            try:
                torch.cuda.synchronize()
                time.sleep(2.0)
                cold = time_steps(step, args.steps, args.warmup, 1, dev)
                extra["c2_cold_window_ms_per_step"] = round(1e3 * cold / args.steps, 4)
                extra["c2_cold_window_images_per_s"] = round(args.batch * args.steps / cold, 1)
            except Exception as e:  # noqa: BLE001
                extra["c2_cold_window_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
            try:
                probe = board_power_probe(step)
                if probe:
                    extra["c2_power_probe"] = probe
                    # the same roofline at the clock the board's power cap leaves the layers (the peak in `roofline` is the
                    # 2.4 GHz figure of the microarchitecture guide; no kernel that draws the cap ever sees that clock)
                    rf = result["roofline"]
                    rf["at_measured_clock"] = {
                        "sclk_mhz_whole_net": probe["sclk_mhz"], "board_power_w": probe["board_power_w"],
                        "peak": round(rf["peak"] * probe["sclk_mhz"] / 2400.0, 1),
                        "frac": round(rf["achieved"] / (rf["peak"] * probe["sclk_mhz"] / 2400.0), 4),
                        "note": "rocm-smi median while the c2 forward loops (extra.c2_power_probe); the dominant layer alone runs "
                                "lower still (profiles/r05_power.txt: 1734 MHz at 1400 W)"}
            except Exception as e:  # noqa: BLE001
                extra["c2_power_probe_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
        if world == 1 and pkg.ops.get_precision() == "mixed":
            extra.update(c2_other_precisions(pkg, net, x, max(5, args.steps // 2), 2, dev))
        if world == 1:
            try:
                extra.update(pixel_shuffle_rates(pkg, dev, args.batch, args.lr_size))
                if copy_gbps:
                    extra["ps_kernel_frac_of_measured_copy"] = round(extra["ps_kernel_GBps"] / copy_gbps, 3)
                # the fused form: the last conv of c2 reads 32 channels and writes the 48 shuffled ones
                hw = args.lr_size - 8
                extra["c2_conv3_ps_store_GBps"] = round(4.0 * args.batch * ((hw + 2) ** 2 * 32 + hw * hw * 48) / (layer_ms[2] * 1e-3) / 1e9, 1)
            except Exception as e:  # noqa: BLE001
                extra["ps_kernel_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
        train_extra(pkg, dev, rank, world, args.extra_steps, extra, cpu_baselines=not args.no_cpu_baseline)
        if dog is not None:
            dog.cancel()
    if rank == 0:
        if extra:
            result["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(lr_size=args.lr_size)
        emit_json(result)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
