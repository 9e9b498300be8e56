#!/usr/bin/env python3
"""Reduce the rocprofv3 output of tools/profile_round.sh to the small files committed under profiles/:
   <tag>_c2_kernel_stats.csv / <tag>_c4_kernel_stats.csv   copies of rocprofv3's --stats kernel summary
   <tag>_c2_pmc_traffic.json   per-kernel mean FETCH_SIZE / WRITE_SIZE per launch -> HBM bytes, with the
                               gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
                               both counters are reported in KiB; FETCH_SIZE counts 128-B requests at 64 B for
                               wide coalesced reads, so it is doubled; WRITE_SIZE is used as reported.
bench.py reads <tag>_c2_pmc_traffic.json for `roofline.traffic` (the counters cannot be collected inside bench.py)."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def find(sub, suffix):
    hits = glob.glob(os.path.join(src, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


for wl in ("c2", "c4", "c3", "c5", "c4s16"):
    f = find(wl + "_kt", "kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, wl)))
        print("copied", f)


def counter_means(sub, counter):
    f = find(sub, "counter_collection.csv")
    if not f:
        return {}
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}



SQ_NAMES = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_WAVE_CYCLES",
            "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"]
COMMANDS = {
    "c2": "python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline (c2: ESPCN x4, 256x256 LR, batch 64)",
    "c3": "python tools/vdsr_step.py 256 (c3: 5 eager VDSR x4 training steps, 41x41, batch 256)",
    "c4": "python tools/edsr_b16.py 128 (c4: 5 eager EDSR x4 training steps, 32x32 LR, batch 128)",
    "c4s16": "python tools/edsr_b16.py 16 (the 16-patch shard of 8-GPU strong scaling)",
    "c5": "python tools/srgan_step.py 16 (c5: SRGAN adversarial steps, batch 16)",
}


def traffic_and_mfma(wl):
    """<tag>_<wl>_pmc_traffic.json (FETCH_SIZE x2 / WRITE_SIZE per launch) and <tag>_<wl>_pmc_mfma.json (SQ pass)."""
    fetch = counter_means(wl + "_fetch", "FETCH_SIZE")
    write = counter_means(wl + "_write", "WRITE_SIZE")
    if fetch or write:
        out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `%s`" % COMMANDS[wl],
               "corrections": "counters in KiB; FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B); WRITE_SIZE as reported",
               "kernels": {}}
        for k in sorted(set(fetch) | set(write)):
            if "srk::" not in k:
                continue
            fk, nf = fetch.get(k, (0.0, 0))
            wk, nw = write.get(k, (0.0, 0))
            out["kernels"][k] = {"launches": max(nf, nw), "fetch_KiB_raw": round(fk, 1), "write_KiB_raw": round(wk, 1),
                                 "hbm_read_bytes": int(fk * 1024 * 2), "hbm_write_bytes": int(wk * 1024),
                                 "hbm_bytes": int(fk * 1024 * 2 + wk * 1024)}
        with open(os.path.join(dst, "%s_%s_pmc_traffic.json" % (tag, wl)), "w") as fh:
            json.dump(out, fh, indent=1)
        print("wrote %s_%s_pmc_traffic.json (%d kernels)" % (tag, wl, len(out["kernels"])))
    # matrix-core / LDS activity (one SQ pass): per-kernel means per launch
    sq = {n: counter_means(wl + "_sq", n) for n in SQ_NAMES}
    sq_out = {"source": "rocprofv3 --pmc " + " ".join(SQ_NAMES) + " over `%s` (own pass, no traces)" % COMMANDS[wl],
              "note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 "
                      "(the counter is summed over the 8 XCDs); SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles per "
                      "v_mfma_f32_16x16x32_bf16 summed over all SIMDs (checked against the instruction count of the layer)",
              "kernels": {}}
    for k in sorted(set().union(*[set(v) for v in sq.values()])):
        if "srk::k_conv" not in k and "srk::k_wgrad" not in k and "srk::k_res2" not in k and "srk::k_c64" not in k:
            continue
        rec = {n: round(sq[n].get(k, (0.0, 0))[0], 1) for n in SQ_NAMES}
        rec["launches"] = max(sq[n].get(k, (0.0, 0))[1] for n in SQ_NAMES)
        if rec["GRBM_GUI_ACTIVE"] > 0:
            rec["mfma_util"] = round(rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * rec["GRBM_GUI_ACTIVE"] / 8.0), 4)
        sq_out["kernels"][k] = rec
    if sq_out["kernels"]:
        with open(os.path.join(dst, "%s_%s_pmc_mfma.json" % (tag, wl)), "w") as fh:
            json.dump(sq_out, fh, indent=1)
        print("wrote %s_%s_pmc_mfma.json (%d kernels)" % (tag, wl, len(sq_out["kernels"])))


for wl in ("c2", "c3", "c4", "c4s16", "c5"):
    traffic_and_mfma(wl)


# round 4: wave-state / LDS-issue counters of the c2 kernels (what holds k_conv_bfw at 3.2 - 3.7 TB/s of mixed traffic)
STALL_NAMES = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS",
               "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F16",
               "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"]
st = {}
for sub in ("c2_stall1", "c2_stall2"):
    for n in STALL_NAMES:
        m = counter_means(sub, n)
        if m:
            st[n] = m
if st:
    out = {"source": "rocprofv3 --pmc <counters below, two passes> over `%s`" % COMMANDS["c2"],
           "note": "per-launch means summed over all waves / SIMDs; SQ_WAVE_CYCLES = wave-resident cycles, SQ_WAIT_ANY = cycles a "
                   "wave waits on any s_waitcnt, SQ_ACTIVE_INST_LDS / SQ_WAIT_INST_LDS = cycles LDS instructions execute / wait "
                   "to issue; fractions are relative to SQ_WAVE_CYCLES", "kernels": {}}
    for k in sorted(set().union(*[set(v) for v in st.values()])):
        if "srk::k_conv" not in k:
            continue
        rec = {n: round(st[n].get(k, (0.0, 0))[0], 1) for n in STALL_NAMES if n in st}
        wc = rec.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU"):
                if n in rec:
                    rec[n + "_frac"] = round(rec[n] / wc, 4)
        out["kernels"][k] = rec
    with open(os.path.join(dst, "%s_c2_pmc_stalls.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote %s_c2_pmc_stalls.json (%d kernels)" % (tag, len(out["kernels"])))

