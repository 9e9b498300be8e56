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


fetch = counter_means("c2_fetch", "FETCH_SIZE")
write = counter_means("c2_write", "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 10 "
                 "--warmup 3 --no-extra --no-cpu-baseline` (c2: ESPCN x4, 256x256 LR, batch 64)",
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
# matrix-core / LDS activity (one SQ pass): per-kernel means per launch
sq_names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_WAVE_CYCLES",
            "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"]
sq = {n: counter_means("c2_sq", n) for n in sq_names}
sq_out = {"source": "rocprofv3 --pmc " + " ".join(sq_names) + " over the c2 bench command (own pass, no traces)",
          "note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 "
                  "(the counter is summed over the 8 XCDs); SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles per "
                  "v_mfma_f32_16x16x32_bf16 summed over all SIMDs (checked against the instruction count of the layer)",
          "kernels": {}}
for k in sorted(set().union(*[set(v) for v in sq.values()])):
    if "srk::k_conv" not in k:
        continue
    rec = {n: round(sq[n].get(k, (0.0, 0))[0], 1) for n in sq_names}
    if rec["GRBM_GUI_ACTIVE"] > 0:
        rec["mfma_util"] = round(rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * rec["GRBM_GUI_ACTIVE"] / 8.0), 4)
    sq_out["kernels"][k] = rec
if sq_out["kernels"]:
    with open(os.path.join(dst, "%s_c2_pmc_mfma.json" % tag), "w") as fh:
        json.dump(sq_out, fh, indent=1)
    print(json.dumps(sq_out, indent=1))
with open(os.path.join(dst, "%s_c2_pmc_traffic.json" % tag), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))
