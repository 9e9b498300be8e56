#!/bin/bash
# Effective shader clock and matrix-pipe occupancy of the kernels of a command (run on the GPU box):
#   tools/pmc_clock.sh <tag> <command ...>
# GRBM_GUI_ACTIVE / 8 = busy cycles of the launch; kernel-trace of the same pass gives its duration.
ROOT=$(pwd); TAG=$1; shift; OUT=$ROOT/gpurun_out/clk_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -o p -- "$@" > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
fc = glob.glob("$OUT/**/p_counter_collection.csv", recursive=True)
fk = glob.glob("$OUT/**/p_kernel_trace.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(fk[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); t = collections.Counter()
seen = set()
for r in csv.DictReader(open(fc[0])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"])); n[k] += 1; t[k] += dur.get(r["Dispatch_Id"], 0.0)
print("%-60s %5s %9s %8s %8s %8s %8s %8s" % ("kernel", "calls", "us", "GHz", "mfma%", "wait%", "stall%", "issue%"))
for k in sorted(acc, key=lambda k: -t[k])[:8]:
    v = acc[k]; c = n[k]; us = t[k] / c
    cyc = v["GRBM_GUI_ACTIVE"] / c / 8
    wc = max(v["SQ_WAVE_CYCLES"], 1)
    print("%-60s %5d %9.1f %8.3f %8.1f %8.1f %8.1f %8.1f" % (k, c, us, cyc / us / 1e3, 100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / c / (cyc * 1024),
          100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc, 100 * v["SQ_ACTIVE_INST_ANY"] / wc))
PY
