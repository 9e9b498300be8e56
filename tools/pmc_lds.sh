#!/bin/bash
# LDS bank-conflict / busy cycles per kernel of a command:  tools/pmc_lds.sh <tag> <command...>
ROOT=$(pwd); TAG=$1; shift; OUT=$ROOT/gpurun_out/lds_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p -- "$@" > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/p_counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_LDS_IDX_ACTIVE"])
print("%-60s %6s %12s %12s %12s  (k cycles per launch; GUI = kernel cycles x 8 XCDs)" % ("kernel", "calls", "conflict", "lds_active", "gui/8"))
for k, v in rows[:14]:
    c = max(n[k], 1)
    print("%-60s %6d %12.1f %12.1f %12.1f" % (k[:60], c, v["SQ_LDS_BANK_CONFLICT"] / c / 1e3, v["SQ_LDS_IDX_ACTIVE"] / c / 1e3, v["GRBM_GUI_ACTIVE"] / c / 8e3))
PY
