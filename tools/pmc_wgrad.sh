#!/bin/bash
# HBM read bytes and LDS activity of the weight-gradient kernel on one workhorse shape (separate --pmc passes, no traces).
#   bash tools/pmc_wgrad.sh [shape]   (on the GPU box; results under gpurun_out/pmc_wgrad)
SHAPE=${1:-vdsr}
OUT=gpurun_out/pmc_wgrad
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o w -- python tools/time_wgrad.py $SHAPE > $OUT/fetch.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o w -- python tools/time_wgrad.py $SHAPE > $OUT/sq.log 2>&1
python - <<'PY'
import csv, glob, collections
for sub in ("fetch", "sq"):
    for f in glob.glob("gpurun_out/pmc_wgrad/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "k_wgrad" not in row["Kernel_Name"]:
                continue
            a = acc[(row["Kernel_Name"][:60], row["Counter_Name"])]
            a[0] += float(row["Counter_Value"]); a[1] += 1
        for (k, c), (v, n) in sorted(acc.items()):
            extra = "  -> HBM read %.1f MB (x2 gfx950 correction, KiB units)" % (v / n * 1024 * 2 / 1e6) if c == "FETCH_SIZE" else ""
            print("%-62s %-28s mean %.4g over %d launches%s" % (k, c, v / n, n, extra))
PY
