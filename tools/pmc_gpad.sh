#!/bin/bash
# usage: pmc_gpad.sh PERM,W16 ...
# LDS bank-conflict cycles of the c2 kernels against the halo group-stride padding of k_conv_bfw (SRK_BFW_GPAD)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/gpad; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for p in "$@"; do
  SRK_BFW_PERM=${p%%,*} SRK_BFW_W16=${p##*,} rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p$p -o c2 -- python $ROOT/bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $OUT/p$p.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/p$p/**/c2_counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "bfw" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_LDS_BANK_CONFLICT": n[k] += 1
for k in acc:
    print("gpad $p", k[:40], {c: round(v / n[k] / 1e6, 2) for c, v in acc[k].items()}, "M per launch")
PY
done
