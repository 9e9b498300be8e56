#!/bin/bash
# TA / TCP / TCC counters of k_conv_rowsr on the c2 first layer, release library vs the build without staging loads
# (variants/rowsw_d1.so = conv_rowsw.hip with -DSRK_KDBG_CONST=1), at most three counters per pass.
# Run on the box; output: gpurun_out/pmc_rowsr.txt
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
PASSES=("TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES"
        "TA_ADDR_STALLED_BY_TD_CYCLES TA_BUFFER_WRITE_WAVEFRONTS TA_FLAT_READ_WAVEFRONTS"
        "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_GATE_EN1"
        "TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCP_TA_ADDR_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES"
        "TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ_LATENCY"
        "TCP_TCC_READ_REQ_LATENCY TCP_WRITE_TAGCONFLICT_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES"
        "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_STALL_INFLIGHT_MAX"
        "TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_TAG_STALL"
        "TCC_IB_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_REQ")
: > gpurun_out/pmc_rowsr.txt
cd /tmp
for v in release noloads; do
  if [ $v = release ]; then L=/root/repo/pytorch_super_resolution_model_collection_amd/libsrk.so; else L=/root/repo/variants/rowsw_d1.so; fi
  i=0
  for P in "${PASSES[@]}"; do
    i=$((i+1))
    rm -rf /tmp/pr_$v$i
    SRK_LIB_PATH=$L timeout 100 rocprofv3 --pmc $P --output-format csv -d /tmp/pr_$v$i -o p -- python /root/repo/tools/time_c2_layers.py 0 > /tmp/pr_$v$i.log 2>&1
    rc=$?
    f=$(find /tmp/pr_$v$i -name "*counter_collection.csv" 2>/dev/null | head -1)
    if [ -z "$f" ]; then echo "$v pass $i: rc=$rc no file ($(tail -1 /tmp/pr_$v$i.log | cut -c1-120))" >> /root/repo/gpurun_out/pmc_rowsr.txt; continue; fi
    python /root/repo/tools/pmc_rowsr_sum.py "$f" $v >> /root/repo/gpurun_out/pmc_rowsr.txt
  done
done
