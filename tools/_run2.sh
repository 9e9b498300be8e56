#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > /root/repo/gpurun_out/counters.txt 2>&1
grep -o "Name:[[:space:]]*[A-Za-z0-9_]*" /root/repo/gpurun_out/counters.txt | sort -u | grep -E "TA_|TCP_|TCC_|TD_|SQ_INST.*VMEM|SQ_WAIT_INST|SQ_INSTS_V" > /root/repo/gpurun_out/counters_mem.txt
wc -l /root/repo/gpurun_out/counters_mem.txt
