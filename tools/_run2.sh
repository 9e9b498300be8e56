#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
s=$(date +%s)
timeout 150 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES TA_BUSY --output-format csv -d /tmp/pq -o p -- python /root/repo/tools/time_c2_layers.py 0 > /tmp/pq.log 2>&1
echo "rc=$? secs=$(( $(date +%s) - s ))" > /root/repo/gpurun_out/pq.txt
tail -5 /tmp/pq.log >> /root/repo/gpurun_out/pq.txt
find /tmp/pq -name "*.csv" | head >> /root/repo/gpurun_out/pq.txt
