#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
for r in 1 2 3; do
echo "sprobe: $(SRK_LIB_PATH=/root/repo/variants/rowsw_sprobe.so python tools/time_c2_layers.py 0 2>&1 | grep layer)"
echo "base:   $(python tools/time_c2_layers.py 0 2>&1 | grep layer)"
done
} > gpurun_out/run30.log 2>&1
