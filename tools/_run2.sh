#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
for r in 1 2 3; do
echo "keep:  $(python tools/time_c2_layers.py 0 2>&1 | grep layer)"
echo "head~: $(SRK_LIB_PATH=/root/repo/variants/rowsw_bufload.so python tools/time_c2_layers.py 0 2>&1 | grep layer)"
done
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "first_layer" 2>&1 | tail -2
} > gpurun_out/run23.log 2>&1
