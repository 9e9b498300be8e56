#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
for r in 1 2 3; do
echo "walk1: $(python tools/time_c2_layers.py 0 2>&1 | grep layer)"
echo "walk0: $(SRK_ROWSR_WALK=0 python tools/time_c2_layers.py 0 2>&1 | grep layer)"
done
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "first_layer" 2>&1 | tail -2
} > gpurun_out/run28.log 2>&1
