#!/bin/bash
# time the c2 3x3 layers with every variants/ring_*.so given (bits):  tools/ring_ablate_run.sh 0 1 2 ...
cd $(dirname $0)/..
for a in "$@"; do
  echo "== BFR_ABL=$a"
  SRK_LIB_PATH=$PWD/variants/ring_$a.so timeout 300 python tools/time_c2_layers.py 1 2 2>&1 | grep -v amdgpu
done
