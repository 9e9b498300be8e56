#!/bin/bash
# Same-box A/B of the ring kernel (k_conv_bfr) against the barrier kernel (k_conv_bfw): the two 3x3 layers of c2 (f16x3),
# then the VDSR step.   tools/ab_ring.sh [extra env assignments ...]
cd $(dirname $0)/..
for round in 1 2; do for v in 0 1; do
  echo "== SRK_BFR=$v $*"
  env SRK_BFR=$v "$@" timeout 300 python tools/time_c2_layers.py 1 2 2>&1 | grep -v amdgpu
done; done
for v in 0 1; do
  echo "== SRK_BFR=$v $*: VDSR step"; env SRK_BFR=$v "$@" timeout 300 python tools/time_vdsr.py 2>&1 | tail -1
done
