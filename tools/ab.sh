#!/bin/bash
# A/B two kernel builds on the same box: tools/libsrk_A.so vs tools/libsrk_B.so (alternating, 2 rounds)
cd $(dirname $0)/..
for round in 1 2; do for v in A B; do
  echo "== $v"; SRK_LIB_PATH=$PWD/tools/libsrk_$v.so python tools/time_shapes.py "$@" 2>&1 | grep -v amdgpu | sed "s/sum=.*//"
done; done
