#!/bin/bash
# same-box A/B of two kernel builds on the hipGraph-captured EDSR x4 train step
cd $(dirname $0)/..
for round in 1 2; do for v in A B; do
  echo "== $v"; SRK_LIB_PATH=$PWD/tools/libsrk_$v.so python tools/edsr_small_batch.py 2>&1 | grep "B=128\|B= 16"
done; done
