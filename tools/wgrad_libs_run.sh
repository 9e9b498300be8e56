#!/bin/bash
# time every tools/libsrk_<variant>.so: weight-gradient layers, then (variants without a digit-letter suffix 'w') the steps
cd $(dirname $0)/..
for round in 1 2; do for f in tools/libsrk_*.so; do
  v=$(basename $f .so); v=${v#libsrk_}
  echo "== $v $(SRK_LIB_PATH=$PWD/$f python tools/time_wgrad.py vdsr edsr128 edsr16 2>&1 | grep wgrad | awk '{printf "%s %s ms %s | ", $1, $3, $8}')"
done; done
for f in tools/libsrk_*.so; do
  v=$(basename $f .so); v=${v#libsrk_}
  case $v in *w) continue;; esac
  echo "== $v steps: $(SRK_LIB_PATH=$PWD/$f python tools/shard_step.py 16 2>&1 | grep -o '[0-9.]* ms/step') | $(SRK_LIB_PATH=$PWD/$f python tools/shard_step.py 128 30 2>&1 | grep -o '[0-9.]* ms/step') | $(SRK_LIB_PATH=$PWD/$f python tools/vdsr_graph_step.py 2>&1 | grep -o 'B=256: [0-9.]* ms')"
done
