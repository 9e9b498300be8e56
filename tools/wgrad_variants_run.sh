#!/bin/bash
# time every tools/libsrk_<variant>.so on the weight-gradient shapes (run on the GPU box)
cd $(dirname $0)/..
for round in 1 2; do for f in tools/libsrk_*.so; do
  v=$(basename $f .so); v=${v#libsrk_}
  echo "== $v $(SRK_LIB_PATH=$PWD/$f python tools/time_wgrad.py vdsr edsr128 2>&1 | grep wgrad | awk '{printf "%s %s ms | ", $1, $3}')"
done; done
