#!/bin/bash
# sustained (3 s loop) time per launch of every tools/libsrk_<variant>.so on the VDSR body layer
cd $(dirname $0)/..
for f in tools/libsrk_*.so; do
  v=$(basename $f .so); v=${v#libsrk_}
  echo "== $v $(SRK_LIB_PATH=$PWD/$f LOOP_SECS=3 python tools/time_wgrad.py ${1:-vdsr} 2>&1 | grep wgrad | awk '{printf "%s %s ms | ", $1, $3}')"
done
