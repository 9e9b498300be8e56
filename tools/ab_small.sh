#!/bin/bash
# does the 64-pixel block pay beyond the current small-problem threshold?  (SRK_BFD_SMALL=1 forces it everywhere)
cd $(dirname $0)/..
for round in 1 2; do for v in default 1; do
  echo "== SRK_BFD_SMALL=$v"
  if [ $v = default ]; then unset SRK_BFD_SMALL; else export SRK_BFD_SMALL=$v; fi
  python tools/srgan_graph_step.py 2>&1 | grep "step B" 
  python tools/shard_step.py 16 2>&1 | grep "ms/step"
  python tools/shard_step.py 32 2>&1 | grep "ms/step"
done; done
