#!/bin/bash
# A/B an environment switch on the training steps:  tools/ab_env.sh VAR valueA valueB
cd $(dirname $0)/..
VAR=$1; shift
for round in 1 2; do for v in "$@"; do
  export $VAR=$v
  echo "== $VAR=$v"
  python tools/shard_step.py 16 2>&1 | grep "ms/step"
  python tools/srgan_graph_step.py 2>&1 | grep "step B"
  python tools/shard_step.py 128 30 2>&1 | grep "ms/step"
done; done
