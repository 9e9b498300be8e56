#!/bin/bash
# A/B two builds (tools/libsrk_A.so, tools/libsrk_B.so) on the weight-gradient shapes + training steps
cd $(dirname $0)/..
for round in 1 2; do for v in A B; do
  echo "== $v"
  SRK_LIB_PATH=$PWD/tools/libsrk_$v.so python tools/time_wgrad.py edsr128 vdsr edsr16 2>&1 | grep wgrad
done; done
for v in A B; do
  echo "== $v steps"
  SRK_LIB_PATH=$PWD/tools/libsrk_$v.so python tools/shard_step.py 16 2>&1 | grep "ms/step"
  SRK_LIB_PATH=$PWD/tools/libsrk_$v.so python tools/shard_step.py 128 30 2>&1 | grep "ms/step"
  SRK_LIB_PATH=$PWD/tools/libsrk_$v.so python tools/vdsr_graph_step.py 2>&1 | grep -i "ms" | tail -1
  SRK_LIB_PATH=$PWD/tools/libsrk_$v.so python tools/srgan_graph_step.py 2>&1 | grep "step B"
done
