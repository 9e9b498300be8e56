#!/bin/bash
# k_wgrad_tr (SRK_WGRAD_TR=1, default) against k_wgrad_bf (SRK_WGRAD_TR=0): layer timings and training steps, same box
cd $(dirname $0)/..
for round in 1 2; do for tr in 0 1; do
  echo "== TR=$tr $(SRK_WGRAD_TR=$tr python tools/time_wgrad.py vdsr edsr128 edsr16 2>&1 | grep wgrad | awk '{printf "%s %s ms %s | ", $1, $3, $8}')"
done; done
for f in tools/libsrk_T2.so tools/libsrk_T4.so; do
  [ -f $f ] && echo "== $f $(SRK_LIB_PATH=$PWD/$f python tools/time_wgrad.py vdsr edsr128 2>&1 | grep wgrad | awk '{printf "%s %s ms | ", $1, $3}')"
done
for tr in 0 1; do
  echo "== TR=$tr steps"
  SRK_WGRAD_TR=$tr python tools/shard_step.py 16 2>&1 | grep "ms/step"
  SRK_WGRAD_TR=$tr python tools/shard_step.py 128 30 2>&1 | grep "ms/step"
  SRK_WGRAD_TR=$tr python tools/vdsr_graph_step.py 2>&1 | grep -i "ms" | tail -1
done
