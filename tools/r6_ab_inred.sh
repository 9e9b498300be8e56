#!/bin/bash
# ON the GPU box: A/B of the in-launch weight-gradient slab reduction (SRK_WGRAD_INRED=0: round-5 reduce launches), same box
for rep in 1 2; do
for v in 0 1; do
  echo "== SRK_WGRAD_INRED=$v (rep $rep)"
  SRK_WGRAD_INRED=$v python tools/shard_step.py 16 50 2>&1 | tail -1
  SRK_WGRAD_INRED=$v python tools/vdsr_graph_step.py 256 2>&1 | tail -1
  SRK_WGRAD_INRED=$v python tools/srgan_graph_step.py 20 2>&1 | tail -1
  SRK_WGRAD_INRED=$v python tools/shard_step.py 128 20 2>&1 | tail -1
done
done
