#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r2w
for round in 1 2; do for v in A B; do
  echo "== $v"
  SRK_LIB_PATH=$PWD/tools/libsrk_$v.so python tools/srgan_graph_step.py 2>&1 | grep -v amdgpu | tail -2
  SRK_LIB_PATH=$PWD/tools/libsrk_$v.so python tools/shard_step.py 2>&1 | grep -v amdgpu | tail -2
done; done
