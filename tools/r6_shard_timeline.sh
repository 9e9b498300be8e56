#!/bin/bash
# ON the GPU box: per-kernel timeline of ONE replay of the graphed 16-patch EDSR shard step (kernel trace of tools/shard_step.py)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r6_tl}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o s16 -- python $ROOT/tools/shard_step.py 16 20 > $OUT/run.log 2>&1
CSV=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
cd $ROOT
python tools/graph_timeline2.py $CSV k_pack_batched_amax 3 > $OUT/timeline.txt 2>&1
tail -45 $OUT/timeline.txt
find $OUT/kt -name "*.csv" -size +20M -delete
