#!/bin/bash
# ON the GPU box: per-kernel timeline of ONE replay of a graphed step.  tools/r6_timeline.sh <tag> <marker-kernel> <python script + args...>
set -u
TAG=$1; MARK=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python $ROOT/"$@" > $OUT/run.log 2>&1
CSV=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
cd $ROOT
python tools/graph_timeline2.py $CSV $MARK 3 > $OUT/timeline.txt 2>&1
tail -3 $OUT/run.log
find $OUT/kt -name "*.csv" -size +20M -delete
