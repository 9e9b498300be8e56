#!/bin/bash
# Run ON the MI355X box (through gpurun) from the repo root: collects the evidence committed under profiles/.
#   1. rocprofv3 --kernel-trace --stats of the c2 bench command (per-kernel average durations)
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a pass: TCC has 4 slots, they cost 3+2)
#   3. the same for one EDSR x4 training step (c4) so the train-path kernels are on record as well
# Output: gpurun_out/prof_<tag>/...; tools/summarize_prof.py turns it into profiles/<tag>_*.{csv,json}.
set -u
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
C2="python $ROOT/bench.py --steps 40 --warmup 5 --no-extra --no-cpu-baseline"   # (enough steps that the cold first launches do not carry the rocprofv3 average)
C4="python $ROOT/tools/edsr_b16.py 128"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_kt -o c2 -- $C2 > $OUT/c2_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c2_fetch -o c2 -- $C2 > $OUT/c2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c2_write -o c2 -- $C2 > $OUT/c2_write.log 2>&1
# matrix-core and LDS activity of the same command (SQ counters share a pass: 8 SQ slots)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/c2_sq -o c2 -- $C2 > $OUT/c2_sq.log 2>&1
# round 4: wave states and LDS issue of the same command (two passes: 8 SQ counter slots each)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/c2_stall1 -o c2 -- $C2 > $OUT/c2_stall1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $OUT/c2_stall2 -o c2 -- $C2 > $OUT/c2_stall2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_kt -o c4 -- $C4 > $OUT/c4_kt.log 2>&1
# the other training configs: c3 (VDSR, batch 256), c5 (SRGAN adversarial step), and the 16-patch EDSR shard that bounds
# 8-GPU strong scaling (kernel traces only)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_kt -o c3 -- python $ROOT/tools/vdsr_step.py 256 > $OUT/c3_kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_kt -o c5 -- python $ROOT/tools/srgan_step.py 16 > $OUT/c5_kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4s16_kt -o c4s16 -- python $ROOT/tools/edsr_b16.py 16 > $OUT/c4s16_kt.log 2>&1
# counter evidence for the training kernels (north star: "rocprof HBM GB/s and MFMA utilisation"): the same three
# separate passes as c2 for c3, c4 and the 16-patch shard; SQ pass only for c5
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
for WL in c3 c4 c4s16; do
  case $WL in c3) CMD="python $ROOT/tools/vdsr_step.py 256";; c4) CMD="$C4";; c4s16) CMD="python $ROOT/tools/edsr_b16.py 16";; esac
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${WL}_fetch -o $WL -- $CMD > $OUT/${WL}_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${WL}_write -o $WL -- $CMD > $OUT/${WL}_write.log 2>&1
  timeout 600 rocprofv3 --pmc $SQ --output-format csv -d $OUT/${WL}_sq -o $WL -- $CMD > $OUT/${WL}_sq.log 2>&1
done
timeout 600 rocprofv3 --pmc $SQ --output-format csv -d $OUT/c5_sq -o c5 -- python $ROOT/tools/srgan_step.py 16 > $OUT/c5_sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c5_fetch -o c5 -- python $ROOT/tools/srgan_step.py 16 > $OUT/c5_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c5_write -o c5 -- python $ROOT/tools/srgan_step.py 16 > $OUT/c5_write.log 2>&1
# raw counter CSVs are large: keep only the reduced files
cd $ROOT
python tools/summarize_prof.py $TAG > $OUT/summary.log 2>&1
cat $OUT/summary.log | tail -30
find $OUT -name "*counter_collection.csv" -size +8M -delete
mkdir -p $OUT/reduced && cp $ROOT/profiles/${TAG}_* $OUT/reduced/ 2>/dev/null
