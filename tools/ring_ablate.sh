#!/bin/bash
# Constant-ablation builds of k_conv_bfr (BFR_ABL bits, csrc/conv_bfr.hip) as variants/ring_<bits>.so: only conv_bfr.hip is
# recompiled, the other objects come from csrc/build.  EXTRA="-DBFR_NSET=4" TAG=n4: further defines, suffix of the name.   tools/ring_ablate.sh 0 1 2 4 ...   (run in the build container)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
C=$ROOT/pytorch_super_resolution_model_collection_amd/csrc
mkdir -p $ROOT/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -fno-gpu-rdc -Wno-unused-function"
for a in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DBFR_ABL=$a ${EXTRA} -c $C/conv_bfr.hip -o /tmp/conv_bfr_$a${TAG}.o 2>/dev/null
    objs=$(ls $C/build/*.o | grep -v conv_bfr.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/ring_$a${TAG}.so $objs /tmp/conv_bfr_$a${TAG}.o
    echo built variants/ring_$a${TAG}.so ) &
done
wait
