#!/bin/bash
# Build a second library from a patched copy of csrc/ for same-box A/B runs:
#   tools/build_variant.sh <out.so> <patch-script> [extra hipcc flags, e.g. -DSRK_EXPERIMENTS]
# the patch script is run inside the copy of csrc/ before compiling
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
OUT=$1; PATCH=$2; shift 2
W=$(mktemp -d /tmp/srkvar.XXXX)
mkdir -p $W/pkg/csrc $W/include
cp $ROOT/pytorch_super_resolution_model_collection_amd/csrc/*.hip $ROOT/pytorch_super_resolution_model_collection_amd/csrc/*.h $W/pkg/csrc/
cp $ROOT/include/srk.h $W/include/
(cd $W/pkg/csrc && bash $PATCH)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -fno-gpu-rdc -Wno-unused-function $*"
cd $W/pkg/csrc
ls *.hip | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc $FLAGS -c {} -o {}.o 2>/dev/null"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT *.o
rm -rf $W
echo built $OUT
