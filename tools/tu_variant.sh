#!/bin/bash
# One library per constant-ablation / option build of ONE translation unit (the other objects are the release ones of csrc/build):
#   tools/tu_variant.sh <name> <file.hip> [-D flags ...]   ->  tools/libsrk_<name>.so
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
C=$ROOT/pytorch_super_resolution_model_collection_amd/csrc
NAME=$1; SRC=$2; shift 2
O=/tmp/tuv_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -fno-gpu-rdc -Wno-unused-function "$@" -c $C/$SRC -o $O
OBJS=$(ls $C/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/libsrk_$NAME.so $OBJS $O
echo built tools/libsrk_$NAME.so
