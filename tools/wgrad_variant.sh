#!/bin/bash
# One library per constant-ablation / option build of conv_wgrad_bf16.hip only (the other objects are the release ones of
# csrc/build):   tools/wgrad_variant.sh <name> [-D flags ...]   ->  tools/libsrk_<name>.so
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
C=$ROOT/pytorch_super_resolution_model_collection_amd/csrc
NAME=$1; shift
O=/tmp/wgv_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -fno-gpu-rdc -Wno-unused-function "$@" -c $C/conv_wgrad_bf16.hip -o $O
OBJS=$(ls $C/build/*.o | grep -v conv_wgrad_bf16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/libsrk_$NAME.so $OBJS $O
echo built tools/libsrk_$NAME.so
