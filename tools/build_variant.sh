#!/bin/bash
# A/B build: the shipped objects with ONE translation unit recompiled under extra flags -> pygemma_amd/lib_dev/<name>/libpygemma_hip.so
# usage: tools/build_variant.sh <name> <unit (e.g. rotate_geno)> [extra hipcc flags...]      (use with PYGEMMA_HIP_LIB=...; build the product first)
set -e
NAME=$1; UNIT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/pygemma_amd/lib_dev/$NAME; mkdir -p $OUT
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable $*"
cd $ROOT/pygemma_amd/csrc
/opt/rocm/bin/hipcc $FLAGS -c $UNIT.hip -o $OUT/$UNIT.o
OBJS=""
for o in ../lib/obj/*.o; do b=$(basename $o .o); if [ $b = $UNIT ]; then OBJS="$OBJS $OUT/$UNIT.o"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o $OUT/libpygemma_hip.so
echo "built $OUT/libpygemma_hip.so"
