#!/bin/bash
# Single-c development build for A/B timing of the association kernel: compiles the library with -DPG_ONLY_C=<c> (assoc.hip then holds
# that one instantiation + its LRT pair) into pygemma_amd/lib_dev/, next to the shipped one.  Use with PYGEMMA_HIP_LIB=... .
# usage: tools/build_dev.sh [c] [extra hipcc flags...]
set -e
C=${1:-5}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/pygemma_amd/lib_dev; mkdir -p $OUT/obj
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable -DPG_ONLY_C=$C $*"
cd $ROOT/pygemma_amd/csrc
for f in api assoc comm rotate rotate_geno syevd; do
  # stale objects are dangerous here (a changed pg_ctx layout in common.hpp once made assoc.o and api.o disagree): rebuild on any header change
  if [ $f = assoc ] || [ ! -f $OUT/obj/$f.o ] || [ $f.hip -nt $OUT/obj/$f.o ] || [ common.hpp -nt $OUT/obj/$f.o ] || [ dgemm.hpp -nt $OUT/obj/$f.o ] || [ ../../include/pygemma_hip.h -nt $OUT/obj/$f.o ]; then
    ( /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OUT/obj/$f.o 2>&1 | grep -v "loop not unrolled" | grep -E "error|warning: var" || true ) &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/obj/{api,assoc,comm,rotate,rotate_geno,syevd}.o -ldl -o $OUT/libpygemma_hip.so
echo "built $OUT/libpygemma_hip.so (c=$C)"
