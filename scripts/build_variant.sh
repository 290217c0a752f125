#!/bin/bash
# An instrumented build of the library next to the product one: scripts/build_variant.sh <name> "<-D flags>" -> build_variants/libeqf_<name>.so
# (load it with EQF_VIO_AMD_LIB=build_variants/libeqf_<name>.so; build_variants/ is git-ignored and removed before the round ends)
set -eu
NAME=$1; FLAGS=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build_variants/obj_$NAME
cd $ROOT/eqf_vio_amd/csrc
make -s eqf_build_id.inc
BASE="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-cuda-compat -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-max-memory-clause=1"
for f in eqf_capi eqf_tiled eqf_tiledf; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $FLAGS -c -o $ROOT/build_variants/obj_$NAME/$f.o $f.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/build_variants/libeqf_$NAME.so $ROOT/build_variants/obj_$NAME/eqf_capi.o $ROOT/build_variants/obj_$NAME/eqf_tiled.o $ROOT/build_variants/obj_$NAME/eqf_tiledf.o
rm -rf $ROOT/build_variants/obj_$NAME
ls -la $ROOT/build_variants/libeqf_$NAME.so
