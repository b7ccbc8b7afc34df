#!/bin/bash
# Build a tuning variant of the library from ONE recompiled source: tools/variant_any.sh <name> <source.hip> <extra hipcc flags...>
#   -> variants/<name>.so (run the normal build first; use with TF_LIB=variants/<name>.so)
set -e
cd "$(dirname "$0")/../texturefusion_amd/csrc"
name=$1; src=$2; shift; shift
mkdir -p ../../variants /tmp/tfvar
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden "$@" -c $src -o /tmp/tfvar/$name.o
objs=""
for f in tf_kernels.hip tf_group.hip tf_xchg.hip tf_mesh.hip tf_atlas.hip tf_pre.hip tf_unit.hip tf_capi.cpp tf_comm.cpp; do
  if [ "$f" = "$src" ]; then objs="$objs /tmp/tfvar/$name.o"; else objs="$objs build/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -lpthread -o ../../variants/$name.so
