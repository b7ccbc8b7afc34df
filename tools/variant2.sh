#!/bin/bash
# Build a tuning variant of the library: tools/variant2.sh <name> <source file> <extra hipcc flags...>  -> variants/<name>.so
# (only the named source of texturefusion_amd/csrc is recompiled with the extra flags; run the normal build first).
# Use it with TF_LIB=variants/<name>.so.
set -e
cd "$(dirname "$0")/../texturefusion_amd/csrc"
name=$1; src=$2; shift; shift
mkdir -p ../../variants /tmp/tfvar
objs=""
for f in tf_kernels.hip tf_mesh.hip tf_atlas.hip tf_pre.hip tf_unit.hip tf_capi.cpp tf_comm.cpp; do
  if [ "$f" = "$src" ]; then
    x=""; case $f in *.cpp) x="-x hip";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden "$@" $x -c $f -o /tmp/tfvar/$name.o
    objs="$objs /tmp/tfvar/$name.o"
  else
    objs="$objs build/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -lpthread -o ../../variants/$name.so
