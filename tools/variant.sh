#!/bin/bash
# Build a tuning variant of the library: tools/variant.sh <name> <extra hipcc flags...>  -> variants/<name>.so
# (only tf_kernels.hip is recompiled with the extra flags; run the normal build first).  Use it with TF_LIB=variants/<name>.so.
set -e
cd "$(dirname "$0")/../texturefusion_amd/csrc"
name=$1; shift
mkdir -p ../../variants /tmp/tfvar
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden "$@" -c tf_kernels.hip -o /tmp/tfvar/$name.o
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/tfvar/$name.o build/tf_mesh.hip.o build/tf_atlas.hip.o build/tf_pre.hip.o build/tf_unit.hip.o \
      build/tf_capi.cpp.o build/tf_comm.cpp.o -ldl -lpthread -o ../../variants/$name.so
