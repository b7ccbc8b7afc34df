#!/bin/bash
# quick bench (textured) of library variants under variants/ (tools/variant_any.sh), "-" = the default build
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then L=""; else L="TF_LIB=variants/$v.so"; fi
  bash tools/r4_sweep.sh ${TAG}_$v "$L"
done
