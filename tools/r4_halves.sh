#!/bin/bash
# A/B of K-A variants built by tools/variant_any.sh (variants/<name>.so): parity tests, then the quick bench, textured and TSDF-only
#   tools/r4_halves.sh <tag> <variant> [<variant> ...]      ("-" = the default library)
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
mkdir -p gpurun_out/r4_$TAG
for v in "$@"; do
  [ "$v" = "-" ] && continue
  echo "== tests with variants/$v.so"
  TF_LIB=variants/$v.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_gpu_group.py tests/test_gpu_textured_soak.py tests/test_gpu_partition.py tests/test_gpu_unit.py -x -q -m gpu 2>&1 | tail -2
done
for v in "$@"; do
  if [ "$v" = "-" ]; then L=""; else L="TF_LIB=variants/$v.so"; fi
  bash tools/r4_sweep.sh ${TAG}_$v "$L"
  BENCH_ARGS="--mode tsdf" bash tools/r4_sweep.sh ${TAG}t_$v "$L"
done
