#!/bin/bash
# AddressSanitizer + UBSan over the CPU oracle (sanitizers run on the CPU build only): builds oracle/tf_oracle.c
# instrumented and runs the oracle-only CPU tests against it.
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -std=gnu11 -fPIC -fopenmp -ffp-contract=off -fno-fast-math -fexcess-precision=standard \
    -fsanitize=address,undefined -fno-omit-frame-pointer oracle/tf_oracle.c -o /tmp/libtf_oracle_asan.so -shared -fopenmp -lm
ASAN=$(gcc -print-file-name=libasan.so)
TF_ORACLE_LIB=/tmp/libtf_oracle_asan.so LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_oracle_kat.py tests/test_oracle_mesh.py tests/test_oracle_pre.py tests/test_resize_properties.py \
                   tests/test_color_compensate.py tests/test_pack_vertices.py tests/test_golden.py tests/test_dataset.py -x -q -m "not gpu" "$@"
