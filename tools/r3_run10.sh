#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_unit.py tests/test_gpu_partition.py tests/test_gpu_texmap.py tests/test_gpu_host_mirror.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -40 $O/pytest.log | grep -v "^$" | tail -30
