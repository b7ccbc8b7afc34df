#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6_job5
timeout 900 python -m pytest tests/test_gpu_partition.py tests/test_gpu_bench_multi.py -x -q > gpurun_out/r6_job5/pytest_part.log 2>&1; tail -3 gpurun_out/r6_job5/pytest_part.log
bash tools/r6_job5.sh
