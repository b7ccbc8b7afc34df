#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_17; mkdir -p $O; rm -f $O/*
timeout 1200 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_atlas.py tests/test_gpu_textured_soak.py -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
