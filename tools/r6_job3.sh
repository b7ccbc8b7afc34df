#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6_job3; mkdir -p $O
bash tools/gpu_job.sh r6j3 tests drv
TF_PATCH_DBG=3 timeout 300 python tools/stamps.py patch > $O/stamps_patch.txt 2>&1; tail -10 $O/stamps_patch.txt
