#!/bin/bash
# round-6 job 2: full GPU tests, driver-flag bench, phase stamps of filter / patch / mesher, K-A timeline
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6_job2; mkdir -p $O
bash tools/gpu_job.sh r6j2 testsx drv
TF_MESH_DBG=10 timeout 300 python tools/stamps.py filter > $O/stamps_filter.txt 2>&1; tail -14 $O/stamps_filter.txt
TF_PATCH_DBG=3 timeout 300 python tools/stamps.py patch > $O/stamps_patch.txt 2>&1; tail -12 $O/stamps_patch.txt
TF_MESH_DBG=9 timeout 300 python tools/stamps.py mesh > $O/stamps_mesh.txt 2>&1; tail -16 $O/stamps_mesh.txt
