#!/bin/bash
# round-6 job 1: parity of the half-wave filter variant, then A/B of filter forms and k_frame dispatch orders
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6_job1; mkdir -p $O
TF_LIB=$PWD/variants/half5.so timeout 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_textured_soak.py tests/test_gpu_partition.py tests/test_gpu_unit.py tests/test_gpu_host_mirror.py -x -q > $O/pytest_half.log 2>&1; echo "half tests rc=$?"; tail -3 $O/pytest_half.log
KT=1 STEPS=100 bash tools/r5_ab.sh j1 "-" "TF_LIB=variants/filt8.so" "TF_LIB=variants/half5.so" "TF_LIB=variants/half5.so TF_FILTER_WG=640" \
  "TF_KFP_ORDER=0 TF_KFP_KA_WG=6" "TF_KFP_ORDER=0 TF_KFP_KA_WG=7" "TF_KFP_PATCH_WG=512" "TF_KFP_PATCH_WG=768 TF_KFP_KA_WG=6" "TF_KFP_KA_WG=6" 2>&1 | tee $O/ab.log
