#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6_job11; mkdir -p $O
KT=1 STEPS=100 bash tools/r5_ab.sh j11 "-" "TF_LIB=variants/kfp7.so" "TF_LIB=variants/kfp7x0.so" "-" "TF_LIB=variants/kfp7.so" 2>&1 | grep -v "k_frame<false\|k_frame<true, false\|k_patch<" | tee $O/ab.log
