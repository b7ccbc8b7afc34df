#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6_job13; mkdir -p $O
KT=1 STEPS=100 bash tools/r5_ab.sh j13 "-" "TF_KFP_ORDER2=1" "-" "TF_KFP_ORDER2=1" 2>&1 | grep -v "k_frame<false\|k_frame<true, false\|k_patch<" | tee $O/ab.log
