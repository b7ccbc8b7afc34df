#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6_job10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_textured_soak.py -x -q -k "randomised" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
KT=1 STEPS=100 bash tools/r5_ab.sh j10 "TF_PATCH_DBG=6" "TF_PATCH_DBG=5" "-" 2>&1 | grep -v "k_frame<false\|k_frame<true, false\|k_patch<" | tee $O/ab.log
