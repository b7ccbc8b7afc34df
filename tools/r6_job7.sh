#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6_job7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_textured_soak.py tests/test_gpu_parity.py tests/test_gpu_atlas.py -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
KT=1 STEPS=100 bash tools/r5_ab.sh j7 "-" "TF_LIB=variants/nolate.so" "TF_LIB=variants/late2.so" "-" "TF_LIB=variants/nolate.so" 2>&1 | grep -v "k_frame<false\|k_frame<true, false\|k_patch<" | tee $O/ab.log
TF_KA_DBG=4096 timeout 300 python tools/timeline_tex.py > $O/timeline_tex.txt 2>&1; head -6 $O/timeline_tex.txt | cut -c1-250; tail -4 $O/timeline_tex.txt
