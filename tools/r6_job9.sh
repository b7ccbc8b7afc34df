#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6_job9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_neighbours.py tests/test_gpu_textured_soak.py tests/test_gpu_partition.py tests/test_gpu_unit.py tests/test_gpu_mesh.py -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
for rep in 1 2; do
  echo "regional:"; timeout 300 python tools/first_orbit.py 2>&1 | tail -1
  echo "global:";   TF_LIB=$PWD/variants/r6_global_stamp.so timeout 300 python tools/first_orbit.py 2>&1 | tail -1
done | tee $O/first_orbit.txt
KT=1 STEPS=100 bash tools/r5_ab.sh j9 "-" "TF_LIB=variants/r6_global_stamp.so" 2>&1 | grep -v "k_frame<false\|k_frame<true, false\|k_patch<" | tee $O/ab.log
