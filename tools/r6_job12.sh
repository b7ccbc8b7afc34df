#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6_job12; mkdir -p $O
bash tools/gpu_job.sh r6j12 tests drv
for lib in "" "variants/fp7.so"; do
  echo "== unit with lib=[$lib]"; TF_LIB=${lib:+$PWD/$lib} timeout 300 python tools/prof_unit.py --run 2>/dev/null | cut -c1-200
  TF_LIB=${lib:+$PWD/$lib} timeout 300 python tools/prof_unit.py --run 2>/dev/null | cut -c1-200
done | tee $O/unit_ab.txt
timeout 900 python tools/soak_random.py 71 12 > $O/soak71.txt 2>&1; tail -2 $O/soak71.txt
timeout 900 python tools/soak_random.py 72 12 > $O/soak72.txt 2>&1; tail -2 $O/soak72.txt
