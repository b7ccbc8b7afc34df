#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6_job4; mkdir -p $O
bash tools/gpu_job.sh r6j4 tests
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/drv.json 2> $O/drv.err ) 2> $O/drv.time; echo "drv rc=$?"; tail -3 $O/drv.time
TF_KA_DBG=4096 timeout 300 python tools/timeline_tex.py > $O/timeline_tex.txt 2>&1; cat $O/timeline_tex.txt | cut -c1-250
timeout 600 python bench.py --gpus 1 --force-exchange --steps 100 --warmup 20 --no-group --cpu-frames 0 > $O/force.json 2> $O/force.err; echo "force rc=$?"
