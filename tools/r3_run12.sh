#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/prof_r3.sh textured --steps 200 --warmup 20
bash tools/prof_r3.sh tsdf --mode tsdf --steps 200 --warmup 20
bash tools/prof_r3.sh hall --scene big --hires --steps 100 --warmup 20 --cpu-frames 8 --cpu-warmup 8
TF_KA_DBG=4096 timeout 300 python tools/timeline3.py > gpurun_out/prof_r3_textured/timeline.txt 2>&1
