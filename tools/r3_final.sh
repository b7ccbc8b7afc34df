#!/bin/bash
# final profile set of the round (run through gpurun from the repo root)
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/prof_r3.sh textured --steps 200 --warmup 20
bash tools/prof_r3.sh tsdf --mode tsdf --steps 200 --warmup 20
bash tools/prof_r3.sh hall --scene big --hires --steps 60 --warmup 10
cd "${GRAFT_REPO_ROOT:-.}"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/driver_style.json 2> gpurun_out/driver_style.err; tail -c 600 gpurun_out/driver_style.json
