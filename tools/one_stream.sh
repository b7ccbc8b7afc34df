#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for o in 0 1 0 1; do
  echo "TF_ONE_STREAM=$o: $(TF_ONE_STREAM=$o python3 bench.py --steps 200 --warmup 20 --cpu-frames 0 --no-host-path --no-roofline 2>/dev/null | python3 -c 'import json,sys; d=json.load(sys.stdin); print(d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"])')"
done
