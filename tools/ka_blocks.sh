#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for b in 1792 1280 1024 768; do
  echo "TF_KA_BLOCKS=$b: $(TF_KA_BLOCKS=$b python3 bench.py --steps 200 --warmup 20 --cpu-frames 0 --no-host-path --no-roofline 2>/dev/null | python3 -c 'import json,sys; d=json.load(sys.stdin); print(d["value"], d["ms_per_step"])')"
done
