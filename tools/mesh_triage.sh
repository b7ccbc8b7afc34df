#!/bin/bash
# k_mesh phase triage: TF_MESH_DBG = 1 filter only, 2 + staging / corner flags, 3 + cell pass, 4 + ranking, 0 = all
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for d in 1 2 3 4 0; do
  out=$R/gpurun_out/triage_$d; mkdir -p $out
  TF_MESH_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --steps 100 --warmup 20 --cpu-frames 0 --no-host-path --no-roofline --no-pmc > /dev/null 2>&1
  f=$(find $out -name "t_kernel_stats.csv" | head -1)
  echo "dbg=$d"; python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_mesh" in r["Name"]: print("  %-40s calls %s avg %.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
