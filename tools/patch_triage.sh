#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for d in 2 1 0; do
  out=$R/gpurun_out/ptriage_$d; mkdir -p $out
  TF_PATCH_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --steps 200 --warmup 20 --cpu-frames 0 --no-host-path --no-roofline --no-pmc > /dev/null 2>&1
  f=$(find $out -name "t_kernel_stats.csv" | head -1)
  python3 - "$f" $d <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_patch<' in r['Name']: print("dbg=%s %-20s %8.1f us"%(sys.argv[2], r['Name'].split('(')[0][-24:], float(r['AverageNs'])/1e3))
PY
done
