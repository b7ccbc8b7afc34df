#!/bin/bash
# VALU/SALU instruction counts per kernel of the call-by-call flow and the fused launch
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/pmc
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O -o split -- python3 $GRAFT_REPO_ROOT/tools/kernel_times.py 30 >/dev/null 2>&1
echo rc=$?
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$O/**/split_counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (s, n) in sorted(acc.items()):
    print("%-62s %-14s per launch %12.0f (%d)" % (k[0], k[1], s / n, n))
PY
