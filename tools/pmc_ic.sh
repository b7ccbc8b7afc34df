#!/bin/bash
# instruction / scalar cache counters of the fused kernel: tools/pmc_ic.sh <tag>
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/pmc
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 120 --unique-frames 60 --cpu-frames 0 --no-roofline"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $O -o $1ic -- $B >/dev/null 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/**/$1ic_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_frame" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in sorted(acc.items()):
        print("%-22s per launch %14.0f  (%d launches)" % (k, s / max(n, 1), n))
PY
