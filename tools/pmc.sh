#!/bin/bash
# SQ counters of the fused per-frame kernel: tools/pmc.sh <tag>   (writes gpurun_out/pmc/<tag>{a,b}_*)
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/pmc
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 120 --unique-frames 60 --cpu-frames 0 --no-roofline"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O -o $1a -- $B >/dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $O -o $1b -- $B >/dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/**/$1*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_frame" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in sorted(acc.items()):
        print("%-22s per launch %14.0f  (%d launches)" % (k, s / max(n, 1), n))
PY
