#!/bin/bash
# TA back-pressure counters for every variants/*.so
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/pmc
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 120 --unique-frames 60 --cpu-frames 0 --no-roofline"
for f in $GRAFT_REPO_ROOT/variants/*.so; do
  n=$(basename $f .so); cp $f $GRAFT_REPO_ROOT/texturefusion_amd/libtexfusion_hip.so
  timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O -o v_$n -- $B >/dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$O/**/v_${n}_counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if "k_frame" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
g = lambda k: acc[k][0] / max(acc[k][1], 1)
print("%-14s VMEM %7.0f  level/inst %5.1f  addr_full %8.0f  cmd_full %8.0f  busy_cu %9.0f  wait_inst %9.0f  wave_cyc %9.0f" % (
    "$n", g("SQ_INSTS_VMEM"), g("SQ_INST_LEVEL_VMEM") / max(g("SQ_INSTS_VMEM"), 1), g("SQ_VMEM_TA_ADDR_FIFO_FULL"),
    g("SQ_VMEM_TA_CMD_FIFO_FULL"), g("SQ_BUSY_CU_CYCLES"), g("SQ_WAIT_INST_ANY"), g("SQ_WAVE_CYCLES")))
PY
done
