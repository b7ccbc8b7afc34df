#!/bin/bash
# vector / scalar memory pipeline counters of the fused kernel: tools/pmc_mem.sh <tag>
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/pmc
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 120 --unique-frames 60 --cpu-frames 0 --no-roofline"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --output-format csv -d $O -o $1m1 -- $B >/dev/null 2>&1
echo "rc=$?"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQC_DCACHE_BUSY_CYCLES SQC_TC_STALL SQ_BUSY_CU_CYCLES --output-format csv -d $O -o $1m2 -- $B >/dev/null 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/**/$1m[12]_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_frame" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in sorted(acc.items()):
        print("%-30s per launch %14.0f  (%d launches)" % (k, s / max(n, 1), n))
PY
