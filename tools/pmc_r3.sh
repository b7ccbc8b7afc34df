#!/bin/bash
# SQ counters of the three per-frame kernels on the timed workload (bench.py --child): tools/pmc_r3.sh
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc3; mkdir -p $O; rm -rf $O/*
B="python3 $R/bench.py --child --steps 60 --warmup 20"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O -o a -- $B >/dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $O -o b -- $B >/dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_FLAT --output-format csv -d $O -o c -- $B >/dev/null 2>&1
python3 - <<PY > $O/summary.txt
import csv, glob, collections
for kern in ("k_frame", "k_mesh_filter", "k_mesh<"):
    print(kern)
    for f in sorted(glob.glob("$O/**/*_counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(lambda: [0.0, 0])
        rows = list(csv.DictReader(open(f)))
        rows = rows[len(rows) // 2:]   # the later half: steady state
        for r in rows:
            if kern in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (s, n) in sorted(acc.items()):
            print("  %-22s per launch %14.0f  (%d launches)" % (k, s / max(n, 1), n))
PY
cat $O/summary.txt
