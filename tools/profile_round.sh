#!/bin/bash
# Final profile set of a round on the GPU box: tools/profile_round.sh <prefix>  ->  gpurun_out/final/<prefix>_*
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O; P=$1
python3 $R/bench.py > $O/${P}_bench_line.json 2> $O/${P}_bench_stderr.txt
python3 $R/bench.py --steps 400 --warmup 400 --unique-frames 60 --cpu-frames 0 > $O/${P}_bench_line_steady_state.json 2>/dev/null
python3 $R/bench.py --atlas-every 10 --cpu-frames 0 --no-roofline > $O/${P}_bench_line_atlas_every_10.json 2>/dev/null
python3 $R/bench.py --hires --cpu-frames 0 > $O/${P}_bench_line_hires.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -o ${P} -- python3 $R/bench.py --cpu-frames 0 >/dev/null 2>&1
find $O/rp -name "${P}_kernel_stats.csv" -exec cp {} $O/${P}_k_frame_kernel_stats.csv \;
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/rp -o ${P}pa -- python3 $R/bench.py --steps 60 --warmup 120 --unique-frames 60 --cpu-frames 0 --no-roofline >/dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $O/rp -o ${P}pb -- python3 $R/bench.py --steps 60 --warmup 120 --unique-frames 60 --cpu-frames 0 --no-roofline >/dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/rp -o ${P}pf -- python3 $R/bench.py --steps 60 --warmup 120 --unique-frames 60 --cpu-frames 0 --no-roofline >/dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/rp -o ${P}pw -- python3 $R/bench.py --steps 60 --warmup 120 --unique-frames 60 --cpu-frames 0 --no-roofline >/dev/null 2>&1
python3 - <<PY
import csv, glob, collections
out = open("$O/${P}_pmc_k_frame.csv", "w")
out.write("counter,per_launch_avg,launches\n")
for tag in ("pa", "pb", "pf", "pw"):
    fs = sorted(glob.glob("$O/rp/**/${P}%s_counter_collection.csv" % tag, recursive=True))
    if not fs:
        continue
    acc = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(fs[-1])):
        if "k_frame" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
    for k, (s, d) in sorted(acc.items()):
        out.write("%s,%.0f,%d\n" % (k, s / max(len(d), 1), len(d)))
out.close()
print(open("$O/${P}_pmc_k_frame.csv").read())
PY
TF_KA_DBG=4096 REPS=2 python3 $R/tools/timeline.py > $O/${P}_wave_timeline.txt 2>/dev/null
head -c 1500 $O/${P}_bench_line.json; echo; cat $O/${P}_bench_line_steady_state.json | head -c 400; echo; cat $O/${P}_bench_line_atlas_every_10.json | head -c 300; echo; cat $O/${P}_bench_line_hires.json | head -c 600; echo; head -8 $O/${P}_k_frame_kernel_stats.csv
