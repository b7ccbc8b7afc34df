#!/bin/bash
# VALU / SALU / LDS instruction counts of k_mesh per phase (TF_MESH_DBG cuts the kernel short after a phase; wrong results)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for d in 1 2 3 4 0; do
  out=$R/gpurun_out/mvalu_$d; rm -rf $out; mkdir -p $out
  TF_MESH_DBG=$d rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $out -o t -- python3 bench.py --child --steps 40 --warmup 10 --resident-headline > /dev/null 2>&1
  python3 - $out $d <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
dur = [0.0, 0]
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows = rows[len(rows) // 2:]
    for r in rows:
        if "k_mesh<" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows = rows[len(rows) // 2:]
    for r in rows:
        if "k_mesh<" in r["Kernel_Name"]:
            dur[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); dur[1] += 1
print("dbg=%s  %.1f us  " % (sys.argv[2], dur[0] / max(1, dur[1]) / 1e3) + "  ".join("%s %.2fM" % (c.replace("SQ_", ""), s / n / 1e6) for c, (s, n) in sorted(acc.items())))
PY
  rm -rf $out
done
