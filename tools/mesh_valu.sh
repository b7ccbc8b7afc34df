#!/bin/bash
# VALU / SALU / LDS instruction counts of k_mesh per phase (TF_MESH_DBG cuts the kernel short after a phase)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for d in 1 2 3 4 0; do
  out=$R/gpurun_out/mvalu_$d; mkdir -p $out
  TF_MESH_DBG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $out -o t -- python3 bench.py --steps 30 --warmup 10 --cpu-frames 0 --no-host-path --no-roofline --no-pmc > /dev/null 2>&1
  python3 - $out $d <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mesh<" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("dbg=%s " % sys.argv[2] + "  ".join("%s %.2fM" % (c.replace("SQ_", ""), s / n / 1e6) for c, (s, n) in sorted(acc.items())))
PY
done
