#!/bin/bash
# SQ counters of the step kernels (one pass, 8 SQ slots): where do the wave cycles go?
#   tools/sq_r2.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
out=$R/gpurun_out/sq_$tag
mkdir -p $out
cd $R
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES \
  --output-format csv -d $out/a -o t -- python3 bench.py "$@" --cpu-frames 0 --no-host-path --no-roofline --no-pmc > $out/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
  --output-format csv -d $out/b -o t -- python3 bench.py "$@" --cpu-frames 0 --no-host-path --no-roofline --no-pmc > $out/b.log 2>&1
python3 - $out <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("tf::", "").strip()
        if not k.startswith("k_"): continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print("   %-24s %14.0f per launch" % (c, s / n))
PY
