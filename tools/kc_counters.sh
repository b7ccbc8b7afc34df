#!/bin/bash
# SQ counters of the selection-only launches (k_frame<false>) of the un-primed per-call flow, for two grid sizes
cd /tmp && export TMPDIR=/tmp
for sb in 512 2048; do
  rm -rf /tmp/kcc_$sb
  TF_SEL_BLOCKS=$sb rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d /tmp/kcc_$sb -o t -- python3 $GRAFT_REPO_ROOT/tools/host_path_probe.py > /dev/null 2>&1
  python3 - $sb <<PY
import csv,glob,sys,collections
f=glob.glob("/tmp/kcc_%s/**/t_counter_collection.csv"%sys.argv[1],recursive=True)[0]
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "k_frame<false>" in r["Kernel_Name"]:
        d[int(r["Grid_Size"])//256][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g in sorted(d):
    print("sel_blocks", sys.argv[1], "grid", g, {k: round(sum(v)/len(v)) for k,v in sorted(d[g].items())})
PY
done
