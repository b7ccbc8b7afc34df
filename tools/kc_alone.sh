cd /tmp && export TMPDIR=/tmp
for sb in 128 256 512 1024 2048; do
  rm -rf /tmp/kc_$sb
  TF_SEL_BLOCKS=$sb rocprofv3 --kernel-trace --output-format csv -d /tmp/kc_$sb -o t -- python3 $GRAFT_REPO_ROOT/tools/host_path_probe.py > /dev/null 2>&1
  python3 - $sb <<PY
import csv,glob,sys
f=glob.glob("/tmp/kc_%s/**/t_kernel_trace.csv"%sys.argv[1],recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_frame" in r["Kernel_Name"]]
import collections
d=collections.defaultdict(list)
for r in rows: d[(r["Kernel_Name"].split("(")[0][-16:], int(r["Grid_Size_X"])//256)].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items()): print("sel_blocks", sys.argv[1], k, "n", len(v), "median %.1f us"%sorted(v)[len(v)//2])
PY
done
