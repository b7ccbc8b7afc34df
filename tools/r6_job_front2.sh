#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/job_front; mkdir -p $O; export TMPDIR=/tmp
for e in 0 1 0 1; do
export TF_UNIT_SERIAL_FRONT=$e
rm -rf $O/trace
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $OLDPWD/tools/prof_unit.py --run > $O/trace.log 2>&1)
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - $T $e <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
g=[r for r in rows if 'k_integrate_group' in r['Kernel_Name']]
g=g[-20:]
per=[(int(b['Start_Timestamp'])-int(a['Start_Timestamp']))/1e3 for a,b in zip(g,g[1:])]
t_lo=int(g[0]['Start_Timestamp'])
d=collections.defaultdict(list)
for r in rows:
    if int(r['Start_Timestamp'])>=t_lo: d[r['Kernel_Name'].replace('void tf::','')[:28]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('serial_front=%s period avg %.1f us (min %.1f max %.1f) | '%(sys.argv[2],sum(per)/len(per),min(per),max(per))+', '.join('%s %.1f'%(k,sum(v)/len(v)) for k,v in d.items() if len(v)>5))
# gaps
ks=[r for r in rows if int(r['Start_Timestamp'])>=t_lo and any(k in r['Kernel_Name'] for k in ('k_integrate_group','k_mesh'))]
gap=collections.defaultdict(list)
for a,b in zip(ks,ks[1:]):
    gap[a['Kernel_Name'].replace('void tf::','')[:14]+'->'+b['Kernel_Name'].replace('void tf::','')[:14]].append((int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3)
print('   gaps: '+', '.join('%s %.1f'%(k,sum(v)/len(v)) for k,v in gap.items()))
PY
done
rm -rf $O/trace
