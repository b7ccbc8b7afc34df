#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3_49; mkdir -p $O; rm -rf $O/*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python3 bench.py --steps 20 --warmup 5 --no-pmc --cpu-frames 0 > $O/line.json 2> $O/err.txt
f=$(find $O/t -name "t_kernel_stats.csv" | head -1); cp $f $O/stats.csv; rm -rf $O/t
python3 - <<'PY'
import csv,json
rows=list(csv.DictReader(open('gpurun_out/r3_49/stats.csv')))
for r in rows[:40]:
    print('%-60s calls %5s avg %8.1f us total %8.1f ms' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
d=json.loads(open('gpurun_out/r3_49/line.json').read().strip().splitlines()[-1])
print(d.get('keyframe_unit'))
PY
