#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_25; mkdir -p $O; rm -f $O/*
TF_PATCH_DBG=3 timeout 300 python tools/stamps.py patch > $O/stamps_patch.txt 2>&1; tail -12 $O/stamps_patch.txt
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --resident-headline"
timeout 400 python bench.py $R > $O/room.json 2> $O/room.err
timeout 400 python bench.py $R --mode tsdf > $O/tsdf.json 2> $O/tsdf.err
python - <<'PY'
import json
for n in ('room','tsdf'):
    d=json.loads(open('gpurun_out/r3_25/%s.json'%n).read().strip().splitlines()[-1])
    ev=d['roofline']['events']['kinds']
    print(n, 'value %.0f' % d['value'], {k:round(v['event_us_minus_pair'],1) for k,v in ev.items() if k in ('integrate','mesh')})
PY
