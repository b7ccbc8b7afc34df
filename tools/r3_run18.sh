#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_18; mkdir -p $O; rm -f $O/*
run() { n=$1; shift; "$@" > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc $?"; }
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --resident-headline"
run base timeout 400 python bench.py $R
TF_LIB=variants/noexact.so run noexact timeout 400 python bench.py $R
TF_LIB=variants/mw6.so run mw6 timeout 400 python bench.py $R
run base2 timeout 400 python bench.py $R
TF_LIB=variants/noexact.so run noexact2 timeout 400 python bench.py $R
TF_LIB=variants/mw6.so run mw6b timeout 400 python bench.py $R
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_18/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'unreadable',e); continue
    r=d.get('roofline',{})
    ev=(r.get('events') or {}).get('kinds') or {}
    print('%-22s value %.0f  events %s' % (f.split('/')[-1][6:-5], d['value'], {k:round(v['event_us_minus_pair'],1) for k,v in ev.items()}))
PY
