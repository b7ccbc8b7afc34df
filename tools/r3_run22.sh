#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_22; mkdir -p $O; rm -f $O/*
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -3
run() { n=$1; shift; "$@" > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc $?"; }
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --resident-headline"
H="--scene big --hires --steps 60 --warmup 10 --no-pmc --cpu-frames 0 --no-group --resident-headline"
run room timeout 400 python bench.py $R
run hall timeout 400 python bench.py $H
run tsdf timeout 400 python bench.py $R --mode tsdf
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_22/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'unreadable',e); continue
    r=d.get('roofline',{})
    ev=(r.get('events') or {}).get('kinds') or {}
    print('%-22s value %.0f frac %.3f events %s' % (f.split('/')[-1][6:-5], d['value'], r.get('frac',0), {k:round(v['event_us_minus_pair'],1) for k,v in ev.items()}))
PY
