#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_19; mkdir -p $O; rm -f $O/*
timeout 1500 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_atlas.py tests/test_gpu_textured_soak.py tests/test_gpu_partition.py tests/test_gpu_unit.py tests/test_gpu_texmap.py -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
run() { n=$1; shift; "$@" > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc $?"; }
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --resident-headline"
H="--scene big --hires --steps 60 --warmup 10 --no-pmc --cpu-frames 0 --no-group --resident-headline"
run new timeout 400 python bench.py $R
TF_FILTER_EXACT=1 run exact timeout 400 python bench.py $R
run new2 timeout 400 python bench.py $R
TF_FILTER_EXACT=1 run exact2 timeout 400 python bench.py $R
run hall timeout 400 python bench.py $H
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_19/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'unreadable',e); continue
    r=d.get('roofline',{})
    ev=(r.get('events') or {}).get('kinds') or {}
    print('%-22s value %.0f frac %.3f events %s' % (f.split('/')[-1][6:-5], d['value'], r.get('frac',0), {k:round(v['event_us_minus_pair'],1) for k,v in ev.items()}), {k:round(v) for k,v in (r.get('per_step') or {}).items() if k in ('dirty','exact','survivors','surface','meshes')})
PY
