#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_16; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_atlas.py tests/test_gpu_textured_soak.py tests/test_gpu_partition.py tests/test_gpu_unit.py tests/test_gpu_texmap.py -m gpu -x -q 2>&1 | tail -5
run() { n=$1; shift; "$@" > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc $?"; }
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --resident-headline"
run fused timeout 400 python bench.py $R
TF_MESH_FUSED=0 run twolaunch timeout 400 python bench.py $R
run fused2 timeout 400 python bench.py $R
TF_MESH_FUSED=0 run twolaunch2 timeout 400 python bench.py $R
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_16/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'unreadable',e); continue
    r=d.get('roofline',{})
    ev=(r.get('events') or {}).get('kinds') or {}
    print('%-22s value %.0f  events %s' % (f.split('/')[-1][6:-5], d['value'], {k:round(v['event_us_minus_pair'],1) for k,v in ev.items()}))
PY
