#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_7; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_atlas.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
B="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc $?"; }
run base TF_X=0
run mesh6 TF_LIB=variants/mesh6.so
run base_b TF_X=0
run mesh6_b TF_LIB=variants/mesh6.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_7/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'unreadable',e); continue
    r=d.get('roofline',{})
    ev=(r.get('events') or {}).get('kinds') or {}
    print('%-28s host %.0f  resident %.0f  events %s' % (f.split('/')[-1][6:-5], d['value'], (d.get('resident') or {}).get('value') or 0, {k:round(v['event_us_minus_pair'],1) for k,v in ev.items()}))
PY
