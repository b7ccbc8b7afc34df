#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_8; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_atlas.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | head -2
B="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc $?"; }
run dyn TF_X=0
run static TF_LIB=variants/static.so
run dyn_kafirst TF_SEL_FIRST=0
run dyn_kap2048 TF_KAP_BLOCKS=2048
run dyn_kap1536 TF_KAP_BLOCKS=1536
run dyn_tsdf TF_X=0
env timeout 300 python bench.py --mode tsdf $B > $O/bench_tsdf_dyn.json 2> $O/bench_tsdf_dyn.err
env TF_LIB=variants/static.so timeout 300 python bench.py --mode tsdf $B > $O/bench_tsdf_static.json 2> $O/bench_tsdf_static.err
TF_KA_DBG=4096 timeout 300 python tools/timeline3.py > $O/timeline_dyn.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_8/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'unreadable',e); continue
    r=d.get('roofline',{})
    ev=(r.get('events') or {}).get('kinds') or {}
    print('%-28s host %.0f  resident %.0f  events %s' % (f.split('/')[-1][6:-5], d['value'], (d.get('resident') or {}).get('value') or 0, {k:round(v['event_us_minus_pair'],1) for k,v in ev.items()}))
PY
head -14 $O/timeline_dyn.txt
