#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_3; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
B="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc $?"; }
run base TF_X=0
run noclaim TF_KA_CLAIM=0
run selfirst1792 TF_SEL_FIRST=1 TF_KAP_BLOCKS=1792
run selfirst1792_noclaim TF_SEL_FIRST=1 TF_KAP_BLOCKS=1792 TF_KA_CLAIM=0
run kfp8 TF_LIB=variants/kfp8.so
run kfp8_selfirst TF_LIB=variants/kfp8.so TF_SEL_FIRST=1
run kfp8_selfirst2304 TF_LIB=variants/kfp8.so TF_SEL_FIRST=1 TF_KAP_BLOCKS=2304
run kfp7_selfirst TF_LIB=variants/kfp7.so TF_SEL_FIRST=1
TF_LIB=variants/kfp8.so TF_SEL_FIRST=1 TF_KA_DBG=4096 timeout 300 python tools/timeline3.py > $O/timeline_kfp8_selfirst.txt 2>&1
TF_LIB=variants/kfp8.so TF_KA_DBG=4096 timeout 300 python tools/timeline3.py > $O/timeline_kfp8.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_3/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'unreadable',e); continue
    r=d.get('roofline',{})
    ev=(r.get('events') or {}).get('kinds') or {}
    print('%-28s host %.0f  resident %.0f  events %s' % (f.split('/')[-1][6:-5], d['value'], (d.get('resident') or {}).get('value') or 0, {k:round(v['event_us_minus_pair'],1) for k,v in ev.items()}))
PY
head -8 $O/timeline_kfp8_selfirst.txt; head -8 $O/timeline_kfp8.txt
