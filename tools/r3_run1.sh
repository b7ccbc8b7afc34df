#!/bin/bash
# round 3, GPU call 1: parity suite, then the patch-stage deferral A/B (tools/r3_run1.sh on the GPU box)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_1; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
B="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group"
for v in base kfp5 kfp7; do
  TF_LIB=variants/$v.so timeout 300 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc $?"
done
TF_PATCH_DEFER=0 timeout 300 python bench.py $B > $O/bench_nodefer.json 2> $O/bench_nodefer.err; echo "nodefer rc $?"
TF_HOST_TRACE=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-roofline --cpu-frames 0 > $O/bench_hosttrace.json 2> $O/bench_hosttrace.err
TF_KA_DBG=4096 timeout 300 python tools/timeline3.py > $O/timeline.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-like rc $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_1/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'unreadable',e); continue
    r=d.get('roofline',{})
    print(f.split('/')[-1], 'value %.0f ms %.4f resident %s frac %s kus %s src %s' % (d['value'], d['ms_per_step'], (d.get('resident') or {}).get('value'), r.get('frac'), r.get('kernel_us_per_step'), (r.get('kernel_time_source') or '')[:20]))
    ev=(r.get('events') or {}).get('kinds') or {}
    print('   events:', {k:round(v['event_us_minus_pair'],1) for k,v in ev.items()}, 'pair', (r.get('events') or {}).get('pair_us'))
    if r.get('kernels'): print('   trace:', {k:round(v['us_per_step'],1) for k,v in r['kernels'].items()})
    if r.get('per_step'): print('   counts:', {k:round(v) for k,v in r['per_step'].items()})
PY
tail -5 $O/bench_hosttrace.err
cat $O/timeline.txt | head -30
