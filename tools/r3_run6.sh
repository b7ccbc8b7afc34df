#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_6; mkdir -p $O
timeout 900 python bench.py --steps 200 --warmup 20 > $O/bench_full.json 2> $O/bench_full.err; echo "full rc $?"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-like rc $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_6/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d.get('roofline',{})
    print(f.split('/')[-1], 'value %.0f ms %.4f resident %s frac %.4f kus %.1f wall %.1f traffic %s alg %.1f MB' % (d['value'], d['ms_per_step'], (d.get('resident') or {}).get('value'), r.get('frac'), r.get('kernel_us_per_step'), r.get('wall_us_per_step'), r.get('traffic'), r.get('algorithmic_bytes_per_step')/1e6))
    print('   src', r.get('kernel_time_source'))
    print('   trace:', {k:round(v['us_per_step'],1) for k,v in (r.get('kernels') or {}).items()})
    print('   groups:', {k:(round(v['us_per_step'],1), round(v['achieved_GBs'])) for k,v in (r.get('groups') or {}).items()})
    print('   events:', {k:round(v['event_us_minus_pair'],1) for k,v in r['events']['kinds'].items()}, 'pair', round(r['events']['pair_us'],2))
    td=r.get('traffic_detail') or {}
    print('   pmc:', {k:round(v.get('bytes_per_step',0)/1e6,1) for k,v in (td.get('kernels') or {}).items()}, td.get('error'))
    print('   counts:', {k:round(v) for k,v in r['per_step'].items()})
    print('   cpu', d.get('cpu_baseline',{}).get('value'), 'kfgroup', (d.get('keyframe_group') or {}).get('kernel_frames_per_s'))
PY
tail -3 $O/bench_full.err
