#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_5; mkdir -p $O
lscpu | grep -i "numa\|socket\|model name\|^CPU(s)" | head
B="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --no-roofline"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc $?"; }
for i in 1 2 3 4 5 6; do run pin$i TF_HOST_TRACE=1; done; ls /sys/class/drm/; cat /sys/class/drm/card*/device/numa_node
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_5/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'unreadable',e); continue
    print('%-28s host %.0f (%.1f us) resident %.0f' % (f.split('/')[-1][6:-5], d['value'], 1e3*d['ms_per_step'], (d.get('resident') or {}).get('value') or 0), d['config'].get('host_affinity'), open(f.replace('.json','.err')).read().split('staging copy')[-1][:8])
PY
