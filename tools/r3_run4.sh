#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
B="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --no-roofline"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc $?"; }
run base TF_HOST_TRACE=1
run base2 TF_HOST_TRACE=1
run alwayswait TF_HOST_ALWAYS_WAIT=1 TF_HOST_TRACE=1
run copy7 TF_COPY_THREADS=7 TF_HOST_TRACE=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_4/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'unreadable',e); continue
    print('%-28s host %.0f (%.1f us, enqueue %.1f us) resident %.0f' % (f.split('/')[-1][6:-5], d['value'], 1e3*d['ms_per_step'], 1e3*d['host_enqueue_ms_per_step'], (d.get('resident') or {}).get('value') or 0))
PY
grep -h "tf host frames" $O/*.err
