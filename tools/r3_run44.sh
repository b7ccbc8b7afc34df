#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_44; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_atlas.py tests/test_gpu_textured_soak.py -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --resident-headline"
one() { python - $O/$1.json "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ev=d['roofline']['events']['kinds']
print('%-28s value %.0f' % (sys.argv[2], d['value']), {k:round(v['event_us_minus_pair'],1) for k,v in ev.items() if k in ('integrate','mesh')})
PY
}
for n in 1 2 3; do timeout 400 python bench.py $R > $O/new_$n.json 2> $O/new_$n.err; one new_$n; done
