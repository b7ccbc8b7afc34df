#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_45; mkdir -p $O; rm -f $O/*
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --resident-headline"
one() { python - $O/$1.json "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ev=d['roofline']['events']['kinds']
print('%-28s value %.0f' % (sys.argv[2], d['value']), {k:round(v['event_us_minus_pair'],1) for k,v in ev.items() if k in ('integrate','mesh')})
PY
}
for n in 1 2; do
for v in p00 p10 p01; do TF_LIB=variants/$v.so timeout 400 python bench.py $R > $O/${v}_$n.json 2> $O/${v}_$n.err; one ${v}_$n; done
timeout 400 python bench.py $R > $O/p11_$n.json 2> $O/p11_$n.err; one p11_$n
done
