#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_51; mkdir -p $O; rm -f $O/*
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --resident-headline"
H="--scene big --hires --steps 60 --warmup 10 --no-pmc --cpu-frames 0 --no-group --resident-headline"
one() { python - $O/$1.json "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ev=d['roofline']['events']['kinds']
print('%-28s value %.0f' % (sys.argv[2], d['value']), {k:round(v['event_us_minus_pair'],1) for k,v in ev.items() if k in ('integrate','mesh')})
PY
}
for n in 1 2; do
timeout 400 python bench.py $R > $O/base_$n.json 2> $O/base_$n.err; one base_$n
TF_LIB=variants/mw6.so timeout 400 python bench.py $R > $O/mw6_$n.json 2> $O/mw6_$n.err; one mw6_$n
done
timeout 400 python bench.py $H > $O/hall_base.json 2> $O/hall_base.err; one hall_base
TF_LIB=variants/mw6.so timeout 400 python bench.py $H > $O/hall_mw6.json 2> $O/hall_mw6.err; one hall_mw6
