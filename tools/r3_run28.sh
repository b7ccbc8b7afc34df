#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_28; mkdir -p $O; rm -f $O/*
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group"
one() { python - $O/$1.json "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-20s host value %.0f (%.1f us)  resident %.0f' % (sys.argv[2], d['value'], 1e3*d['ms_per_step'], d['resident']['value']))
PY
}
for n in 1 4 1 4 8; do TF_HOST_EVENT_EVERY=$n timeout 400 python bench.py $R > $O/ev_$n.json 2> $O/ev_$n.err; one ev_$n; done
