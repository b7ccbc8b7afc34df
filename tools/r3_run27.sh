#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_27; mkdir -p $O; rm -f $O/*
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group"
one() { python - $O/$1.json "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-20s host value %.0f (%.1f us)  resident %.0f' % (sys.argv[2], d['value'], 1e3*d['ms_per_step'], d['resident']['value']))
PY
}
TF_HOST_TRACE=1 timeout 400 python bench.py $R > $O/base.json 2> $O/base.err; one base; grep -i "host trace\|trace" $O/base.err | tail -3
TF_HOST_TRACE=1 TF_HOST_NOH2D=1 timeout 400 python bench.py $R > $O/noh2d.json 2> $O/noh2d.err; one noh2d; grep -i "trace" $O/noh2d.err | tail -3
TF_HOST_TRACE=1 TF_COPY_THREADS=0 timeout 400 python bench.py $R > $O/nohelpers.json 2> $O/nohelpers.err; one nohelpers; grep -i "trace" $O/nohelpers.err | tail -3
timeout 400 python bench.py $R > $O/base2.json 2> $O/base2.err; one base2
