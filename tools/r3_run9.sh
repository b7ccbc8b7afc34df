#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_9; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -30 $O/pytest.log | grep -v "^$" | tail -25
B="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group"
timeout 300 python bench.py $B > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_9/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; ev=r['events']['kinds']
print('host %.0f resident %.0f' % (d['value'], d['resident']['value']), {k:round(v['event_us_minus_pair'],1) for k,v in ev.items()})
PY
