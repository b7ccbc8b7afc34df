#!/bin/bash
# the round's closing check (through gpurun): every -m gpu test, the smoke entry, the bench line with the driver's flags
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_full; mkdir -p $O; rm -f $O/*
timeout 2700 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed\|error" $O/tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > $O/driver_style.json 2> $O/driver_style.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_full/driver_style.json').read().strip().splitlines()[-1])
r=d['roofline']
print('driver flags: value %.0f (%.1f us) resident %.0f frac %.3f kernels %.1f traffic %s cpu %.1f' % (d['value'], 1e3*d['ms_per_step'], d['resident']['value'], r['frac'], r['kernel_us_per_step'], r['traffic'], d['cpu_baseline']['value']))
PY
