#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_34; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_textured_soak.py tests/test_gpu_atlas.py tests/test_gpu_parity.py tests/test_gpu_host_mirror.py -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
B="--no-pmc --cpu-frames 0 --no-group"
for n in 1 2 3; do
  python bench.py $B --steps 20 --warmup 5 > $O/k20_$n.json 2> $O/k20_$n.err
  python bench.py $B --steps 200 --warmup 20 > $O/k200_$n.json 2> $O/k200_$n.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_34/k*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print('%-10s host value %.0f (%.1f us)  resident %.0f' % (f.split('/')[-1][:-5], d['value'], 1e3*d['ms_per_step'], d['resident']['value']))
PY
