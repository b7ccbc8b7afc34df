#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_35; mkdir -p $O; rm -f $O/*
B="--no-pmc --cpu-frames 0 --no-group"
for n in 1 2 3 4; do
  for lf in 1 0; do
    TF_HOST_LAUNCH_FIRST=$lf python bench.py $B --steps 20 --warmup 5 > $O/k20_lf${lf}_$n.json 2> $O/k20_lf${lf}_$n.err
    TF_HOST_LAUNCH_FIRST=$lf python bench.py $B --steps 200 --warmup 20 > $O/k200_lf${lf}_$n.json 2> $O/k200_lf${lf}_$n.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_35/k*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print('%-14s host value %.0f (%.1f us)  resident %.0f' % (f.split('/')[-1][:-5], d['value'], 1e3*d['ms_per_step'], d['resident']['value']))
PY
