#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_42; mkdir -p $O; rm -f $O/*
B="--no-pmc --cpu-frames 0 --no-group --no-roofline"
for n in 1 2 3 4 5 6; do
  for r in 8 6 7; do
    TF_HOST_RING=$r python bench.py $B --steps 20 --warmup 5 > $O/k20_r${r}_$n.json 2> $O/k20_r${r}_$n.err
  done
done
for n in 1 2; do
  for r in 8 6; do
    TF_HOST_RING=$r python bench.py $B --steps 200 --warmup 20 > $O/k200_r${r}_$n.json 2> $O/k200_r${r}_$n.err
  done
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r3_42/k*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    acc[f.split('/')[-1].rsplit('_',1)[0]].append(1e3*d['ms_per_step'])
for k,v in sorted(acc.items()):
    print('%-12s us/frame:' % k, ' '.join('%.1f' % x for x in v))
PY
