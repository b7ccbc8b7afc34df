#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_47; mkdir -p $O; rm -f $O/*
B="--no-pmc --cpu-frames 0 --no-group --no-roofline"
for n in 1 2 3; do
  for lf in 1 0; do
    TF_HOST_TRACE=1 TF_HOST_LAUNCH_FIRST=$lf python bench.py $B --steps 200 --warmup 20 > $O/k200_lf${lf}_$n.json 2> $O/k200_lf${lf}_$n.err
    grep "tf host frames" $O/k200_lf${lf}_$n.err | tail -1 | sed "s/.*copies a launch/lf$lf waits/"
  done
done
for n in 1 2 3 4 5 6; do
  for lf in 1 0; do
    TF_HOST_LAUNCH_FIRST=$lf python bench.py $B --steps 20 --warmup 5 > $O/k20_lf${lf}_$n.json 2> $O/k20_lf${lf}_$n.err
  done
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r3_47/k*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    acc[f.split('/')[-1].rsplit('_',1)[0]].append(1e3*d['ms_per_step'])
for k,v in sorted(acc.items()):
    print('%-12s us/frame:' % k, ' '.join('%.1f' % x for x in v))
PY
