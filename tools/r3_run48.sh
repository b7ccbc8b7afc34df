#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_48; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_textured_soak.py tests/test_gpu_atlas.py tests/test_gpu_parity.py tests/test_gpu_host_mirror.py -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
B="--no-pmc --cpu-frames 0 --no-group --no-roofline"
for n in 1 2 3; do
  TF_HOST_TRACE=1 python bench.py $B --steps 200 --warmup 20 > $O/k200_$n.json 2> $O/k200_$n.err
  grep "tf host frames" $O/k200_$n.err | tail -1 | sed "s/.*copies a launch/textured waits/"
  TF_HOST_TRACE=1 python bench.py $B --steps 200 --warmup 20 --mode tsdf > $O/tsdf_$n.json 2> $O/tsdf_$n.err
  grep "tf host frames" $O/tsdf_$n.err | tail -1 | sed "s/.*copies a launch/tsdf waits/"
done
for n in 1 2 3 4 5 6; do python bench.py $B --steps 20 --warmup 5 > $O/k20_$n.json 2> $O/k20_$n.err; done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r3_48/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    acc[f.split('/')[-1].rsplit('_',1)[0]].append(1e3*d['ms_per_step'])
for k,v in sorted(acc.items()):
    print('%-12s us/frame:' % k, ' '.join('%.1f' % x for x in v))
PY
