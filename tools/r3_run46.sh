#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3_46; mkdir -p $O; rm -rf $O/*
B="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --no-roofline"
for lf in 1 0 1 0; do
  TF_HOST_TRACE=1 TF_HOST_LAUNCH_FIRST=$lf python bench.py $B > $O/lf$lf.json 2> $O/lf$lf.err
  python - $O/lf$lf.json $lf <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('launch_first=%s host value %.0f (%.1f us)' % (sys.argv[2], d['value'], 1e3*d['ms_per_step']))
PY
  grep "tf host frames" $O/lf$lf.err | tail -1
done
prof() { tag=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o t -- python3 bench.py --child --steps 200 --warmup 20 > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name "t_kernel_trace.csv" | head -1); echo "== $tag"; python3 tools/gaps.py $f 200 | head -5; rm -rf $O/$tag; }
export TF_HOST_LAUNCH_FIRST=1; prof lf1
export TF_HOST_LAUNCH_FIRST=0; prof lf0
