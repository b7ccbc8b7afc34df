#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3_30; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests/test_gpu_textured_soak.py tests/test_gpu_atlas.py tests/test_gpu_parity.py -m gpu -x -q -k "host or deferr or stretch" > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
prof() { tag=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o t -- python3 bench.py --child "$@" > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name "t_kernel_trace.csv" | head -1); echo "== $tag"; python3 tools/gaps.py $f 200 | head -5; rm -rf $O/$tag; }
prof host --steps 200 --warmup 20
prof host_tsdf --steps 200 --warmup 20 --mode tsdf
B="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group"
for n in a b; do python bench.py $B > $O/room_$n.json 2> $O/room_$n.err; python bench.py $B --mode tsdf > $O/tsdf_$n.json 2> $O/tsdf_$n.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_30/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print('%-12s host value %.0f (%.1f us)  resident %.0f' % (f.split('/')[-1][:-5], d['value'], 1e3*d['ms_per_step'], d['resident']['value']))
PY
