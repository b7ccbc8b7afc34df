#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_33; mkdir -p $O; rm -f $O/*
B="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group"
for n in 1 2 3 4 5 6; do
  TF_HOST_TRACE=1 python bench.py $B --mode tsdf > $O/tsdf_$n.json 2> $O/tsdf_$n.err
  python - $O/tsdf_$n.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('tsdf host value %.0f (%.1f us)  resident %.0f' % (d['value'], 1e3*d['ms_per_step'], d['resident']['value']))
PY
  grep "tf host frames" $O/tsdf_$n.err | tail -1
done
for n in 1 2 3; do
  TF_HOST_TRACE=1 python bench.py $B > $O/tex_$n.json 2> $O/tex_$n.err
  python - $O/tex_$n.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('tex host value %.0f (%.1f us)  resident %.0f' % (d['value'], 1e3*d['ms_per_step'], d['resident']['value']))
PY
  grep "tf host frames" $O/tex_$n.err | tail -1
done
