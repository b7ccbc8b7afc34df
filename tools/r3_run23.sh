#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_23; mkdir -p $O; rm -f $O/*
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --resident-headline"
for cfg in "1024 512" "768 512" "512 512" "384 512" "256 512" "1024 256" "512 256" "768 128" "1024 512"; do
  set -- $cfg
  TF_PATCH_BLOCKS=$1 TF_SEL_BLOCKS=$2 timeout 400 python bench.py $R > $O/b_$1_$2.json 2> $O/b_$1_$2.err
  python - $O/b_$1_$2.json "$cfg" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ev=d['roofline']['events']['kinds']
print('patch/sel blocks %-10s value %.0f' % (sys.argv[2], d['value']), {k:round(v['event_us_minus_pair'],1) for k,v in ev.items() if k in ('integrate','mesh')})
PY
done
