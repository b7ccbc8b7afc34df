#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_24; mkdir -p $O; rm -f $O/*
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --resident-headline"
H="--scene big --hires --steps 60 --warmup 10 --no-pmc --cpu-frames 0 --no-group --resident-headline"
one() { tag=$1; shift
  python - $O/$tag.json "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ev=d['roofline']['events']['kinds']
print('%-28s value %.0f' % (sys.argv[2], d['value']), {k:round(v['event_us_minus_pair'],1) for k,v in ev.items() if k in ('integrate','mesh')})
PY
}
for cfg in "1024 64" "1024 96" "1024 128" "1024 192" "768 64" "768 96" "768 192" "896 128"; do
  set -- $cfg
  TF_PATCH_BLOCKS=$1 TF_SEL_BLOCKS=$2 timeout 400 python bench.py $R > $O/room_$1_$2.json 2> $O/room_$1_$2.err; one room_$1_$2
done
for s in 64 128 256 512; do
  TF_SEL_BLOCKS=$s timeout 400 python bench.py $R --mode tsdf > $O/tsdf_$s.json 2> $O/tsdf_$s.err; one tsdf_$s
  TF_SEL_BLOCKS=$s timeout 400 python bench.py $H > $O/hall_$s.json 2> $O/hall_$s.err; one hall_$s
done
