#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_26; mkdir -p $O; rm -f $O/*
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --resident-headline"
one() { python - $O/$1.json "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ev=d['roofline']['events']['kinds']
print('%-28s value %.0f' % (sys.argv[2], d['value']), {k:round(v['event_us_minus_pair'],1) for k,v in ev.items() if k in ('integrate','mesh')})
PY
}
for k in 1280 1536 1792 2048 2304; do TF_KAP_BLOCKS=$k timeout 400 python bench.py $R > $O/kap_$k.json 2> $O/kap_$k.err; one kap_$k; done
for k in 32 64 256; do TF_BBOX_BLOCKS=$k timeout 400 python bench.py $R > $O/bbox_$k.json 2> $O/bbox_$k.err; one bbox_$k; done
for k in 3072 8192; do TF_MESH_GRID=$k timeout 400 python bench.py $R > $O/mgrid_$k.json 2> $O/mgrid_$k.err; one mgrid_$k; done
