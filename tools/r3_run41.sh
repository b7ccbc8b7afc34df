#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_41; mkdir -p $O; rm -f $O/*
R="--steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --resident-headline"
one() { python - $O/$1.json "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ev=d['roofline']['events']['kinds']
print('%-28s value %.0f' % (sys.argv[2], d['value']), {k:round(v['event_us_minus_pair'],1) for k,v in ev.items() if k in ('integrate','mesh')})
PY
}
timeout 400 python bench.py $R > $O/base.json 2> $O/base.err; one base
# 7 waves per SIMD: 8 workgroups of K-A per CU; 5 waves: 6
TF_LIB=variants/kfp7.so TF_KAP_BLOCKS=2048 timeout 400 python bench.py $R > $O/kfp7.json 2> $O/kfp7.err; one kfp7
TF_LIB=variants/kfp7.so TF_KAP_BLOCKS=1792 timeout 400 python bench.py $R > $O/kfp7b.json 2> $O/kfp7b.err; one kfp7b
TF_LIB=variants/kfp5.so TF_KAP_BLOCKS=1536 timeout 400 python bench.py $R > $O/kfp5.json 2> $O/kfp5.err; one kfp5
TF_LIB=variants/kfp5.so TF_KAP_BLOCKS=1792 timeout 400 python bench.py $R > $O/kfp5b.json 2> $O/kfp5b.err; one kfp5b
timeout 400 python bench.py $R > $O/base2.json 2> $O/base2.err; one base2
