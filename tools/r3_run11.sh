#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_11; mkdir -p $O
timeout 600 python bench.py --steps 100 --warmup 10 --no-pmc --cpu-frames 0 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_11/bench.json').read().strip().splitlines()[-1])
print('host %.0f resident %.0f' % (d['value'], d['resident']['value']))
print(json.dumps(d.get('keyframe_unit'), indent=1)[:1500])
print({k:v for k,v in d.get('keyframe_group',{}).items() if k!='note'})
PY
