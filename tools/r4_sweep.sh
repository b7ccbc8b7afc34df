#!/bin/bash
# A/B sweep of environment knobs over the quick bench (resident + host frames, no profiler child passes):
#   tools/r4_sweep.sh <tag> "<ENV=.. ENV=..>" "<ENV..>" ...     each quoted argument = one configuration ("-" = defaults)
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
O=gpurun_out/r4_$TAG; mkdir -p $O
i=0
for cfg in "$@"; do
  i=$((i+1))
  [ "$cfg" = "-" ] && cfg=""
  env $cfg timeout 600 python bench.py --steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --repeats 2 ${BENCH_ARGS:-} > $O/cfg$i.json 2> $O/cfg$i.err
  python - $O/cfg$i.json "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ev=d['roofline']['events']['kinds']
    print('%-70s host %.1f us (med %.1f) resident %.1f us | %s' % (sys.argv[2] or 'defaults', 1e3*d['ms_per_step'], 1e3*d['repeats']['ms_per_step_median'],
          1e3*d['resident']['ms_per_step'], {k:round(v['event_us_minus_pair'],1) for k,v in ev.items() if k in ('integrate','mesh')}))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
