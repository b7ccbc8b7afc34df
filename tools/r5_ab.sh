#!/bin/bash
# Round-5 A/B: quick bench (host frames + resident + HIP-event kernel times) per configuration, all in ONE job so that the
# numbers are comparable:  tools/r5_ab.sh <tag> "<ENV=.. ENV=..>" ...   ("-" = defaults; TF_LIB=variants/x.so picks a variant build)
# KT=1: also a rocprofv3 --kernel-trace --stats pass of `bench.py --child` per configuration (average duration per kernel).
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
O=$PWD/gpurun_out/r5_$TAG; mkdir -p $O
export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  [ "$cfg" = "-" ] && cfg=""
  env $cfg timeout 600 python bench.py --steps ${STEPS:-200} --warmup 20 --no-pmc --cpu-frames 0 --no-group --repeats 2 ${BENCH_ARGS:-} > $O/cfg$i.json 2> $O/cfg$i.err
  python - $O/cfg$i.json "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ev=d['roofline']['events']['kinds']
    print('%-60s host %.1f us (med %.1f) resident %.1f us | %s' % (sys.argv[2] or 'defaults', 1e3*d['ms_per_step'], 1e3*d['repeats']['ms_per_step_median'],
          1e3*d['resident']['ms_per_step'], {k:round(v['event_us_minus_pair'],1) for k,v in ev.items()}))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
  if [ -n "$KT" ]; then
    U=$O/kt$i; rm -rf $U; mkdir -p $U
    ( for kv in $cfg; do export "$kv"; done
      [ -n "$TF_LIB" ] && export TF_LIB=$(cd ${GRAFT_REPO_ROOT:-.} && readlink -f $TF_LIB)
      cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $U -o t -- python3 $OLDPWD/bench.py --child --steps ${STEPS:-200} --warmup 20 ${BENCH_ARGS:-} > $U/log 2>&1 )
    S=$(find $U -name "*kernel_stats.csv" | head -1)
    if [ -n "$S" ]; then cp $S $O/cfg${i}_kernel_stats.csv; python - $S <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('k_frame','k_mesh','k_dirty','k_patch','k_step')):
        print('    %-46s calls %6s avg %8.2f us' % (n.replace('void tf::','')[:46], r['Calls'], float(r['AverageNs'])/1e3))
PY
    fi
    rm -rf $U
  fi
done
