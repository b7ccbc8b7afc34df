#!/bin/bash
# Round-3 profile set: the bench line (with its own rocprofv3 child passes -> kernel times, roofline.traffic) and the
# rocprofv3 --kernel-trace --stats summary of the same command (run through gpurun from the repo root).
#   tools/prof_r3.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
out=$R/gpurun_out/prof_r3_$tag
mkdir -p $out
cd $R
python3 bench.py "$@" > $out/bench_line.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py "$@" --cpu-frames 0 --no-group --no-pmc > $out/bench_line_profiled.json 2>> $out/bench.err
f=$(find $out/trace -name "t_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && head -12 $out/kernel_stats.csv | cut -c1-160
rm -rf $out/trace
python3 - "$out/bench_line.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get('roofline',{})
print('value %.0f (%.1f us)  resident %s  frac %s  kernel us %s  traffic %s  alg %s' % (d['value'], 1e3*d['ms_per_step'], (d.get('resident') or {}).get('value'), r.get('frac'), r.get('kernel_us_per_step'), r.get('traffic'), r.get('algorithmic_bytes_per_step')))
print({k:round(v['us_per_step'],1) for k,v in (r.get('kernels') or {}).items()})
print('cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
