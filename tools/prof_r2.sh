#!/bin/bash
# Round-2 profile set: bench line (with its own PMC child passes -> roofline.traffic) + rocprofv3 kernel trace of the
# same command (run through gpurun from the repo root).
#   tools/prof_r2.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd $R
python3 bench.py "$@" > $out/bench_line.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py "$@" --cpu-frames 0 --no-host-path --no-pmc > $out/bench_line_profiled.json 2>> $out/bench.err
f=$(find $out/trace -name "t_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && head -12 $out/kernel_stats.csv | cut -c1-160
rm -rf $out/trace
