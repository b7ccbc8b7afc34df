#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3_20; mkdir -p $O; rm -rf $O/*
prof() { tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o t -- python3 bench.py --child "$@" > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name "t_kernel_stats.csv" | head -1); echo "== $tag"; head -6 $f | cut -d, -f1-8 | cut -c1-200; cp $f $O/${tag}_stats.csv; rm -rf $O/$tag; }
prof new --steps 100 --warmup 20
export TF_FILTER_EXACT=1
prof exact --steps 100 --warmup 20
unset TF_FILTER_EXACT
prof hall --scene big --hires --steps 40 --warmup 10
