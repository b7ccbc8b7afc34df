#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3_39; mkdir -p $O; rm -rf $O/*
timeout 2700 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2; grep -n "Error\|assert" $O/tests.log | head -5
prof() { tag=$1; n=$2; shift; shift
  rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o t -- python3 bench.py --child "$@" > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name "t_kernel_trace.csv" | head -1); echo "== $tag"; python3 tools/gaps.py $f $n | head -6; rm -rf $O/$tag; }
prof room 200 --steps 200 --warmup 20 --resident-headline
prof hall 60 --scene big --hires --steps 60 --warmup 10 --resident-headline
