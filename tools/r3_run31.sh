#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3_31; mkdir -p $O; rm -rf $O/*
prof() { tag=$1; n=$2; shift; shift
  rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o t -- python3 bench.py --child "$@" > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name "t_kernel_trace.csv" | head -1); echo "== $tag"; python3 tools/gaps.py $f $n | head -8; rm -rf $O/$tag; }
prof hall_resident 60 --scene big --hires --steps 60 --warmup 10 --resident-headline
prof hall_host 60 --scene big --hires --steps 60 --warmup 10
