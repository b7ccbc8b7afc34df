#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3_32; mkdir -p $O; rm -rf $O/*
prof() { tag=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o t -- python3 bench.py --child "$@" > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name "t_kernel_trace.csv" | head -1); echo "== $tag"; python3 tools/gaps.py $f 200 | head -6; rm -rf $O/$tag; }
prof base --steps 200 --warmup 20
export TF_HOST_EVENT_NOFENCE=1
prof nofence --steps 200 --warmup 20
unset TF_HOST_EVENT_NOFENCE
export HSA_ENABLE_SDMA=0
prof nosdma --steps 200 --warmup 20
