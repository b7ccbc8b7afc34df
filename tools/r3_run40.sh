#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3_40; mkdir -p $O; rm -rf $O/*
prof() { tag=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o t -- python3 bench.py --child --steps 100 --warmup 20 --resident-headline > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name "t_kernel_trace.csv" | head -1); echo "== $tag"; python3 tools/gaps.py $f 100 | grep "filter\|k_mesh<"; rm -rf $O/$tag; }
prof base
for k in 1 2 3; do export TF_LIB=variants/xf$k.so; prof xf$k; done
