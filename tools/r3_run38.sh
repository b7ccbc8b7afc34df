#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3_38; mkdir -p $O; rm -rf $O/*
timeout 1500 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_atlas.py tests/test_gpu_textured_soak.py tests/test_gpu_partition.py tests/test_gpu_unit.py -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
prof() { tag=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o t -- python3 bench.py --child "$@" > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name "t_kernel_trace.csv" | head -1); echo "== $tag"; python3 tools/gaps.py $f 200 | head -5; rm -rf $O/$tag; }
prof half --steps 200 --warmup 20 --resident-headline
export TF_FILTER_HALF=0
prof wave --steps 200 --warmup 20 --resident-headline
