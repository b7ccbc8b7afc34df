#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3_21; mkdir -p $O; rm -rf $O/*
prof() { tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o t -- python3 bench.py --child "$@" > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name "t_kernel_stats.csv" | head -1); cp $f $O/${tag}_stats.csv; rm -rf $O/$tag
  python3 - $O/${tag}_stats.csv $tag <<'PY'
import csv,sys
print('==',sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'k_mesh' in n or 'k_frame<true, true' in n:
        print(' ',n[:40], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['MinNs'], r['MaxNs'])
PY
}
prof new --steps 100 --warmup 20
for v in NO_SFL NO_STORE; do
  export TF_LIB=variants/x_$v.so
  prof $v --steps 100 --warmup 20
done
unset TF_LIB
export TF_FILTER_EXACT=1
prof exact --steps 100 --warmup 20
unset TF_FILTER_EXACT
prof hall --scene big --hires --steps 40 --warmup 10
