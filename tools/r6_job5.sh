#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r6_job5; mkdir -p $O
export TMPDIR=/tmp
for mode in "" "--no-overlap"; do
  tag=ovl; [ -n "$mode" ] && tag=ser
  U=$O/tr_$tag; rm -rf $U; mkdir -p $U
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $U -o t -- python3 $OLDPWD/bench.py --gpus 1 --force-exchange --resident-headline --child --steps 60 --warmup 20 $mode > $U/log 2>&1)
  T=$(find $U -name "*kernel_trace.csv" | head -1)
  echo "== force-exchange $tag"; python3 tools/gaps.py $T 50 | tee $O/gaps_$tag.txt
  rm -rf $U
done
