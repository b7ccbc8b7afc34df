#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for g in 1536 4096 8192 16384; do
  out=$R/gpurun_out/mgrid_$g; mkdir -p $out
  TF_MESH_GRID=$g rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --steps 200 --warmup 20 --cpu-frames 0 --no-host-path --no-roofline > /dev/null 2>&1
  f=$(find $out -name "t_kernel_stats.csv" | head -1)
  echo "grid=$g"; grep "k_mesh" $f | awk -F'","' '{print substr($1,1,24), $4}'
done
