#!/bin/bash
# HBM-traffic counters, collected as MI355X_MICROARCH.md prescribes: one rocprofv3 --pmc pass per counter, no trace
# domains in the same run, the program itself after "--".  Run through gpurun from the repo root:
#   tools/pmc_r2.sh <tag> [bench args...]
# Leaves gpurun_out/pmc_<tag>/{fetch,write,calib_fetch,calib_write}/..._counter_collection.csv; tools/pmc_summary.py
# turns them into per-kernel bytes per launch with the calibration applied.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  d=$out/$(echo $c | tr A-Z a-z | sed 's/_size//')
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $d -o t -- python3 bench.py "$@" --cpu-frames 0 --no-host-path --no-roofline > $d.log 2>&1
  timeout 300 rocprofv3 --pmc $c --output-format csv -d ${d/pmc_$tag\//pmc_$tag\/calib_} -o t -- tools/calib_fetch > $out/calib_$(basename $d).log 2>&1
done
find $out -name "*counter_collection.csv" | xargs ls -la
python3 tools/pmc_summary.py $out | tee $out/summary.json
