#!/bin/bash
# One parameterised GPU job script for round 4 (replaces the r3_run*.sh one-offs):
#   tools/r4_run.sh <tag> <step> [<step> ...]      steps: tests | testsx (stop at first failure) | bench | tsdf | hall |
#                                                  prof (rocprofv3 --kernel-trace --stats of the default bench) | kf (keyframe unit)
# Output under gpurun_out/r4_<tag>/.
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
O=gpurun_out/r4_$TAG; mkdir -p $O
export TMPDIR=/tmp
for step in "$@"; do
  case $step in
    tests)  timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -5 $O/pytest.log ;;
    testsx) timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -30 $O/pytest.log ;;
    bench)  timeout 900 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3000 $O/bench.json ;;
    drv)    timeout 900 python bench.py --steps 20 --warmup 5 > $O/drv.json 2> $O/drv.err; echo "drv rc=$?"; tail -c 1500 $O/drv.json ;;
    tsdf)   timeout 900 python bench.py --mode tsdf --steps 200 --warmup 20 > $O/tsdf.json 2> $O/tsdf.err; echo "tsdf rc=$?"; tail -c 1500 $O/tsdf.json ;;
    hall)   timeout 900 python bench.py --scene big --hires --steps 60 --warmup 10 > $O/hall.json 2> $O/hall.err; echo "hall rc=$?"; tail -c 1500 $O/hall.json ;;
    quick)  timeout 600 python bench.py --steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group > $O/quick.json 2> $O/quick.err; echo "quick rc=$?"; tail -c 2500 $O/quick.json ;;
    prof)   (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o prof -- python3 $OLDPWD/bench.py --steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group > $OLDPWD/$O/prof.log 2>&1); echo "prof rc=$?"; ls $O/prof* | head ;;
    unit|unit_moved)
            # the keyframe unit under rocprofv3: kernel trace, FETCH_SIZE, WRITE_SIZE (separate passes), exact counts, summary
            MV=""; [ $step = unit_moved ] && MV="--moved"
            U=$PWD/$O/$step; mkdir -p $U
            python3 tools/prof_unit.py --run $MV > $U/run_line.json 2> $U/run.err
            python3 tools/prof_unit.py --count $MV > $U/counts.json 2> $U/count.err
            (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $U/trace -o t -- python3 $OLDPWD/tools/prof_unit.py --run $MV > $U/trace.log 2>&1)
            (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $U/fetch -o t -- python3 $OLDPWD/tools/prof_unit.py --run $MV > $U/fetch.log 2>&1)
            (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $U/write -o t -- python3 $OLDPWD/tools/prof_unit.py --run $MV > $U/write.log 2>&1)
            T=$(find $U/trace -name "*kernel_trace.csv" | head -1); F=$(find $U/fetch -name "*counter_collection.csv" | head -1); W=$(find $U/write -name "*counter_collection.csv" | head -1)
            S=$(find $U/trace -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $U/kernel_stats.csv
            python3 tools/prof_unit.py --summarize "$T" "$F" "$W" $U/counts.json > $U/summary.json 2> $U/summary.err
            rm -rf $U/trace $U/fetch $U/write
            echo "$step rc=$?"; cat $U/run_line.json; tail -c 1800 $U/summary.json; tail -3 $U/summary.err $U/count.err ;;
    *)      echo "unknown step $step" ;;
  esac
done
