#!/bin/bash
# One parameterised GPU job script:
#   tools/gpu_job.sh <tag> <step> [<step> ...]      steps: tests | testsx (stop at first failure) | bench | tsdf | hall |
#                                                  prof (rocprofv3 --kernel-trace --stats of the default bench) | kf (keyframe unit)
# Output under gpurun_out/r4_<tag>/.
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
O=gpurun_out/job_$TAG; mkdir -p $O
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
    prof|prof_tsdf|prof_hall)   # rocprofv3 --kernel-trace --stats of the same command as bench / tsdf / hall (whole run: pre-roll, windows, replay)
            case $step in prof) A="--steps 200 --warmup 20";; prof_tsdf) A="--mode tsdf --steps 200 --warmup 20";; prof_hall) A="--scene big --hires --steps 60 --warmup 10";; esac
            U=$PWD/$O/$step; mkdir -p $U
            (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $U/trace -o t -- python3 $OLDPWD/bench.py $A --no-pmc --cpu-frames 0 --no-group --repeats 0 > $U/bench_line_profiled.json 2> $U/prof.err)
            rc=$?  # (the profiler's / the benchmark's, not rm's)
            S=$(find $U/trace -name "*kernel_stats.csv" | head -1)
            if [ $rc -eq 0 ] && [ -n "$S" ]; then cp $S $O/${step}_kernel_stats.csv; head -8 $O/${step}_kernel_stats.csv | cut -c1-150; fi
            rm -rf $U/trace; echo "$step rc=$rc" ;;
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
            python3 tools/prof_unit.py --summarize "$T" "$F" "$W" $U/counts.json --run-line $U/run_line.json > $U/summary.json 2> $U/summary.err
            rc=$?
            rm -rf $U/trace $U/fetch $U/write
            echo "$step rc=$rc"; cat $U/run_line.json; tail -c 1800 $U/summary.json; tail -n 3 $U/summary.err; tail -n 3 $U/count.err ;;
    sq)     # SQ counters of the per-frame kernels over the timed workload (bench.py --child), three passes
            U=$PWD/$O/sq; mkdir -p $U; rm -rf $U/*
            B="python3 $PWD/bench.py --child --steps 60 --warmup 20"
            (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $U -o a -- $B >/dev/null 2>&1)
            (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $U -o b -- $B >/dev/null 2>&1)
            (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_FLAT --output-format csv -d $U -o c -- $B >/dev/null 2>&1)
            python3 - $U > $O/sq_summary.txt <<'PY'
import csv, glob, collections, sys
U = sys.argv[1]
print("SQ counters per launch, steady-state half of `bench.py --child --steps 60 --warmup 20` (rocprofv3 --pmc, three passes)")
for kern in ("k_frame", "k_mesh_filter", "k_mesh<"):
    print(kern)
    for f in sorted(glob.glob(U + "/**/*_counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(lambda: [0.0, 0])
        rows = list(csv.DictReader(open(f)))
        rows = rows[len(rows) // 2:]
        for r in rows:
            if kern in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (s, n) in sorted(acc.items()):
            print("  %-22s per launch %14.0f  (%d launches)" % (k, s / max(n, 1), n))
PY
            rc=$?
            rm -rf $U; echo "sq rc=$rc"; cat $O/sq_summary.txt | head -80 ;;
    *)      echo "unknown step $step" ;;
  esac
done
