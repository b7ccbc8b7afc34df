cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD
for cfg in "1 0" "2 0" "6 0"; do
  set -- $cfg
  U=$R/gpurun_out/nl_$1_$2; rm -rf $U; mkdir -p $U
  ( export UNIT_N_LOCAL=$1; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $U -o t -- python3 $R/tools/prof_unit.py --run > $U/log 2>&1 )
  S=$(find $U -name "*kernel_stats.csv" | head -1)
  echo "== n_local=$1"; tail -1 $U/log | cut -c1-120
  python3 - $S <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('k_integrate','k_mesh','k_select','k_pre','k_bbox')):
        print('    %-50s calls %6s avg %8.2f us' % (n.replace('void tf::','')[:50], r['Calls'], float(r['AverageNs'])/1e3))
PY
  rm -rf $U
done
