cd "${GRAFT_REPO_ROOT:-.}"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; nproc; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
python - <<'PY'
import os; print("affinity", len(os.sched_getaffinity(0)))
PY
for a in "--staged-host-frames" ""; do
  cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "nr_throttled|throttled_usec|nr_periods" | tr '\n' ' '; echo
  python bench.py --steps 20 --warmup 5 --no-pmc --cpu-frames 0 --no-group --repeats 11 --no-roofline $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['repeats']
print('$a first %.1f us | 12 windows med %.1f min %.1f max %.1f' % (1e3*d['ms_per_step'],1e3*r['ms_per_step_median'],1e3*r['ms_per_step_min'],1e3*r['ms_per_step_max']))"
done
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "nr_throttled|throttled_usec|nr_periods" | tr '\n' ' '; echo
uptime
