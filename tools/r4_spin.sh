#!/bin/bash
# alternating A/B over the driver's bench flags, six windows per run: host frames out of registered caller buffers (default)
# against the staging copy (--staged-host-frames), the latter with and without the helpers' spin (TF_COPY_SPIN_US)
cd "${GRAFT_REPO_ROOT:-.}"
N=${1:-4}
for i in $(seq 1 $N); do for cfg in "reg" "staged200" "staged0"; do
case $cfg in reg) A=""; E="";; staged200) A="--staged-host-frames"; E="TF_COPY_SPIN_US=200";; staged0) A="--staged-host-frames"; E="TF_COPY_SPIN_US=0";; esac
env $E python bench.py --steps 20 --warmup 5 --no-pmc --cpu-frames 0 --no-group --repeats 5 --no-roofline $A ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_phases_us_per_step']; r=d['repeats']
print('%-10s first %.1f us | six windows med %.1f min %.1f max %.1f | staging %.1f wait-upload %.1f wait-device %.1f launches %.1f' % ('$cfg',1e3*d['ms_per_step'],1e3*r['ms_per_step_median'],1e3*r['ms_per_step_min'],1e3*r['ms_per_step_max'],h['staging_copy_us'],h['wait_for_upload_us'],h['wait_for_device_us'],h['launches_us']))"
done; done
