#!/bin/bash
# the keyframe unit's forked front end: the unit's run line with the fork and with TF_UNIT_SERIAL_FRONT=1, alternating
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/job_front; mkdir -p $O
for i in 1 2 3; do
  for e in 0 1; do
    TF_UNIT_SERIAL_FRONT=$e timeout 300 python3 tools/prof_unit.py --run > $O/run_${e}_$i.json 2> $O/run.err; echo "serial_front=$e: $(cat $O/run_${e}_$i.json | cut -c1-120)"
  done
done
