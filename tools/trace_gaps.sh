#!/bin/bash
# Kernel durations and back-to-back gaps of the fused launch from a rocprofv3 kernel trace:
#   tools/trace_gaps.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/trace; tag=$1; shift
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-frames 0 --no-roofline "$@" >/dev/null 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$O/**/${tag}_kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if "k_frame" in r["Kernel_Name"]]
rows.sort()
rows = rows[len(rows) // 2:]          # the timed half
dur = [e - s for s, e in rows]
gap = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
gap = [g for g in gap if g < 100000]
print("$tag: launches %d  duration avg %.2f us  gap avg %.2f us  period avg %.2f us" % (
    len(rows), sum(dur) / len(dur) / 1e3, sum(gap) / len(gap) / 1e3, (sum(dur) / len(dur) + sum(gap) / len(gap)) / 1e3))
PY
