#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_50; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_unit.py tests/test_gpu_group.py tests/test_gpu_texmap.py tests/test_gpu_atlas.py -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2; grep -n "Error\|assert" $O/tests.log | head -5
python bench.py --steps 20 --warmup 5 --no-pmc --cpu-frames 0 > $O/line.json 2> $O/err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_50/line.json').read().strip().splitlines()[-1])
k=d['keyframe_unit']
print('keyframe unit: %.0f kf/s (%.1f us), with moved %.0f kf/s (%.1f us)' % (k['new_keyframes_only']['keyframes_per_s'], 1e3*k['new_keyframes_only']['ms_per_keyframe'], k['with_one_moved_keyframe_every_other_call']['keyframes_per_s'], 1e3*k['with_one_moved_keyframe_every_other_call']['ms_per_keyframe']))
PY
