#!/bin/bash
# round-6 profile set: bench lines (driver flags, 200 steps, TSDF only, hall), rocprofv3 --stats of each, SQ counters, keyframe unit
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/gpu_job.sh r6prof drv bench tsdf hall prof prof_tsdf prof_hall sq unit unit_moved
