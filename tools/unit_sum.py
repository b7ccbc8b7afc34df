#!/usr/bin/env python3
"""one-line digest of a `gpu_job.sh unit` output directory: sum of kernel time, per-kernel us, the run line"""
import json, sys
d = json.load(open(sys.argv[1] + "/summary.json"))
print(round(d["kernel_us_per_keyframe"], 1), {k: (round(v["us"], 1), v["launches"]) for k, v in d["kernels"].items()})
print(open(sys.argv[1] + "/run_line.json").read()[:200])
