#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from rocprofv3 --pmc passes (tools/pmc_r2.sh), with the access-shape
calibration of tools/calib_fetch applied.

  python3 tools/pmc_summary.py gpurun_out/pmc_<tag>  ->  JSON: {"calibration": {...}, "kernels": {name: {...}}}

FETCH_SIZE and WRITE_SIZE are reported by rocprofv3 in KiB-like units that are NOT trusted here: the unit and the
gfx950 under-count are both folded into one factor per access shape, counter / known bytes, measured by the
calibration kernels in the same pass on the same box."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def read_pass(d, counter):
    """{kernel: [sum of counter, dispatches]}"""
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").strip()
                acc[k][0] += float(row["Counter_Value"])
                acc[k][1] += 1
    return acc


def main():
    out = sys.argv[1]
    res = {"calibration": {}, "kernels": {}}
    calib_bytes = 2 << 30
    for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        cal = read_pass(os.path.join(out, "calib_" + name), counter)
        for k, (s, n) in sorted(cal.items()):
            if "k_calib" not in k or n == 0:
                continue
            res["calibration"].setdefault(k, {})[counter + "_per_launch"] = s / n
            res["calibration"][k][counter + "_per_byte"] = s / n / calib_bytes
        run = read_pass(os.path.join(out, name), counter)
        for k, (s, n) in sorted(run.items()):
            if not k.startswith("k_") and "tf::" not in k:
                continue
            res["kernels"].setdefault(k, {})[counter + "_sum"] = s
            res["kernels"][k][counter + "_launches"] = n
            res["kernels"][k][counter + "_per_launch"] = s / n
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
