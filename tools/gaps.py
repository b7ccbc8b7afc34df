#!/usr/bin/env python3
"""Kernel time and idle time of the last K frames of a rocprofv3 --kernel-trace CSV (bench.py --child):
    python tools/gaps.py <kernel_trace.csv> K
A frame starts at a k_frame<true, ...> launch.  Prints, per frame: the sum of the kernel durations, the idle time between
consecutive kernels, the frame period; and the same per kernel."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
K = int(sys.argv[2])
starts = [i for i, r in enumerate(rows) if "k_frame<true" in r[2]]
lo, hi = starts[-K - 1], starts[-1]
win = rows[lo:hi]
busy = sum(e - s for s, e, _ in win)
gap = sum(max(0, win[i + 1][0] - win[i][1]) for i in range(len(win) - 1)) + max(0, rows[hi][0] - win[-1][1])
period = rows[hi][0] - rows[lo][0]
print("frames %d: period %.1f us, kernels %.1f us, idle between kernels %.1f us" % (K, period / K / 1e3, busy / K / 1e3, gap / K / 1e3))
acc = collections.defaultdict(lambda: [0, 0])
gaps_after = collections.defaultdict(lambda: [0, 0])
for i, (s, e, n) in enumerate(win):
    k = n.split("(")[0][:40]
    acc[k][0] += e - s; acc[k][1] += 1
    nxt = win[i + 1][0] if i + 1 < len(win) else rows[hi][0]
    gaps_after[k][0] += max(0, nxt - e); gaps_after[k][1] += 1
for k, (t, c) in sorted(acc.items(), key=lambda x: -x[1][0]):
    print("  %-42s %6.1f us/frame (%d launches)  idle behind it %5.2f us" % (k, t / K / 1e3, c, gaps_after[k][0] / max(1, gaps_after[k][1]) / 1e3))
