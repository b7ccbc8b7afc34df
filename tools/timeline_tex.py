#!/usr/bin/env python3
"""Wave timeline of one steady-state TEXTURED frame launch (k_frame<true, true>: K-A(f), patch stage(f-1), K-C(f+1), K-B(f+2)).
Run with TF_KA_DBG=4096 (the tuning instance stamps {start, end, role, XCC} per wave, 100 MHz chip-wide clock)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from texturefusion_amd import capi, synth
from tests.util import HipBuffer

cam = synth.Camera()
gv = capi.Volume(np.float32(0.005), cam, max_chunks=1 << 18)
n = int(os.environ.get("FRAMES", "70"))
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(n)]
bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in frames]
poses = np.stack([f[3].reshape(12) for f in frames])
pinv = np.stack([synth.pose_inverse16(f[3]) for f in frames])
for rep in range(2):  # second pass over the same frames: steady state
    gv.stream_frames_textured_device([b[0].ptr for b in bufs], [b[1].ptr for b in bufs], poses, pinv, rep * n)
    gv.sync()
raw = gv.debug_phase_raw()
t0, t1, role = raw[:, 10].astype(np.int64), raw[:, 11].astype(np.int64), raw[:, 12].astype(np.int64)
m = role > 0
if not m.any():
    sys.exit("no stamps: set TF_KA_DBG=4096")
b0 = t0[m].min(); t0 = t0 - b0; t1 = t1 - b0
print("waves stamped %d, span %.2f us" % (m.sum(), t1[m].max() / 100.0))
names = {1: "K-A", 2: "K-C select", 3: "K-B bbox", 4: "patch stage"}
for r in (4, 2, 3, 1):
    k = m & (role == r)
    if not k.any():
        continue
    s_, e_ = t0[k] / 100.0, t1[k] / 100.0
    d = e_ - s_
    print("%-12s waves %5d  start min/med/p90/max %6.2f %6.2f %6.2f %6.2f   end min/med/p90/max %6.2f %6.2f %6.2f %6.2f   dur med %6.2f p90 %6.2f max %6.2f us"
          % (names[r], k.sum(), s_.min(), np.median(s_), np.percentile(s_, 90), s_.max(), e_.min(), np.median(e_), np.percentile(e_, 90), e_.max(),
             np.median(d), np.percentile(d, 90), d.max()))
span = t1[m].max()
bins = np.linspace(0, span, 21)
for r in (4, 2, 3, 1):
    k = m & (role == r)
    print("%-12s resident per 5%% bin:" % names[r], [int(np.sum((t0[k] < bins[i + 1]) & (t1[k] > bins[i]))) for i in range(20)])
k = m & (role == 1)
nch = raw[:, 8].astype(np.int64)[k]; endt = t1[k] / 100.0; st = t0[k] / 100.0
for c in sorted(set(nch.tolist())):
    sel = nch == c
    print("K-A waves with %d chunks: %5d  start med %.2f  end med %.2f max %.2f us  dur med %.2f" % (c, sel.sum(), np.median(st[sel]), np.median(endt[sel]), endt[sel].max(), np.median(endt[sel] - st[sel])))
k4 = m & (role == 4)
d4 = (t1[k4] - t0[k4]) / 100.0
print("patch waves: %d with dur < 2 us (no patch), %d longer; longer ones: dur med %.2f p90 %.2f" % ((d4 < 2).sum(), (d4 >= 2).sum(), np.median(d4[d4 >= 2]), np.percentile(d4[d4 >= 2], 90)))
