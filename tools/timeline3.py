#!/usr/bin/env python3
"""Wave timeline of one steady-state launch of the textured stream's k_frame (run with TF_KA_DBG=4096): K-A of frame f,
the patch stage of frame f - 1, K-C of f + 1, K-B of f + 2 as block ranges of one kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from texturefusion_amd import capi, synth
cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
N = int(os.environ.get("FRAMES", "200")); EXTRA = int(os.environ.get("EXTRA", "37"))
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(N)]
dd = [torch.from_numpy(f[0]).to(dev) for f in frames]; dc = [torch.from_numpy(f[1]).to(dev) for f in frames]
poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
pinv = np.stack([synth.pose_inverse16(f[3]) for f in frames]).astype(np.float32)
v = capi.Volume(res, cam, max_chunks=1 << 19, max_list=1 << 18)
idx = [k % N for k in range(N + EXTRA + 2)]
v.stream_frames_textured_device([dd[i].data_ptr() for i in idx], [dc[i].data_ptr() for i in idx], poses[idx], pinv[idx], 0, n_ahead=2)
v.sync()
raw = v.debug_phase_raw()
t0, t1, role = raw[:, 10].astype(np.int64), raw[:, 11].astype(np.int64), raw[:, 12].astype(np.int64)
m = role > 0
b0 = t0[m].min(); t0 -= b0; t1 -= b0   # s_memrealtime: 100 MHz, chip-wide
print("waves stamped %d, span %.2f us" % (m.sum(), t1[m].max() / 100.0))
names = ((1, "K-A"), (4, "patch"), (2, "K-C select"), (3, "K-B bbox"))
for r, name in names:
    k = m & (role == r)
    if not k.any():
        continue
    s_, e_ = t0[k] / 100.0, t1[k] / 100.0
    busy = k & ((t1 - t0) > 150)  # waves that did more than look at an empty list (> 1.5 us)
    print("%-12s waves %5d (busy %5d)  start min/med/max %6.2f %6.2f %6.2f   end min/med/max %6.2f %6.2f %6.2f   dur med %6.2f max %6.2f us" % (
        name, k.sum(), busy.sum(), s_.min(), np.median(s_), s_.max(), e_.min(), np.median(e_), e_.max(), np.median((e_ - s_)[(e_ - s_) > 1.5]) if busy.any() else 0, (e_ - s_).max()))
span = t1[m].max()
bins = np.linspace(0, span, 21)
for r, name in names:
    k = m & (role == r)
    print("%-6s waves resident per 5%% bin:" % name, [int(np.sum((t0[k] < bins[i + 1]) & (t1[k] > bins[i]))) for i in range(20)])
k = m & (role == 1)
nch = raw[:, 8].astype(np.int64)[k]; endt = t1[k] / 100.0
for c in sorted(set(nch.tolist())):
    sel = nch == c
    print("K-A waves with %d chunks: %5d  end med %.2f max %.2f us" % (c, sel.sum(), np.median(endt[sel]), endt[sel].max()))
