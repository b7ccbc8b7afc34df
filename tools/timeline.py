#!/usr/bin/env python3
"""Wave timeline of one steady-state fused launch (run with TF_KA_DBG=4096)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from texturefusion_amd import capi, synth
cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(60)]
frac = float(os.environ.get("FRAC", "1"))   # only the first frac*W image columns carry depth
for f in frames:
    f[0][:, int(frac * cam.width):] = 0.0
dd = [torch.from_numpy(f[0]).to(dev) for f in frames]; dc = [torch.from_numpy(f[1]).to(dev) for f in frames]
poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
v = capi.Volume(res, cam, max_chunks=1 << 19, max_list=1 << 18)
# REPS=2: the second pass over the same 60 frames (steady state); REPS=1: first touch (the default
# bench's regime).  Stamps are those of the last launch that had all three roles.
for rep in range(int(os.environ.get("REPS", "2"))):
    v.integrate_frames_device([t.data_ptr() for t in dd], [t.data_ptr() for t in dc], poses); v.sync()
raw = v.debug_phase_raw()
t0, t1, role = raw[:, 10].astype(np.int64), raw[:, 11].astype(np.int64), raw[:, 12].astype(np.int64)
m = role > 0
xcd = raw[:, 13].astype(np.int64)
b0 = t0[m].min(); t0 -= b0; t1 -= b0   # s_memrealtime: 100 MHz, chip-wide
print("waves stamped %d, span %.2f us" % (m.sum(), t1[m].max() / 100.0))
for r, name in ((1, "K-A"), (2, "K-C select"), (3, "K-B bbox")):
    k = m & (role == r)
    if not k.any():
        continue
    s_, e_ = t0[k] / 100.0, t1[k] / 100.0
    print("%-12s waves %5d  start min/med/max %6.2f %6.2f %6.2f   end min/med/max %6.2f %6.2f %6.2f   dur med %6.2f max %6.2f us" % (
        name, k.sum(), s_.min(), np.median(s_), s_.max(), e_.min(), np.median(e_), e_.max(), np.median(e_ - s_), (e_ - s_).max()))
span = t1[m].max()
bins = np.linspace(0, span, 21)
for r, name in ((1, "K-A"), (2, "K-C"), (3, "K-B")):
    k = m & (role == r)
    print(name, "waves resident per 5% bin:", [int(np.sum((t0[k] < bins[i + 1]) & (t1[k] > bins[i]))) for i in range(20)])
for x in range(8):
    k = m & (xcd == x) & (role == 1)
    print("xcd %d K-A: first start %.2f last end %.2f" % (x, t0[k].min() / 100.0, t1[k].max() / 100.0))
k = m & (role == 1)
pro = (raw[:, 14].astype(np.int64) - b0 - t0)[k] / 100.0
print("K-A prologue (wave start -> chunk loop): min/med/max %.2f %.2f %.2f us" % (pro.min(), np.median(pro), pro.max()))
ent = (raw[:, 15].astype(np.int64) - raw[:, 14].astype(np.int64))[k & (raw[:, 15] > 0)] / 100.0
print("K-A first list entry (loop start -> scalars loaded): min/med/max %.2f %.2f %.2f us" % (ent.min(), np.median(ent), ent.max()))
# work per K-A wave vs when it finished
nch = raw[:, 8].astype(np.int64)[k]; nrow = raw[:, 9].astype(np.int64)[k]; endt = t1[k] / 100.0; dur = (t1[k] - t0[k]) / 100.0
for c in sorted(set(nch.tolist())):
    sel = nch == c
    print("waves with %d chunks: %5d  end med %.2f max %.2f us  rows med %d" % (c, sel.sum(), np.median(endt[sel]), endt[sel].max(), np.median(nrow[sel])))
late = endt > np.percentile(endt, 95)
print("latest 5%% of the waves: chunks mean %.2f, rows mean %.0f (all waves: %.2f, %.0f)" % (nch[late].mean(), nrow[late].mean(), nch.mean(), nrow.mean()))
print("corr(duration, rows) = %.2f   corr(duration, chunks) = %.2f" % (np.corrcoef(dur, nrow)[0, 1], np.corrcoef(dur, nch)[0, 1]))
