#!/usr/bin/env python3
"""Where does a chunk / a patch spend its time INSIDE the mesher and the patch kernel?

The kernels write s_memrealtime (100 MHz, chip-wide) at their phase boundaries into the library's debug table when
their triage knob asks for it; this script streams 70 textured frames of the room orbit and prints, for the LAST
frame's launch, the median / p90 / max duration of every phase plus when workgroups start and end.

    TF_MESH_DBG=9  python tools/stamps.py mesh     # k_mesh, one row per workgroup (its first chunk)
    TF_PATCH_DBG=3 python tools/stamps.py patch    # k_patch, one row per wave (= patch); phases closed with a wait
    TF_MESH_DBG=10 python tools/stamps.py filter   # k_mesh_filter (wave form), one row per wave: its first entry
(K-A has its own: TF_KA_DBG=4096 python tools/timeline.py)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from texturefusion_amd import capi, synth
from tests.util import HipBuffer

what = sys.argv[1] if len(sys.argv) > 1 else "mesh"
cam = synth.Camera()
gv = capi.Volume(np.float32(0.005), cam, max_chunks=1 << 18)
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(int(os.environ.get("FRAMES", "70")))]
bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in frames]
poses = np.stack([f[3].reshape(12) for f in frames])
pinv = np.stack([synth.pose_inverse16(f[3]) for f in frames])
gv.stream_frames_textured_device([b[0].ptr for b in bufs], [b[1].ptr for b in bufs], poses, pinv, 0)
gv.sync()
P = gv.debug_phase_raw().astype(np.int64)[:16384 if what == "filter" else 4096]
t0 = P[:, 0]
if not (t0 > 0).any():
    sys.exit("no stamps: set TF_MESH_DBG=9 / TF_PATCH_DBG=3")
base = t0[t0 > 0].min()


def line(name, d):
    print("%-26s median %5.2f  p90 %5.2f  max %5.2f us" % (name, np.median(d), np.percentile(d, 90), d.max()))


if what == "filter":
    st = (t0[t0 > 0] - base) / 100.0
    line("start of a wave", st)
    print("waves stamped: %d; started later than 3 us: %d, later than 6 us: %d" % ((t0 > 0).sum(), (st > 3).sum(), (st > 6).sum()))
    has = (t0 > 0) & (P[:, 6] > 0)
    print("waves with an entry: %d" % has.sum())
    line("wave start -> entry + counter arrived", (P[has, 1] - P[has, 0]) / 100.0)
    line("-> row + alive arrived", (P[has & (P[:, 2] > 0), 2] - P[has & (P[:, 2] > 0), 1]) / 100.0)
    g3 = has & (P[:, 3] > 0)
    line("-> summaries arrived", (P[g3, 3] - P[g3, 2]) / 100.0)
    g4 = has & (P[:, 4] > 0)
    line("ruled out: -> record reset done", (P[g4, 4] - P[g4, 3]) / 100.0)
    g5 = has & (P[:, 5] > 0)
    line("exact test + row append", (P[g5, 5] - P[g5, 3]) / 100.0)
    line("first entry total", (P[has, 6] - P[has, 0]) / 100.0)
    print("ruled out by the summaries %d, exact %d" % (g4.sum(), g5.sum()))
    print("last end - first start: %.2f us" % ((P[has, 6].max() - base) / 100.0))
    ends = np.sort((P[has, 6] - base) / 100.0)
    print("ends: median %.2f p90 %.2f p99 %.2f" % (np.median(ends), np.percentile(ends, 90), np.percentile(ends, 99)))
elif what == "mesh":
    ok = P[:, 8] > P[:, 0]
    print("workgroups with a chunk: %d of %d" % (ok.sum(), (t0 > 0).sum()))
    for a, b, name in [(0, 1, "row + own voxel loads"), (1, 3, "own + halo staged"), (3, 4, "corner flags"), (4, 5, "cell pass"),
                       (5, 6, "ranks"), (6, 7, "vertices (thread 0)"), (7, 8, "triangles + record")]:
        line(name, (P[ok, b] - P[ok, a]) / 100.0)
    line("chunk total", (P[ok, 8] - P[ok, 0]) / 100.0)
    st = (P[ok, 0] - base) / 100.0
    line("start of a busy workgroup", st)
    print("busy workgroups that start later than 5 us: %d (row numbers %s ...)" % ((st > 5).sum(), np.nonzero(ok)[0][st > 5][:8]))
    xcd = np.nonzero(ok)[0] % 8
    print("busy workgroups per XCD (row %% 8):", [int((xcd == x).sum()) for x in range(8)])
    print("last end - first start: %.2f us" % ((P[ok, 8].max() - base) / 100.0))
else:
    ok = (P[:, 1] > 0) & (t0 > 0)
    print("waves with a patch: %d of %d" % (ok.sum(), (t0 > 0).sum()))
    prev = 0
    for k, name in [(1, "list entry + record"), (2, "exchange + slot rank"), (3, "vertex loads"), (4, "projection + gathers"),
                    (5, "tap arithmetic, box, stores"), (6, "blit loads")]:
        good = ok & (P[:, k] > 0)
        line(name, (P[good, k] - P[good, k - 1]) / 100.0)
    good = ok & (P[:, 6] > 0)
    line("wave total", (P[good, 6] - P[good, 0]) / 100.0)
    line("start of a wave", (t0[t0 > 0] - base) / 100.0)
    print("last blit-load stamp - first start: %.2f us" % ((P[good, 6].max() - base) / 100.0))
