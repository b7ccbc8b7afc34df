#!/usr/bin/env python3
"""The textured unit on a volume that is being EXPLORED: the first orbit of the room stream (every frame inserts chunks at the
frontier of what has been seen), resident frames, one streaming call per 10 frames.  Prints microseconds per frame over frames
20 .. N and what the neighbour table looks like at the end (tf_check_neighbours: rows, known words, trusted rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from texturefusion_amd import capi, synth
from tests.util import HipBuffer

cam = synth.Camera()
n = int(os.environ.get("FRAMES", "200"))
gv = capi.Volume(np.float32(0.005), cam, max_chunks=1 << 19, max_list=1 << 18, mesh_blocks=1 << 17)
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(n)]
bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in frames]
poses = np.stack([f[3].reshape(12) for f in frames])
pinv = np.stack([synth.pose_inverse16(f[3]) for f in frames])
t0 = None
for b in range(0, n, 10):
    if b == 20:
        gv.sync(); t0 = time.perf_counter()
    e = min(b + 10, n)
    ahead = min(2, n - e)
    idx = list(range(b, e + ahead))
    gv.stream_frames_textured_device([bufs[i][0].ptr for i in idx], [bufs[i][1].ptr for i in idx], poses[idx], pinv[idx], b, n_ahead=ahead)
gv.sync()
dt = time.perf_counter() - t0
c = gv.check_neighbours()
st = gv.stats()
print("first orbit, frames 20..%d: %.1f us per frame; chunks %d; table: rows %d, known words %d, trusted rows %d (%.0f %%), wrong %d, missed %d"
      % (n, 1e6 * dt / (n - 20), st.n_chunks, c[0], c[1], c[3], 100.0 * c[3] / max(1, c[0]), c[2], c[4]))
gv.close()
