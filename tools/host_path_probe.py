#!/usr/bin/env python3
"""Where does a tf_integrate_frame_host call spend its time?  (run under rocprofv3 --kernel-trace --stats for the device side)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from texturefusion_amd import capi, synth

cam = synth.Camera()
res = np.float32(0.005)
vol = capi.Volume(res, cam, max_chunks=1 << 19, max_list=1 << 18, max_coarse=1 << 20)
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(60)]
pinv = [synth.pose_inverse16(f[3]) for f in frames]
tex = len(sys.argv) > 1 and sys.argv[1] == "tex"
for k in range(20):
    f = frames[k]
    vol.integrate_frame_host(f[0], f[1], f[3].reshape(12), pinv[k] if tex else None, k)
vol.sync()
t0 = time.perf_counter()
th = []
for k in range(20, 60):
    f = frames[k]
    a = time.perf_counter()
    vol.integrate_frame_host(f[0], f[1], f[3].reshape(12), pinv[k] if tex else None, k)
    th.append(time.perf_counter() - a)
t1 = time.perf_counter()
vol.sync()
t2 = time.perf_counter()
print("tex" if tex else "tsdf", "per call host ms: median %.3f max %.3f; enqueue total %.3f ms, drain %.3f ms, per frame %.3f ms"
      % (1e3 * np.median(th), 1e3 * max(th), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t2 - t0) / 40))
