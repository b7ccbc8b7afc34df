"""First calls of tf_integrate_frame_host in tf_host_frame_set_async(1) mode: per-call host time (TSDF-only stream out of
registered arrays).  PYTHONPATH=. python tools/async_probe.py"""
import numpy as np, time
from texturefusion_amd import capi, synth
cam = synth.Camera(); res = np.float32(0.005)
n = 40
fr = [synth.room_frame(k, cam, with_quality=False) for k in range(n)]
h_depth = np.stack([f[0] for f in fr]); h_rgba = np.stack([f[1] for f in fr]); poses = np.stack([f[3].reshape(12) for f in fr]).astype(np.float32)
pinv = np.stack([synth.pose_inverse16(f[3]) for f in fr]).astype(np.float32)
for textured in (False,):
    vol = capi.Volume(res, cam, max_chunks=1 << 18, mesh_blocks=1 << 16)
    vol.host_register(h_depth); vol.host_register(h_rgba)
    def run(first, count, rec=None):
        for j in range(count):
            i = (first + j) % n
            t0 = time.perf_counter()
            vol.integrate_frame_host_addr(h_depth[i].ctypes.data, h_rgba[i].ctypes.data, poses[i].ctypes.data, pinv[i].ctypes.data if textured else 0, first + j)
            if rec is not None: rec.append(1e6 * (time.perf_counter() - t0))
    run(0, 200); vol.sync()
    for mode in (True, False, True):
        vol.host_frame_set_async(mode)
        rec = []
        t0 = time.perf_counter(); run(100, 160, rec); vol.host_frame_fence(); vol.sync(); dt = time.perf_counter() - t0
        print("async" if mode else "sync", "%.1f us per frame" % (1e6 * dt / 160), "calls > 300 us:", [(i, int(x)) for i, x in enumerate(rec) if x > 300][:12], "median %.1f" % np.median(rec))
    vol.close()
