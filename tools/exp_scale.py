#!/usr/bin/env python3
"""How the fused launch scales with the number of selected chunks: S-room frames with only the
first f*W image columns valid (the rest has no depth)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from texturefusion_amd import capi, synth
cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(60)]
poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
dc = [torch.from_numpy(f[1]).to(dev) for f in frames]
for frac in [float(x) for x in os.environ.get('FRACS', '0,0.125,0.25,0.5,0.75,1').split(',')]:
    dd = []
    for f in frames:
        d = f[0].copy(); d[:, int(frac * cam.width):] = 0.0
        dd.append(torch.from_numpy(d).to(dev))
    v = capi.Volume(res, cam, max_chunks=1 << 19, max_list=1 << 18)
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        v.integrate_frames_device([t_.data_ptr() for t_ in dd], [t_.data_ptr() for t_ in dc], poses); v.sync()
        dt = time.perf_counter() - t
        if rep >= 2: best = min(best, dt)
    # chunks of the last frame through the call-by-call API
    v.frame_upload(dd[-1].cpu().numpy(), frames[-1][1], None)
    ids, _ = v.prepare(poses[-1].reshape(3, 4))
    print("valid columns %4.0f%%: %6d chunks selected in the last frame  %.2f us/frame" % (100 * frac, len(ids), 1e6 * best / 60))
    v.close()
