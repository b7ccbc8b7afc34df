#!/usr/bin/env python3
"""How much slack does one stream leave?  N independent volumes on N HIP streams, same frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from texturefusion_amd import capi, synth
cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(100)]
dd = [torch.from_numpy(f[0]).to(dev) for f in frames]; dc = [torch.from_numpy(f[1]).to(dev) for f in frames]
poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
pd, pc = [t.data_ptr() for t in dd], [t.data_ptr() for t in dc]
for n in (1, 2, 3, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
    vols = [capi.Volume(res, cam, max_chunks=1 << 18, max_list=1 << 17, stream=s.cuda_stream) for s in streams]
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for v in vols:
            v.integrate_frames_device(pd, pc, poses)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("%d streams: %.2f us per frame per stream, aggregate %.0f frames/s" % (n, 1e6 * dt / 100, n * 100 / dt))
    for v in vols:
        v.close()
