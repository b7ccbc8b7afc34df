#!/usr/bin/env python3
"""Experiment: constant depth / colour images, so gather addresses do not change any result."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from texturefusion_amd import capi, synth
cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(60)]
d = torch.full((cam.height, cam.width), 1.5, dtype=torch.float32, device=dev)
c = torch.full((cam.height, cam.width, 4), 77, dtype=torch.uint8, device=dev)
poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
v = capi.Volume(res, cam, max_chunks=1 << 19, max_list=1 << 18)
dd = [d.data_ptr()] * 60; dc = [c.data_ptr()] * 60
for rep in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    v.integrate_frames_device(dd, dc, poses); v.sync()
    dt = time.perf_counter() - t
    if rep >= 2:
        print("rep %d: %.2f us/frame, chunks/frame %.0f" % (rep, 1e6 * dt / 60, v.stats().n_selected / 60.0 if hasattr(v.stats(), 'n_selected') else -1))
