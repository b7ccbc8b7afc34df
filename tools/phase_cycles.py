#!/usr/bin/env python3
"""Per-phase wave cycles of the voxel-update kernel (run with TF_KA_DBG=2048)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from texturefusion_amd import capi, synth
cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(60)]
dd = [torch.from_numpy(f[0]).to(dev) for f in frames]; dc = [torch.from_numpy(f[1]).to(dev) for f in frames]
poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
v = capi.Volume(res, cam, max_chunks=1 << 19, max_list=1 << 18)
for rep in range(2):
    v.integrate_frames_device([t.data_ptr() for t in dd], [t.data_ptr() for t in dc], poses); v.sync()
    cyc = v.debug_phase_cycles(reset=True)
names = ["list entry", "geometry + gather issue", "slot resolve", "predicates + row loads (wait depth)",
         "arithmetic (wait rows)", "stores issue", "finalize"]
n = max(cyc[9], 1); tot = sum(cyc[:7])
for k in range(7):
    print("%-38s %8.0f cycles/chunk  %5.1f%%" % (names[k], cyc[k] / n, 100.0 * cyc[k] / max(tot, 1)))
print("total %.0f cycles/chunk over %d chunks; slow-path slot resolutions %d; RMW passes %d (%.2f per chunk)" % (
    tot / n, n, cyc[7], cyc[8], cyc[8] / n))
