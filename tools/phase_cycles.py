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
names = ["list/scalars", "geom0+gather", "slot resolve", "predicates(wait depth)", "voxel load issue", "next geometry",
         "arith(wait voxels)", "gathers+stores issue", "finalize", "chunks"]
n = max(cyc[9], 1); tot = sum(cyc[:9])
for k in range(9):
    print("%-26s %8.0f cycles/chunk  %5.1f%%" % (names[k], cyc[k] / n, 100.0 * cyc[k] / max(tot, 1)))
print("total %.0f cycles/chunk over %d chunks" % (tot / n, n))
