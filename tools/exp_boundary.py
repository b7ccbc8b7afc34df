#!/usr/bin/env python3
"""N-rank layouts simulated on one GPU, one rank after the other: time per frame of every rank (the
slowest bounds the job) and boundary records per 40-frame batch, for x slabs and for balanced slabs
of the diagonal key x + y + z."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from texturefusion_amd import capi, synth
from texturefusion_amd import partition as part
cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
NF = 200
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(NF)]
dd = [torch.from_numpy(f[0]).to(dev) for f in frames]; dc = [torch.from_numpy(f[1]).to(dev) for f in frames]
poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
buf = torch.empty((1 << 14) * capi.TF_BOUNDARY_RECORD_BYTES, dtype=torch.uint8, device=dev)
v0 = capi.Volume(res, cam, max_chunks=1 << 18, max_list=1 << 17)
keys = {}
for axis in ((1, 0, 0), (1, 1, 1)):
    ks = []
    for i in range(0, NF, 25):
        v0.frame_upload(frames[i][0], None, None)
        ids, _ = v0.prepare(frames[i][3])
        ks.append(part.key_of(ids, axis))
    v0.reset()
    keys[axis] = np.concatenate(ks)
v0.close()
for world in (2, 4, 8):
    for name, axis, edges in (("x slabs, equal width", (1, 0, 0), part.slab_bounds(part.room_extent_chunks(res), world)),
                              ("x slabs, balanced", (1, 0, 0), part.balanced_edges(keys[(1, 0, 0)], world)),
                              ("x+y+z slabs, balanced", (1, 1, 1), part.balanced_edges(keys[(1, 1, 1)], world))):
        us, recs = [], []
        for rank in range(world):
            v = capi.Volume(res, cam, max_chunks=1 << 18, max_list=1 << 17)
            v.set_partition(edges[rank], edges[rank + 1], axis)
            n = 0
            for rep in range(2):
                torch.cuda.synchronize(); t = time.perf_counter()
                for b in range(0, NF, 40):
                    v.integrate_frames_device([x.data_ptr() for x in dd[b:b + 40]], [x.data_ptr() for x in dc[b:b + 40]], poses[b:b + 40])
                    if rep == 1:
                        v.sync(); n += v.boundary_pack(buf.data_ptr(), 1 << 14)
                v.sync(); dt = time.perf_counter() - t
            us.append(1e6 * dt / NF); recs.append(n // (NF // 40))
            v.close()
        print("world %d  %-22s per-rank us/frame %s  -> slowest %.1f;  records/batch max %d" % (
            world, name, " ".join("%.1f" % u for u in us), max(us), max(recs)))
