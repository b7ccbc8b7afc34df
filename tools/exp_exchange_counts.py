#!/usr/bin/env python3
"""Per-frame record counts of the boundary exchange for 2 / 4 / 8 balanced x+y+z slabs of the S-room stream (one GPU,
rank after rank; ghosts are not exchanged here -- the count a rank SENDS only depends on its own updates), and the
textured step time of every rank without the exchange.  Sizes --exchange-cap."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from texturefusion_amd import capi, synth
from texturefusion_amd import partition as part

cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
NF = int(os.environ.get("NF", "80"))
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(NF)]
dd = [torch.from_numpy(f[0]).to(dev) for f in frames]; dc = [torch.from_numpy(f[1]).to(dev) for f in frames]
poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
pinv = np.stack([synth.pose_inverse16(f[3]) for f in frames]).astype(np.float32)
buf = torch.empty((1 << 13) * capi.TF_BOUNDARY_RECORD_BYTES, dtype=torch.uint8, device=dev)
axis = (1, 1, 1)
v0 = capi.Volume(res, cam, max_chunks=1 << 18, max_list=1 << 17)
ks = []
for i in range(0, 200, 25):
    f = synth.room_frame(i, cam, with_quality=False)
    v0.frame_upload(f[0], None, None)
    ids, _ = v0.prepare(f[3])
    ks.append(part.key_of(ids, axis))
v0.close()
keys = np.concatenate(ks)
for world in (2, 4, 8):
    edges = part.balanced_edges(keys, world)
    worst, times = 0, []
    for rank in range(world):
        v = capi.Volume(res, cam, max_chunks=1 << 18, max_list=1 << 17)
        v.set_partition(edges[rank], edges[rank + 1], axis)
        counts = []
        t_tot = 0.0
        for k in range(NF):
            sub = [k, min(k + 1, NF - 1), min(k + 2, NF - 1)]
            torch.cuda.synchronize(); t = time.perf_counter()
            v.stream_frames_device([dd[i].data_ptr() for i in sub], [dc[i].data_ptr() for i in sub], poses[sub], n_ahead=2)
            v.texture_frame_device(pinv[k], k)
            v.sync(); t_tot += time.perf_counter() - t
            counts.append(v.boundary_pack(buf.data_ptr(), 1 << 13))
        v.close()
        worst = max(worst, max(counts))
        times.append(1e6 * t_tot / NF)
        print("world %d rank %d: records per frame max %d mean %.0f; %.0f us per frame (synchronised per frame)"
              % (world, rank, max(counts), np.mean(counts), times[-1]))
    print("world %d: largest block any rank sends: %d records = %.1f MB; slowest rank %.0f us" % (world, worst, worst * 8208e-6, max(times)))
