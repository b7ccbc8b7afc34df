#!/usr/bin/env python3
"""Host time per entry point of the reference's own call sequence for one keyframe (INTEGRATION.md approach A: the header
swap) on the bench's S-room orbit: prepare, integrate (colour), integrate (depth) x 6, finalize, update_meshes,
compress_meshes, keyframe_cache + generate_patches, update_atlas -- every call synchronous.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from texturefusion_amd import capi, synth

cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
N = 200
fr = [synth.room_frame(k, cam, with_quality=False) for k in range(N)]
dd = [torch.from_numpy(f[0]).to(dev) for f in fr]; dc = [torch.from_numpy(f[1]).to(dev) for f in fr]
poses = np.stack([f[3].reshape(12) for f in fr]).astype(np.float32)
pinv = np.stack([synth.pose_inverse16(f[3]) for f in fr]).astype(np.float32)
vol = capi.Volume(res, cam, max_chunks=1 << 19, max_list=1 << 18, max_coarse=1 << 20)
acc = {}
def T(name, fn):
    t = time.perf_counter(); r = fn(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
def keyframe(g, timed):
    k0 = (7 * g) % N; loc = [(k0 + 1 + i) % N for i in range(6)]
    tt = T if timed else (lambda n, f: f())
    vol.frame_bind_device(dd[k0].data_ptr(), dc[k0].data_ptr(), 0)
    ids, new = tt("prepare", lambda: vol.prepare(poses[k0]))
    needs = np.zeros(len(ids), np.uint8)
    tt("integrate_colour", lambda: vol.integrate(poses[k0], ids, needs, 1, True, False))
    for k in loc:
        vol.frame_bind_device(dd[k].data_ptr(), 0, 0)
        tt("integrate_depth_x6", lambda: vol.integrate(poses[k], ids, needs, 1, False, False))
    tt("finalize", lambda: vol.finalize(ids, needs, new))
    tt("update_meshes", lambda: vol.update_meshes())
    upd = tt("compress_meshes", lambda: vol.compress_meshes())
    tt("keyframe_cache+pose", lambda: vol.keyframe_cache_device(1000 + g, dc[k0].data_ptr(), dd[k0].data_ptr(), stride=4, pose_inv16=pinv[k0]))
    tt("generate_patches", lambda: vol.generate_patches(upd, np.full(len(upd), 1000 + g, np.int32)))
    tt("update_atlas", lambda: vol.update_atlas(upd))
    if g >= 8: vol.keyframe_release(1000 + g - 8)
    return len(ids), len(upd)
for g in range(28): keyframe(g, False)
vol.sync()
K = 16
t0 = time.perf_counter()
for g in range(28, 28 + K): n_ids, n_upd = keyframe(g, True)
vol.sync()
dt = time.perf_counter() - t0
print(json.dumps({"keyframes": K, "ms_per_keyframe": 1e3 * dt / K, "list_entries": n_ids, "chunks_to_update": n_upd,
                  "us_per_keyframe_by_entry_point": {k: round(1e6 * v / K, 1) for k, v in acc.items()}}))
