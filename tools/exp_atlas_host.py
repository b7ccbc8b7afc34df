#!/usr/bin/env python3
"""Host-side cost of the device-resident atlas update and of a 10-frame batch enqueue."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from texturefusion_amd import capi, synth
cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(40)]
dd = [torch.from_numpy(f[0]).to(dev) for f in frames]; dc = [torch.from_numpy(f[1]).to(dev) for f in frames]
poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
v = capi.Volume(res, cam, max_chunks=1 << 18, max_list=1 << 17)
A = {}
for i in range(0, 40, 10):
    rgb = torch.from_numpy(np.ascontiguousarray(frames[i][1][..., :3])).to(dev)
    ids, voff, verts, cols = synth.mesh_from_depth(frames[i][0], frames[i][1], frames[i][3], cam, res, 4)
    nv = int(voff[-1])
    A[i] = dict(ids=ids, voff=voff, kf=np.full(len(ids), i, np.int32), T=np.tile(synth.pose_inverse16(frames[i][3]), (len(ids), 1)),
                dv=torch.from_numpy(verts.astype(np.float32)).to(dev), dc=torch.from_numpy(cols.astype(np.float32)).to(dev),
                tc=torch.empty(nv * 2, device=dev), tcol=torch.empty(nv * 3, device=dev), po=torch.empty(len(ids) * 8, dtype=torch.int32, device=dev), rgb=rgb)
    v.keyframe_cache_device(i, rgb.data_ptr(), dd[i].data_ptr())
for rep in range(3):
    t_int = t_atl = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in range(0, 40, 10):
        t = time.perf_counter()
        v.integrate_frames_device([x.data_ptr() for x in dd[b:b + 10]], [x.data_ptr() for x in dc[b:b + 10]], poses[b:b + 10])
        t_int += time.perf_counter() - t
        a = A[b]; t = time.perf_counter()
        v.patches_update_device(a["ids"], a["kf"], a["T"], a["voff"], a["dv"].data_ptr(), a["dc"].data_ptr(), a["tc"].data_ptr(), a["tcol"].data_ptr(), a["po"].data_ptr())
        t_atl += time.perf_counter() - t
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print("rep %d: per 10-frame batch: integrate enqueue %.0f us, atlas call %.0f us (%d patches); enqueue total %.0f us, wall %.0f us" % (
        rep, 1e6 * t_int / 4, 1e6 * t_atl / 4, len(A[0]["ids"]), 1e6 * t_enq / 4, 1e6 * t_all / 4))
