#!/usr/bin/env python3
"""What the SIZED neighbour exchange moves on the bench stream (S-room orbit, steady state after one orbit of pre-roll),
for 2 / 4 / 8 balanced x+y+z slabs -- one GPU, rank after rank (what a rank SENDS depends only on its own updates, and
the capacities come from the selection every rank runs in full): per rank the records actually packed, the record
capacity tf_boundary_band_bounds gave the two blocks, and bytes on the wire / (records x 8208 B).  Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from texturefusion_amd import capi, synth
from texturefusion_amd import partition as part

cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
ORBIT = 200
NF = int(os.environ.get("NF", "100"))
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(ORBIT)]
dd = [torch.from_numpy(f[0]).to(dev) for f in frames]; dc = [torch.from_numpy(f[1]).to(dev) for f in frames]
poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
axis = (1, 1, 1)
v0 = capi.Volume(res, cam, max_chunks=1 << 18, max_list=1 << 17)
ks = []
for i in range(0, ORBIT, 25):
    v0.frame_upload(frames[i][0], None, None)
    ids, _ = v0.prepare(frames[i][3])
    ks.append(part.key_of(ids, axis))
v0.close()
keys = np.concatenate(ks)
cap = 4096
bb = capi.boundary_block_bytes(cap)
dn = torch.zeros(bb, dtype=torch.uint8, device=dev); up = torch.zeros(bb, dtype=torch.uint8, device=dev)
out = {}
for world in (2, 4, 8):
    edges = part.balanced_edges(keys, world)
    ranks = []
    for rank in range(world):
        v = capi.Volume(res, cam, max_chunks=1 << 18, max_list=1 << 17)
        v.set_partition(edges[rank], edges[rank + 1], axis)
        idx = list(range(ORBIT))
        v.stream_frames_device([dd[i].data_ptr() for i in idx], [dc[i].data_ptr() for i in idx], poses[idx])  # pre-roll
        v.boundary_pack_bands2(dn.data_ptr(), cap, up.data_ptr(), cap)  # drop what the pre-roll flagged
        v.sync()
        rec = cap_rec = wire = 0
        worst = 0
        for k in range(NF):
            v.stream_frames_device([dd[k].data_ptr()], [dc[k].data_ptr()], poses[k:k + 1])
            sd, su, rb, ra = v.boundary_band_bounds(cap)
            v.boundary_pack_bands2(dn.data_ptr(), sd, up.data_ptr(), su)
            v.sync()   # (TF_ERR_CAPACITY would surface at the receiver; here the headers say it)
            n_dn = int(dn[:4].cpu().numpy().view(np.uint32)[0]); n_up = int(up[:4].cpu().numpy().view(np.uint32)[0])
            assert n_dn <= sd and n_up <= su, (world, rank, k, n_dn, sd, n_up, su)
            if rank > 0:
                rec += n_dn; cap_rec += sd; wire += capi.boundary_block_bytes(sd); worst = max(worst, sd)
            if rank + 1 < world:
                rec += n_up; cap_rec += su; wire += capi.boundary_block_bytes(su); worst = max(worst, su)
        v.close()
        ranks.append({"rank": rank, "records_per_frame": rec / NF, "capacity_per_frame": cap_rec / NF,
                      "wire_MB_per_frame": 1e-6 * wire / NF, "wire_over_records": wire / (rec * 8208.0) if rec else None,
                      "largest_block_records": worst})
    tot_w = sum(r["wire_MB_per_frame"] for r in ranks); tot_r = sum(r["records_per_frame"] for r in ranks)
    out["world_%d" % world] = {"edges": [int(e) for e in edges[1:-1]], "ranks": ranks,
                               "wire_over_records_all_ranks": 1e6 * tot_w / (tot_r * 8208.0),
                               "busiest_rank_wire_MB_per_frame": max(r["wire_MB_per_frame"] for r in ranks),
                               "fixed_blocks_r3_MB_per_frame_per_rank": 2 * 1e-6 * capi.boundary_block_bytes(1024)}
print(json.dumps(out))
