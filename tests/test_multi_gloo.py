"""N > 1 path on CPU: two processes over gloo.  Each rank owns a ChunkID.x slab, runs selection in
full, integrates only its slab (oracle as the per-rank compute -- test infrastructure), exchanges
the [count | records] blocks of the updated ghost-band chunks with ONE all-gather (the product's exchange
helper, same block layout as tf_boundary_pack_block), and the union of the partitions must equal the
single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pack_block(vol, ids, needs, lo, hi, cap):
    """numpy twin of tf_boundary_pack_block: [u32 count, pad | records of int4 id | {sdf,w}[512] | colour[512][4]]."""
    from texturefusion_amd import exchange, partition as part
    face = part.boundary_mask(ids, lo, hi) & (needs != 0)
    sel = ids[face]
    assert len(sel) <= cap
    blk = np.zeros(exchange.block_bytes(cap), np.uint8)
    blk[:4] = np.array([len(sel)], np.uint32).view(np.uint8)
    rec = blk[exchange.HEADER_BYTES:].reshape(cap, exchange.RECORD_BYTES)
    for i, cid in enumerate(sel):
        s, w, c = vol.get_chunk(cid)
        rec[i, :12] = np.asarray(cid, np.int32).view(np.uint8)
        tw = np.stack([s, w], 1).astype(np.float32)
        rec[i, 16:16 + 4096] = tw.reshape(-1).view(np.uint8)
        rec[i, 16 + 4096:] = c.view(np.uint8)
    return blk


def _unpack_blocks(vol, allb, world, own, cap):
    """numpy twin of tf_boundary_unpack_blocks."""
    from texturefusion_amd import exchange
    bb = exchange.block_bytes(cap)
    for r in range(world):
        if r == own:
            continue
        blk = allb[r * bb:(r + 1) * bb]
        n = int(blk[:4].view(np.uint32)[0])
        rec = blk[exchange.HEADER_BYTES:].reshape(cap, exchange.RECORD_BYTES)
        for i in range(n):
            cid = rec[i, :12].view(np.int32).copy()
            tw = rec[i, 16:16 + 4096].view(np.float32).reshape(512, 2)
            col = rec[i, 16 + 4096:].view(np.uint16).copy()
            vol.set_chunk(cid, tw[:, 0].copy(), tw[:, 1].copy(), col)


def _frame(k, cam):
    from texturefusion_amd import synth
    pose = synth.pose_euler(0.15 * (k + 1), 0.05 * k, 0.0, (0.05 * k, 0.0, 0.0))
    depth, rgba, q, _ = synth.wall_frame(1.2, cam, pose=pose, seed=k)
    return depth, rgba, q, pose


def _worker(rank, port, out_dir):
    import torch
    import torch.distributed as dist
    from oracle import api as O
    from texturefusion_amd import exchange, partition as part, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    res = np.float32(0.01)
    cam = synth.Camera()
    C, ig = O.camera_from(cam), O.default_integrator()
    ext = part.room_extent_chunks(res, half_x=0.6, margin=0.1)
    lo, hi = part.slab_for_rank(ext, rank, WORLD)
    vol = O.Volume(res, C, ig)
    cap = 256
    all_needs = []
    for k in (0, 1):
        depth, rgba, q, pose = _frame(k, cam)
        ids, new = vol.prepare(depth, pose)  # full selection on every rank
        own = part.owner_of(ids, ext, WORLD) == rank
        mine = ids[own]
        nd = np.zeros(len(mine), np.uint8)
        vol.integrate(depth, rgba, None, pose, mine, nd, 1, -1)
        needs = np.zeros(len(ids), np.uint8)
        needs[own] = nd
        # chunks created by prepare but owned elsewhere are not this rank's business
        newm = new.copy()
        vol.finalize(ids[own], nd, newm[own])
        for cid in ids[~own]:
            pass
        # the path's one collective: a fixed-capacity all-gather of [count | records] blocks, counts in-band
        blk = torch.from_numpy(_pack_block(vol, ids, needs, lo, hi, cap))
        allb = exchange.allgather_blocks(blk)
        _unpack_blocks(vol, allb.numpy(), WORLD, rank, cap)
        # merged needsUpdate flags in reference list order
        tn = torch.from_numpy(needs.copy())
        gl = [torch.zeros_like(tn) for _ in range(WORLD)]
        dist.all_gather(gl, tn)
        merged = part.merge_needs([g.numpy() for g in gl], ids, ext, WORLD)
        all_needs.append(merged)
    owned = [cid for cid in vol.list_chunks() if lo <= cid[0] < hi]
    ghosts = [cid for cid in vol.list_chunks() if not (lo <= cid[0] < hi)]
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
             owned=np.asarray(owned, np.int32).reshape(-1, 3),
             ghosts=np.asarray(ghosts, np.int32).reshape(-1, 3),
             owned_sdf=np.stack([vol.get_chunk(c)[0] for c in owned]),
             owned_w=np.stack([vol.get_chunk(c)[1] for c in owned]),
             owned_col=np.stack([vol.get_chunk(c)[2] for c in owned]),
             ghost_sdf=np.stack([vol.get_chunk(c)[0] for c in ghosts]) if ghosts else np.zeros((0, 512), np.float32),
             needs=np.concatenate(all_needs))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partition_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    from oracle import api as O
    from texturefusion_amd import synth

    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)

    res = np.float32(0.01)
    cam = synth.Camera()
    ref = O.Volume(res, O.camera_from(cam), O.default_integrator())
    ref_needs = []
    for k in (0, 1):
        depth, rgba, q, pose = _frame(k, cam)
        ids, new = ref.prepare(depth, pose)
        nd = np.zeros(len(ids), np.uint8)
        ref.integrate(depth, rgba, None, pose, ids, nd, 1, -1)
        ref.finalize(ids, nd, new)
        ref_needs.append(nd)
    ref_needs = np.concatenate(ref_needs)
    ref_ids = {tuple(c) for c in ref.list_chunks()}
    seen = set()
    for r in range(WORLD):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(z["needs"], ref_needs)          # merged flags == single-process flags
        for i, cid in enumerate(z["owned"]):
            t = tuple(int(x) for x in cid)
            if t not in ref_ids:
                # created by the full selection but never updated on the owner: garbage in the reference
                assert np.all(z["owned_w"][i] == 0)
                continue
            seen.add(t)
            s, w, c = ref.get_chunk(cid)
            assert np.array_equal(s.view(np.uint32), z["owned_sdf"][i].view(np.uint32))
            assert np.array_equal(w.view(np.uint32), z["owned_w"][i].view(np.uint32))
            assert np.array_equal(c, z["owned_col"][i])
        for i, cid in enumerate(z["ghosts"]):  # ghost copies carry the owner's current TSDF
            t = tuple(int(x) for x in cid)
            if t in ref_ids and z["ghost_sdf"][i].min() < 999.0:
                s, _, _ = ref.get_chunk(cid)
                assert np.array_equal(s.view(np.uint32), z["ghost_sdf"][i].view(np.uint32))
    assert seen == ref_ids


# ---- the neighbour form: three ranks, blocks only between adjacent slabs -------------------------------------
def _pack_bands(vol, ids, needs, lo, hi, cap):
    """numpy twin of tf_boundary_pack_bands (ownership key = ChunkID.x, so a + b + c = 1): the down block holds the
    updated chunks with lo <= x <= lo + 1, the up block those with x == hi - 1."""
    from texturefusion_amd import exchange
    x = np.asarray(ids, np.int64)[:, 0]
    upd = needs != 0
    out = []
    for face in ((x >= lo) & (x <= lo + 1) & upd, (x == hi - 1) & upd):
        sel = ids[face]
        assert len(sel) <= cap
        blk = np.zeros(exchange.block_bytes(cap), np.uint8)
        blk[:4] = np.array([len(sel)], np.uint32).view(np.uint8)
        rec = blk[exchange.HEADER_BYTES:].reshape(cap, exchange.RECORD_BYTES)
        for i, cid in enumerate(sel):
            s_, w, c = vol.get_chunk(cid)
            rec[i, :12] = np.asarray(cid, np.int32).view(np.uint8)
            rec[i, 16:16 + 4096] = np.stack([s_, w], 1).astype(np.float32).reshape(-1).view(np.uint8)
            rec[i, 16 + 4096:] = c.view(np.uint8)
        out.append(blk)
    return out


def _worker3(rank, port, out_dir, sized=False):
    import torch
    import torch.distributed as dist
    from oracle import api as O
    from texturefusion_amd import exchange, partition as part, synth

    world = 3
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = np.float32(0.01)
    cam = synth.Camera()
    ext = part.room_extent_chunks(res, half_x=0.6, margin=0.1)
    lo, hi = part.slab_for_rank(ext, rank, world)
    vol = O.Volume(res, O.camera_from(cam), O.default_integrator())
    cap = 256
    received = 0
    for k in (0, 1, 2):
        depth, rgba, q, pose = _frame(k, cam)
        ids, new = vol.prepare(depth, pose)
        own = part.owner_of(ids, ext, world) == rank
        nd = np.zeros(int(own.sum()), np.uint8)
        vol.integrate(depth, rgba, None, pose, ids[own], nd, 1, -1)
        needs = np.zeros(len(ids), np.uint8)
        needs[own] = nd
        vol.finalize(ids[own], nd, new[own])
        down, up = _pack_bands(vol, ids, needs, lo, hi, cap)
        if sized:
            # every rank sizes its four blocks from ITS copy of the full selection; gloo rejects a receive whose length
            # differs from the matching send, so a disagreement between the two sides of a transfer fails here
            cnt = part.band_counts(ids, lo, hi)
            b = [part.xchg_bucket(c, cap) for c in cnt]
            assert int(down[:4].view(np.uint32)[0]) <= b[0] and int(up[:4].view(np.uint32)[0]) <= b[1]  # selected >= updated
            below, above = exchange.neighbour_exchange_sized(
                torch.from_numpy(down[:exchange.block_bytes(b[0])].copy()), torch.from_numpy(up[:exchange.block_bytes(b[1])].copy()),
                exchange.block_bytes(b[2]), exchange.block_bytes(b[3]))
            moved = (exchange.block_bytes(b[2]) if rank > 0 else 0) + (exchange.block_bytes(b[3]) if rank + 1 < world else 0)
            bb = exchange.block_bytes(cap)
            two = np.zeros(2 * bb, np.uint8)
            two[:len(below)] = below.numpy()
            two[bb:bb + len(above)] = above.numpy()
            wire = locals().get("wire", 0) + moved
        else:
            below, above = exchange.neighbour_exchange(torch.from_numpy(down), torch.from_numpy(up))
            two = np.concatenate([below.numpy(), above.numpy()])
            wire = locals().get("wire", 0) + (rank > 0) * len(down) + (rank + 1 < world) * len(up)
        received += int(below.numpy()[:4].view(np.uint32)[0]) + int(above.numpy()[:4].view(np.uint32)[0])
        _unpack_blocks(vol, two, 2, -1, cap)
    chunks = vol.list_chunks()
    np.savez(os.path.join(out_dir, "n%d.npz" % rank), ids=np.asarray(chunks, np.int32).reshape(-1, 3),
             sdf=np.stack([vol.get_chunk(c)[0] for c in chunks]), w=np.stack([vol.get_chunk(c)[1] for c in chunks]),
             lo=lo, hi=hi, received=received, wire=wire)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sized", [False, True], ids=["fixed_capacity", "sized_by_selection"])
def test_three_rank_neighbour_exchange_equals_single_process(tmp_path, sized):
    """Slabs [.., e1) [e1, e2) [e2, ..): every rank sends its down band to the rank below and its up band to the rank
    above (two blocks received whatever the number of ranks).  Owned chunks and the ghost copies a rank's mesher would
    read (x in [lo - 1, hi + 1]) equal the single-process volume bit for bit."""
    import torch.multiprocessing as mp
    from oracle import api as O
    from texturefusion_amd import synth

    port = _free_port()
    mp.spawn(_worker3, args=(port, str(tmp_path), sized), nprocs=3, join=True)
    res = np.float32(0.01)
    cam = synth.Camera()
    ref = O.Volume(res, O.camera_from(cam), O.default_integrator())
    for k in (0, 1, 2):
        depth, rgba, q, pose = _frame(k, cam)
        ids, new = ref.prepare(depth, pose)
        nd = np.zeros(len(ids), np.uint8)
        ref.integrate(depth, rgba, None, pose, ids, nd, 1, -1)
        ref.finalize(ids, nd, new)
    ref_ids = {tuple(c) for c in ref.list_chunks()}
    owned_seen, ghosts_checked, received, wire = set(), 0, 0, 0
    for r in range(3):
        z = np.load(os.path.join(str(tmp_path), "n%d.npz" % r))
        lo, hi = int(z["lo"]), int(z["hi"])
        received += int(z["received"])
        wire += int(z["wire"])
        for i, cid in enumerate(z["ids"]):
            t = tuple(int(x) for x in cid)
            mine = lo <= t[0] < hi
            if t not in ref_ids:
                assert np.all(z["w"][i] == 0)  # selected everywhere, updated nowhere: garbage in the reference
                continue
            if mine:
                owned_seen.add(t)
            elif not (lo - 1 <= t[0] <= hi + 1) or z["sdf"][i].min() >= 999.0:
                continue  # outside the band this rank reads, or never received (not updated on its owner)
            else:
                ghosts_checked += 1
            s_, w, _ = ref.get_chunk(cid)
            assert np.array_equal(s_.view(np.uint32), z["sdf"][i].view(np.uint32)), (r, t, mine)
            assert np.array_equal(w.view(np.uint32), z["w"][i].view(np.uint32)), (r, t, mine)
    assert owned_seen == ref_ids and ghosts_checked > 20 and received > 20
    if sized:  # what moved is proportional to what changed (8-record buckets on a small scene: allow 2x + one bucket each)
        from texturefusion_amd import exchange
        assert wire <= 2.0 * received * exchange.RECORD_BYTES + 12 * exchange.block_bytes(8), (wire, received)
    else:
        assert wire > 4 * received * 8208   # the fixed blocks this replaces
