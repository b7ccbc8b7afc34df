"""Chunk-range partitions on ONE GPU: two handles own complementary ChunkID.x slabs, run the fused unit
on the same frames, exchange boundary records through tf_boundary_pack / tf_boundary_unpack; the union
of the partitions must equal the single-handle result bit for bit."""
import numpy as np
import pytest

from texturefusion_amd import capi, synth
from tests.util import HipBuffer, sorted_ids

pytestmark = pytest.mark.gpu
RES5 = np.float32(0.005)


@pytest.mark.parametrize("axis,split", [((1, 0, 0), 3), ((1, 1, 1), 35), ((1, 0, 1), 45)])
def test_two_partitions_equal_one(gpu_required, axis, split):
    cam = synth.Camera()
    single = capi.Volume(RES5, cam, max_chunks=1 << 16)
    parts = [capi.Volume(RES5, cam, max_chunks=1 << 16), capi.Volume(RES5, cam, max_chunks=1 << 16)]
    parts[0].set_partition(-(1 << 31), split, axis)
    parts[1].set_partition(split, (1 << 31) - 1, axis)
    cap = 4096
    bufs = [HipBuffer(cap * capi.TF_BOUNDARY_RECORD_BYTES) for _ in range(2)]
    for k in (0, 1, 2):
        depth, rgba, q, pose = synth.room_frame(k, cam)
        for v in [single] + parts:
            v.frame_upload(depth, rgba, None)
            v.integrate_frame(pose, True)
            v.sync()
        n = [parts[r].boundary_pack(bufs[r].ptr, cap) for r in range(2)]
        assert n[0] > 0 and n[1] > 0
        parts[0].boundary_unpack(bufs[1].ptr, n[1])
        parts[1].boundary_unpack(bufs[0].ptr, n[0])
        for v in parts:
            v.sync()
        # every packed record is an updated chunk of the packer's ghost band: the layer below the upper
        # face, and the sum(axis) + 1 layers above the lower face (what the mesher of the rank below reads:
        # c + {0,1}^3 and the face neighbours of those, Structure/ChunkManager.cpp:288-315,618-632)
        for r in range(2):
            rec = bufs[r].to_host(n[r] * capi.TF_BOUNDARY_RECORD_BYTES).reshape(n[r], -1)
            ids = rec[:, :12].copy().view(np.int32).reshape(-1, 3)
            k = ids.astype(np.int64) @ np.array(axis)
            if r == 0:
                assert np.all(k == split - 1)
            else:
                assert np.all((k >= split) & (k <= split + sum(axis)))
                assert set(np.unique(k)) == set(range(split, split + sum(axis) + 1))
    ref_ids = sorted_ids(single.list_chunks())
    s_ref, w_ref, c_ref = single.get_chunks(ref_ids)
    key = {tuple(c): i for i, c in enumerate(ref_ids)}
    seen = set()
    for r, v in enumerate(parts):
        ids = v.list_chunks()
        lo, hi = (-(1 << 31), split) if r == 0 else (split, (1 << 31) - 1)
        s, w, c = v.get_chunks(ids)
        for i, cid in enumerate(ids):
            t = tuple(int(x) for x in cid)
            owned = lo <= int(np.dot(cid.astype(np.int64), axis)) < hi
            if owned:
                assert t in key, "partition %d holds a chunk the single volume does not" % r
                seen.add(t)
            if t in key:  # owned chunks and ghost copies both carry the reference state
                j = key[t]
                assert np.array_equal(s[i].view(np.uint32), s_ref[j].view(np.uint32))
                assert np.array_equal(w[i].view(np.uint32), w_ref[j].view(np.uint32))
                assert np.array_equal(c[i], c_ref[j])
    assert seen == set(key)
    # every chunk the mesher of an owned chunk reads is present (and, by the loop above, bit-identical)
    offs = set()
    for o in np.ndindex(2, 2, 2):
        offs.add(o)
        for a in range(3):
            for d in (-1, 1):
                q = list(o)
                q[a] += d
                offs.add(tuple(q))
    for r, v in enumerate(parts):
        have = set(map(tuple, v.list_chunks().tolist()))
        lo, hi = (-(1 << 31), split) if r == 0 else (split, (1 << 31) - 1)
        for t in key:
            if not (lo <= int(np.dot(np.array(t, np.int64), axis)) < hi):
                continue
            for o in offs:
                q = (t[0] + o[0], t[1] + o[1], t[2] + o[2])
                if q in key:
                    assert q in have, "partition %d lacks mesher neighbour %s of owned chunk %s" % (r, q, t)
    dirty = set(map(tuple, single.dirty()))
    dparts = set(map(tuple, parts[0].dirty())) | set(map(tuple, parts[1].dirty()))
    assert dirty == dparts
    for v in [single] + parts:
        v.close()
    for b in bufs:
        b.free()


@pytest.mark.parametrize("axis,edges", [((1, 0, 0), (-2, 4)), ((1, 1, 1), (20, 45))])
def test_three_partitions_with_neighbour_band_blocks_equal_one(gpu_required, axis, edges):
    """The neighbour form of the exchange on one GPU: three handles own consecutive slabs; after every frame a handle's
    DOWN block (tf_boundary_pack_bands) goes to the handle below and its UP block to the one above, nobody else sees
    them.  Every owned chunk and every chunk an owned chunk's mesher reads equals the single volume bit for bit."""
    cam = synth.Camera()
    single = capi.Volume(RES5, cam, max_chunks=1 << 16)
    bounds = [-(1 << 31), edges[0], edges[1], (1 << 31) - 1]
    parts = [capi.Volume(RES5, cam, max_chunks=1 << 16) for _ in range(3)]
    for r, v in enumerate(parts):
        v.set_partition(bounds[r], bounds[r + 1], axis)
    cap = 2048
    bb = capi.boundary_block_bytes(cap)
    blk = [[HipBuffer(bb), HipBuffer(bb)] for _ in range(3)]  # [rank][down, up]
    pair = HipBuffer(2 * bb)
    sent = 0
    for k in (0, 1, 2, 3):
        depth, rgba, q, pose = synth.room_frame(2 * k, cam)
        for v in [single] + parts:
            v.frame_upload(depth, rgba, None)
            v.integrate_frame(pose, True)
        for r, v in enumerate(parts):
            v.boundary_pack_bands(blk[r][0].ptr, blk[r][1].ptr, cap)
            v.sync()
        for r in range(3):
            n_dn = int(blk[r][0].to_host(4).view(np.uint32)[0]); n_up = int(blk[r][1].to_host(4).view(np.uint32)[0])
            assert n_dn <= cap and n_up <= cap
            sent += n_dn + n_up
            for side, n in ((0, n_dn), (1, n_up)):  # every record sits in the band its block is for
                if n:
                    rec = blk[r][side].to_host(bb)[16:16 + n * capi.TF_BOUNDARY_RECORD_BYTES].reshape(n, -1)
                    key = rec[:, :12].copy().view(np.int32).reshape(-1, 3).astype(np.int64) @ np.array(axis)
                    if side == 0:
                        assert np.all((key >= bounds[r]) & (key <= bounds[r] + sum(axis)))
                    else:
                        assert np.all(key == bounds[r + 1] - 1)
        for r, v in enumerate(parts):  # from below: its UP block; from above: its DOWN block
            srcs = [blk[r - 1][1] if r > 0 else None, blk[r + 1][0] if r < 2 else None]
            host = np.zeros(2 * bb, np.uint8)
            for j, src in enumerate(srcs):
                if src is not None:
                    host[j * bb:(j + 1) * bb] = src.to_host(bb)
            pair.from_host(host)
            v.boundary_unpack_blocks(pair.ptr, 2, -1, cap, join_dirty=False)
            v.sync()
    assert sent > 100
    ref_ids = sorted_ids(single.list_chunks())
    s_ref, w_ref, c_ref = single.get_chunks(ref_ids)
    key = {tuple(c): i for i, c in enumerate(ref_ids)}
    offs = set()
    for o in np.ndindex(2, 2, 2):
        offs.add(o)
        for a in range(3):
            for d in (-1, 1):
                q = list(o); q[a] += d
                offs.add(tuple(q))
    seen = set()
    for r, v in enumerate(parts):
        ids = v.list_chunks()
        s, w, c = v.get_chunks(ids)
        have = {tuple(int(x) for x in cid): i for i, cid in enumerate(ids)}
        lo, hi = bounds[r], bounds[r + 1]
        for t, i in have.items():
            if lo <= int(np.dot(np.array(t, np.int64), axis)) < hi:
                assert t in key
                seen.add(t)
        for t in key:  # owned chunks and everything their meshers read
            if not (lo <= int(np.dot(np.array(t, np.int64), axis)) < hi):
                continue
            for o in offs:
                q = (t[0] + o[0], t[1] + o[1], t[2] + o[2])
                if q not in key:
                    continue
                assert q in have, "partition %d lacks mesher neighbour %s of owned chunk %s" % (r, q, t)
                i, j = have[q], key[q]
                assert np.array_equal(s[i].view(np.uint32), s_ref[j].view(np.uint32)), (r, q)
                assert np.array_equal(w[i].view(np.uint32), w_ref[j].view(np.uint32)), (r, q)
                assert np.array_equal(c[i], c_ref[j]), (r, q)
    assert seen == set(key)
    for v in [single] + parts:
        v.close()
    for b in [x for pr in blk for x in pr] + [pair]:
        b.free()


def test_async_pack_leaves_the_same_records_and_count(gpu_required):
    """tf_boundary_pack_async (count on the device, no host round trip) == tf_boundary_pack."""
    cam = synth.Camera()
    vols = [capi.Volume(RES5, cam, max_chunks=1 << 16) for _ in range(2)]
    for v in vols:
        v.set_partition(-2, 6)
    cap = 4096
    bufs = [HipBuffer(cap * capi.TF_BOUNDARY_RECORD_BYTES) for _ in range(2)]
    cnt = HipBuffer(4)
    for k in (0, 1):
        depth, rgba, q, pose = synth.room_frame(k, cam)
        for v in vols:
            v.frame_upload(depth, rgba, None)
            v.integrate_frame(pose, True)
    n_sync = vols[0].boundary_pack(bufs[0].ptr, cap)
    vols[1].boundary_pack_async(bufs[1].ptr, cap, cnt.ptr)
    vols[1].sync()
    n_async = int(cnt.to_host(4).view(np.uint32)[0])
    assert n_sync == n_async and n_sync > 0
    a = bufs[0].to_host(n_sync * capi.TF_BOUNDARY_RECORD_BYTES).reshape(n_sync, -1)
    b = bufs[1].to_host(n_sync * capi.TF_BOUNDARY_RECORD_BYTES).reshape(n_sync, -1)
    # record order follows the hash scan of each handle; compare as sets keyed by chunk id
    ka = {bytes(r[:12]): bytes(r) for r in a}
    kb = {bytes(r[:12]): bytes(r) for r in b}
    assert ka == kb
    # the touched bits are consumed: a second pack is empty
    assert vols[0].boundary_pack(bufs[0].ptr, cap) == 0
    for v in vols:
        v.close()
    for x in bufs + [cnt]:
        x.free()


def _copy_d2d(dst_ptr, src_ptr, nbytes):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(dst_ptr, src_ptr, nbytes, 3) == 0  # hipMemcpyDeviceToDevice
    assert hip.hipDeviceSynchronize() == 0  # a D2D hipMemcpy may return before the copy has landed


@pytest.mark.parametrize("caps", [[4096] * 8, [24, 24, 24, 4096, 4096, 4096, 4096, 4096]], ids=["fits", "overflow_then_recover"])
def test_textured_two_partitions_equal_one(gpu_required, caps):
    """The textured per-frame unit on two chunk-range partitions (x + y + z slabs) of one GPU: per frame voxel
    update on both, fixed-capacity blocks [count | records] exchanged as an all-gather would deliver them
    (tf_boundary_pack_block / tf_boundary_unpack_blocks with join_dirty), then the texture stage.  The union of
    the partitions' chunks and meshes equals the single volume bit for bit; every mesh is owned by exactly one.

    overflow_then_recover: the first three exchanges have room for 24 records (a ghost band has hundreds).  Those frames
    report TF_ERR_CAPACITY and mesh against stale ghosts; the senders keep what did not fit flagged, and the frame in
    which a ghost finally arrives puts its owned neighbours into THAT frame's dirty set -- so at the end voxels and mesh
    geometry are those of the single volume again."""
    cam = synth.Camera()
    axis, split = (1, 1, 1), 35
    single = capi.Volume(RES5, cam, max_chunks=1 << 16)
    parts = [capi.Volume(RES5, cam, max_chunks=1 << 16) for _ in range(2)]
    parts[0].set_partition(-(1 << 31), split, axis)
    parts[1].set_partition(split, (1 << 31) - 1, axis)
    bb_max = capi.boundary_block_bytes(max(caps))
    blocks = [HipBuffer(2 * bb_max) for _ in range(2)]  # what each rank holds after the all-gather
    mine = [HipBuffer(bb_max) for _ in range(2)]
    n = len(caps)
    frames = [synth.room_frame(3 * k, cam, with_quality=False) for k in range(n)]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in frames]
    overflows = 0
    for k, f in enumerate(frames):
        cap = caps[k]
        bb = capi.boundary_block_bytes(cap)
        T = synth.pose_inverse16(f[3])
        dd, dr = [bufs[k][0].ptr], [bufs[k][1].ptr]
        single.stream_frames_textured_device(dd, dr, f[3].reshape(1, 12), T.reshape(1, 16), k)
        for r, v in enumerate(parts):
            v.stream_frames_device(dd, dr, f[3].reshape(1, 12))
            v.boundary_pack_block(mine[r].ptr, cap)
        for v in parts:
            v.sync()
        for r in range(2):  # "all-gather": every rank ends up with [block of rank 0 | block of rank 1]
            for q in range(2):
                _copy_d2d(blocks[r].ptr + q * bb, mine[q].ptr, bb)
        for r, v in enumerate(parts):
            v.boundary_unpack_blocks(blocks[r].ptr, 2, r, cap, join_dirty=True)
            v.texture_frame_device(T, k)
            try:
                v.sync()
            except capi.TFError as e:
                assert cap < 4096 and e.code == capi.TF_ERR_CAPACITY, e
                overflows += 1
    assert overflows == (0 if min(caps) == 4096 else 6)
    single.sync()
    ref_ids = sorted_ids(single.list_chunks())
    key = {tuple(c): i for i, c in enumerate(ref_ids)}
    s_ref, w_ref, c_ref = single.get_chunks(ref_ids)
    ref_m = sorted_ids(single.list_meshes())
    assert len(ref_m) > 300
    mvoff, mioff, mV, mN, mC, mI, madj, msimp = single.get_meshes(ref_m)
    mkey = {tuple(c): i for i, c in enumerate(ref_m)}
    seen_m = set()
    for r, v in enumerate(parts):
        lo, hi = (-(1 << 31), split) if r == 0 else (split, (1 << 31) - 1)
        ids = v.list_chunks()
        s, w, c = v.get_chunks(ids)
        for i, cid in enumerate(ids):
            t = tuple(int(x) for x in cid)
            if t in key:
                j = key[t]
                assert np.array_equal(s[i].view(np.uint32), s_ref[j].view(np.uint32)), (r, t)
                assert np.array_equal(c[i], c_ref[j])
        pm = sorted_ids(v.list_meshes())
        own = [m for m in pm if lo <= int(np.dot(m.astype(np.int64), axis)) < hi]
        assert len(own) == len(pm), "a rank meshes only the chunks it owns"
        voff, ioff, V, N, Cc, I, adj, simp = v.get_meshes(pm)
        for i, cid in enumerate(pm):
            t = tuple(int(x) for x in cid)
            assert t in mkey and t not in seen_m
            seen_m.add(t)
            j = mkey[t]
            assert np.array_equal(V[voff[i]:voff[i + 1]].view(np.uint32), mV[mvoff[j]:mvoff[j + 1]].view(np.uint32)), (r, t)
            assert np.array_equal(N[voff[i]:voff[i + 1]].view(np.uint32), mN[mvoff[j]:mvoff[j + 1]].view(np.uint32)), (r, t)
            assert np.array_equal(I[ioff[i]:ioff[i + 1]], mI[mioff[j]:mioff[j + 1]]), (r, t)
    assert seen_m == set(mkey)
    for v in [single] + parts:
        v.close()
    for b in blocks + mine + [x for p in bufs for x in p]:
        b.free()


def test_rccl_single_rank_plumbing(gpu_required):
    """tf_comm_init / tf_exchange_boundary with one rank: librccl is opened, the communicator comes up, the
    all-gather runs on the handle's stream, the own block is skipped -- and the textured flow with an exchange
    after every frame equals the plain one."""
    cam = synth.Camera()
    a = capi.Volume(RES5, cam, max_chunks=1 << 15)
    b = capi.Volume(RES5, cam, max_chunks=1 << 15)
    b.set_partition(-40, 60, (1, 1, 1))
    a.set_partition(-40, 60, (1, 1, 1))
    b.comm_init(0, 1, capi.comm_unique_id())
    b.comm_exchange_every_frame(2048)
    frames = [synth.room_frame(2 * k, cam, with_quality=False) for k in range(4)]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in frames]
    for k, f in enumerate(frames):
        T = synth.pose_inverse16(f[3])
        for v in (a, b):
            v.stream_frames_textured_device([bufs[k][0].ptr], [bufs[k][1].ptr], f[3].reshape(1, 12), T.reshape(1, 16), k)
    b.exchange_boundary(2048)
    a.sync(); b.sync()
    ia, ib = sorted_ids(a.list_chunks()), sorted_ids(b.list_chunks())
    assert np.array_equal(ia, ib)
    sa, wa, ca = a.get_chunks(ia[::9]); sb, wb, cb = b.get_chunks(ia[::9])
    assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32)) and np.array_equal(ca, cb)
    assert np.array_equal(sorted_ids(a.list_meshes()), sorted_ids(b.list_meshes()))
    a.close(); b.close()
    for p in bufs:
        p[0].free(); p[1].free()


@pytest.mark.parametrize("edges,min_recv", [((24, 46), 300), ((-400, -390), 0), ((40, 44), 100)])
def test_sized_neighbour_exchange_three_partitions_textured(gpu_required, edges, min_recv):
    """The SIZED neighbour exchange (tf_boundary_band_bounds / tf_boundary_pack_bands2 / tf_boundary_unpack_pair) with
    the textured unit on three consecutive x + y + z slabs of one GPU.  Every rank sizes its four blocks from its own
    copy of the frame's selection; the test checks that the two sides of every transfer computed the SAME capacity (on a
    real transport a mismatch is a hang), that nothing overflowed, that the bytes moved stay within 1.5x of what the
    records need, and that chunks and meshes of the union equal the single volume bit for bit."""
    # (second case: two slabs far outside the scene -- empty lists, empty blocks of the minimum size, one rank does all the
    # work; third case: a middle slab of the minimum width the neighbour form allows, a + b + c + 1 = 4 keys)
    cam = synth.Camera()
    axis = (1, 1, 1)
    bounds = [-(1 << 31), edges[0], edges[1], (1 << 31) - 1]
    single = capi.Volume(RES5, cam, max_chunks=1 << 16)
    parts = [capi.Volume(RES5, cam, max_chunks=1 << 16) for _ in range(3)]
    for r, v in enumerate(parts):
        v.set_partition(bounds[r], bounds[r + 1], axis)
    cap = 4096
    bb = capi.boundary_block_bytes(cap)
    out = [[HipBuffer(bb), HipBuffer(bb)] for _ in range(3)]    # [rank][down, up]
    inn = [[HipBuffer(bb), HipBuffer(bb)] for _ in range(3)]    # [rank][from below, from above]
    n = 8
    frames = [synth.room_frame(3 * k, cam, with_quality=False) for k in range(n)]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in frames]
    wire = 0
    zero_hdr = np.zeros(16, np.uint8)
    for k, f in enumerate(frames):
        T = synth.pose_inverse16(f[3])
        dd, dr = [bufs[k][0].ptr], [bufs[k][1].ptr]
        single.stream_frames_textured_device(dd, dr, f[3].reshape(1, 12), T.reshape(1, 16), k)
        bnd = []
        for r, v in enumerate(parts):
            v.stream_frames_device(dd, dr, f[3].reshape(1, 12))
            bnd.append(v.boundary_band_bounds(cap))
        for r in range(3):
            sd, su, rb, ra = bnd[r]
            assert all(8 <= x <= cap and x % 8 == 0 for x in bnd[r])
            if r > 0:
                assert sd == bnd[r - 1][3] and rb == bnd[r - 1][1], (k, r, bnd)   # both ends of a transfer agree
            if r < 2:
                assert su == bnd[r + 1][2] and ra == bnd[r + 1][0], (k, r, bnd)
        for r, v in enumerate(parts):
            v.boundary_pack_bands2(out[r][0].ptr, bnd[r][0], out[r][1].ptr, bnd[r][1])
            v.sync()
        for r in range(3):  # the "wire": exactly tf_boundary_block_bytes(bound) bytes per transfer
            if r > 0:
                nb = capi.boundary_block_bytes(bnd[r][0])
                _copy_d2d(inn[r - 1][1].ptr, out[r][0].ptr, nb); wire += nb
            else:
                inn[r][0].from_host(zero_hdr)
            if r < 2:
                nb = capi.boundary_block_bytes(bnd[r][1])
                _copy_d2d(inn[r + 1][0].ptr, out[r][1].ptr, nb); wire += nb
            else:
                inn[r][1].from_host(zero_hdr)
        for r, v in enumerate(parts):
            v.boundary_unpack_pair(inn[r][0].ptr, bnd[r][2], inn[r][1].ptr, bnd[r][3], join_dirty=True)
            v.texture_frame_device(T, k)
            v.sync()    # (raises TF_ERR_CAPACITY if a block overflowed: the bound must hold everything the frame flagged)
    single.sync()
    sent = sum(v.comm_stats_ex()["records_sent"] for v in parts)
    recv = sum(v.comm_stats_ex()["records_received"] for v in parts)
    # rank 0's down block and rank 2's up block go nowhere (no such neighbour): records_sent counts them, the wire does not
    assert recv >= min_recv and sent >= recv
    # (a cold volume, frames three orbit steps apart: selected / updated is at its largest here, ~1.4, + 8-record buckets;
    # tools/exp_sized_exchange.py measures the steady-state stream of the bench)
    if min_recv >= 300:
        assert wire <= 1.6 * recv * capi.TF_BOUNDARY_RECORD_BYTES, (wire, recv)
    ref_ids = sorted_ids(single.list_chunks())
    key = {tuple(c): i for i, c in enumerate(ref_ids)}
    s_ref, w_ref, c_ref = single.get_chunks(ref_ids)
    ref_m = sorted_ids(single.list_meshes())
    assert len(ref_m) > 300
    mvoff, mioff, mV, mN, mC, mI, madj, msimp = single.get_meshes(ref_m)
    mkey = {tuple(c): i for i, c in enumerate(ref_m)}
    seen_m, seen_c = set(), set()
    for r, v in enumerate(parts):
        lo, hi = bounds[r], bounds[r + 1]
        ids = v.list_chunks()
        s, w, c = v.get_chunks(ids)
        for i, cid in enumerate(ids):
            t = tuple(int(x) for x in cid)
            if lo <= sum(t) < hi:
                assert t in key
                seen_c.add(t)
            if t in key:
                j = key[t]
                assert np.array_equal(s[i].view(np.uint32), s_ref[j].view(np.uint32)), (r, t)
                assert np.array_equal(w[i].view(np.uint32), w_ref[j].view(np.uint32)), (r, t)
                assert np.array_equal(c[i], c_ref[j])
        pm = sorted_ids(v.list_meshes())
        voff, ioff, V, N, Cc, I, adj, simp = v.get_meshes(pm)
        for i, cid in enumerate(pm):
            t = tuple(int(x) for x in cid)
            assert lo <= sum(t) < hi and t in mkey and t not in seen_m
            seen_m.add(t)
            j = mkey[t]
            assert np.array_equal(V[voff[i]:voff[i + 1]].view(np.uint32), mV[mvoff[j]:mvoff[j + 1]].view(np.uint32)), (r, t)
            assert np.array_equal(N[voff[i]:voff[i + 1]].view(np.uint32), mN[mvoff[j]:mvoff[j + 1]].view(np.uint32)), (r, t)
            assert np.array_equal(I[ioff[i]:ioff[i + 1]], mI[mioff[j]:mioff[j + 1]]), (r, t)
    assert seen_c == set(key) and seen_m == set(mkey)
    for v in [single] + parts:
        v.close()
    for b in [x for pr in out + inn for x in pr] + [x for p in bufs for x in p]:
        b.free()


def test_pack_bands_overflow_on_one_side_keeps_the_other_block_whole(gpu_required):
    """A chunk of a thin slab belongs to both bands.  When only ONE block is too small, the other still gets every
    record its count announces (no counted-but-unwritten record), and what did not fit stays flagged for a retry."""
    cam = synth.Camera()
    v = capi.Volume(RES5, cam, max_chunks=1 << 16)
    v.set_partition(30, 33, (1, 1, 1))  # three keys wide: every owned chunk is in the down band, key 32 also in the up band
    depth, rgba, q, pose = synth.room_frame(0, cam)
    v.frame_upload(depth, rgba, None)
    v.integrate_frame(pose, True)
    big, small = 4096, 8
    dn, up = HipBuffer(capi.boundary_block_bytes(big)), HipBuffer(capi.boundary_block_bytes(big))
    v.boundary_pack_bands2(dn.ptr, big, up.ptr, small)
    v.sync()
    n_dn = int(dn.to_host(4).view(np.uint32)[0]); n_up = int(up.to_host(4).view(np.uint32)[0])
    assert n_dn > 100 and n_up > small
    rec = dn.to_host(capi.boundary_block_bytes(big))[16:16 + n_dn * capi.TF_BOUNDARY_RECORD_BYTES].reshape(n_dn, -1)
    ids = rec[:, :12].copy().view(np.int32).reshape(-1, 3)
    keys = ids.astype(np.int64).sum(1)
    assert np.all((keys >= 30) & (keys <= 32)) and len({tuple(x) for x in ids.tolist()}) == n_dn   # every record written, once
    # retry with room: exactly the chunks whose up side did not fit come again (in both blocks: still flagged as a whole)
    v.boundary_pack_bands2(dn.ptr, big, up.ptr, big)
    v.sync()
    n_dn2 = int(dn.to_host(4).view(np.uint32)[0]); n_up2 = int(up.to_host(4).view(np.uint32)[0])
    assert n_up2 == n_up - small and n_dn2 == n_up2
    v.boundary_pack_bands2(dn.ptr, big, up.ptr, big)
    v.sync()
    assert int(dn.to_host(4).view(np.uint32)[0]) == 0 and int(up.to_host(4).view(np.uint32)[0]) == 0
    v.close(); dn.free(); up.free()


@pytest.mark.parametrize("ahead", [0, 2], ids=["call_by_call", "primed_pipeline"])
def test_rccl_single_rank_sized_per_frame_exchange(gpu_required, ahead):
    """The in-library per-frame exchange in its sized form with one rank (what can run without a second GPU): the band
    counts travel from the selection role to the host (published by the previous exchange's unpack launch when the stream
    is fed with frames ahead, by a launch of its own otherwise), the blocks are packed with the sized capacities, nothing
    overflows, and the volume equals one that never exchanged."""
    cam = synth.Camera()
    a = capi.Volume(RES5, cam, max_chunks=1 << 15)
    b = capi.Volume(RES5, cam, max_chunks=1 << 15)
    for v in (a, b):
        v.set_partition(20, 48, (1, 1, 1))
    b.comm_init(0, 1, capi.comm_unique_id())
    b.comm_exchange_every_frame(2048)
    n = 6
    frames = [synth.room_frame(2 * k, cam, with_quality=False) for k in range(n + ahead)]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in frames]
    poses = np.stack([f[3].reshape(12) for f in frames])
    pinv = np.stack([synth.pose_inverse16(f[3]) for f in frames])
    for k in range(n):
        idx = list(range(k, k + 1 + ahead))
        for v in (a, b):
            v.stream_frames_textured_device([bufs[i][0].ptr for i in idx], [bufs[i][1].ptr for i in idx], poses[idx], pinv[idx], k,
                                            n_ahead=ahead)
    a.sync(); b.sync()   # (TF_ERR_CAPACITY here = a sized block was too small)
    st = b.comm_stats_ex()
    assert st["exchanges"] == n and st["checked"] == 1 and st["mode"] == 0
    assert st["overlapped"] == n             # every exchange ran on the second stream next to the interior mesh pass
    assert st["records_sent"] > 200          # packed (the single rank has nobody to send to)
    assert st["bytes_sent"] == 0 and st["bytes_received"] == 0
    ia, ib = sorted_ids(a.list_chunks()), sorted_ids(b.list_chunks())
    assert np.array_equal(ia, ib)
    sa, wa, ca = a.get_chunks(ia[::7]); sb, wb, cb = b.get_chunks(ia[::7])
    assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32)) and np.array_equal(ca, cb)
    ma, mb = sorted_ids(a.list_meshes()), sorted_ids(b.list_meshes())
    assert np.array_equal(ma, mb) and len(ma) > 100
    ga, gb = a.get_meshes(ma), b.get_meshes(mb)
    assert np.array_equal(ga[2].view(np.uint32), gb[2].view(np.uint32)) and np.array_equal(ga[5], gb[5])
    a.close(); b.close()
    for p in bufs:
        p[0].free(); p[1].free()


@pytest.mark.parametrize("flow", ["texture_frame_device", "keyframe_unit"])
def test_rccl_single_rank_exchange_behind_lists_without_band_counts(gpu_required, flow):
    """The per-frame exchange behind entry points whose lists carry NO band counts (tf_texture_frame_device after a plain
    voxel update; the keyframe unit, whose selection is the unordered non-fused one): the blocks must have the caller's fixed
    capacity there -- sizing them from FrameCtl::band_cnt, which only the fused stream's selection role counts, gave
    8-record blocks and TF_ERR_CAPACITY as soon as more than eight band chunks were touched."""
    cam = synth.Camera()
    a = capi.Volume(RES5, cam, max_chunks=1 << 15)
    b = capi.Volume(RES5, cam, max_chunks=1 << 15)
    for v in (a, b):
        v.set_partition(20, 48, (1, 1, 1))
    b.comm_init(0, 1, capi.comm_unique_id())
    b.comm_exchange_every_frame(2048)
    n = 7 if flow == "keyframe_unit" else 4
    frames = [synth.room_frame(2 * k, cam, with_quality=False) for k in range(n)]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in frames]
    if flow == "texture_frame_device":
        for k, f in enumerate(frames):
            T = synth.pose_inverse16(f[3])
            for v in (a, b):
                v.stream_frames_device([bufs[k][0].ptr], [bufs[k][1].ptr], f[3].reshape(1, 12))
                v.texture_frame_device(T, k)
        want_x = n
    else:
        for v in (a, b):
            g = capi.Volume.unit_group(77, (bufs[0][0].ptr, bufs[0][1].ptr, 0, frames[0][3]),
                                       [(bufs[k][0].ptr, frames[k][3]) for k in range(1, 7)])
            v.keyframe_unit(fresh=g, moved=[], texture=True, pose_inv16=synth.pose_inverse16(frames[0][3]))
        want_x = 1
    a.sync(); b.sync()   # (TF_ERR_CAPACITY here = a block was sized from counts nobody made)
    st = b.comm_stats_ex()
    assert st["exchanges"] == want_x and st["records_sent"] > 8
    ia, ib = sorted_ids(a.list_chunks()), sorted_ids(b.list_chunks())
    assert np.array_equal(ia, ib) and len(ia) > 500
    sa, wa, ca = a.get_chunks(ia[::7]); sb, wb, cb = b.get_chunks(ia[::7])
    assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32)) and np.array_equal(ca, cb)
    ma, mb = sorted_ids(a.list_meshes()), sorted_ids(b.list_meshes())
    assert np.array_equal(ma, mb)
    a.close(); b.close()
    for p in bufs:
        p[0].free(); p[1].free()


@pytest.mark.parametrize("edges", [(24, 46), (40, 44), "hall"])
def test_three_partitions_textured_exchange_overlapped_with_interior_meshes(gpu_required, edges):
    """The overlapped order: every rank meshes the INTERIOR chunks of its dirty set (27-neighbourhood
    owned) BEFORE the ghosts arrive (tf_texture_frame_device_phase 1), the exchange happens, then the boundary chunks
    and what the ghosts added (phase 2).  Chunks, meshes, patches' slots of the union equal the single volume bit for bit:
    the interior pass read nothing the exchange brings."""
    hall = edges == "hall"  # (the 1280x960 hall: dirty lists of ~70 k entries -- the filter's workgroup-batch form in the interior pass)
    cam = synth.Camera.hires() if hall else synth.Camera()
    axis = (1, 1, 1)
    pool = 1 << 19 if hall else 1 << 16
    n = 6 if hall else 8
    if hall:
        frames = [synth.room_frame(k, cam, half=(4.0, 3.0, 4.0), radius=3.2, with_quality=False) for k in range(18, 18 + n)]
    else:
        frames = [synth.room_frame(3 * k, cam, with_quality=False) for k in range(n)]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in frames]
    single = capi.Volume(RES5, cam, max_chunks=pool)
    for k, f in enumerate(frames):  # the reference run first (the hall's slab edges are cut where its meshes are)
        T = synth.pose_inverse16(f[3])
        single.stream_frames_textured_device([bufs[k][0].ptr], [bufs[k][1].ptr], f[3].reshape(1, 12), T.reshape(1, 16), k)
    single.sync()
    if hall:
        mk = np.sort(single.list_meshes().astype(np.int64).sum(1))
        edges = (int(mk[len(mk) // 3]), int(mk[2 * len(mk) // 3]) + 4)
    bounds = [-(1 << 31), edges[0], edges[1], (1 << 31) - 1]
    parts = [capi.Volume(RES5, cam, max_chunks=pool) for _ in range(3)]
    for r, v in enumerate(parts):
        v.set_partition(bounds[r], bounds[r + 1], axis)
    cap = 16384 if hall else 4096
    bb = capi.boundary_block_bytes(cap)
    out = [[HipBuffer(bb), HipBuffer(bb)] for _ in range(3)]
    inn = [[HipBuffer(bb), HipBuffer(bb)] for _ in range(3)]
    zero_hdr = np.zeros(16, np.uint8)
    for k, f in enumerate(frames):
        T = synth.pose_inverse16(f[3])
        dd, dr = [bufs[k][0].ptr], [bufs[k][1].ptr]
        for v in parts:
            v.stream_frames_device(dd, dr, f[3].reshape(1, 12))
            v.texture_frame_device_phase(T, k, 1)          # interior meshes: no ghost of this frame has arrived
        for r, v in enumerate(parts):
            v.boundary_pack_bands(out[r][0].ptr, out[r][1].ptr, cap)
            v.sync()
        for r in range(3):
            if r > 0:
                _copy_d2d(inn[r - 1][1].ptr, out[r][0].ptr, bb)
            else:
                inn[r][0].from_host(zero_hdr)
            if r < 2:
                _copy_d2d(inn[r + 1][0].ptr, out[r][1].ptr, bb)
            else:
                inn[r][1].from_host(zero_hdr)
        for r, v in enumerate(parts):
            v.boundary_unpack_pair(inn[r][0].ptr, cap, inn[r][1].ptr, cap, join_dirty=True)
            v.texture_frame_device_phase(T, k, 2)          # boundary meshes + what the ghosts added
            v.sync()
    single.sync()
    ref_ids = sorted_ids(single.list_chunks())
    key = {tuple(c): i for i, c in enumerate(ref_ids)}
    s_ref, w_ref, c_ref = single.get_chunks(ref_ids)
    ref_m = sorted_ids(single.list_meshes())
    assert len(ref_m) > (150 if hall else 300)
    mvoff, mioff, mV, mN, mC, mI, madj, msimp = single.get_meshes(ref_m)
    mkey = {tuple(c): i for i, c in enumerate(ref_m)}
    seen_m, n_interior, n_boundary = set(), 0, 0
    for r, v in enumerate(parts):
        lo, hi = bounds[r], bounds[r + 1]
        ids = v.list_chunks()
        s, w, c = v.get_chunks(ids)
        for i, cid in enumerate(ids):
            t = tuple(int(x) for x in cid)
            if t in key:
                j = key[t]
                assert np.array_equal(s[i].view(np.uint32), s_ref[j].view(np.uint32)), (r, t)
                assert np.array_equal(c[i], c_ref[j])
        pm = sorted_ids(v.list_meshes())
        voff, ioff, V, N, Cc, I, adj, simp = v.get_meshes(pm)
        for i, cid in enumerate(pm):
            t = tuple(int(x) for x in cid)
            assert lo <= sum(t) < hi and t in mkey and t not in seen_m
            seen_m.add(t)
            j = mkey[t]
            if sum(t) - 3 >= lo and sum(t) + 3 < hi:
                n_interior += 1
            else:
                n_boundary += 1
            assert np.array_equal(V[voff[i]:voff[i + 1]].view(np.uint32), mV[mvoff[j]:mvoff[j + 1]].view(np.uint32)), (r, t)
            assert np.array_equal(N[voff[i]:voff[i + 1]].view(np.uint32), mN[mvoff[j]:mvoff[j + 1]].view(np.uint32)), (r, t)
            assert np.array_equal(I[ioff[i]:ioff[i + 1]], mI[mioff[j]:mioff[j + 1]]), (r, t)
            assert np.array_equal(adj[i], madj[j]), (r, t)
    assert seen_m == set(mkey)
    assert n_boundary > (3 if hall else 20) and n_interior > 100   # both passes had meshes to make
    for v in [single] + parts:
        v.close()
    for b in [x for pr in out + inn for x in pr] + [x for p in bufs for x in p]:
        b.free()


def test_phase_two_without_phase_one_is_refused(gpu_required):
    cam = synth.Camera()
    v = capi.Volume(RES5, cam, max_chunks=1 << 14)
    f = synth.room_frame(0, cam, with_quality=False)
    d, c = HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])
    v.stream_frames_device([d.ptr], [c.ptr], f[3].reshape(1, 12))
    T = synth.pose_inverse16(f[3])
    with pytest.raises(capi.TFError):
        v.texture_frame_device_phase(T, 0, 2)
    v.texture_frame_device_phase(T, 0, 1)
    with pytest.raises(capi.TFError):
        v.texture_frame_device(T, 0)            # half a stage pending
    v.texture_frame_device_phase(T, 0, 2)
    v.sync()
    assert len(v.list_meshes()) >= 0
    v.close(); d.free(); c.free()


@pytest.mark.parametrize("n_local", [0, 3])
def test_keyframe_unit_two_partitions_equal_one(gpu_required, n_local):
    """tf_keyframe_unit_device on two slabs of the chunk-range partition (TSDF-only calls, the band records exchanged by
    hand between the calls): the keyframe's own pass inside the group kernel (no quality image, local frames behind it),
    the lazy slot lookup, the finalize and the band flags it leaves for the pack -- the union of the slabs and every ghost
    copy must carry the single volume's voxels bit for bit, and the same chunks must be dirty."""
    cam = synth.Camera()
    axis, split = (1, 1, 1), 35
    single = capi.Volume(RES5, cam, max_chunks=1 << 16)
    parts = [capi.Volume(RES5, cam, max_chunks=1 << 16), capi.Volume(RES5, cam, max_chunks=1 << 16)]
    parts[0].set_partition(-(1 << 31), split, axis)
    parts[1].set_partition(split, (1 << 31) - 1, axis)
    cap = 4096
    xb = [HipBuffer(cap * capi.TF_BOUNDARY_RECORD_BYTES) for _ in range(2)]
    per = 1 + n_local
    frames = [synth.room_frame(2 * k, cam, with_quality=False) for k in range(3 * per)]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in frames]
    for g in range(3):
        k0 = g * per
        grp = capi.Volume.unit_group(40 + g, (bufs[k0][0].ptr, bufs[k0][1].ptr, None, frames[k0][3]),
                                     [(bufs[k][0].ptr, frames[k][3]) for k in range(k0 + 1, k0 + per)])
        for v in [single] + parts:
            v.keyframe_unit(fresh=grp, texture=False)
            v.sync()
        n = [parts[r].boundary_pack(xb[r].ptr, cap) for r in range(2)]
        assert n[0] > 0 and n[1] > 0
        parts[0].boundary_unpack(xb[1].ptr, n[1])
        parts[1].boundary_unpack(xb[0].ptr, n[0])
        for v in parts:
            v.sync()
    ref_ids = sorted_ids(single.list_chunks())
    assert len(ref_ids) > 1500
    s_ref, w_ref, c_ref = single.get_chunks(ref_ids)
    key = {tuple(c): i for i, c in enumerate(ref_ids)}
    seen = set()
    for r, v in enumerate(parts):
        ids = v.list_chunks()
        lo, hi = (-(1 << 31), split) if r == 0 else (split, (1 << 31) - 1)
        s, w, c = v.get_chunks(ids)
        for i, cid in enumerate(ids):
            t = tuple(int(x) for x in cid)
            if lo <= int(np.dot(cid.astype(np.int64), axis)) < hi:
                assert t in key, "partition %d holds a chunk the single volume does not" % r
                seen.add(t)
            if t in key:  # owned chunks and ghost copies both carry the reference state
                j = key[t]
                assert np.array_equal(s[i].view(np.uint32), s_ref[j].view(np.uint32)), (r, t)
                assert np.array_equal(w[i].view(np.uint32), w_ref[j].view(np.uint32)), (r, t)
                assert np.array_equal(c[i], c_ref[j]), (r, t)
    assert seen == set(key)
    dirty = set(map(tuple, single.dirty()))
    dparts = set(map(tuple, parts[0].dirty())) | set(map(tuple, parts[1].dirty()))
    assert dirty == dparts and len(dirty) > 1000
    for v in [single] + parts:
        v.close()
    for b in xb + [x for p in bufs for x in p]:
        b.free()
