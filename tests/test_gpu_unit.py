"""The keyframe unit (tf_keyframe_unit_device = MobileFusion::tsdfFusion, GCFusion/MobileFusion.cpp:274-406, as one
asynchronous call) against the oracle driven call by call in the reference's order: a first keyframe group; then a
second group together with the first one MOVED (retract, de-integrate over its stored validChunks at the old poses,
re-integrate at new ones); meshes, CompressMeshes, GeneratePatches with the new keyframe as label, UpdateAtlas.
Volume, observations, meshes, patches and atlas texels must be identical."""
import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import capi, synth
from tests.util import HipBuffer, assert_chunks_equal, sorted_ids

pytestmark = pytest.mark.gpu
RES5 = np.float32(0.005)


def _oracle_group(ov, kf_id, key, local, flag, ids=None):
    """ReIntegrateKeyframe (MobileFusion.cpp:114-221); key = (depth, rgba, quality, pose), local = [(depth, pose)]"""
    depth, rgba, quality, pose = key
    if flag == 1:
        ids, new = ov.prepare(depth, pose)
        needs = np.zeros(len(ids), np.uint8)
    else:
        new = np.zeros(len(ids), np.uint8)
        needs = np.ones(len(ids), np.uint8)
    ov.integrate(depth, rgba, quality, pose, ids, needs, flag, kf_id)
    for d, p in local:
        ov.integrate(d, None, None, p, ids, needs, flag, -1)
    return ov.finalize(ids, needs, new)


@pytest.mark.parametrize("with_q", [True, False])
def test_keyframe_unit_with_a_moved_keyframe(gpu_required, with_q):
    """with_q = False: keyframes without a quality image -- their own depth + colour pass then runs as the first frame of
    the group kernel's visit (k_integrate_group<., KEY>) instead of a launch of its own"""
    cam = synth.Camera()
    ov = O.Volume(RES5, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(RES5, cam, max_chunks=1 << 16)
    oa = O.Atlas(RES5)
    fr = [synth.room_frame(k, cam, with_quality=True) for k in range(12)]
    if not with_q:
        fr = [(f[0], f[1], None, f[3]) for f in fr]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1]),
             HipBuffer(f[2].nbytes).from_host(f[2]) if with_q else None) for f in fr]

    def dev_key(k, pose):
        return (bufs[k][0].ptr, bufs[k][1].ptr, bufs[k][2].ptr if with_q else None, pose)

    # ---- keyframe 5 = frame 0 with local frames 1..3; textured with itself
    A_loc = [1, 2, 3]
    gA = capi.Volume.unit_group(5, dev_key(0, fr[0][3]), [(bufs[k][0].ptr, fr[k][3]) for k in A_loc])
    gv.keyframe_unit(fresh=gA, texture=True, pose_inv16=synth.pose_inverse16(fr[0][3]))
    validA = _oracle_group(ov, 5, fr[0], [(fr[k][0], fr[k][3]) for k in A_loc], 1)
    ov.update_meshes()
    ids = ov.compress_meshes()
    kfs = {5: (np.ascontiguousarray(fr[0][1][..., :3]), fr[0][0], synth.pose_inverse16(fr[0][3]))}
    ov.generate_patches(oa, ids, np.full(len(ids), 5, np.int32), kfs)
    ov.update_atlas(oa, ids)
    # ---- keyframe 9 = frame 6 with local frames 7..10, and keyframe 5 MOVED: its frames get the poses of the frames
    # one step further along the orbit (a loop closure that shifted the whole group)
    B_loc = [7, 8, 9, 10]
    newA = [fr[k + 1][3] for k in [0] + A_loc]
    gB = capi.Volume.unit_group(9, dev_key(6, fr[6][3]), [(bufs[k][0].ptr, fr[k][3]) for k in B_loc])
    gA2 = capi.Volume.unit_group(5, dev_key(0, newA[0]), [(bufs[k][0].ptr, newA[1 + i]) for i, k in enumerate(A_loc)],
                                 old_keyframe_pose=fr[0][3], old_local_poses=[fr[k][3] for k in A_loc])
    gv.keyframe_unit(fresh=gB, moved=[gA2], texture=True, pose_inv16=synth.pose_inverse16(fr[6][3]))
    assert ov.retract_observations(5, validA) > 0 or not with_q  # (no quality image: no observation is ever recorded)
    _oracle_group(ov, 5, fr[0], [(fr[k][0], fr[k][3]) for k in A_loc], 0, ids=validA)
    movedA = (fr[0][0], fr[0][1], fr[0][2], newA[0])
    _oracle_group(ov, 5, movedA, [(fr[k][0], newA[1 + i]) for i, k in enumerate(A_loc)], 1)
    _oracle_group(ov, 9, fr[6], [(fr[k][0], fr[k][3]) for k in B_loc], 1)
    ov.update_meshes()
    ids = ov.compress_meshes()
    kfs[9] = (np.ascontiguousarray(fr[6][1][..., :3]), fr[6][0], synth.pose_inverse16(fr[6][3]))
    ov.generate_patches(oa, ids, np.full(len(ids), 9, np.int32), kfs)
    ov.update_atlas(oa, ids)
    gv.sync()
    # ---- volume, observations
    oids = sorted_ids(ov.list_chunks())
    assert np.array_equal(oids, sorted_ids(gv.list_chunks())) and len(oids) > 3000
    assert_chunks_equal(ov, gv, oids[::9], "keyframe unit")
    want = np.zeros((len(oids), 3), np.float32)
    for i, cid in enumerate(oids):
        obs = ov.observations(cid)
        want[i] = [obs.get(9, 0.0), obs.get(5, 0.0), obs.get(7, 0.0)]
    got = gv.export_datacost(oids, 9, [5, 7])
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    if with_q:
        assert (want[:, 0] > 0).sum() > 500 and (want[:, 1] > 0).sum() > 500
    # ---- meshes, patches, atlas
    mids = sorted_ids(ov.list_meshes())
    assert np.array_equal(mids, sorted_ids(gv.list_meshes())) and len(mids) > 300
    voff, ioff, V, N, Cc, I, adj, simp = gv.get_meshes(mids)
    g = gv.get_patches(mids)
    for i, cid in enumerate(mids):
        m = ov.get_mesh(cid)
        assert np.array_equal(V[voff[i]:voff[i + 1]].view(np.uint32), m["verts"].view(np.uint32)), cid
        assert np.array_equal(I[ioff[i]:ioff[i + 1]], m["indices"]), cid
        assert bool(simp[i]) == m["simplified"] and np.array_equal(adj[i], m["adj"]), cid
        o = ov.get_patch(cid)
        tl = o["texloc"] if o["flags"] & 1 else (1 << 64) - 1
        assert int(g["texloc"][i]) == tl and g["frameid"][i] == o["frameid"], cid
        if o["flags"] & 1:
            a, b = g["voff"][i], g["voff"][i + 1]
            assert np.array_equal(g["bbox"][i], o["bbox"]), cid
            assert np.array_equal(g["texcoord"][a:b].view(np.uint32), o["texcoord"].view(np.uint32)), cid
            assert np.array_equal(g["texcolor"][a:b].view(np.uint32), o["texcolor"].view(np.uint32)), cid
    assert gv.atlas_loc_next() == oa.loc_next()
    used = g["texloc"][g["texloc"] != np.uint64((1 << 64) - 1)]
    hot = oa.hot_range(used)
    r0, r1 = hot[0] // 13824, hot[1] // 13824
    assert r1 > r0 and np.array_equal(gv.atlas_rows(r0, r1, 13824), oa.buffer()[r0:r1])
    assert len(gv.dirty()) == 0 and len(ov.dirty()) == 0
    for t in bufs:
        for b in t:
            if b is not None:
                b.free()
    gv.close()


def test_keyframe_store_reuses_regions_and_compacts(gpu_required, monkeypatch):
    """The keyframes' validChunks store (tf_unit.hip): a chain of four keyframe groups in which every earlier keyframe is
    moved again each time, with exact-fit regions (TF_UNIT_NO_SLACK: any growth of a list needs a new region) in an arena
    of 6.5 first-keyframe lists for four keyframes whose lists grow to 1.1-1.6 of that (TF_UNIT_ARENA) -- lists are rewritten in place, regions die, the live ones are moved together.  A list
    that came back wrong would de-integrate the wrong chunks: the volume is compared with the oracle's, chunk by chunk."""
    cam = synth.Camera()
    fr = [synth.room_frame(2 * k, cam, with_quality=True) for k in range(14)]
    probe = O.Volume(RES5, O.camera_from(cam), O.default_integrator())
    list_len = len(_oracle_group(probe, 1, fr[0], [(fr[1][0], fr[1][3])], 1))
    monkeypatch.setenv("TF_UNIT_NO_SLACK", "1")
    monkeypatch.setenv("TF_UNIT_ARENA", str(int(6.5 * list_len)))
    ov = O.Volume(RES5, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(RES5, cam, max_chunks=1 << 16)
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1]), HipBuffer(f[2].nbytes).from_host(f[2]))
            for f in fr]
    groups = {1: (0, [1]), 2: (3, [4]), 3: (6, [7]), 4: (9, [10])}  # keyframe id -> (its frame, its local frames)
    pose_now = {}   # keyframe id -> [keyframe pose, local poses...] of its last integration
    valid = {}      # keyframe id -> the oracle's validChunks

    def dev_group(kf, poses, old=None):
        key, loc = groups[kf]
        return capi.Volume.unit_group(kf, (bufs[key][0].ptr, bufs[key][1].ptr, bufs[key][2].ptr, poses[0]),
                                      [(bufs[k][0].ptr, poses[1 + i]) for i, k in enumerate(loc)],
                                      **({} if old is None else dict(old_keyframe_pose=old[0], old_local_poses=old[1:])))

    def oracle_group(kf, poses, flag):
        key, loc = groups[kf]
        k = (fr[key][0], fr[key][1], fr[key][2], poses[0])
        return _oracle_group(ov, kf, k, [(fr[j][0], poses[1 + i]) for i, j in enumerate(loc)], flag,
                             ids=valid.get(kf) if flag == 0 else None)

    for step, kf in enumerate([1, 2, 3, 4]):
        key, loc = groups[kf]
        fresh_poses = [fr[key][3]] + [fr[j][3] for j in loc]
        moved = []
        for m in sorted(pose_now):  # every earlier keyframe moves: its frames get the poses `step` frames further along
            mk, ml = groups[m]
            new = [fr[min(j + step, len(fr) - 1)][3] for j in [mk] + ml]
            moved.append((m, pose_now[m], new))
        gv.keyframe_unit(fresh=dev_group(kf, fresh_poses), moved=[dev_group(m, new, old) for m, old, new in moved], texture=False)
        for m, old, new in moved:
            assert ov.retract_observations(m, valid[m]) > 0
            oracle_group(m, old, 0)
            valid[m] = oracle_group(m, new, 1)
            pose_now[m] = new
        valid[kf] = oracle_group(kf, fresh_poses, 1)
        pose_now[kf] = fresh_poses
    gv.sync()
    st = gv.keyframe_unit_stats()
    assert st["compactions"] >= 1 and st["reuses"] >= 1 and st["regions"] > 4 and st["top"] <= st["capacity"], st
    oids = sorted_ids(ov.list_chunks())
    assert np.array_equal(oids, sorted_ids(gv.list_chunks())) and len(oids) > 3000
    assert_chunks_equal(ov, gv, oids[::5], "keyframe store")
    for t in bufs:
        for b in t:
            b.free()
    gv.close()


def test_keyframe_store_grows_past_its_first_table_and_arena(gpu_required, monkeypatch):
    """No fixed ceiling on keyframes (the reference sizes its database for 20 000 frames, main.cpp:81): with a first table
    of 8 slots (TF_UNIT_SLOTS) and the default growth rule, 40 keyframes go in, the table doubles three times, and keyframes
    stored BEFORE the doublings are moved afterwards -- their validChunks must have survived the copies: the volume is
    compared with one that never grew (TF_UNIT_SLOTS = 64) and with the oracle for the moved groups' chunks."""
    cam = synth.Camera()
    n_kf = 40
    fr = [synth.room_frame(3 * k, cam, with_quality=False) for k in range(n_kf + 2)]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in fr]

    def run(slots):
        monkeypatch.setenv("TF_UNIT_SLOTS", str(slots))
        gv = capi.Volume(RES5, cam, max_chunks=1 << 17, max_list=1 << 16)   # (arena 16 x 65536 entries: never grows here)

        def grp(k, shift=0, old=False):
            kw = dict(old_keyframe_pose=fr[k][3], old_local_poses=[]) if old else {}
            return capi.Volume.unit_group(100 + k, (bufs[k][0].ptr, bufs[k][1].ptr, 0, fr[k + shift][3]), [], **kw)

        for k in range(n_kf):
            moved = []
            if k == 30:
                moved = [grp(2, 1, True), grp(7, 1, True)]      # stored while the table had 8 slots
            if k == 39:
                moved = [grp(20, 1, True)]                      # ... and while it had 32
            gv.keyframe_unit(fresh=grp(k), moved=moved, texture=False)
        gv.sync()
        return gv

    a = run(8)
    sa = a.keyframe_unit_stats_ex()
    assert sa["keyframes"] == n_kf and sa["slots"] == 64 and sa["doublings"] >= 3, sa
    b = run(64)
    sb = b.keyframe_unit_stats_ex()
    assert sb["slots"] == 64 and sb["doublings"] == 0, sb
    ia, ib = sorted_ids(a.list_chunks()), sorted_ids(b.list_chunks())
    assert np.array_equal(ia, ib) and len(ia) > 5000
    for lo in range(0, len(ia), 4096):
        s1, w1, c1 = a.get_chunks(ia[lo:lo + 4096:3])
        s2, w2, c2 = b.get_chunks(ia[lo:lo + 4096:3])
        assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32)) and np.array_equal(w1.view(np.uint32), w2.view(np.uint32))
        assert np.array_equal(c1, c2)
    # the moved keyframes against the oracle: after de-integrating frame 2 at its old pose and re-integrating it at frame
    # 3's, the oracle volume that did the same holds the same voxels in the chunks of that keyframe's new list
    ov = O.Volume(RES5, O.camera_from(cam), O.default_integrator())
    valid = {}
    for k in range(n_kf):
        if k == 30:
            for m in (2, 7):
                _oracle_group(ov, 100 + m, (fr[m][0], fr[m][1], None, fr[m][3]), [], 0, ids=valid[m])
                valid[m] = _oracle_group(ov, 100 + m, (fr[m][0], fr[m][1], None, fr[m + 1][3]), [], 1)
        if k == 39:
            _oracle_group(ov, 120, (fr[20][0], fr[20][1], None, fr[20][3]), [], 0, ids=valid[20])
            valid[20] = _oracle_group(ov, 120, (fr[20][0], fr[20][1], None, fr[21][3]), [], 1)
        valid[k] = _oracle_group(ov, 100 + k, (fr[k][0], fr[k][1], None, fr[k][3]), [], 1)
    oids = sorted_ids(ov.list_chunks())
    assert np.array_equal(oids, ia)
    assert_chunks_equal(ov, a, oids[::7], "grown keyframe store")
    a.close(); b.close()
    for p in bufs:
        p[0].free(); p[1].free()


def test_keyframe_arena_doubles_when_the_live_lists_fill_it(gpu_required, monkeypatch):
    """The arena side of the same: started at a size the first few lists fill (the initial size follows tf_config.max_list),
    it doubles instead of reporting TF_ERR_CAPACITY, and a keyframe stored before the doubling is moved after it."""
    cam = synth.Camera()
    n_kf = 24
    fr = [synth.room_frame(4 * k, cam, with_quality=False) for k in range(n_kf + 1)]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in fr]
    gv = capi.Volume(RES5, cam, max_chunks=1 << 17, max_list=1 << 14)   # arena = 16 x 16384 entries: ~26 room lists with slack
    ov = O.Volume(RES5, O.camera_from(cam), O.default_integrator())
    valid = {}
    cap0 = None
    for k in range(n_kf):
        moved = []
        if k == n_kf - 1:
            moved = [capi.Volume.unit_group(100 + 1, (bufs[1][0].ptr, bufs[1][1].ptr, 0, fr[2][3]), [], old_keyframe_pose=fr[1][3],
                                            old_local_poses=[])]
            _oracle_group(ov, 101, (fr[1][0], fr[1][1], None, fr[1][3]), [], 0, ids=valid[1])
            valid[1] = _oracle_group(ov, 101, (fr[1][0], fr[1][1], None, fr[2][3]), [], 1)
        gv.keyframe_unit(fresh=capi.Volume.unit_group(100 + k, (bufs[k][0].ptr, bufs[k][1].ptr, 0, fr[k][3]), []), moved=moved,
                         texture=False)
        valid[k] = _oracle_group(ov, 100 + k, (fr[k][0], fr[k][1], None, fr[k][3]), [], 1)
        if cap0 is None:
            cap0 = gv.keyframe_unit_stats_ex()["arena"]
    gv.sync()   # (an arena that ran full would raise TF_ERR_CAPACITY here)
    st, sx = gv.keyframe_unit_stats(), gv.keyframe_unit_stats_ex()
    assert sx["arena"] > cap0 and sx["doublings"] >= 1 and st["top"] <= st["capacity"] == sx["arena"], (st, sx, cap0)
    oids = sorted_ids(ov.list_chunks())
    assert np.array_equal(oids, sorted_ids(gv.list_chunks()))
    assert_chunks_equal(ov, gv, oids[::9], "doubled arena")
    gv.close()
    for p in bufs:
        p[0].free(); p[1].free()


@pytest.mark.parametrize("with_q", [True, False])
@pytest.mark.parametrize("n_local", [0, 1, 6])
def test_keyframe_groups_of_every_size(gpu_required, n_local, with_q):
    """a keyframe alone (no local frame), with one, and with the maximum of six (integrateLocalFrameNum): TSDF-only unit
    calls (texture = 0) against the oracle's ReIntegrateKeyframe, two keyframes in a row"""
    cam = synth.Camera()
    ov = O.Volume(RES5, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(RES5, cam, max_chunks=1 << 16)
    fr = [synth.room_frame(k, cam, with_quality=True) for k in range(2 * (1 + n_local))]
    if not with_q:
        fr = [(f[0], f[1], None, f[3]) for f in fr]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1]),
             HipBuffer(f[2].nbytes).from_host(f[2]) if with_q else None) for f in fr]
    for g in range(2):
        k0 = g * (1 + n_local)
        loc = list(range(k0 + 1, k0 + 1 + n_local))
        grp = capi.Volume.unit_group(20 + g, (bufs[k0][0].ptr, bufs[k0][1].ptr, bufs[k0][2].ptr if with_q else None, fr[k0][3]),
                                     [(bufs[k][0].ptr, fr[k][3]) for k in loc])
        gv.keyframe_unit(fresh=grp, texture=False)
        _oracle_group(ov, 20 + g, fr[k0], [(fr[k][0], fr[k][3]) for k in loc], 1)
    gv.sync()
    ids = sorted_ids(ov.list_chunks())
    assert np.array_equal(ids, sorted_ids(gv.list_chunks())) and len(ids) > 1000
    assert_chunks_equal(ov, gv, ids[::7], "group of 1 + %d frames" % n_local)
    assert np.array_equal(sorted_ids(ov.dirty()), sorted_ids(gv.dirty()))
    for b in bufs:
        for x in b:
            if x is not None:
                x.free()
    gv.close()


def run_unit_sequence(cam, res, frames, plan, with_q, max_chunks=1 << 17, stride=5, gpu_ahead=False, junk_then_reset=0):
    """A sequence of tf_keyframe_unit_device calls against the oracle driven call by call.  frames[k] = (depth, rgba,
    quality, pose); plan = [(kf_id, key_frame_index, [local frame indices], [(moved kf_id, pose shift)])]: every call
    integrates one new keyframe group, textured with its keyframe, and first moves the listed earlier keyframes -- their
    frames take the poses of the frames `shift` further along.  Compared at the end: chunk set, voxels (every stride-th
    chunk), observations, meshes, patches and the atlas fill.  Returns the number of meshes."""
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(res, cam, max_chunks=max_chunks)
    oa = O.Atlas(res)
    if not with_q:
        frames = [(f[0], f[1], None, f[3]) for f in frames]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1]),
             HipBuffer(f[2].nbytes).from_host(f[2]) if with_q else None) for f in frames]

    def dev_key(k, pose):
        return (bufs[k][0].ptr, bufs[k][1].ptr, bufs[k][2].ptr if with_q else None, pose)

    state = {}  # kf_id -> {key, loc, poses (current), valid (oracle's validChunks)}
    kfs = {}
    if junk_then_reset:
        # unit calls of another "session" first, then tf_volume_reset: the chain of calls (the unit's own stream, its events, the
        # ring of selection sets, the keyframe store) starts over -- what follows must equal a fresh volume's results
        for j in range(junk_then_reset):
            k = len(frames) - 1 - 7 * j
            grp = capi.Volume.unit_group(9000 + j, dev_key(k - 6, frames[k - 6][3]), [(bufs[k - 5 + i][0].ptr, frames[k - 5 + i][3]) for i in range(6)])
            gv.keyframe_unit(fresh=grp, moved=[], texture=True, pose_inv16=synth.pose_inverse16(frames[k - 6][3]))
        gv.reset()
    if gpu_ahead:
        # every call is enqueued before the oracle starts: the device works through the calls back to back, which is when a
        # call's front end (selection, records) overlaps the previous call's filter || patch stage and mesher (tf_unit.hip)
        ahead = {}  # kf_id -> (key, loc, current poses): what the moved groups of later calls need (no oracle result)
        for kf_id, key, loc, moves in plan:
            moved = []
            for mid, shift in moves:
                mk, ml, mp = ahead[mid]
                newp = [frames[min(k + shift, len(frames) - 1)][3] for k in [mk] + ml]
                moved.append(capi.Volume.unit_group(mid, dev_key(mk, newp[0]), [(bufs[k][0].ptr, newp[1 + i]) for i, k in enumerate(ml)],
                                                    old_keyframe_pose=mp[0], old_local_poses=mp[1:]))
                ahead[mid] = (mk, ml, newp)
            fresh = capi.Volume.unit_group(kf_id, dev_key(key, frames[key][3]), [(bufs[k][0].ptr, frames[k][3]) for k in loc])
            gv.keyframe_unit(fresh=fresh, moved=moved, texture=True, pose_inv16=synth.pose_inverse16(frames[key][3]))
            ahead[kf_id] = (key, list(loc), [frames[key][3]] + [frames[k][3] for k in loc])
    for kf_id, key, loc, moves in plan:
        moved = []
        for mid, shift in moves:
            st = state[mid]
            idx = [st["key"]] + st["loc"]
            newp = [frames[min(k + shift, len(frames) - 1)][3] for k in idx]
            moved.append(capi.Volume.unit_group(mid, dev_key(st["key"], newp[0]),
                                                [(bufs[k][0].ptr, newp[1 + i]) for i, k in enumerate(st["loc"])],
                                                old_keyframe_pose=st["poses"][0], old_local_poses=st["poses"][1:]))
            # oracle: retract, de-integrate at the old poses over the stored validChunks, integrate at the new ones
            ov.retract_observations(mid, st["valid"])
            fk = frames[st["key"]]
            _oracle_group(ov, mid, (fk[0], fk[1], fk[2], st["poses"][0]),
                          [(frames[k][0], st["poses"][1 + i]) for i, k in enumerate(st["loc"])], 0, ids=st["valid"])
            st["valid"] = _oracle_group(ov, mid, (fk[0], fk[1], fk[2], newp[0]),
                                        [(frames[k][0], newp[1 + i]) for i, k in enumerate(st["loc"])], 1)
            st["poses"] = newp
        fresh = capi.Volume.unit_group(kf_id, dev_key(key, frames[key][3]), [(bufs[k][0].ptr, frames[k][3]) for k in loc])
        T = synth.pose_inverse16(frames[key][3])
        if not gpu_ahead:
            gv.keyframe_unit(fresh=fresh, moved=moved, texture=True, pose_inv16=T)
        valid = _oracle_group(ov, kf_id, frames[key], [(frames[k][0], frames[k][3]) for k in loc], 1)
        state[kf_id] = dict(key=key, loc=list(loc), poses=[frames[key][3]] + [frames[k][3] for k in loc], valid=valid)
        ov.update_meshes()
        ids = ov.compress_meshes()
        kfs[kf_id] = (np.ascontiguousarray(frames[key][1][..., :3]), frames[key][0], T)
        ov.generate_patches(oa, ids, np.full(len(ids), kf_id, np.int32), kfs)
        ov.update_atlas(oa, ids)
    gv.sync()
    oids = sorted_ids(ov.list_chunks())
    assert np.array_equal(oids, sorted_ids(gv.list_chunks())) and len(oids) > 200
    assert_chunks_equal(ov, gv, oids[::stride], "unit sequence")
    all_kf = [p[0] for p in plan]
    want = np.zeros((len(oids), len(all_kf)), np.float32)
    for i, cid in enumerate(oids):
        obs = ov.observations(cid)
        want[i] = [obs.get(k, 0.0) for k in all_kf]
    got = gv.export_datacost(oids, all_kf[0], all_kf[1:])
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    mids = sorted_ids(ov.list_meshes())
    assert np.array_equal(mids, sorted_ids(gv.list_meshes()))
    voff, ioff, V, N, Cc, I, adj, simp = gv.get_meshes(mids)
    g = gv.get_patches(mids)
    for i, cid in enumerate(mids):
        m = ov.get_mesh(cid)
        assert np.array_equal(V[voff[i]:voff[i + 1]].view(np.uint32), m["verts"].view(np.uint32)), cid
        assert np.array_equal(I[ioff[i]:ioff[i + 1]], m["indices"]), cid
        assert bool(simp[i]) == m["simplified"] and np.array_equal(adj[i], m["adj"]), cid
        o = ov.get_patch(cid)
        tl = o["texloc"] if o["flags"] & 1 else (1 << 64) - 1
        assert int(g["texloc"][i]) == tl and g["frameid"][i] == o["frameid"], cid
        if o["flags"] & 1:
            a, b = g["voff"][i], g["voff"][i + 1]
            assert np.array_equal(g["texcoord"][a:b].view(np.uint32), o["texcoord"].view(np.uint32)), cid
            assert np.array_equal(g["texcolor"][a:b].view(np.uint32), o["texcolor"].view(np.uint32)), cid
    assert gv.atlas_loc_next() == oa.loc_next()
    assert len(gv.dirty()) == 0 and len(ov.dirty()) == 0
    for t in bufs:
        for b in t:
            if b is not None:
                b.free()
    gv.close()
    return len(mids)


def test_keyframe_unit_calls_back_to_back(gpu_required):
    """Ten keyframe groups (1 + 6 frames each) along the orbit, all ten calls enqueued before anything else happens: the
    device runs the calls back to back, every call's front end on its own stream beside the previous call's texture
    stage.  Chunks, voxels, observations, meshes, patches and the atlas fill must equal the oracle's, call by call order."""
    cam = synth.Camera()  # (640 x 480 at 6 mm: a call is ~150 us of device work, more than the host needs to enqueue the next)
    frames = [synth.room_frame(2 * k, cam, with_quality=False, wobble=0.05) for k in range(56)]
    plan = [(100 + g, 7 * g, [7 * g + 1 + i for i in range(6)], []) for g in range(8)]
    assert run_unit_sequence(cam, np.float32(0.006), frames, plan, False, max_chunks=1 << 18, stride=3, gpu_ahead=True) > 1000


def test_keyframe_unit_calls_after_a_reset(gpu_required):
    """three unit calls, tf_volume_reset, then the back-to-back sequence: as on a fresh volume"""
    cam = synth.Camera()
    frames = [synth.room_frame(2 * k, cam, with_quality=False, wobble=0.05) for k in range(36)]
    plan = [(100 + g, 7 * g, [7 * g + 1 + i for i in range(6)], []) for g in range(3)]
    assert run_unit_sequence(cam, np.float32(0.006), frames, plan, False, max_chunks=1 << 18, stride=3, gpu_ahead=True,
                             junk_then_reset=2) > 500


def test_keyframe_unit_calls_back_to_back_with_moved_keyframes(gpu_required):
    """... and with a moved keyframe in every other call (the keyframe integrated two calls earlier takes the poses of the frames
    one further along): the fresh group's selection then runs on the unit's stream beside the moved keyframe's de- and
    re-integration."""
    cam = synth.Camera()
    frames = [synth.room_frame(2 * k, cam, with_quality=False, wobble=0.05) for k in range(44)]
    plan = [(100 + g, 7 * g, [7 * g + 1 + i for i in range(6)], [(100 + g - 2, 1)] if g >= 2 and g % 2 == 0 else []) for g in range(6)]
    assert run_unit_sequence(cam, np.float32(0.006), frames, plan, False, max_chunks=1 << 18, stride=3, gpu_ahead=True) > 800


@pytest.mark.parametrize("with_q", [True, False])
def test_keyframe_unit_sequence_with_repeated_moves(gpu_required, with_q):
    """four keyframe groups of different sizes in a row (0, 2, 6 and 1 local frames); the first keyframe is moved twice, the
    second once, a keyframe without local frames is moved too"""
    cam = synth.Camera(320, 240, 262.5, 262.5, 159.5, 119.5, 0.01, 5.0)
    frames = [synth.room_frame(3 * k, cam, with_quality=True, wobble=0.05) for k in range(16)]
    plan = [(3, 0, [], []),
            (8, 1, [2, 3], [(3, 1)]),
            (11, 4, [5, 6, 7, 8, 9, 10], [(8, 2)]),
            (12, 11, [12], [(3, 2), (11, 1)])]
    assert run_unit_sequence(cam, np.float32(0.008), frames, plan, with_q) > 100


@pytest.mark.parametrize("with_q", [True, False])
def test_keyframe_unit_then_the_callers_view_selection(gpu_required, with_q):
    """The product's own order (MobileFusion::tsdfFusion, :274-406): the unit WITHOUT its texture stage -- integration of
    the keyframe groups and UpdateMeshes on the device, asynchronous --, then the caller's CompressMeshes, its view
    selection (host code: here a label per chunk that alternates between the keyframes seen so far), GeneratePatches with
    those labels and UpdateAtlas through the call-by-call entry points.  chunksToUpdate, meshes, patches and atlas rows must
    equal the oracle's after every keyframe."""
    cam = synth.Camera(320, 240, 262.5, 262.5, 159.5, 119.5, 0.01, 5.0)
    res = np.float32(0.008)
    frames = [synth.room_frame(k, cam, with_quality=True, wobble=0.02) for k in range(16)]
    if not with_q:
        frames = [(f[0], f[1], None, f[3]) for f in frames]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1]),
             HipBuffer(f[2].nbytes).from_host(f[2]) if with_q else None) for f in frames]
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(res, cam, max_chunks=1 << 16)
    oa = O.Atlas(res)
    kfs, seen = {}, []
    plan = [(4, 0, [1, 2, 3, 4]), (9, 5, [6, 7, 8, 9, 10]), (13, 11, [12, 13]), (14, 14, [15])]
    for kf_id, key, loc in plan:
        grp = capi.Volume.unit_group(kf_id, (bufs[key][0].ptr, bufs[key][1].ptr, bufs[key][2].ptr if with_q else None, frames[key][3]),
                                     [(bufs[k][0].ptr, frames[k][3]) for k in loc])
        gv.keyframe_unit(fresh=grp, texture=False)
        _oracle_group(ov, kf_id, frames[key], [(frames[k][0], frames[k][3]) for k in loc], 1)
        ov.update_meshes()
        ids = ov.compress_meshes()
        gids = gv.compress_meshes()
        assert np.array_equal(ids, gids)
        T = synth.pose_inverse16(frames[key][3])
        kfs[kf_id] = (np.ascontiguousarray(frames[key][1][..., :3]), frames[key][0], T)
        gv.keyframe_cache_device(kf_id, bufs[key][1].ptr, bufs[key][0].ptr, stride=4, pose_inv16=T)
        seen.append(kf_id)
        labels = np.array([seen[i % len(seen)] for i in range(len(ids))], np.int32)  # the caller's view selection
        ov.generate_patches(oa, ids, labels, kfs)
        ov.update_atlas(oa, ids)
        rc, hot = gv.generate_patches(ids, labels)
        assert rc == 0
        gv.update_atlas(ids)
    gv.sync()
    mids = sorted_ids(ov.list_meshes())
    assert np.array_equal(mids, sorted_ids(gv.list_meshes())) and len(mids) > 100
    voff, ioff, V, N, Cc, I, adj, simp = gv.get_meshes(mids)
    g = gv.get_patches(mids)
    for i, cid in enumerate(mids):
        m = ov.get_mesh(cid)
        assert np.array_equal(V[voff[i]:voff[i + 1]].view(np.uint32), m["verts"].view(np.uint32)), cid
        assert np.array_equal(I[ioff[i]:ioff[i + 1]], m["indices"]), cid
        o = ov.get_patch(cid)
        tl = o["texloc"] if o["flags"] & 1 else (1 << 64) - 1
        assert int(g["texloc"][i]) == tl and g["frameid"][i] == o["frameid"], cid
        if o["flags"] & 1:
            a, b = g["voff"][i], g["voff"][i + 1]
            assert np.array_equal(g["bbox"][i], o["bbox"]), cid
            assert np.array_equal(g["texcoord"][a:b].view(np.uint32), o["texcoord"].view(np.uint32)), cid
            assert np.array_equal(g["texcolor"][a:b].view(np.uint32), o["texcolor"].view(np.uint32)), cid
    assert gv.atlas_loc_next() == oa.loc_next()
    used = g["texloc"][g["texloc"] != np.uint64((1 << 64) - 1)]
    hot = oa.hot_range(used)
    r0, r1 = hot[0] // 13824, hot[1] // 13824
    assert r1 > r0 and np.array_equal(gv.atlas_rows(r0, r1, 13824), oa.buffer()[r0:r1])
    for t in bufs:
        for b in t:
            if b is not None:
                b.free()
    gv.close()


@pytest.mark.parametrize("with_q", [False, True])
def test_keyframe_pass_inside_the_group_kernel_on_hard_images(gpu_required, with_q):
    """with_q: the keyframes carry a quality image (k_integrate_group<., KEY, QUAL>: observationQualitySum kept in row order --
    random qualities make every sum order-sensitive -- and compared through the observations table).
    The keyframe's own pass as the first frame of the group kernel's visit (local frames behind it) on
    the images K-A's own tests use against integrate_body: NaN / -Inf / negative / denormal / near- and far-plane depth pixels, a wall
    35 cm from the camera (chunks near the image border and the near plane: the generic division path, stalled rows,
    off-image lanes -> the out-of-observation constant), and the same keyframe integrated often enough for the colour
    counts to be halved at 120 (ProjectionIntegrator.cpp:274-292); then everything de-integrated again (flag 0: the
    colour subtraction)."""
    cam = synth.Camera()
    res = np.float32(0.005)
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(res, cam, max_chunks=1 << 17)
    rng = np.random.default_rng(5)
    frames = []
    for k in (3, 4, 5, 6):
        depth, rgba, q, pose = synth.room_frame(k, cam, with_quality=False)
        depth = depth.copy()
        H, W = depth.shape
        # (no +Inf / 1e30 here: one such pixel makes the frame's bounding box overflow the candidate grid and the whole frame
        # is skipped, on both sides -- tests/test_gpu_parity.py has that case)
        bad = [np.nan, -np.inf, -1.5, 1e-40, 0.0, 5.0, 4.999, 0.01, 0.0101]
        ys = rng.integers(0, H, 4000); xs = rng.integers(0, W, 4000)
        depth[ys, xs] = np.float32(rng.choice(bad, 4000))
        depth[100:104, 200:260] = np.nan
        depth[300:303, 50:90] = -1.0
        frames.append((depth, rgba, None, pose))
    for s in (1, 2, 3):
        w = synth.wall_frame(0.35, cam, seed=s)
        frames.append((w[0], w[1], None, w[3]))
    if with_q:
        frames = [(f[0], f[1], rng.uniform(0.01, 1.0, f[0].shape).astype(np.float32), f[3]) for f in frames]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1]),
             HipBuffer(f[2].nbytes).from_host(f[2]) if with_q else None) for f in frames]

    def call(kf_id, key, loc, old=None):
        kw = {}
        if old is not None:
            kw = dict(old_keyframe_pose=frames[key][3], old_local_poses=[frames[k][3] for k in loc])
        return capi.Volume.unit_group(kf_id, (bufs[key][0].ptr, bufs[key][1].ptr, bufs[key][2].ptr if with_q else None, frames[key][3]),
                                      [(bufs[k][0].ptr, frames[k][3]) for k in loc], **kw)

    valid = {}
    plan = [(1, 0, [1]), (2, 2, [3, 1]), (3, 4, [5, 6])]
    for kf_id, key, loc in plan:
        gv.keyframe_unit(fresh=call(kf_id, key, loc), texture=False)
        valid[kf_id] = _oracle_group(ov, kf_id, frames[key], [(frames[k][0], frames[k][3]) for k in loc], 1)
    # the wall keyframe again and again: colour counts pass 120 and are quartered
    for rep in range(130):
        gv.keyframe_unit(fresh=call(100 + rep, 4, [5]), texture=False)
        _oracle_group(ov, 100 + rep, frames[4], [(frames[5][0], frames[5][3])], 1)
    gv.sync()
    ids = sorted_ids(ov.list_chunks())
    assert np.array_equal(ids, sorted_ids(gv.list_chunks())) and len(ids) > 1000
    assert_chunks_equal(ov, gv, ids[::3], "hard images, integrated")
    assert np.array_equal(sorted_ids(ov.dirty()), sorted_ids(gv.dirty()))
    kf_all = [1, 2, 3, 100, 164, 229]  # chunk->observations of these keyframes (Chisel.h:244-247), every chunk
    want = np.zeros((len(ids), len(kf_all)), np.float32)
    for i, cid in enumerate(ids):
        obs = ov.observations(cid)
        want[i] = [obs.get(k, 0.0) for k in kf_all]
    got = gv.export_datacost(ids, kf_all[0], kf_all[1:])
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    if with_q:
        assert (want != 0).sum() > 1000
    # the three keyframes moved onto themselves: de-integration (flag 0) and re-integration at the same poses
    for kf_id, key, loc in plan:
        gv.keyframe_unit(fresh=None, moved=[call(kf_id, key, loc, old=True)], texture=False)
        ov.retract_observations(kf_id, valid[kf_id])
        _oracle_group(ov, kf_id, frames[key], [(frames[k][0], frames[k][3]) for k in loc], 0, ids=valid[kf_id])
        valid[kf_id] = _oracle_group(ov, kf_id, frames[key], [(frames[k][0], frames[k][3]) for k in loc], 1)
    gv.sync()
    ids = sorted_ids(ov.list_chunks())
    assert np.array_equal(ids, sorted_ids(gv.list_chunks()))
    assert_chunks_equal(ov, gv, ids[::3], "hard images, moved")
    for b in bufs:
        for x in b:
            if x is not None:
                x.free()
    gv.close()
