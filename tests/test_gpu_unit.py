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


def test_keyframe_unit_with_a_moved_keyframe(gpu_required):
    cam = synth.Camera()
    ov = O.Volume(RES5, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(RES5, cam, max_chunks=1 << 16)
    oa = O.Atlas(RES5)
    fr = [synth.room_frame(k, cam, with_quality=True) for k in range(12)]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1]), HipBuffer(f[2].nbytes).from_host(f[2]))
            for f in fr]

    def dev_key(k, pose):
        return (bufs[k][0].ptr, bufs[k][1].ptr, bufs[k][2].ptr, pose)

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
    assert ov.retract_observations(5, validA) > 0
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
