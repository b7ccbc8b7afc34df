"""The atlas stage on device-resident meshes -- Chisel::GeneratePatches / UpdateAtlas / CompensateColor /
DrawMeshes (Structure/Chisel.cpp:149-355), Atlas::AddPatch / UpdateBuffer (Structure/Atlas.cpp:43-91),
Patch::CalculateTexCoords (Structure/Patch.cpp:40-108) -- against the oracle, on meshes both sides produced
by marching cubes from the same integrated frames.  Everything integer (slots, boxes, flags, texels) and the
per-vertex projections are compared bit for bit; colour compensation within the tolerance stated in
tests/test_color_compensate.py."""
import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import capi, synth
from tests.util import RES5, HipBuffer, make_pair, sorted_ids

pytestmark = pytest.mark.gpu
TOL = 2e-5  # colour compensation only (see tests/test_color_compensate.py)


def _integrate(ov, gv, frames):
    for depth, rgba, pose in frames:
        ov.integrate_frame(depth, rgba, pose)
        gv.frame_upload(depth, rgba, None)
        gv.integrate_frame(pose, True)
    ov.update_meshes()
    gv.update_meshes()
    oc, gc = ov.compress_meshes(), gv.compress_meshes()
    assert np.array_equal(oc, gc) and len(oc) > 0
    return oc


def _keyframe(frame):
    depth, rgba, pose = frame
    return np.ascontiguousarray(rgba[..., :3]), depth, synth.pose_inverse16(pose)


def _cache(gv, kfs):
    for k, (rgb, depth, T) in kfs.items():
        gv.keyframe_cache(k, rgb, depth, T)


def _compare_patches(ov, gv, ids, what="", labs=False):
    g = gv.get_patches(ids)
    for i, cid in enumerate(ids):
        o = ov.get_patch(cid)
        a, b = g["voff"][i], g["voff"][i + 1]
        tl = o["texloc"] if o["flags"] & 1 else (1 << 64) - 1
        assert int(g["texloc"][i]) == tl, "%s: texloc of %s" % (what, cid)
        assert g["frameid"][i] == o["frameid"], "%s: frameid of %s" % (what, cid)
        assert (int(g["flags"][i]) & 31) == (o["flags"] & 31), "%s: flags of %s: %d vs %d" % (what, cid, g["flags"][i], o["flags"])
        if not (o["flags"] & 1):
            continue
        assert np.array_equal(g["bbox"][i], o["bbox"]), "%s: bbox of %s" % (what, cid)
        assert np.array_equal(g["ratio"][i].view(np.uint32), o["ratio"].view(np.uint32)), "%s: ratio of %s" % (what, cid)
        assert b - a == len(o["texcoord"])
        assert np.array_equal(g["texcoord"][a:b].view(np.uint32), o["texcoord"].view(np.uint32)), "%s: texcoord of %s" % (what, cid)
        assert np.array_equal(g["texcolor"][a:b].view(np.uint32), o["texcolor"].view(np.uint32)), "%s: texcolor of %s" % (what, cid)
        if labs and (o["flags"] & 16) and (o["flags"] & 32):
            assert np.abs(g["labs"][a:b] - o["labs"]).max() <= TOL, "%s: labs of %s" % (what, cid)
    return g


def _compare_atlas(oa, gv, hot, width=13824):
    r0, r1 = hot[0] // width, hot[1] // width
    assert r1 > r0
    ga = gv.atlas_rows(r0, r1, width)
    assert np.array_equal(ga, oa.buffer()[r0:r1]), "atlas rows %d..%d differ" % (r0, r1)


def test_generate_patches_update_atlas(gpu_required):
    cam = synth.Camera()
    ov, gv, cam, ig = make_pair(RES5, cam, max_chunks=1 << 15)
    poses = [synth.pose_identity(), synth.pose_euler(0.12, -0.05, 0.02, (0.04, -0.02, 0.0))]
    frames = []
    for k in range(6):
        d, rgba, q, pose = synth.wall_frame(1.2, cam, pose=poses[k % 2], seed=k, rgba_value=(40 * k + 10, 100, 250 - 30 * k, 1))
        rgba = synth._hash_colour(np.stack(np.meshgrid(np.arange(cam.width) * 0.01, np.arange(cam.height) * 0.01), -1)[..., [0, 1, 1]] + k, 5)
        frames.append((d, rgba, pose))
    ids = _integrate(ov, gv, frames)
    oa = O.Atlas(RES5)
    kfs = {3: _keyframe(frames[4]), 8: _keyframe(frames[5])}
    _cache(gv, kfs)
    labels = np.where(np.arange(len(ids)) % 3 == 0, 8, 3).astype(np.int32)
    orc, ohot = ov.generate_patches(oa, ids, labels, kfs)
    grc, ghot = gv.generate_patches(ids, labels)
    assert orc == 0 and grc == 0 and ohot == ghot
    assert gv.atlas_loc_next() == oa.loc_next()
    _compare_patches(ov, gv, ids, "first GeneratePatches")
    ov.update_atlas(oa, ids)
    gv.update_atlas(ids)
    _compare_patches(ov, gv, ids, "after UpdateAtlas")  # ratio is written by UpdateBuffer
    _compare_atlas(oa, gv, ohot)
    # second keyframe round: slots are kept (Patch::clear), labels swap, a sub-list in another order
    sub = ids[::-1][: len(ids) // 2]
    labels2 = np.where(np.arange(len(sub)) % 2 == 0, 8, 3).astype(np.int32)
    orc, ohot2 = ov.generate_patches(oa, sub, labels2, kfs)
    grc, ghot2 = gv.generate_patches(sub, labels2)
    assert orc == 0 and grc == 0 and ohot2 == ghot2 and gv.atlas_loc_next() == oa.loc_next()
    ov.update_atlas(oa, sub)
    gv.update_atlas(sub)
    _compare_patches(ov, gv, ids, "second round")
    _compare_atlas(oa, gv, ohot)
    # chunks without a mesh in the list are skipped on both sides (Chisel.cpp:157)
    extra = np.concatenate([ids[:5], np.array([[900, 900, 900]], np.int32), ids[5:9]])
    lab3 = np.full(len(extra), 3, np.int32)
    assert ov.generate_patches(oa, extra, lab3, kfs)[1] == gv.generate_patches(extra, lab3)[1]
    gv.close()


def test_resize_branch_close_wall(gpu_required):
    """Wall at 0.55 m: a 4 cm chunk spans ~38 px, the ROI exceeds the 24 x 18 slot -> cv::resize branch."""
    cam = synth.Camera()
    ov, gv, cam, ig = make_pair(RES5, cam, max_chunks=1 << 14)
    frames = []
    for k in range(4):
        d, rgba, q, pose = synth.wall_frame(0.55, cam, seed=k)
        rgba = synth._hash_colour(np.stack(np.meshgrid(np.arange(cam.width) * 0.013, np.arange(cam.height) * 0.017), -1)[..., [0, 1, 1]], 7)
        frames.append((d, rgba, pose))
    ids = _integrate(ov, gv, frames)
    oa = O.Atlas(RES5)
    kfs = {0: _keyframe(frames[0])}
    _cache(gv, kfs)
    labels = np.zeros(len(ids), np.int32)
    _, ohot = ov.generate_patches(oa, ids, labels, kfs)
    _, ghot = gv.generate_patches(ids, labels)
    ov.update_atlas(oa, ids)
    gv.update_atlas(ids)
    g = _compare_patches(ov, gv, ids, "close wall")
    assert (g["ratio"] < 1).any(), "the scene was meant to exercise the resize branch"
    _compare_atlas(oa, gv, ohot)
    gv.close()


def test_resize_branch_constant_image_property(gpu_required):
    """The resize branch on the device, checked against a fact instead of the oracle: a constant keyframe image
    fills every resized slot with exactly that constant (all tap-coefficient pairs sum to 2048), and the ratios
    written back are slot / ROI."""
    cam = synth.Camera()
    ov, gv, cam, ig = make_pair(RES5, cam, max_chunks=1 << 14)
    frames = []
    for k in range(4):
        d, rgba, q, pose = synth.wall_frame(0.55, cam, seed=k, rgba_value=(91, 203, 17, 1))
        frames.append((d, rgba, pose))
    ids = _integrate(ov, gv, frames)
    kfs = {0: _keyframe(frames[0])}
    _cache(gv, kfs)
    labels = np.zeros(len(ids), np.int32)
    _, ghot = gv.generate_patches(ids, labels)
    gv.update_atlas(ids)
    P = gv.get_patches(ids)
    width, pw, ph = 13824, 24, 18
    r0, r1 = int(ghot[0]) // width, int(ghot[1]) // width + ph
    rows = gv.atlas_rows(r0, r1, width)
    n_resized = 0
    for i in range(len(ids)):
        if not (P["flags"][i] & 1) or P["bbox"][i][2] <= 0:
            continue
        cols, rws = int(P["bbox"][i][2]), int(P["bbox"][i][3])
        x, y = int(P["texloc"][i]) % width, int(P["texloc"][i]) // width - r0
        if cols > pw or rws > ph:
            n_resized += 1
            assert (rows[y:y + ph, x:x + pw] == np.array([91, 203, 17], np.uint8)).all()
            want = (np.float32(pw) / np.float32(cols) if cols > pw else np.float32(1),
                    np.float32(ph) / np.float32(rws) if rws > ph else np.float32(1))
            assert P["ratio"][i][0] == want[0] and P["ratio"][i][1] == want[1]
        else:
            assert (rows[y:y + rws, x:x + cols] == np.array([91, 203, 17], np.uint8)).all()
    assert n_resized > 50
    gv.close()


def test_atlas_overflow_is_minus_one(gpu_required):
    """A 96 x 36 atlas holds 2 bands x 4 slots: the ninth AddPatch throws (Atlas.cpp:52-53), GeneratePatches
    returns -1 (Chisel.cpp:170-173), the entries before it are processed."""
    cam = synth.Camera()
    ov = O.Volume(RES5, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(RES5, cam, max_chunks=1 << 14, atlas_w=96, atlas_h=36)
    frames = [(f[0], f[1], f[3]) for f in (synth.wall_frame(1.2, cam, seed=k) for k in range(5))]
    ids = _integrate(ov, gv, frames)[:20]
    oa = O.Atlas(RES5, 96, 36)
    kfs = {0: _keyframe(frames[0])}
    _cache(gv, kfs)
    labels = np.zeros(len(ids), np.int32)
    orc, _ = ov.generate_patches(oa, ids, labels, kfs)
    grc, _ = gv.generate_patches(ids, labels)
    assert orc == -1 and grc == capi.TF_ERR_ATLAS_FULL
    g = gv.get_patches(ids)
    for i, cid in enumerate(ids):
        o = ov.get_patch(cid)
        assert bool(g["flags"][i] & 1) == bool(o["flags"] & 1) == (i < 8)
        if i < 8:
            assert int(g["texloc"][i]) == o["texloc"]
    gv.close()


def test_compensate_color_and_draw_meshes(gpu_required):
    cam = synth.Camera()
    ov, gv, cam, ig = make_pair(RES5, cam, max_chunks=1 << 15)
    frames = []
    for k in range(6):
        d, rgba, q, pose = synth.wall_frame(1.2, cam, seed=k)
        rgba = synth._hash_colour(np.stack(np.meshgrid(np.arange(cam.width) * 0.01, np.arange(cam.height) * 0.01), -1)[..., [0, 1, 1]], 5)
        frames.append((d, rgba, pose))
    ids = _integrate(ov, gv, frames)
    oa = O.Atlas(RES5)
    dark = (frames[1][0], (frames[1][1].astype(np.float32) * 0.7).astype(np.uint8), frames[1][2])
    far = (np.where(frames[2][0] > 0, frames[2][0] + 1.0, 0).astype(np.float32), frames[2][1], frames[2][2])  # depth test fails
    kfs = {2: _keyframe(frames[0]), 5: _keyframe(dark), 7: _keyframe(far)}
    _cache(gv, kfs)
    labels = np.array([(2, 5, 7)[i % 3] for i in range(len(ids))], np.int32)
    ov.generate_patches(oa, ids, labels, kfs)
    gv.generate_patches(ids, labels)
    g = _compare_patches(ov, gv, ids, "three keyframes")
    assert (g["flags"] & 4).any() and not (g["flags"] & 4).all()  # wrong_mapping on the "far" keyframe only
    ov.update_atlas(oa, ids)
    gv.update_atlas(ids)
    assert ov.compensate_color() == gv.compensate_color() == 3
    _compare_patches(ov, gv, ids, "after CompensateColor", labs=True)
    # a second call finds every patch adjusted -- except the cluster of the "far" keyframe, whose patches all map
    # wrongly: nothing was learnt, has_adjusted stayed false (Chisel.cpp:242)
    assert ov.compensate_color() == gv.compensate_color() == 1
    # DrawMeshes: identical streams except the packed colour delta (column 5), which quantises labs - texcolor
    oV, oI = ov.draw_meshes(oa)
    gV, gI = gv.draw_meshes()
    assert oV.shape == gV.shape and np.array_equal(oI, gI) and len(oI) > 0
    cols = [c for c in range(12) if c != 5]
    assert np.array_equal(oV[:, cols].view(np.uint32), gV[:, cols].view(np.uint32))

    def unpack(a):  # the 27-bit code sits in a float32: its low 3 bits (third channel) are rounded away
        a = a.astype(np.int64)
        return np.stack([(a >> 18) & 511, (a >> 9) & 511], -1)
    assert np.abs(unpack(oV[:, 5]) - unpack(gV[:, 5])).max() <= 1
    # ... and exact given the device's own labs (the oracle's packer on the downloaded patch arrays)
    voff, ioff, V, N, Cc, I, adj, simp = gv.get_meshes(ids)
    p = gv.get_patches(ids)
    complete = ((np.diff(voff) > 0) & (simp > 0) & ((p["flags"] & 8) > 0) & (p["frameid"] >= 0)).astype(np.uint8)
    wrong = ((p["flags"] & 4) > 0).astype(np.uint8)
    labs_valid = (((p["flags"] & 16) > 0) & (wrong == 0)).astype(np.uint8)
    rV, rI = O.pack_vertices(complete, wrong, labs_valid, p["texloc"], p["ratio"], 13824, 13824, voff, V, Cc, N,
                             p["texcoord"], p["texcolor"], p["labs"], ioff, I)
    assert np.array_equal(rV.view(np.uint32), gV.view(np.uint32)) and np.array_equal(rI, gI)
    gv.close()


def test_meshes_upload_feeds_the_atlas_stage(gpu_required):
    """A host that keeps its own mesher: tf_meshes_upload puts its meshes into allMeshes, the atlas stage runs
    on them like on device-built ones."""
    cam = synth.Camera()
    ov, gv, cam, ig = make_pair(RES5, cam, max_chunks=1 << 14)
    frames = [(f[0], f[1], f[3]) for f in (synth.wall_frame(1.2, cam, seed=k) for k in range(5))]
    for depth, rgba, pose in frames:
        ov.integrate_frame(depth, rgba, pose)
    ov.update_meshes()
    ids = ov.compress_meshes()
    ms = [ov.get_mesh(c) for c in ids]
    voff = np.concatenate([[0], np.cumsum([len(m["verts"]) for m in ms])]).astype(np.int64)
    ioff = np.concatenate([[0], np.cumsum([len(m["indices"]) for m in ms])]).astype(np.int64)
    gv.meshes_upload(ids, voff, ioff, np.concatenate([m["verts"] for m in ms]), np.concatenate([m["normals"] for m in ms]),
                     np.concatenate([m["colors"] for m in ms]), np.concatenate([m["indices"] for m in ms]))
    assert np.array_equal(sorted_ids(gv.list_meshes()), sorted_ids(ids))
    v2, i2, V, N, Cc, I, adj, simp = gv.get_meshes(ids)
    assert np.array_equal(v2, voff) and np.array_equal(i2, ioff)
    assert np.array_equal(V.view(np.uint32), np.concatenate([m["verts"] for m in ms]).view(np.uint32))
    assert np.array_equal(I, np.concatenate([m["indices"] for m in ms]))
    oa = O.Atlas(RES5)
    kfs = {1: _keyframe(frames[0])}
    _cache(gv, kfs)
    labels = np.ones(len(ids), np.int32)
    ov.generate_patches(oa, ids, labels, kfs)
    gv.generate_patches(ids, labels)
    _compare_patches(ov, gv, ids, "uploaded meshes")
    gv.close()


def test_fused_textured_stream(gpu_required):
    """tf_stream_frames_textured_device (integrate -> UpdateMeshes -> CompressMeshes -> GeneratePatches with label
    = the frame -> UpdateAtlas per frame, asynchronous, frames resident in HBM) == the oracle's per-frame unit:
    volume, meshes, patches, slot order (ascending chunk id among a frame's new patches), atlas texels."""
    cam = synth.Camera()
    ov, gv, cam, ig = make_pair(RES5, cam, max_chunks=1 << 16)
    oa = O.Atlas(RES5)
    n = 9
    fr = [synth.room_frame(2 * k, cam, with_quality=False) for k in range(n)]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in fr]
    poses = np.stack([f[3].reshape(12) for f in fr])
    pinv = np.stack([synth.pose_inverse16(f[3]) for f in fr])
    for k, f in enumerate(fr):
        ov.frame_textured(oa, f[0], f[1], f[3], pinv[k], 100 + k)
    # three calls: 4 frames (2 selected ahead), 3 frames starting primed (1 ahead), 2 frames
    dd, dr = [b[0].ptr for b in bufs], [b[1].ptr for b in bufs]
    gv.stream_frames_textured_device(dd[0:6], dr[0:6], poses[0:6], pinv[0:6], 100, n_ahead=2)
    gv.stream_frames_textured_device(dd[4:8], dr[4:8], poses[4:8], pinv[4:8], 104, n_ahead=1)
    gv.stream_frames_textured_device(dd[7:9], dr[7:9], poses[7:9], pinv[7:9], 107, n_ahead=0)
    gv.sync()
    from tests.util import assert_chunks_equal
    oids = sorted_ids(ov.list_chunks())
    assert np.array_equal(oids, sorted_ids(gv.list_chunks()))
    assert_chunks_equal(ov, gv, oids[::7], "fused stream")
    mids = sorted_ids(ov.list_meshes())
    assert np.array_equal(mids, sorted_ids(gv.list_meshes())) and len(mids) > 500
    voff, ioff, V, N, Cc, I, adj, simp = gv.get_meshes(mids)
    for i, cid in enumerate(mids):
        m = ov.get_mesh(cid)
        assert np.array_equal(V[voff[i]:voff[i + 1]].view(np.uint32), m["verts"].view(np.uint32)), cid
        assert np.array_equal(I[ioff[i]:ioff[i + 1]], m["indices"]), cid
        assert bool(simp[i]) == m["simplified"] and np.array_equal(adj[i], m["adj"]), cid
    assert gv.atlas_loc_next() == oa.loc_next()
    g = _compare_patches(ov, gv, mids, "fused stream")
    used = g["texloc"][g["texloc"] != np.uint64((1 << 64) - 1)]
    hot = oa.hot_range(used)
    _compare_atlas(oa, gv, hot)
    st = gv.texture_stats()
    assert st.n_slots == len(used) and st.n_dirty > 0 and st.n_meshes > 0
    assert len(gv.dirty()) == 0  # CompressMeshes cleared meshesToUpdate after every frame
    for a, b in bufs:
        a.free(); b.free()
    gv.close()


def test_deferred_patch_stage_cannot_be_observed(gpu_required):
    """The patch stage of a textured frame rides on the NEXT frame's launch (AtlasState::pend_patch).  Whatever looks
    at patches / atlas / meshes in between sees it done: a download right behind a streaming call that left the stage
    pending (n_ahead > 0), then host frames (the stage crosses from one entry point to the other), a TSDF-only frame
    (the stage goes out on its own), and a final comparison -- each against the oracle's state at that point."""
    cam = synth.Camera()
    ov, gv, cam, ig = make_pair(RES5, cam, max_chunks=1 << 16)
    oa = O.Atlas(RES5)
    n = 11
    fr = [synth.room_frame(2 * k, cam, with_quality=False) for k in range(n)]
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in fr]
    poses = np.stack([f[3].reshape(12) for f in fr])
    pinv = np.stack([synth.pose_inverse16(f[3]) for f in fr])
    dd, dr = [b[0].ptr for b in bufs], [b[1].ptr for b in bufs]

    def check(tag):
        mids = sorted_ids(ov.list_meshes())
        assert np.array_equal(mids, sorted_ids(gv.list_meshes())), tag
        assert len(mids) > 300, tag
        g = _compare_patches(ov, gv, mids, tag)
        used = g["texloc"][g["texloc"] != np.uint64((1 << 64) - 1)]
        assert len(used) > 300, tag
        _compare_atlas(oa, gv, oa.hot_range(used))
        assert gv.atlas_loc_next() == oa.loc_next(), tag

    for k in range(6):
        ov.frame_textured(oa, fr[k][0], fr[k][1], fr[k][3], pinv[k], 10 + k)
    gv.stream_frames_textured_device(dd[0:8], dr[0:8], poses[0:8], pinv[0:8], 10, n_ahead=2)  # frame 5's stage stays pending
    check("behind a streaming call")
    for k in range(6, 9):  # host frames: deferred two calls, the pending stage crosses the entry points
        ov.frame_textured(oa, fr[k][0], fr[k][1], fr[k][3], pinv[k], 10 + k)
        gv.integrate_frame_host(fr[k][0], fr[k][1], fr[k][3], pinv[k], 10 + k)
    check("behind host frames")
    ov.integrate_frame(fr[9][0], fr[9][1], fr[9][3])  # a TSDF-only frame between textured ones
    gv.integrate_frame_host(fr[9][0], fr[9][1], fr[9][3], None, 0)
    ov.frame_textured(oa, fr[10][0], fr[10][1], fr[10][3], pinv[10], 20)
    gv.integrate_frame_host(fr[10][0], fr[10][1], fr[10][3], pinv[10], 20)
    gv.sync()
    from tests.util import assert_chunks_equal
    oids = sorted_ids(ov.list_chunks())
    assert np.array_equal(oids, sorted_ids(gv.list_chunks()))
    assert_chunks_equal(ov, gv, oids[::11], "deferred patch stage")
    check("at the end")
    for a, b in bufs:
        a.free(); b.free()
    gv.close()


def test_wrongly_sized_host_images_are_rejected(gpu_required):
    cam = synth.Camera()
    gv = capi.Volume(RES5, cam, max_chunks=1 << 12)
    d, rgba, q, pose = synth.wall_frame(1.2, cam, seed=0)
    with pytest.raises(capi.TFError) as e:
        gv.integrate_frame_host(d[:100], rgba, pose, None, 0)
    assert e.value.code == capi.TF_ERR_INVALID
    with pytest.raises(capi.TFError):
        gv.integrate_frame_host(d, rgba[:, :320], pose, None, 0)
    with pytest.raises(capi.TFError):
        gv.frame_upload(d, rgba[:10])
    gv.close()


def test_fused_stream_atlas_overflow(gpu_required):
    """The fused per-frame unit with an atlas of 2 bands x 4 slots: from the frame whose new patches no longer fit,
    the entries behind the first failing AddPatch (ascending chunk id) are skipped in every frame, exactly as the
    oracle's GeneratePatches returns -1 there (Chisel.cpp:170-173); the error surfaces at the next synchronisation
    and everything up to it is identical."""
    cam = synth.Camera()
    ov = O.Volume(RES5, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(RES5, cam, max_chunks=1 << 14, atlas_w=96, atlas_h=36)
    oa = O.Atlas(RES5, 96, 36)
    n = 8
    fr = []
    for k in range(n):
        d, rgba, q, pose = synth.wall_frame(1.2, cam, seed=k)
        rgba = synth._hash_colour(np.stack(np.meshgrid(np.arange(cam.width) * 0.01, np.arange(cam.height) * 0.01), -1)[..., [0, 1, 1]] + k, 5)
        fr.append((d, rgba, pose))
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in fr]
    poses = np.stack([f[2].reshape(12) for f in fr])
    pinv = np.stack([synth.pose_inverse16(f[2]) for f in fr])
    for k, f in enumerate(fr):
        ov.frame_textured(oa, f[0], f[1], f[2], pinv[k], 50 + k)
    gv.stream_frames_textured_device([b[0].ptr for b in bufs], [b[1].ptr for b in bufs], poses, pinv, 50)
    with pytest.raises(capi.TFError) as e:
        gv.sync()
    assert e.value.code == capi.TF_ERR_ATLAS_FULL
    mids = sorted_ids(ov.list_meshes())
    assert np.array_equal(mids, sorted_ids(gv.list_meshes())) and len(mids) > 100
    g = _compare_patches(ov, gv, mids, "fused overflow")
    have = g["texloc"] != np.uint64((1 << 64) - 1)
    assert have.sum() == 8, "2 bands x 4 slots"
    assert gv.atlas_loc_next() == oa.loc_next()
    assert np.array_equal(gv.atlas_rows(0, 36, 96), oa.buffer()[0:36])
    for a, b in bufs:
        a.free(); b.free()
    gv.close()
