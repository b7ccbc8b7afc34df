"""HIP path vs CPU oracle, through the C ABI, on identical seeded inputs.

Bar (north_star): chunk / voxel indices, list order, flags and colour accumulators bit-exact; TSDF
values within a stated float tolerance -- the tolerance used here is ZERO (bit-exact): the kernels
round every operation exactly as the oracle does.
"""
import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import capi, synth
from tests.util import RES5, RES10, HipBuffer, assert_chunks_equal, make_pair, sorted_ids

pytestmark = pytest.mark.gpu


def _frame_flow(ov, gv, depth, rgba, quality, pose, kf_id=-1, use_quality=False):
    """prepare -> integrate -> finalize on both sides, compared after every call."""
    oids, onew = ov.prepare(depth, pose)
    gv.frame_upload(depth, rgba, quality)
    gids, gnew = gv.prepare(pose)
    assert np.array_equal(oids, gids), "visible-chunk list (ids or order) differs"
    assert np.array_equal(onew, gnew), "newChunkFlag differs"
    oneeds = np.zeros(len(oids), np.uint8)
    gneeds = np.zeros(len(gids), np.uint8)
    oq = ov.integrate(depth, rgba, quality if use_quality else None, pose, oids, oneeds, 1, kf_id)
    gq = gv.integrate(pose, gids, gneeds, 1, rgba is not None, use_quality)
    assert np.array_equal(oneeds, gneeds), "needsUpdateFlag differs"
    if rgba is not None:
        assert np.array_equal(oq.view(np.uint32), gq.view(np.uint32)), "chunkObservationQuality differs"
    assert_chunks_equal(ov, gv, oids, "after integrate")
    ovalid = ov.finalize(oids, oneeds, onew)
    gvalid = gv.finalize(gids, gneeds, gnew)
    assert np.array_equal(ovalid, gvalid), "validChunks differs"
    return oids, oneeds, onew


def test_wall_known_answer_chunk(gpu_required):
    """SURVEY.md A.1-9 probe: chunk (0,0,24), wall at 1 m -- w = 17.676939, sdf = 0.0024999836, q = 48."""
    ov, gv, cam, ig = make_pair(max_chunks=1 << 12)
    depth, rgba, quality, pose = synth.wall_frame(1.0, cam, hole_stride=0)
    ids = np.array([[0, 0, 24], [0, -12, 24]], np.int32)
    for cid in ids:
        ov.set_chunk(cid, *O.fresh_chunk())
        gv.set_chunk(cid, *O.fresh_chunk())
    gv.frame_upload(depth, rgba, quality)
    oneeds = np.zeros(2, np.uint8)
    gneeds = np.zeros(2, np.uint8)
    oq = ov.integrate(depth, rgba, quality, pose, ids, oneeds, 1, 3)
    gq = gv.integrate(pose, ids, gneeds, 1, True, True)
    assert list(gneeds) == [1, 0] and list(oneeds) == [1, 0]
    assert gq[0] == np.float32(48.0) and oq[0] == np.float32(48.0)
    s, w, c = gv.get_chunk(ids[0])
    assert w.max() == np.float32(17.676939)
    assert np.float32(0.0024999836) in s
    assert_chunks_equal(ov, gv, ids)


@pytest.mark.parametrize("res", [RES5, RES10])
def test_wall_frames_full_flow(gpu_required, res):
    ov, gv, cam, ig = make_pair(res, max_chunks=1 << 15)
    depth, rgba, quality, pose = synth.wall_frame(1.5, cam)
    for it in range(3):
        _frame_flow(ov, gv, depth, rgba, quality, pose, kf_id=it, use_quality=True)
    assert ov.num_chunks() == gv.stats().n_chunks
    assert np.array_equal(sorted_ids(ov.list_chunks()), sorted_ids(gv.list_chunks()))
    assert np.array_equal(sorted_ids(ov.dirty()), sorted_ids(gv.dirty()))


def test_room_stream_full_flow(gpu_required):
    ov, gv, cam, ig = make_pair(max_chunks=1 << 16)
    for k in (0, 1, 2, 50):
        depth, rgba, quality, pose = synth.room_frame(k, cam)
        _frame_flow(ov, gv, depth, rgba, quality, pose, kf_id=k, use_quality=True)
    assert np.array_equal(sorted_ids(ov.list_chunks()), sorted_ids(gv.list_chunks()))
    assert np.array_equal(sorted_ids(ov.dirty()), sorted_ids(gv.dirty()))
    st = gv.stats()
    ost = ov.rowstats()
    assert st.n_chunks == ov.num_chunks()


def test_fused_frame_matches_call_by_call(gpu_required):
    """tf_integrate_frame (5-arg unit, asynchronous, device-resident list) == oracle's 5-arg unit."""
    ov, gv, cam, ig = make_pair(max_chunks=1 << 16)
    for k in (0, 3, 6):
        depth, rgba, quality, pose = synth.room_frame(k, cam)
        ov.rowstats(clear=True)
        nv, ns = ov.integrate_frame(depth, rgba, pose)
        gv.frame_upload(depth, rgba, None)
        gv.integrate_frame(pose, True)
        gv.sync()
        st = gv.stats()
        ost = ov.rowstats()
        assert st.n_selected == ns
        assert st.n_updated == nv
        assert st.rows_tsdf == ost.rows_tsdf and st.rows_color == ost.rows_color
        assert st.n_chunks == ov.num_chunks()
    ids = ov.list_chunks()
    assert np.array_equal(sorted_ids(ids), sorted_ids(gv.list_chunks()))
    assert_chunks_equal(ov, gv, ids, "fused")
    assert np.array_equal(sorted_ids(ov.dirty()), sorted_ids(gv.dirty()))


def test_depth_only_and_deintegrate(gpu_required):
    """Keyframe group: colour frame + depth-only local frames on one list, then de-integration (flag 0)."""
    ov, gv, cam, ig = make_pair(max_chunks=1 << 16)
    depth, rgba, quality, pose = synth.room_frame(10, cam)
    oids, onew = ov.prepare(depth, pose)
    gv.frame_upload(depth, rgba, quality)
    gids, gnew = gv.prepare(pose)
    assert np.array_equal(oids, gids)
    on = np.zeros(len(oids), np.uint8)
    gn = np.zeros(len(oids), np.uint8)
    ov.integrate(depth, rgba, quality, pose, oids, on, 1, 10)
    gv.integrate(pose, gids, gn, 1, True, True)
    for k in (11, 12):  # local frames: depth only, their own pose, same list (MobileFusion.cpp:187-203)
        d2, _, _, p2 = synth.room_frame(k, cam)
        ov.integrate(d2, None, None, p2, oids, on, 1, -1)
        gv.frame_upload(d2, None, None)
        gv.integrate(p2, gids, gn, 1, False, False)
        assert np.array_equal(on, gn)
    assert_chunks_equal(ov, gv, oids, "keyframe group")
    ovalid = ov.finalize(oids, on, onew)
    gvalid = gv.finalize(gids, gn, gnew)
    assert np.array_equal(ovalid, gvalid)
    # de-integrate the colour frame over its validChunks (MobileFusion.cpp:135-143)
    on2 = np.ones(len(ovalid), np.uint8)
    gn2 = np.ones(len(ovalid), np.uint8)
    oq = ov.integrate(depth, rgba, quality, pose, ovalid, on2, 0, 10)
    gv.frame_upload(depth, rgba, quality)
    gq = gv.integrate(pose, gvalid, gn2, 0, True, True)
    assert np.array_equal(oq.view(np.uint32), gq.view(np.uint32))
    assert_chunks_equal(ov, gv, ovalid, "de-integrated")


def test_flags_changed_by_the_caller_between_calls(gpu_required):
    """The library remembers which needsUpdate / isNew flags the device list holds and skips the upload when the caller hands
    the same ones back; a caller that clears, sets or flips flags between calls must see exactly the reference's
    `needsUpdateFlag[i] |= updated` on WHAT IT PASSED, and finalize must act on the flags it is given."""
    ov, gv, cam, ig = make_pair(max_chunks=1 << 16)
    depth, rgba, quality, pose = synth.room_frame(20, cam)
    oids, onew = ov.prepare(depth, pose)
    gv.frame_upload(depth, rgba, quality)
    gids, gnew = gv.prepare(pose)
    assert np.array_equal(oids, gids) and np.array_equal(onew, gnew)
    n = len(oids)
    on, gn = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    ov.integrate(depth, rgba, quality, pose, oids, on, 1, 20)
    gv.integrate(pose, gids, gn, 1, True, True)
    assert np.array_equal(on, gn) and on.any()
    rng = np.random.default_rng(7)
    for k, mode in ((21, "same"), (22, "cleared"), (23, "random"), (24, "same"), (25, "ones")):
        d2, _, _, p2 = synth.room_frame(k, cam)
        if mode == "cleared":
            on[:] = 0; gn[:] = 0
        elif mode == "random":
            r = rng.integers(0, 2, n).astype(np.uint8)
            on[:] = r; gn[:] = r
        elif mode == "ones":
            on[:] = 1; gn[:] = 1
        ov.integrate(d2, None, None, p2, oids, on, 1, -1)
        gv.frame_upload(d2, None, None)
        gv.integrate(p2, gids, gn, 1, False, False)
        assert np.array_equal(on, gn), mode
    assert_chunks_equal(ov, gv, oids, "flags changed between calls")
    # finalize on flags the device has never seen: every second updated chunk dropped, isNew flipped on a few
    on[::2] = 0; gn[::2] = 0
    onew2, gnew2 = onew.copy(), gnew.copy()
    onew2[1::5] ^= 1; gnew2[1::5] ^= 1
    ovalid = ov.finalize(oids, on, onew2)
    gvalid = gv.finalize(gids, gn, gnew2)
    assert np.array_equal(ovalid, gvalid)
    assert np.array_equal(sorted_ids(ov.list_chunks()), sorted_ids(gv.list_chunks()))
    assert np.array_equal(sorted_ids(ov.dirty()), sorted_ids(gv.dirty()))
    # chunks removed WITH data (new, updated, but finalized as not updated: RemoveChunk) come back fresh when a later frame
    # selects them again
    removed = oids[(on == 0) & (onew2 == 1)]
    assert len(removed) > 100 and not any(ov.has_chunk(c) for c in removed[:20])
    for k in (26, 27):
        d3, c3, q3, p3 = synth.room_frame(k, cam)
        ids3, _, _ = _frame_flow(ov, gv, d3, c3, q3, p3, kf_id=k, use_quality=True)
    assert [bool(gv.has_chunk(c)) for c in ids3] == [bool(ov.has_chunk(c)) for c in ids3]
    assert_chunks_equal(ov, gv, np.array([c for c in ids3 if ov.has_chunk(c)], np.int32), "after removal with data")
    back = [c for c in removed if ov.has_chunk(c)]
    assert len(back) > 20
    assert_chunks_equal(ov, gv, np.array(back, np.int32), "removed chunks that came back")
    gv.close()


def test_colour_saturation_cap(gpu_required):
    """Colour count never exceeds 120: 121 -> all four channels >> 2 (ProjectionIntegrator.cpp:281-287)."""
    ov, gv, cam, ig = make_pair(max_chunks=1 << 12)
    depth, rgba, quality, pose = synth.wall_frame(1.0, cam, hole_stride=0, rgba_value=(255, 128, 7, 1))
    cid = np.array([[0, 0, 24]], np.int32)
    sdf, w, col = O.fresh_chunk()
    col[:] = np.tile(np.array([119 * 255, 119 * 128, 119 * 7, 119], np.uint16), 512)
    ov.set_chunk(cid[0], sdf, w, col)
    gv.set_chunk(cid[0], sdf, w, col)
    gv.frame_upload(depth, rgba, None)
    for it in range(3):
        on = np.zeros(1, np.uint8)
        gn = np.zeros(1, np.uint8)
        ov.integrate(depth, rgba, None, pose, cid, on, 1, -1)
        gv.integrate(pose, cid, gn, 1, True, False)
        assert_chunks_equal(ov, gv, cid, "saturation pass %d" % it)
    _, _, c = gv.get_chunk(cid[0])
    assert c.reshape(-1, 4)[:, 3].max() <= 120


def test_rotated_poses_and_borders(gpu_required):
    """Tilted cameras exercise off-image rows (the pos-stall quirk) and the out-of-observation quality."""
    ov, gv, cam, ig = make_pair(max_chunks=1 << 16)
    for j, (yaw, pitch, roll) in enumerate([(0.35, 0.2, 0.1), (-0.6, -0.15, 0.3), (1.2, 0.05, -0.2)]):
        depth, rgba, quality, _ = synth.room_frame(7 * j, cam)
        pose = synth.pose_euler(yaw, pitch, roll, (0.1 * j, -0.05, 0.2))
        _frame_flow(ov, gv, depth, rgba, quality, pose, kf_id=j, use_quality=True)


def test_empty_depth_selects_nothing(gpu_required):
    ov, gv, cam, ig = make_pair(max_chunks=1 << 12)
    depth = np.zeros((cam.height, cam.width), np.float32)
    pose = synth.pose_identity()
    oids, _ = ov.prepare(depth, pose)
    gv.frame_upload(depth, None, None)
    gids, _ = gv.prepare(pose)
    assert len(oids) == 0 and len(gids) == 0
    gv.integrate_frame(pose, False)
    gv.sync()
    assert gv.stats().n_chunks == 0


def test_hires_camera(gpu_required):
    cam = synth.Camera.hires()
    ov, gv, cam, ig = make_pair(cam=cam, max_chunks=1 << 16)
    depth, rgba, quality, pose = synth.room_frame(5, cam)
    _frame_flow(ov, gv, depth, rgba, quality, pose, kf_id=5, use_quality=True)


def test_missing_chunk_is_an_error(gpu_required):
    from texturefusion_amd import capi
    ov, gv, cam, ig = make_pair(max_chunks=1 << 12)
    depth, rgba, quality, pose = synth.wall_frame(1.0, cam)
    gv.frame_upload(depth, rgba, None)
    ids = np.array([[1000, 1000, 1000]], np.int32)
    with pytest.raises(capi.TFError) as e:
        gv.integrate(pose, ids, np.zeros(1, np.uint8), 1, True, False)
    assert e.value.code == capi.TF_ERR_MISSING_CHUNK


def test_pool_capacity_is_reported(gpu_required):
    from texturefusion_amd import capi
    ov, gv, cam, ig = make_pair(max_chunks=256)
    depth, rgba, quality, pose = synth.wall_frame(1.5, cam)
    gv.frame_upload(depth, rgba, None)
    with pytest.raises(capi.TFError) as e:
        gv.prepare(pose)
    assert e.value.code == capi.TF_ERR_CAPACITY


def test_batched_two_stream_pipeline_matches_oracle(gpu_required):
    """tf_integrate_frames_device: selection of frame f+1 overlaps integration of frame f on a second
    stream (double-buffered selection scratch); results must equal the oracle's frame-by-frame unit."""
    ov, gv, cam, ig = make_pair(max_chunks=1 << 16)
    ks = [0, 1, 2, 3, 4, 5, 6]
    frames = [synth.room_frame(k, cam, with_quality=False) for k in ks]
    dd = [HipBuffer(f[0].nbytes).from_host(f[0]) for f in frames]
    dc = [HipBuffer(f[1].nbytes).from_host(f[1]) for f in frames]
    poses = np.stack([f[3].reshape(12) for f in frames])
    # odd and even batch lengths exercise both selection sets as the "last" one
    for lo, hi in ((0, 3), (3, 7)):
        gv.integrate_frames_device([b.ptr for b in dd[lo:hi]], [b.ptr for b in dc[lo:hi]], poses[lo:hi])
        gv.sync()
        for f in frames[lo:hi]:
            ov.rowstats(clear=True)
            nv, ns = ov.integrate_frame(f[0], f[1], f[3])
        st = gv.stats()
        ost = ov.rowstats()
        assert (st.n_selected, st.n_updated) == (ns, nv)
        assert (st.rows_tsdf, st.rows_color) == (ost.rows_tsdf, ost.rows_color)
        assert st.n_chunks == ov.num_chunks()
    ids = ov.list_chunks()
    assert np.array_equal(sorted_ids(ids), sorted_ids(gv.list_chunks()))
    assert_chunks_equal(ov, gv, ids, "batched")
    assert np.array_equal(sorted_ids(ov.dirty()), sorted_ids(gv.dirty()))
    # the call-by-call flow still works right after a batch
    depth, rgba, quality, pose = synth.room_frame(8, cam)
    _frame_flow(ov, gv, depth, rgba, quality, pose, kf_id=8, use_quality=True)
    for b in dd + dc:
        b.free()


def test_meshes_to_update_clear_and_erase_semantics(gpu_required):
    """meshesToUpdate is expanded lazily from per-chunk mark / erase epochs: check it against the
    oracle's explicit set across fused frames, call-by-call frames, a clear, and garbage-collection."""
    ov, gv, cam, ig = make_pair(max_chunks=1 << 16)
    for k in (0, 1):
        depth, rgba, q, pose = synth.room_frame(k, cam)
        ov.integrate_frame(depth, rgba, pose)
        gv.frame_upload(depth, rgba, None)
        gv.integrate_frame(pose, True)
    gv.sync()
    assert np.array_equal(sorted_ids(ov.dirty()), sorted_ids(gv.dirty()))
    ov.clear_dirty()
    gv.clear_dirty()
    assert len(gv.dirty()) == 0
    depth, rgba, q, pose = synth.room_frame(30, cam)          # mostly new region after the clear
    _frame_flow(ov, gv, depth, rgba, q, pose, kf_id=30, use_quality=True)
    assert np.array_equal(sorted_ids(ov.dirty()), sorted_ids(gv.dirty()))
    depth, rgba, q, pose = synth.room_frame(31, cam)
    ov.integrate_frame(depth, rgba, pose)
    gv.frame_upload(depth, rgba, None)
    gv.integrate_frame(pose, True)
    gv.sync()
    assert np.array_equal(sorted_ids(ov.dirty()), sorted_ids(gv.dirty()))
    assert gv.stats().n_dirty == len(ov.dirty())


def _random_pose(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    t = rng.uniform(-0.5, 0.5, size=3)
    return np.concatenate([R, t.reshape(3, 1)], 1).astype(np.float32)


def test_random_rigid_poses_and_explicit_lists(gpu_required):
    """Arbitrary rotations and hand-made chunk lists: chunks around and behind the camera plane
    (p.z near 0 or negative -> generic division path), far chunks, chunks that project off-image
    (pos-stall), all against the oracle bit for bit."""
    ov, gv, cam, ig = make_pair(max_chunks=1 << 16)
    rng = np.random.default_rng(20261001)
    for it in range(4):
        depth, rgba, quality, _ = synth.room_frame(11 * it, cam)
        pose = _random_pose(rng)
        _frame_flow(ov, gv, depth, rgba, quality, pose, kf_id=it, use_quality=True)
    # explicit list: a cube of chunks around the camera centre of the last pose, incl. behind it
    c = np.floor(pose[:, 3] / (8 * float(RES5))).astype(np.int32)
    ids = np.array([[c[0] + i, c[1] + j, c[2] + k] for i in range(-3, 4) for j in range(-3, 4) for k in range(-3, 4)],
                   np.int32)
    far = np.array([[c[0] + 80 + i, c[1] + j, c[2] + 60] for i in range(3) for j in range(3)], np.int32)
    ids = np.concatenate([ids, far])
    for cid in ids:
        if not ov.has_chunk(cid):
            ov.set_chunk(cid, *O.fresh_chunk())
            gv.set_chunk(cid, *O.fresh_chunk())
    depth = np.full((cam.height, cam.width), 0.06, np.float32)   # a surface 6 cm in front of the camera
    depth[::7, ::5] = 2.5
    gv.frame_upload(depth, rgba, quality)
    on, gn = np.zeros(len(ids), np.uint8), np.zeros(len(ids), np.uint8)
    oq = ov.integrate(depth, rgba, quality, pose, ids, on, 1, 9)
    gq = gv.integrate(pose, ids, gn, 1, True, True)
    assert np.array_equal(on, gn)
    assert np.array_equal(oq.view(np.uint32), gq.view(np.uint32))
    assert_chunks_equal(ov, gv, ids, "explicit list")


def test_rgb_plus_valid_mask_packs_like_the_callers_loop(gpu_required):
    """tf_frame_upload_rgb == the RGBA staging loop of MobileFusion.cpp:144-163 done on the host."""
    cam = synth.Camera()
    depth, rgba, quality, pose = synth.room_frame(4, cam)
    rng = np.random.default_rng(3)
    valid = (rng.random((cam.height, cam.width)) > 0.2).astype(np.uint8) * rng.integers(1, 255, (cam.height, cam.width)).astype(np.uint8)
    rgb = np.ascontiguousarray(rgba[..., :3])
    packed = np.zeros_like(rgba)
    packed[..., :3] = np.where(valid[..., None] > 0, rgb, 0)
    packed[..., 3] = (valid > 0).astype(np.uint8)
    a = capi.Volume(RES5, cam, max_chunks=1 << 16)
    b = capi.Volume(RES5, cam, max_chunks=1 << 16)
    a.frame_upload(depth, packed, quality)
    b.frame_upload_rgb(depth, rgb, valid, quality)
    for v in (a, b):
        v.integrate_frame(pose, True)
        v.sync()
    ids = sorted_ids(a.list_chunks())
    assert np.array_equal(ids, sorted_ids(b.list_chunks()))
    sa, wa, ca = a.get_chunks(ids)
    sb, wb, cb = b.get_chunks(ids)
    assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32)) and np.array_equal(wa.view(np.uint32), wb.view(np.uint32))
    assert np.array_equal(ca, cb) and ca.any()


def test_non_finite_and_negative_depth_pixels(gpu_required):
    """Depth images with NaN, +-Inf, negative, huge and denormal pixels (what a filter chain ahead of the path can leave
    behind): the reference's comparisons are all ordered (false on NaN), its min / max reductions keep the running value --
    bounding box, selection, voxel update, colour and the fused per-frame unit must agree with the oracle bit for bit."""
    ov, gv, cam, ig = make_pair(max_chunks=1 << 16)
    rng = np.random.default_rng(99)
    for it, k in enumerate((3, 4, 5)):
        depth, rgba, quality, pose = synth.room_frame(k, cam)
        depth = depth.copy()
        H, W = depth.shape
        bad = [np.nan, np.inf, -np.inf, -1.5, 1e30, 1e-40, 0.0, 5.0, 0.01]
        ys = rng.integers(0, H, 4000); xs = rng.integers(0, W, 4000)
        depth[ys, xs] = np.float32(rng.choice(bad, 4000))
        depth[100:104, 200:260] = np.nan        # a block of NaNs: whole rows of a chunk's footprint
        depth[300:303, 50:90] = np.inf
        if it < 2:
            _frame_flow(ov, gv, depth, rgba, quality, pose, kf_id=it, use_quality=True)
        else:  # the fused unit on the same kind of image
            ov.integrate_frame(depth, rgba, pose)
            gv.frame_upload(depth, rgba, None)
            gv.integrate_frame(pose, True)
            gv.sync()
    ids = sorted_ids(ov.list_chunks())
    assert np.array_equal(ids, sorted_ids(gv.list_chunks()))
    assert_chunks_equal(ov, gv, ids[::3], "non-finite depth")
    assert np.array_equal(sorted_ids(ov.dirty()), sorted_ids(gv.dirty()))
    gv.close()


def test_list_and_candidate_grid_capacity_errors_leave_no_trace(gpu_required):
    """A frame whose visible list (tf_config.max_list) or candidate grid (max_coarse) does not fit is an error of THAT call
    -- call by call and in the fused unit (where the frame is skipped as a whole) --, the sticky status is cleared by the
    call that reports it, and the volume goes on exactly like one that never saw the frame."""
    cam = synth.Camera()
    big = synth.room_frame(7, cam)            # ~10 k chunks
    small = synth.wall_frame(0.35, cam, seed=3)  # a wall 35 cm away: a few hundred chunks
    for kw, text in ((dict(max_list=512), "visible list full"), (dict(max_coarse=64), "candidate grid full")):
        ov, gv, cam, ig = make_pair(cam=cam, max_chunks=1 << 16, **kw)
        gv.frame_upload(big[0], big[1], None)
        with pytest.raises(capi.TFError) as e:
            gv.prepare(big[3])
        assert e.value.code == capi.TF_ERR_CAPACITY and text in str(e.value)
        with pytest.raises(capi.TFError) as e:   # the fused unit: enqueued, reported by the next synchronising call
            gv.integrate_frame(big[3], True)
            gv.sync()
        assert e.value.code == capi.TF_ERR_CAPACITY and text in str(e.value)
        gv.sync()                                 # (cleared)
        assert len(gv.list_chunks()) == 0 or kw.get("max_coarse")  # nothing of the oversized frame was integrated ...
        n_before = len(gv.list_chunks())
        if kw.get("max_coarse") is None:
            # ... and a frame that fits runs as on a fresh volume
            ov.integrate_frame(small[0], small[1], small[3])
            gv.frame_upload(small[0], small[1], None)
            gv.integrate_frame(small[3], True)
            gv.sync()
            ids = sorted_ids(ov.list_chunks())
            assert len(ids) > 50 and np.array_equal(ids, sorted_ids(gv.list_chunks()))
            assert_chunks_equal(ov, gv, ids, "after a capacity error")
        else:
            assert n_before == 0
        gv.close()
