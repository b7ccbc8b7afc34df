"""The textured per-frame unit away from the bench stream: a hand-held orbit (general rotations), other image and
voxel sizes, the 1280x960 hall of configs[3], the host-frames entry point -- each against the oracle's per-frame
unit, bit for bit (voxels, meshes, patches, slots, atlas texels).  Every case takes a few seconds."""
import os

import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import capi, synth
from tests.util import assert_chunks_equal, sorted_ids, HipBuffer
from tests.test_gpu_atlas import _compare_patches, _compare_atlas

pytestmark = pytest.mark.gpu


def _run(cam, res, frames, host_frames=False, max_chunks=1 << 17, stride=1):
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    # (TF_SOAK_MESH_BLOCKS: a small mesh store -- the runs of tools/soak_random.py then live off recycled blocks)
    mb = int(os.environ.get("TF_SOAK_MESH_BLOCKS", "0"))
    gv = capi.Volume(res, cam, max_chunks=max_chunks, max_list=1 << 17, max_coarse=1 << 20, **({"mesh_blocks": mb} if mb else {}))
    oa = O.Atlas(res)
    pinv = [synth.pose_inverse16(f[3]) for f in frames]
    for k, f in enumerate(frames):
        ov.frame_textured(oa, f[0], f[1], f[3], pinv[k], 10 + k)
    bufs = []
    if host_frames == "rgb":  # Frame::rgb + Frame::colorValidFlag (or none) as the caller holds them; packed on the device
        for k, f in enumerate(frames):
            valid = np.ascontiguousarray(f[1][..., 3])
            rgb = np.ascontiguousarray(f[1][..., :3]).copy()
            rgb[valid == 0] = 77  # (whatever the camera delivered where the flag says invalid)
            gv.integrate_frame_host_rgb(f[0], rgb, None if valid.all() else valid, f[3].reshape(12), pinv[k], 10 + k)
    elif host_frames == "registered":  # ONE pair of caller buffers, registered once, refilled for every frame and
        d = np.empty_like(frames[0][0]); c = np.empty_like(frames[0][1])  # scribbled over as soon as the call returns
        gv.host_register(d); gv.host_register(c)
        gv.host_register(d)  # (inside a registered range already: a no-op)
        for k, f in enumerate(frames):
            d[...] = f[0]; c[...] = f[1]
            gv.integrate_frame_host(d, c, f[3].reshape(12), pinv[k], 10 + k)
            d[...] = 123.0; c[...] = 200
        gv.sync()
        gv.host_unregister(c); gv.host_unregister(d)
    elif host_frames == "registered_async":  # a ring of four registered buffer pairs, calls that do not wait for their upload,
        ring = [(np.empty_like(frames[0][0]), np.empty_like(frames[0][1])) for _ in range(4)]  # a fence before a pair is refilled
        for d, c in ring:
            gv.host_register(d); gv.host_register(c)
        gv.host_frame_set_async(True)
        for k, f in enumerate(frames):
            d, c = ring[k % 4]
            if k >= 4:
                gv.host_frame_fence()   # every upload queued so far is through: the pair may be written again
            d[...] = f[0]; c[...] = f[1]
            gv.integrate_frame_host(d, c, f[3].reshape(12), pinv[k], 10 + k)
        gv.host_frame_fence()
        for d, c in ring:
            d[...] = 7.0; c[...] = 9
        gv.sync()
        gv.host_frame_set_async(False)
        for d, c in ring:
            gv.host_unregister(c); gv.host_unregister(d)
    elif host_frames:
        if host_frames == "no_deferral":  # integrate in the call that brings the frame (tf_host_frame_set_deferral)
            gv.host_frame_set_deferral(False)
            assert gv.host_frame_deferral()[0] == 0
        for k, f in enumerate(frames):
            gv.integrate_frame_host(f[0], f[1], f[3].reshape(12), pinv[k], 10 + k)
    else:
        bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in frames]
        poses = np.stack([f[3].reshape(12) for f in frames])
        gv.stream_frames_textured_device([b[0].ptr for b in bufs], [b[1].ptr for b in bufs], poses, np.stack(pinv), 10)
    gv.sync()
    oids = sorted_ids(ov.list_chunks())
    assert np.array_equal(oids, sorted_ids(gv.list_chunks()))
    assert_chunks_equal(ov, gv, oids[::stride], "textured soak")
    mids = sorted_ids(ov.list_meshes())
    assert np.array_equal(mids, sorted_ids(gv.list_meshes()))
    sub = mids[::stride]
    voff, ioff, V, N, Cc, I, adj, simp = gv.get_meshes(sub)
    for i, cid in enumerate(sub):
        m = ov.get_mesh(cid)
        assert np.array_equal(V[voff[i]:voff[i + 1]].view(np.uint32), m["verts"].view(np.uint32)), cid
        assert np.array_equal(N[voff[i]:voff[i + 1]].view(np.uint32), m["normals"].view(np.uint32)), cid
        assert np.array_equal(I[ioff[i]:ioff[i + 1]], m["indices"]), cid
        assert bool(simp[i]) == m["simplified"] and np.array_equal(adj[i], m["adj"]), cid
    assert gv.atlas_loc_next() == oa.loc_next()
    g = _compare_patches(ov, gv, sub, "textured soak")
    gall = gv.get_patches(mids)
    used = gall["texloc"][gall["texloc"] != np.uint64((1 << 64) - 1)]
    if len(used):
        _compare_atlas(oa, gv, oa.hot_range(used))
    n_meshes = len(mids)
    for a, b in bufs:
        a.free(); b.free()
    gv.close()
    return n_meshes


def random_case(rng, stride=None):
    """One randomised case of the textured unit -- image size, focal length, voxel size, orbit stretch, hand-held wobble, entry
    point -- compared with the oracle by _run (also what tools/soak_random.py loops over); returns (description, meshes)."""
    W = int(rng.choice([320, 400, 480, 640])); H = int(rng.choice([240, 304, 360, 480]))
    f = float(rng.uniform(0.7, 1.0) * W * 525.0 / 640.0)
    res = float(rng.choice([0.005, 0.006, 0.008, 0.01]))
    cam = synth.Camera(W, H, f, f, W / 2 - 0.5, H / 2 - 0.5, 0.01, 5.0)
    k0 = int(rng.integers(0, 180)); step = int(rng.integers(1, 4)); n = int(rng.integers(10, 26))
    wob = float(rng.uniform(0, 0.1)); radius = float(rng.uniform(0.2, 1.2))
    frames = [synth.room_frame(k0 + step * i, cam, with_quality=False, wobble=wob, radius=radius) for i in range(n)]
    host = [False, True, "registered", "registered_async", "no_deferral", "rgb"][int(rng.integers(0, 6))]  # entry point / host-frame path
    st = int(rng.integers(1, 6)) if stride is None else stride
    nm = _run(cam, np.float32(res), frames, host_frames=host, max_chunks=1 << 18, stride=st)
    return "%dx%d f %.0f res %.3f frames %d step %d wobble %.2f radius %.2f host %s" % (W, H, f, res, n, step, wob, radius, host), nm


@pytest.mark.parametrize("seed", [61, 62])
def test_randomised_cases(gpu_required, seed):
    """three cases per seed of tools/soak_random.py's generator, every chunk / mesh / patch compared (the tool itself runs
    as many as one likes: 46 cases passed at the end of round 2, 24 more in round 6)"""
    rng = np.random.default_rng(seed)
    for _ in range(3):
        what, nm = random_case(rng, stride=1)
        assert nm > 0, what


def test_hand_held_orbit(gpu_required):
    """general rotation matrices (pitch / roll wobble on the orbit): the summation-order-sensitive case"""
    cam = synth.Camera()
    frames = [synth.room_frame(k, cam, with_quality=False, wobble=0.08) for k in range(12)]
    assert _run(cam, np.float32(0.005), frames, stride=3) > 500


def test_long_orbit_of_the_bench_stream(gpu_required):
    """90 consecutive frames of the stream bench.py times (almost half the orbit): weights pass the mesher's
    threshold of 50, colour counts reach their cap of 120, meshes lose and regain vertices, patches are re-projected
    dozens of times and the filter's class summaries (kept by the voxel writers, made exact by the filter) live
    through thousands of updates per chunk."""
    cam = synth.Camera()
    frames = [synth.room_frame(k, cam, with_quality=False) for k in range(90)]
    assert _run(cam, np.float32(0.005), frames, max_chunks=1 << 18, stride=1) > 5000  # EVERY chunk, mesh and patch compared


@pytest.mark.parametrize("W,H,f,res", [(320, 240, 262.5, 0.01), (400, 304, 330.0, 0.008)])
def test_other_cameras_and_voxel_sizes(gpu_required, W, H, f, res):
    cam = synth.Camera(W, H, f, f, W / 2 - 0.5, H / 2 - 0.5, 0.01, 5.0)
    frames = [synth.room_frame(k, cam, with_quality=False, wobble=0.03) for k in range(16)]
    assert _run(cam, np.float32(res), frames, stride=2) > 50


def test_host_frames_entry_point(gpu_required):
    cam = synth.Camera()
    frames = [synth.room_frame(2 * k, cam, with_quality=False) for k in range(10)]
    assert _run(cam, np.float32(0.005), frames, host_frames=True, stride=3) > 400


def test_host_frames_async_with_fence(gpu_required):
    """tf_host_frame_set_async + tf_host_frame_fence: calls out of a ring of registered buffers return before their upload
    is through, the caller fences before it refills a buffer; results as with waiting calls"""
    cam = synth.Camera()
    frames = [synth.room_frame(2 * k, cam, with_quality=False) for k in range(10)]
    assert _run(cam, np.float32(0.005), frames, host_frames="registered_async", stride=3) > 400


def test_host_frames_without_deferral_per_handle(gpu_required):
    """tf_host_frame_set_deferral(v, 0): a live caller's setting -- every frame is in the volume when its call returns to the
    stream (no four-frame latency); the default of other handles is untouched"""
    cam = synth.Camera()
    frames = [synth.room_frame(2 * k, cam, with_quality=False) for k in range(8)]
    assert _run(cam, np.float32(0.005), frames, host_frames="no_deferral", stride=3) > 300
    assert capi.host_frame_deferral()[0] in (0, 4)  # (the build's default; 0 only with TF_HOST_DEFER=0 in the environment)


def test_host_frames_out_of_registered_caller_buffers(gpu_required):
    """tf_host_register: frames uploaded straight out of the caller's own (page-locked in place) buffers; the buffers are the
    caller's again when the call returns -- they are overwritten at once here"""
    cam = synth.Camera()
    frames = [synth.room_frame(2 * k, cam, with_quality=False) for k in range(10)]
    assert _run(cam, np.float32(0.005), frames, host_frames="registered", stride=3) > 400


def test_host_frames_as_rgb_and_valid_flags(gpu_required):
    """tf_integrate_frame_host_rgb: the caller's RGBA staging loops (MobileFusion.cpp:144-163, :232-243) on the device --
    once with every pixel valid (no flags: alpha = 1 everywhere), once with flagged-out regions (rgba = 0 there)"""
    cam = synth.Camera()
    frames = [synth.room_frame(2 * k, cam, with_quality=False) for k in range(10)]
    assert all(f[1][..., 3].all() for f in frames)
    assert _run(cam, np.float32(0.005), frames, host_frames="rgb", stride=3) > 400
    holes = []
    for k, f in enumerate(frames):
        rgba = f[1].copy()
        rgba[40 + 7 * k:200, 100:300 + 11 * k] = 0   # colorValidFlag == 0: the staging loop zeroes all four bytes
        rgba[::5, ::3] = 0
        holes.append((f[0], rgba, f[2], f[3]))
    assert _run(cam, np.float32(0.005), holes, host_frames="rgb", stride=3) > 400


def test_hall_1280x960(gpu_required):
    """BASELINE configs[3]'s camera and hall (8 x 6 x 8 m, 1280 x 960), a few frames (the oracle needs ~1.5 s per
    frame here).  The bench orbits at 0.5 m from the centre, where walls at 3-4 m need ~20 frames to pass the mesher's
    weight threshold; the test orbits at 3.2 m instead, 0.8 m from the wall it faces, so that six frames produce
    meshes -- and patches larger than their atlas slots (the resize branch) -- while the far walls still fill the
    volume with tens of thousands of chunks."""
    cam = synth.Camera.hires()
    frames = [synth.room_frame(k, cam, half=(4.0, 3.0, 4.0), radius=3.2, with_quality=False) for k in range(18, 24)]
    n = _run(cam, np.float32(0.005), frames, max_chunks=1 << 19, stride=11)
    assert n > 200


def test_hall_bench_stream_30_frames(gpu_required):
    """the 1280x960 hall exactly as bench.py streams it (configs[3]: orbit at 0.5 m, 70-80 k chunks per frame): long
    dirty lists take the filter's batch form (eight lanes per entry), the visible list is two-ended with ~66 k costly
    chunks, and the walls' weights pass the mesher's threshold after a couple of dozen frames"""
    cam = synth.Camera.hires()
    frames = [synth.room_frame(k, cam, half=(4.0, 3.0, 4.0), radius=0.5, with_quality=False) for k in range(30)]
    n = _run(cam, np.float32(0.005), frames, max_chunks=1 << 20, stride=1)  # EVERY chunk, mesh and patch compared
    assert n > 2000


def test_host_frames_deferral_is_not_observable(gpu_required):
    """tf_integrate_frame_host runs two frames behind internally; any other entry point has to see every frame that was
    handed over: state queries after 1, 2, 3, 5 and 6 calls (pipeline depths 1, 2, 2, 2, 1 at the flush), a
    call-by-call integration in between, and another stream of host frames behind it."""
    cam = synth.Camera()
    res = np.float32(0.005)
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(res, cam, max_chunks=1 << 16)
    oa = O.Atlas(res)
    frames = [synth.room_frame(3 * k, cam, with_quality=False, wobble=0.04) for k in range(9)]
    pinv = [synth.pose_inverse16(f[3]) for f in frames]

    def check(tag):
        oids = sorted_ids(ov.list_chunks())
        assert np.array_equal(oids, sorted_ids(gv.list_chunks())), tag  # (list_chunks is one of the flushing entry points)
        assert_chunks_equal(ov, gv, oids[::9], tag)
        assert np.array_equal(sorted_ids(ov.list_meshes()), sorted_ids(gv.list_meshes())), tag

    for k in range(6):
        f = frames[k]
        ov.frame_textured(oa, f[0], f[1], f[3], pinv[k], 70 + k)
        gv.integrate_frame_host(f[0], f[1], f[3].reshape(12), pinv[k], 70 + k)
        if k in (0, 1, 2, 4, 5):
            check("after %d host frames" % (k + 1))
    # a call-by-call frame in between (prepare / integrate / finalize), then host frames again
    f = frames[6]
    oids, onew = ov.prepare(f[0], f[3])
    gv.frame_upload(f[0], f[1], None)
    gids, gnew = gv.prepare(f[3])
    assert np.array_equal(oids, gids)
    on, gn = np.zeros(len(oids), np.uint8), np.zeros(len(oids), np.uint8)
    ov.integrate(f[0], f[1], None, f[3], oids, on, 1, -1)
    gv.integrate(f[3], gids, gn, 1, True, False)
    assert np.array_equal(on, gn)
    assert np.array_equal(ov.finalize(oids, on, onew), gv.finalize(gids, gn, gnew))
    for k in (7, 8):
        f = frames[k]
        ov.integrate_frame(f[0], f[1], f[3])
        gv.integrate_frame_host(f[0], f[1], f[3].reshape(12), None, 0)
    oids = sorted_ids(ov.list_chunks())
    assert np.array_equal(oids, sorted_ids(gv.list_chunks()))
    assert_chunks_equal(ov, gv, oids[::9], "host frames behind a call-by-call frame")
    gv.close()


def test_textured_frames_after_a_tsdf_only_stretch(gpu_required):
    """Chisel::meshesToUpdate collects marks until CompressMeshes clears it: the first textured frame behind frames that
    were integrated without the textured unit meshes and textures everything those frames touched, not its own chunks
    alone (and the summaries those frames left as 'anything' are made exact on the way)."""
    cam = synth.Camera()
    res = np.float32(0.005)
    frames = [synth.room_frame(2 * k, cam, with_quality=False, wobble=0.03) for k in range(14)]
    pinv = [synth.pose_inverse16(f[3]) for f in frames]
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(res, cam, max_chunks=1 << 17)
    oa = O.Atlas(res)
    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in frames]
    poses = np.stack([f[3].reshape(12) for f in frames])

    def tsdf(lo, hi):
        for k in range(lo, hi):
            ov.integrate_frame(frames[k][0], frames[k][1], frames[k][3])
        gv.stream_frames_device([b[0].ptr for b in bufs[lo:hi]], [b[1].ptr for b in bufs[lo:hi]], poses[lo:hi])

    def textured(lo, hi):
        for k in range(lo, hi):
            ov.frame_textured(oa, frames[k][0], frames[k][1], frames[k][3], pinv[k], 40 + k)
        gv.stream_frames_textured_device([b[0].ptr for b in bufs[lo:hi]], [b[1].ptr for b in bufs[lo:hi]], poses[lo:hi],
                                         np.stack(pinv[lo:hi]), 40 + lo)

    def check(tag):
        gv.sync()
        mids = sorted_ids(ov.list_meshes())
        assert np.array_equal(mids, sorted_ids(gv.list_meshes())), tag
        assert len(gv.dirty()) == len(ov.dirty()), tag
        voff, ioff, V, N, Cc, I, adj, simp = gv.get_meshes(mids)
        for i, cid in enumerate(mids):
            m = ov.get_mesh(cid)
            assert np.array_equal(V[voff[i]:voff[i + 1]].view(np.uint32), m["verts"].view(np.uint32)), (tag, cid)
            assert bool(simp[i]) == m["simplified"] and np.array_equal(adj[i], m["adj"]), (tag, cid)
        assert gv.atlas_loc_next() == oa.loc_next(), tag
        _compare_patches(ov, gv, mids, tag)
        return len(mids)

    tsdf(0, 6)
    textured(6, 8)      # frame 6 inherits the marks of frames 0..5
    n1 = check("behind six TSDF-only frames")
    tsdf(8, 10)
    gv.update_meshes(); ov.update_meshes()   # call-by-call mesher in between: the marks stay (no CompressMeshes)
    textured(10, 14)
    n2 = check("behind two more and an UpdateMeshes")
    assert n1 > 300 and n2 >= n1
    for a, b in bufs:
        a.free(); b.free()
    gv.close()


def test_pool_of_four_million_chunks(gpu_required):
    """tf_config.max_chunks beyond 2^21 (the ceiling of rounds 1-4: pool slots were 21-bit fields of the patch lists): a
    2^22-slot pool -- 32 GiB of voxels, the mesh store allocated on demand -- takes the hall's frames bit for bit like any
    other.  The reference's ChunkMap is an unordered_map (Structure/ChunkManager.h:111): only memory bounds it."""
    cam = synth.Camera.hires()
    frames = [synth.room_frame(k, cam, half=(4.0, 3.0, 4.0), radius=3.2, with_quality=False) for k in range(18, 24)]
    n = _run(cam, np.float32(0.005), frames, max_chunks=1 << 22, stride=17)
    assert n > 200
