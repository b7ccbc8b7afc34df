"""Texture-atlas path on the GPU vs the oracle: slot assignment / bbox / flags bit-exact, texcoord and
texcolour bit-exact (same operation order), atlas texels exact in the copy branch and exact against the
oracle's INTER_LINEAR restatement in the resize branch (that restatement itself is parity-unpinned
third-party arithmetic)."""
import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import capi, synth

pytestmark = pytest.mark.gpu
RES5 = np.float32(0.005)


def _scene(z, cam, pose):
    depth = np.full((cam.height, cam.width), z, np.float32)
    rays = synth._rays(cam)
    R = pose[:, :3].astype(np.float64)
    world = pose[:, 3].astype(np.float64) + (rays * z) @ R.T
    rgb = np.ascontiguousarray(synth._hash_colour(world, 17)[..., :3])
    return depth, rgb


def _chunk_grid(z, res, nx=10, ny=8):
    edge = 8 * float(res)
    kz = int(np.floor(z / edge))
    return np.array([[i, j, kz] for i in range(-nx, nx) for j in range(-ny, ny)], np.int32)


def _run_both(gv, oa, cam, C, ids, voff, verts, cols, kf, T, rgb, depth):
    n = len(ids)
    out = gv.patches_update(ids, np.full(n, kf, np.int32), np.tile(T, (n, 1)), voff, verts, cols)
    assert out["rc"] == 0
    texlocs = []
    for p in range(n):
        sl = slice(voff[p], voff[p + 1])
        rc, tl = oa.alloc() if tuple(ids[p]) not in _run_both.seen else (0, _run_both.seen[tuple(ids[p])])
        assert rc == 0
        _run_both.seen[tuple(ids[p])] = tl
        texlocs.append(tl)
        o = O.patch_project(verts[sl], cols[sl], T, rgb, depth, C)
        assert out["texloc"][p] == tl
        assert np.array_equal(out["bbox"][p], o["bbox"]), (p, out["bbox"][p], o["bbox"])
        assert (out["flags"][p] & 1) == (1 if o["flag"] == -1 else 0)
        assert bool(out["flags"][p] & 2) == o["wrong_mapping"]
        assert np.array_equal(out["texcoord"][sl].view(np.uint32), o["texcoord"].view(np.uint32))
        assert np.array_equal(out["texcolor"][sl].view(np.uint32), o["texcolor"].view(np.uint32))
        rc, ratio = oa.blit(tl, rgb, o["bbox"])
        assert rc == 0
        assert np.array_equal(out["ratio"][p].view(np.uint32), ratio.view(np.uint32))
    hs, he = oa.hot_range(texlocs)
    assert out["hot"] == (hs, he)
    return out, texlocs


_run_both.seen = {}


@pytest.mark.parametrize("z,tilt", [(1.5, 0.0), (0.8, 0.0), (1.1, 0.25)])
def test_patches_match_oracle(gpu_required, z, tilt):
    _run_both.seen = {}
    cam = synth.Camera()
    C = O.camera_from(cam)
    pose = synth.pose_euler(tilt, tilt / 2, 0.0, (0.02, -0.01, 0.0))
    depth, rgb = _scene(z, cam, pose)
    AH = 18 * 40
    gv = capi.Volume(RES5, cam, max_chunks=1 << 12, atlas_w=13824, atlas_h=AH)
    oa = O.Atlas(RES5, 13824, AH)
    assert gv.atlas_patch_size() == (oa.pw, oa.ph) == (24, 18)
    gv.keyframe_cache(3, rgb, depth)
    # plane z = const in CAMERA space expressed in world space: build vertices in camera space, move to world
    ids, voff, verts_c, cols = synth.wall_mesh_for_chunks(_chunk_grid(z, RES5), RES5, z)
    R, t = pose[:, :3].astype(np.float64), pose[:, 3].astype(np.float64)
    verts = (verts_c.astype(np.float64) @ R.T + t).astype(np.float32)
    T = synth.pose_inverse16(pose)
    out, texlocs = _run_both(gv, oa, cam, C, ids, voff, verts, cols, 3, T, rgb, depth)
    big = (out["ratio"] < 1).any(axis=1).sum()
    if z < 1.0:
        assert big > 0          # close wall: ROI larger than the 24x18 slot -> resize branch exercised
    rows = gv.atlas_rows(0, AH, 13824)
    assert np.array_equal(rows, oa.buffer()[:AH])
    # second update of the same chunks keeps every slot (Patch::clear keeps texloc, Atlas.cpp:60-62)
    nxt = gv.atlas_loc_next()
    out2, _ = _run_both(gv, oa, cam, C, ids, voff, verts, cols, 3, T, rgb, depth)
    assert gv.atlas_loc_next() == nxt == oa.loc_next()
    assert np.array_equal(out2["texloc"], out["texloc"])
    gv.close()


def test_atlas_overflow_is_minus_one(gpu_required):
    cam = synth.Camera()
    pose = synth.pose_identity()
    depth, rgb = _scene(1.5, cam, pose)
    gv = capi.Volume(RES5, cam, max_chunks=1 << 10, atlas_w=96, atlas_h=36)  # 8 slots
    gv.keyframe_cache(0, rgb, depth)
    ids, voff, verts, cols = synth.wall_mesh_for_chunks(_chunk_grid(1.5, RES5, 3, 2), RES5, 1.5)
    assert len(ids) > 8
    T = synth.pose_inverse16(pose)
    out = gv.patches_update(ids, np.zeros(len(ids), np.int32), np.tile(T, (len(ids), 1)), voff, verts, cols)
    assert out["rc"] == capi.TF_ERR_ATLAS_FULL == -1   # Chisel::GeneratePatches' return value
    gv.close()


def test_vertices_outside_the_image_are_flagged(gpu_required):
    cam = synth.Camera()
    C = O.camera_from(cam)
    pose = synth.pose_identity()
    depth, rgb = _scene(1.5, cam, pose)
    gv = capi.Volume(RES5, cam, max_chunks=1 << 10, atlas_w=13824, atlas_h=36)
    gv.keyframe_cache(1, rgb, depth)
    verts = np.array([[0.0, 0.0, 1.5], [0.05, 0.02, 1.5], [5.0, 0.0, 1.5], [0.0, -4.0, 1.5]], np.float32)
    cols = np.full((4, 3), 0.5, np.float32)
    ids = np.array([[0, 0, 37]], np.int32)
    T = synth.pose_inverse16(pose)
    out = gv.patches_update(ids, np.array([1], np.int32), T[None], np.array([0, 4], np.int64), verts, cols)
    o = O.patch_project(verts, cols, T, rgb, depth, C)
    assert o["flag"] == -1 and (out["flags"][0] & 1)
    assert np.array_equal(out["bbox"][0], o["bbox"])
    # in-image vertices must agree bit for bit; clamped ones read outside the image in the reference
    assert np.array_equal(out["texcoord"].view(np.uint32), o["texcoord"].view(np.uint32))
    assert np.array_equal(out["texcolor"][:2].view(np.uint32), o["texcolor"][:2].view(np.uint32))
    gv.close()


def test_room_frame_patches_match_oracle(gpu_required):
    """General pose, depth-derived vertices (synth.mesh_from_depth), ~1.5 k patches in one batch."""
    _run_both.seen = {}
    cam = synth.Camera()
    C = O.camera_from(cam)
    depth, rgba, q, pose = synth.room_frame(12, cam)
    rgb = np.ascontiguousarray(rgba[..., :3])
    AH = 18 * 8
    gv = capi.Volume(RES5, cam, max_chunks=1 << 12, atlas_w=13824, atlas_h=AH)
    oa = O.Atlas(RES5, 13824, AH)
    gv.keyframe_cache(12, rgb, depth)
    ids, voff, verts, cols = synth.mesh_from_depth(depth, rgba, pose, cam, RES5, stride=4, max_chunks=1500)
    assert len(ids) == 1500 and voff[-1] > 5000
    T = synth.pose_inverse16(pose)
    out, texlocs = _run_both(gv, oa, cam, C, ids, voff, verts, cols, 12, T, rgb, depth)
    assert np.array_equal(gv.atlas_rows(0, AH, 13824), oa.buffer()[:AH])
    gv.close()


def test_device_resident_update_equals_host_variant(gpu_required):
    """tf_patches_update_device (meshes and results in HBM, asynchronous) == tf_patches_update, bit for bit:
    slots, per-patch records, texcoords, texcolours and the atlas texels."""
    from tests.util import HipBuffer
    cam = synth.Camera()
    AH = 18 * 40
    vols = [capi.Volume(RES5, cam, max_chunks=1 << 10, atlas_w=13824, atlas_h=AH) for _ in range(2)]
    pose = synth.pose_euler(0.2, -0.1, 0.05, (0.05, 0.0, 0.1))
    depth, rgb = _scene(1.4, cam, pose)
    ids, voff, verts_c, cols = synth.wall_mesh_for_chunks(_chunk_grid(1.4, RES5, nx=8, ny=6), RES5, 1.4)
    R, t = pose[:, :3].astype(np.float64), pose[:, 3].astype(np.float64)
    verts = (verts_c.astype(np.float64) @ R.T + t).astype(np.float32)
    T = synth.pose_inverse16(pose)
    n = len(ids)
    for v in vols:
        v.keyframe_cache(3, rgb, depth)
    ref = vols[0].patches_update(ids, np.full(n, 3, np.int32), np.tile(T, (n, 1)), voff, verts, cols)
    nv = int(voff[-1])
    bufs = dict(verts=HipBuffer(nv * 12), cols=HipBuffer(nv * 12), tc=HipBuffer(nv * 8), tcol=HipBuffer(nv * 12),
                po=HipBuffer(n * 32))
    bufs["verts"].from_host(np.ascontiguousarray(verts, np.float32))
    bufs["cols"].from_host(np.ascontiguousarray(cols, np.float32))
    for rep in range(2):  # the second call re-uses the slots (Patch::clear keeps texloc) and the descriptor ring
        rc, texloc, hot = vols[1].patches_update_device(ids, np.full(n, 3, np.int32), np.tile(T, (n, 1)), voff,
                                                        bufs["verts"].ptr, bufs["cols"].ptr, bufs["tc"].ptr,
                                                        bufs["tcol"].ptr, bufs["po"].ptr)
        assert rc == 0
    vols[1].sync()
    assert np.array_equal(texloc, ref["texloc"]) and hot == ref["hot"]
    tc = bufs["tc"].to_host(nv * 8).view(np.float32).reshape(-1, 2)
    tcol = bufs["tcol"].to_host(nv * 12).view(np.float32).reshape(-1, 3)
    po = bufs["po"].to_host(n * 32).view(np.int32).reshape(n, 8)
    assert np.array_equal(tc.view(np.uint32), ref["texcoord"].view(np.uint32))
    assert np.array_equal(tcol.view(np.uint32), ref["texcolor"].view(np.uint32))
    assert np.array_equal(po[:, :4], ref["bbox"]) and np.array_equal(po[:, 4], ref["flags"])
    assert np.array_equal(po[:, 5:7].view(np.float32).view(np.uint32), ref["ratio"].view(np.uint32))
    assert np.array_equal(vols[0].atlas_rows(0, AH, 13824), vols[1].atlas_rows(0, AH, 13824))
    for v in vols:
        v.close()
    for b in bufs.values():
        b.free()
