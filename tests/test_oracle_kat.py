"""Known-answer tests that pin the CPU oracle.

Sources of the expected values:
  * SURVEY.md App. A.1-9 -- numbers the survey recorded from the reference's own voxelUpdateSIMD
    translation unit (w = 17.676939, sdf = 0.00249998365 -> 0.00249996944, q = 48, colour cap 121 -> 30,
    border chunk (0,-12,24) not updated), committed as data in tests/golden/survey_kat.json;
  * BASELINE.md s.2 sizing figures of the reference kernel (4606 chunks updated, 70656 colour rows
    for a wall at 1.5 m);
  * closed-form values of the formulas the reference states in source (truncation, weights, atlas slots).
The reference has no tests / golden vectors of its own and cannot be built here, so
everything beyond these is "parity unpinned" (see oracle/tf_oracle.h).
"""
import json
import os

import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "survey_kat.json")))
RES5 = np.float32(0.005)


@pytest.fixture(scope="module")
def env():
    cam = synth.Camera()
    return cam, O.camera_from(cam), O.default_integrator()


def test_truncation_and_weight_closed_form(env):
    cam, C, ig = env
    z = np.float32(0.96)  # origin z of chunk (0,0,24) at 5 mm
    expect = abs(0.0019 * 0.96 ** 2 + 0.00152 * 0.96 + 0.001504) * 6.0
    tr = O.truncation(ig, z)
    assert abs(tr - expect) < 1e-8
    assert np.float32(1.0) / (np.float32(2.0) * np.float32(tr)) == np.float32(KAT["first_pass"]["weight"])


def test_survey_probe_first_pass(env):
    cam, C, ig = env
    k = KAT["first_pass"]
    depth, rgba, q, pose = synth.wall_frame(1.0, cam, hole_stride=0, rgba_value=tuple(k["rgba"]),
                                            quality_value=k["quality_px"])
    sdf, w, col = O.fresh_chunk()
    upd, qs, st = O.voxel_update(depth, rgba, q, C, ig, pose, 1, k["chunk"], RES5, sdf, w, col)
    assert upd
    assert qs == np.float32(k["q"])                      # 0.25 x 192 colour-band voxels
    assert w.max() == np.float32(k["weight"])            # 17.676939
    assert np.float32(k["sdf"]) in sdf                   # 0.00249998365
    c4 = col.reshape(-1, 4)
    assert (c4[:, 3] > 0).sum() == k["colour_voxels"]
    assert np.array_equal(np.unique(c4[c4[:, 3] > 0], axis=0), np.array([k["rgba"]], np.uint16))


def test_survey_probe_decay_of_untouched_lanes(env):
    """Second pass with depth zeroed on even columns: those lanes keep their weight and their sdf
    becomes exactly sdf*w/(w+1e-4f) (0.00249998365 -> 0.00249996944)."""
    cam, C, ig = env
    k = KAT["second_pass"]
    depth, rgba, q, pose = synth.wall_frame(1.0, cam, hole_stride=0)
    sdf, w, col = O.fresh_chunk()
    O.voxel_update(depth, rgba, q, C, ig, pose, 1, KAT["first_pass"]["chunk"], RES5, sdf, w, col)
    s0, w0 = sdf.copy(), w.copy()
    d2 = depth.copy()
    d2[:, 0::2] = 0
    O.voxel_update(d2, rgba, q, C, ig, pose, 1, KAT["first_pass"]["chunk"], RES5, sdf, w, col)
    same_w = (w == w0) & (w0 > 0)
    sel = same_w & (s0 == np.float32(k["sdf_before"]))
    assert sel.any()
    assert np.all(sdf[sel] == np.float32(k["sdf_after"]))
    w32, s32 = w0[sel][0], s0[sel][0]
    assert np.float32(np.float32(s32 * w32) / np.float32(w32 + np.float32(1e-4))) == np.float32(k["sdf_after"])
    grown = (w > w0)
    assert grown.any() and np.all(w[grown] == np.float32(w0[grown] + np.float32(KAT["first_pass"]["weight"])))


def test_survey_probe_colour_cap_and_deintegration(env):
    cam, C, ig = env
    depth, rgba, q, pose = synth.wall_frame(1.0, cam, hole_stride=0, rgba_value=(200, 100, 50, 1))
    sdf, w, col = O.fresh_chunk()
    cid = KAT["first_pass"]["chunk"]
    for i in range(121):
        O.voxel_update(depth, rgba, None, C, ig, pose, 1, cid, RES5, sdf, w, col)
        assert col.reshape(-1, 4)[:, 3].max() <= 120
    c4 = col.reshape(-1, 4)
    assert c4[:, 3].max() == KAT["cap"]["count_after_121"]  # 121 -> all four channels >> 2 -> 30
    # one flag=0 pass subtracts exactly one weight / colour sample and reports the same positive q
    sdf, w, col = O.fresh_chunk()
    O.voxel_update(depth, rgba, q, C, ig, pose, 1, cid, RES5, sdf, w, col)
    O.voxel_update(depth, rgba, q, C, ig, pose, 1, cid, RES5, sdf, w, col)
    w2, c2 = w.copy(), col.copy()
    upd, qs, _ = O.voxel_update(depth, rgba, q, C, ig, pose, 0, cid, RES5, sdf, w, col)
    assert qs == np.float32(KAT["first_pass"]["q"])
    m = c2.reshape(-1, 4)[:, 3] == 2
    assert np.all(col.reshape(-1, 4)[m] == np.array([200, 100, 50, 1], np.uint16))
    assert np.allclose(w[w2 > 0], w2[w2 > 0] - np.float32(KAT["first_pass"]["weight"]), atol=1e-5)


def test_survey_probe_border_chunk_pos_stall(env):
    """Chunk (0,-12,24) at z = 1 m: a fully off-image row stalls `pos`; nothing is updated (A.1-4)."""
    cam, C, ig = env
    depth, rgba, q, pose = synth.wall_frame(1.0, cam, hole_stride=0)
    sdf, w, col = O.fresh_chunk()
    upd, qs, st = O.voxel_update(depth, rgba, q, C, ig, pose, 1, KAT["border_chunk"], RES5, sdf, w, col)
    assert not upd and st.rows_tsdf == 0
    assert np.all(sdf == 999.0) and np.all(w == 0)


def test_analytic_plane_inside_band(env):
    """sdf = d - z_voxel and w = n / (2 trunc) for voxels inside the band of a fronto-parallel plane."""
    cam, C, ig = env
    zwall = np.float32(1.0)
    depth, rgba, q, pose = synth.wall_frame(float(zwall), cam, hole_stride=0)
    cid = (2, -3, 24)
    sdf, w, col = O.fresh_chunk()
    n = 3
    for _ in range(n):
        O.voxel_update(depth, None, None, C, ig, pose, 1, cid, RES5, sdf, w, col)
    o = np.float32(8 * 24) * RES5
    tr = np.float32(O.truncation(ig, o))
    zc = np.array([np.float32(o + np.float32(np.float32(z) * RES5 + RES5 * np.float32(0.5))) for z in range(8)])
    for z in range(8):
        sl = slice(z * 64, (z + 1) * 64)
        sd = np.float32(zwall - zc[z])
        if -0.03 < sd < tr + np.float32(np.sqrt(3.0) * 0.005):
            assert np.allclose(sdf[sl], sd, atol=2e-6)
            assert np.allclose(w[sl], n / (2 * tr), rtol=1e-5)
        else:
            assert np.all(sdf[sl] == 999.0)


def test_reference_sizing_wall_1p5m(env):
    """wall at 1.5 m, 640x480, 5 mm -> 4606 chunks updated, 70656 colour rows."""
    cam, C, ig = env
    depth, rgba, q, pose = synth.wall_frame(1.5, cam)
    v = O.Volume(RES5, C, ig)
    nv, ns = v.integrate_frame(depth, rgba, pose)
    st = v.rowstats()
    assert nv == KAT["sizing_wall_1p5m"]["chunks_updated"]
    assert st.rows_color == KAT["sizing_wall_1p5m"]["rows_color"]
    assert abs(st.rows_tsdf - KAT["sizing_wall_1p5m"]["rows_tsdf_approx"]) < 200
    assert v.num_chunks() == nv  # new-but-untouched chunks were garbage-collected


def test_hole_free_wall_selects_nothing(env):
    """Only depth+0.2 points enter the AABB (ChunkManager.h:331): a hole-free fronto-parallel wall
    at identity pose selects zero chunks (SURVEY.md A.3 quirk i)."""
    cam, C, ig = env
    depth, rgba, q, pose = synth.wall_frame(1.5, cam, hole_stride=0)
    ids, nc = O.select(depth, C, ig, pose, RES5)
    assert len(ids) == 0
    mn, mx = O.bbox(depth, C, pose, RES5)
    assert mn[2] == mx[2] == 42


def test_selection_order_and_uniqueness(env):
    cam, C, ig = env
    depth, rgba, q, pose = synth.room_frame(3, cam)
    ids, nc = O.select(depth, C, ig, pose, RES5)
    assert len(ids) > 1000
    assert len(np.unique(ids, axis=0)) == len(ids)
    # x-outer .. z-inner over 4x4x4 blocks, then i,j,k inside a block
    mn, mx = O.bbox(depth, C, pose, RES5)
    blk = (ids - (mn - 1)) // 4
    key = (blk[:, 0].astype(np.int64) << 40) | (blk[:, 1].astype(np.int64) << 20) | blk[:, 2]
    assert np.all(np.diff(key) >= 0)


def test_atlas_patch_size_and_slots():
    for res, pw, ph in ((0.005, 24, 18), (0.01, 48, 36)):
        a = O.Atlas(np.float32(res), 13824, 64)  # short atlas: same arithmetic, little memory
        assert (a.pw, a.ph) == (pw, ph)
        a.close()
    a = O.Atlas(np.float32(0.005), 13824, 64)
    locs = []
    for i in range(580):
        rc, t = a.alloc()
        assert rc == 0
        locs.append(t)
    assert locs[0] == 0 and locs[1] == 24 and locs[575] == 13800   # 576 slots per band
    assert locs[576] == 18 * 13824                                   # wrap to the next band of PH rows
    a.close()


def test_atlas_overflow_returns_minus_one():
    a = O.Atlas(np.float32(0.005), 96, 36)  # 4 slots per band (x = 0,24,48,72; the >= test wraps after 72), 2 bands
    n_ok = 0
    while True:
        rc, t = a.alloc()
        if rc != 0:
            break
        n_ok += 1
        assert n_ok < 100
    assert n_ok == 8
    a.close()


def test_finalize_dirty_set_and_garbage(env):
    cam, C, ig = env
    v = O.Volume(RES5, C, ig)
    ids = np.array([[0, 0, 0], [5, 5, 5], [9, 9, 9]], np.int32)
    for cid in ids:
        v.set_chunk(cid, *O.fresh_chunk())
    needs = np.array([1, 0, 0], np.uint8)
    new = np.array([0, 1, 0], np.uint8)
    valid = v.finalize(ids, needs, new)
    assert np.array_equal(valid, ids[:1])
    d = {tuple(x) for x in v.dirty()}
    assert d == {(0, 0, 0), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)}
    assert not v.has_chunk(ids[1]) and v.has_chunk(ids[2]) and v.has_chunk(ids[0])


def test_avx2_baseline_kernel_is_bit_identical_to_the_scalar_restatement(env):
    """The CPU-baseline kernel (AVX2 rows, like the reference's own kernel) must not drift from the checker."""
    cam, C, ig = env
    if not O.lib().tfo_have_avx2():
        pytest.skip("no AVX2 on this host")
    a = O.Volume(RES5, C, ig)
    b = O.Volume(RES5, C, ig)
    b.set_kernel(1)
    poses = [synth.pose_identity(), synth.pose_euler(0.5, 0.2, -0.1, (0.1, 0.0, 0.1)), synth.pose_euler(-0.9, -0.3, 0.4)]
    for k, pose in enumerate(poses):
        depth, rgba, q, _ = synth.room_frame(3 * k, cam)
        ids, new = a.prepare(depth, pose)
        ids_b, new_b = b.prepare(depth, pose)
        assert np.array_equal(ids, ids_b)
        na, nb = np.zeros(len(ids), np.uint8), np.zeros(len(ids), np.uint8)
        qa = a.integrate(depth, rgba, q, pose, ids, na, 1, k)
        qb = b.integrate(depth, rgba, q, pose, ids, nb, 1, k)
        assert np.array_equal(na, nb) and np.array_equal(qa.view(np.uint32), qb.view(np.uint32))
        qa = a.integrate(depth, None, None, pose, ids, na, 1, -1)
        qb = b.integrate(depth, None, None, pose, ids, nb, 1, -1)
        a.finalize(ids, na, new)
        b.finalize(ids, nb, new)
        if k == 1:  # de-integration path
            va = np.ones(len(ids), np.uint8)
            keep = np.array([a.has_chunk(c) for c in ids])
            a.integrate(depth, rgba, q, pose, ids[keep], va[keep].copy(), 0, k)
            b.integrate(depth, rgba, q, pose, ids[keep], va[keep].copy(), 0, k)
    ids = a.list_chunks()
    assert np.array_equal(ids, b.list_chunks())
    for cid in ids[:: max(1, len(ids) // 400)]:
        sa, wa, ca = a.get_chunk(cid)
        sb, wb, cb = b.get_chunk(cid)
        assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32))
        assert np.array_equal(wa.view(np.uint32), wb.view(np.uint32))
        assert np.array_equal(ca, cb)
    sa, sb = a.rowstats(), b.rowstats()
    assert (sa.rows_tsdf, sa.rows_color, sa.chunks_updated) == (sb.rows_tsdf, sb.rows_color, sb.chunks_updated)
