"""The neighbour table (VolumeDev::nbr) under the situations that can make it lie: chunks created right next
to chunks whose rows already hold a trusted "no chunk there" (the reference resolves the same neighbours by id on every
call -- ChunkManager.cpp:618-632 for the mesher's corner voxels, Chisel.cpp:134-137 for CompressMeshes' exchange -- so any
stale word shows up as a different mesh or adjacency flag), tf_volume_reset, and the call-by-call flow that brings no pool
slots with its dirty list.  Results are compared with the oracle in full; tf_check_neighbours compares the table with the hash."""
import ctypes as C

import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import capi, synth
from tests.util import assert_chunks_equal, sorted_ids
from tests.test_gpu_atlas import _compare_patches

pytestmark = pytest.mark.gpu


def _compare_all(ov, gv, what):
    oids = sorted_ids(ov.list_chunks())
    assert np.array_equal(oids, sorted_ids(gv.list_chunks())), what
    assert_chunks_equal(ov, gv, oids, what)
    mids = sorted_ids(ov.list_meshes())
    assert np.array_equal(mids, sorted_ids(gv.list_meshes())), what
    voff, ioff, V, N, Cc, I, adj, simp = gv.get_meshes(mids)
    for i, cid in enumerate(mids):
        m = ov.get_mesh(cid)
        assert np.array_equal(V[voff[i]:voff[i + 1]].view(np.uint32), m["verts"].view(np.uint32)), (what, cid)
        assert np.array_equal(I[ioff[i]:ioff[i + 1]], m["indices"]), (what, cid)
        assert bool(simp[i]) == m["simplified"] and np.array_equal(adj[i], m["adj"]), (what, cid)
    _compare_patches(ov, gv, mids, what)
    return len(mids)


def test_chunks_created_next_to_trusted_absences(gpu_required):
    """A wall is fused and textured until every dirty chunk's row is checked and trusted (steady frames insert no key); then
    the wall moves towards the camera by one chunk, three times: whole layers of NEW chunks appear as -z neighbours of chunks
    whose rows said "nothing there".  Every mesh, adjacency flag, patch and voxel must still equal the oracle's."""
    cam = synth.Camera()
    res = np.float32(0.005)
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    oa = O.Atlas(res)
    # (walls fused 4 cm apart leave chunks with two conflicting surfaces: meshes of a thousand vertices -- full-size blocks)
    gv = capi.Volume(res, cam, max_chunks=1 << 14, mesh_max_vertices=2240, mesh_max_triangles=2560)
    k = 0
    trusted_before = None
    for step, z in enumerate((1.30, 1.30, 1.26, 1.22, 1.18)):
        for rep in range(4 if step < 2 else 3):
            depth, rgba, _, pose = synth.wall_frame(z, cam, seed=k)
            T = synth.pose_inverse16(pose)
            gv.integrate_frame_host(depth, rgba, pose, T, k)
            ov.frame_textured(oa, depth, rgba, pose, T, k)
            k += 1
        gv.sync()
        c = gv.check_neighbours()
        assert c[0] > 0 and c[1] > 0 and c[2] == 0 and c[4] == 0, c.tolist()
        if step == 1:
            trusted_before = int(c[3])
            assert trusted_before > 500, "steady frames should leave the dirty chunks' rows trusted: %s" % c.tolist()
        if step == 2:  # keys were inserted: rows checked before that are void until a filter launch checks them again
            assert int(c[3]) < trusted_before + 5000
        n = _compare_all(ov, gv, "wall at %.2f" % z)
    assert n > 800
    gv.close()


def test_table_is_emptied_by_reset_and_filled_again(gpu_required):
    cam = synth.Camera()
    res = np.float32(0.005)
    gv = capi.Volume(res, cam, max_chunks=1 << 14)
    for round_ in range(2):
        ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
        oa = O.Atlas(res)
        for k in range(5):
            depth, rgba, _, pose = synth.wall_frame(1.1 + 0.3 * round_, cam, seed=10 * round_ + k)
            T = synth.pose_inverse16(pose)
            gv.integrate_frame_host(depth, rgba, pose, T, k)
            ov.frame_textured(oa, depth, rgba, pose, T, k)
        gv.sync()
        assert _compare_all(ov, gv, "round %d" % round_) > 300
        c = gv.check_neighbours()
        assert c[1] > 0 and c[2] == 0 and c[4] == 0, c.tolist()
        gv.reset()
        c = gv.check_neighbours()
        assert c[0] == 0 and c[1] == 0, "tf_volume_reset must forget every row: %s" % c.tolist()
    gv.close()


def test_call_by_call_flow_fills_the_table_too(gpu_required):
    """tf_update_meshes hands the filter a dirty list WITHOUT pool slots (the reference's own call sequence): the chunk's slot
    comes from one hash lookup, its neighbours from the table."""
    cam = synth.Camera()
    res = np.float32(0.005)
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(res, cam, max_chunks=1 << 14)
    for k in range(4):
        depth, rgba, _, pose = synth.wall_frame(1.2 - 0.04 * (k // 2), cam, seed=k)
        gv.frame_upload(depth, rgba, None)
        gv.integrate_frame(pose, True)
        ov.integrate_frame(depth, rgba, pose)
        gv.update_meshes()
        ov.update_meshes()
        gv.compress_meshes()
        ov.compress_meshes()
    mids = sorted_ids(ov.list_meshes())
    assert len(mids) > 300 and np.array_equal(mids, sorted_ids(gv.list_meshes()))
    voff, ioff, V, N, Cc, I, adj, simp = gv.get_meshes(mids)
    for i, cid in enumerate(mids):
        m = ov.get_mesh(cid)
        assert np.array_equal(V[voff[i]:voff[i + 1]].view(np.uint32), m["verts"].view(np.uint32)), cid
        assert np.array_equal(adj[i], m["adj"]), cid
    c = gv.check_neighbours()
    assert c[1] > 0 and c[2] == 0 and c[4] == 0, c.tolist()
    gv.close()


def test_create_with_an_older_shorter_config(gpu_required):
    """tf_volume_create_sized: a caller built against a tf_config that ended before mesh_blocks hands over fewer bytes; the
    library takes the defaults for what it does not get (every pool slot can own a mesh)."""
    L = capi.lib()
    L.tf_volume_create_sized.argtypes = [C.POINTER(C.c_int32), C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    cfg = capi.Config(0, 1 << 12, 1 << 12, 1 << 16, 0, 0, 0, 0, 0, 0, 12345678)  # (the trailing field must NOT be read)
    short = capi.Config.mesh_blocks.offset
    assert 0 < short < C.sizeof(capi.Config)
    h = C.c_void_p()
    dims = (C.c_int32 * 3)(8, 8, 8)
    rc = L.tf_volume_create_sized(dims, C.c_float(0.005), 1, C.byref(cfg), short, C.byref(h))
    assert rc == 0, L.tf_last_error()
    st = capi.Stats()
    assert L.tf_get_stats(h, C.byref(st)) == 0
    assert L.tf_volume_destroy(h) == 0


def test_registering_views_inside_a_locked_arena(gpu_required):
    """tf_host_register keeps ONE process-wide registry of page-locked ranges: a second handle that registers views inside an
    arena another handle locked shares that range (hipHostRegister on locked pages would fail), and an overlap that is not
    contained is refused with a message that says so."""
    cam = synth.Camera()
    a, b = capi.Volume(np.float32(0.005), cam, max_chunks=1 << 12), capi.Volume(np.float32(0.005), cam, max_chunks=1 << 12)
    arena = np.zeros(8 << 20, np.uint8)
    a.host_register(arena)
    depth_view, rgba_view = arena[4096:4096 + 640 * 480 * 4], arena[2 << 20:(2 << 20) + 640 * 480 * 4]
    b.host_register(depth_view)
    b.host_register(rgba_view)
    a.host_unregister(arena)          # (b still holds the range: the pages stay locked)
    b.host_unregister(rgba_view)
    b.host_unregister(depth_view)     # last reference: unlocked
    c = np.zeros(1 << 20, np.uint8)
    a.host_register(c[:1 << 19])
    with pytest.raises(capi.TFError) as e:
        b.host_register(c[1 << 18:])  # overlaps, not contained
    assert "overlap" in str(e.value)
    a.host_unregister(c[:1 << 19])
    a.close(); b.close()
