"""The mesh store hands a chunk's block back when its mesh comes out empty (mesh_block_for / blk_release, tf_devfn.h): the
reference's Mesh::Clear() frees the vectors of a mesh that lost its surface while the mesh stays in allMeshes
(ChunkManager.cpp:254-262), so a surface that drifts through the volume holds storage for the chunks it crosses NOW, not
for every chunk it ever crossed.  A wall drifts away from the camera one centimetre at a time: 2884 chunks get a mesh over
the stream, at most ~1400 hold vertices at any moment.  Blocks released in a launch serve the NEXT one (a layer that empties and the layer that fills do so in the same mesher launch), so
the stream needs about two layers of blocks: a store of 2000 must carry it, bit-exact -- one block per mesh ever made would
need 2884."""
import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import capi, synth
from tests.test_gpu_neighbours import _compare_all

pytestmark = pytest.mark.gpu


def _drift(gv, ov, oa, steps, reps=3):
    k = 0
    for step in range(steps):
        for _ in range(reps):
            depth, rgba, _, pose = synth.wall_frame(1.0 + 0.01 * step, synth.Camera(), seed=k)
            T = synth.pose_inverse16(pose)
            gv.integrate_frame_host(depth, rgba, pose, T, k)
            if ov is not None:
                ov.frame_textured(oa, depth, rgba, pose, T, k)
            k += 1
    gv.sync()


def test_a_drifting_surface_reuses_the_blocks_it_left(gpu_required):
    cam = synth.Camera()
    res = np.float32(0.005)
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    oa = O.Atlas(res)
    gv = capi.Volume(res, cam, max_chunks=1 << 14, mesh_blocks=2000)
    _drift(gv, ov, oa, 14)
    n = _compare_all(ov, gv, "drifting wall")
    assert n > 2500, n                    # meshes in allMeshes: every chunk the wall ever crossed
    holding = sum(1 for cid in ov.list_meshes() if len(ov.get_mesh(cid)["verts"]))
    assert holding < 1000, holding        # ... of which these hold vertices at the end
    gv.close()


def test_the_large_pool_recycles_too(gpu_required):
    """Small blocks of 64 vertices / 64 triangles: a planar wall's meshes (81 / 128) all live in the LARGE pool.  2000 large
    blocks carry the drifting wall's 2884 meshes the same way."""
    cam = synth.Camera()
    res = np.float32(0.005)
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    oa = O.Atlas(res)
    gv = capi.Volume(res, cam, max_chunks=1 << 14, mesh_max_vertices=64, mesh_max_triangles=64, mesh_overflow_blocks=2000)
    _drift(gv, ov, oa, 14)
    assert _compare_all(ov, gv, "drifting wall, large pool") > 2500
    gv.close()


def test_the_store_still_reports_exhaustion(gpu_required):
    """Fewer blocks than meshes that hold vertices at ONE moment: TF_ERR_CAPACITY with the message that names the knob."""
    cam = synth.Camera()
    gv = capi.Volume(np.float32(0.005), cam, max_chunks=1 << 14, mesh_blocks=400)
    with pytest.raises(capi.TFError) as e:
        _drift(gv, None, None, 2)
    assert "mesh_blocks" in str(e.value)
    gv.close()
