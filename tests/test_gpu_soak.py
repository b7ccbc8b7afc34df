"""Randomised parity soak: odd image sizes, three voxel sizes, near / far planes, arbitrary rigid poses,
integration, de-integration, with and without colour / quality, call-by-call and fused flows -- all
bit-exact against the oracle.  Seeds are fixed; every case finishes in a second or two."""
import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import capi, synth
from tests.util import assert_chunks_equal, sorted_ids

pytestmark = pytest.mark.gpu


def _rand_pose(rng, spread):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    t = rng.uniform(-spread, spread, size=3)
    return np.concatenate([R, t.reshape(3, 1)], 1).astype(np.float32)


CASES = [
    # (seed, W, H, f, res, near, far)
    (1, 640, 480, 525.0, 0.005, 0.01, 5.0),
    (2, 328, 242, 260.0, 0.01, 0.01, 5.0),     # odd height, width a multiple of 8 only (the ABI's constraint)
    (3, 168, 121, 140.0, 0.02, 0.3, 2.5),      # tight near / far: depth-validity predicate bites
    (4, 400, 300, 300.7, 0.01, 0.01, 4.0),     # non-integer intrinsics are truncated (PinholeCamera.h:46-49)
    (5, 640, 480, 525.0, 0.008, 0.5, 1.8),
]


@pytest.mark.parametrize("seed,W,H,f,res,near,far", CASES)
def test_random_frames_both_flows(gpu_required, seed, W, H, f, res, near, far):
    rng = np.random.default_rng(1000 + seed)
    cam = synth.Camera(W, H, f, f * 1.01, W / 2 - 0.5 + 0.3, H / 2 - 0.5 - 0.2, near, far)
    res = np.float32(res)
    ig = O.default_integrator()
    ov = O.Volume(res, O.camera_from(cam), ig)
    gv = capi.Volume(res, cam, max_chunks=1 << 16)
    ov2 = O.Volume(res, O.camera_from(cam), ig)
    gv2 = capi.Volume(res, cam, max_chunks=1 << 16)
    kept = []
    for it in range(4):
        depth, rgba, quality, pose0 = synth.room_frame(int(rng.integers(0, 200)), cam)
        pose = pose0 if it % 2 == 0 else _rand_pose(rng, 0.4)
        use_q = bool(it & 1)
        use_c = it != 2
        # ---- call-by-call flow
        oids, onew = ov.prepare(depth, pose)
        gv.frame_upload(depth, rgba if use_c else None, quality if use_q else None)
        gids, gnew = gv.prepare(pose)
        assert np.array_equal(oids, gids) and np.array_equal(onew, gnew)
        on, gn = np.zeros(len(oids), np.uint8), np.zeros(len(oids), np.uint8)
        oq = ov.integrate(depth, rgba if use_c else None, quality if (use_q and use_c) else None, pose, oids, on, 1, it)
        gq = gv.integrate(pose, gids, gn, 1, use_c, use_q and use_c)
        assert np.array_equal(on, gn)
        if use_c:
            assert np.array_equal(oq.view(np.uint32), gq.view(np.uint32))
        assert_chunks_equal(ov, gv, oids, "case %d frame %d" % (seed, it))
        assert np.array_equal(ov.finalize(oids, on, onew), gv.finalize(gids, gn, gnew))
        kept.append((depth, rgba, quality, pose, oids, use_c))
        # ---- fused flow on a second pair of volumes
        ov2.integrate_frame(depth, rgba, pose)
        gv2.frame_upload(depth, rgba, None)
        gv2.integrate_frame(pose, True)
    gv2.sync()
    ids2 = ov2.list_chunks()
    assert np.array_equal(sorted_ids(ids2), sorted_ids(gv2.list_chunks()))
    assert_chunks_equal(ov2, gv2, ids2, "case %d fused" % seed)
    assert np.array_equal(sorted_ids(ov2.dirty()), sorted_ids(gv2.dirty()))
    # ---- de-integration of the second frame on its own list (ReIntegrateKeyframe, flag 0)
    depth, rgba, quality, pose, oids, use_c = kept[1]
    alive = np.array([ov.has_chunk(c) for c in oids], bool)
    ids = oids[alive]
    if len(ids):
        gv.frame_upload(depth, rgba if use_c else None, None)
        on, gn = np.zeros(len(ids), np.uint8), np.zeros(len(ids), np.uint8)
        ov.integrate(depth, rgba if use_c else None, None, pose, ids, on, 0, -1)
        gv.integrate(pose, ids, gn, 0, use_c, False)
        assert np.array_equal(on, gn)
        assert_chunks_equal(ov, gv, ids, "case %d de-integrate" % seed)
    for v in (gv, gv2):
        v.close()
