"""tf_integrate_depth_group: the local frames of a keyframe group (GCFusion/MobileFusion.cpp:187-203 -- depth-only
frames integrated over the keyframe's chunk list, each with its own pose) in one visit per chunk, against the oracle
running them one after the other, bit for bit: voxels, needsUpdate flags, validChunks; integrate and de-integrate."""
import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import capi, synth
from tests.util import RES5, HipBuffer, assert_chunks_equal, make_pair, sorted_ids

pytestmark = pytest.mark.gpu


def _group(cam, k0, n_local, wobble):
    kf = synth.room_frame(k0, cam, wobble=wobble)
    local = [synth.room_frame(k0 + 1 + i, cam, with_quality=False, wobble=wobble) for i in range(n_local)]
    return kf, local


@pytest.mark.parametrize("n_local,wobble", [(6, 0.0), (3, 0.06), (1, 0.02)])
def test_group_equals_sequential_oracle(gpu_required, n_local, wobble):
    cam = synth.Camera()
    ov, gv, cam, ig = make_pair(RES5, cam, max_chunks=1 << 16)
    bufs = []
    for rep, k0 in enumerate((10, 13)):  # two keyframe groups, the second over partly the same chunks
        kf, local = _group(cam, k0, n_local, wobble)
        depth, rgba, quality, pose = kf
        oids, onew = ov.prepare(depth, pose)
        gv.frame_upload(depth, rgba, quality)
        gids, gnew = gv.prepare(pose)
        assert np.array_equal(oids, gids) and np.array_equal(onew, gnew)
        on, gn = np.zeros(len(oids), np.uint8), np.zeros(len(oids), np.uint8)
        oq = ov.integrate(depth, rgba, quality, pose, oids, on, 1, 40 + rep)
        gq = gv.integrate(pose, gids, gn, 1, True, True)
        assert np.array_equal(on, gn) and np.array_equal(oq.view(np.uint32), gq.view(np.uint32))
        for f in local:  # the oracle: one IntegrateDepthScanColor per local frame (MobileFusion.cpp:200-202)
            ov.integrate(f[0], None, None, f[3], oids, on, 1, -1)
        db = [HipBuffer(f[0].nbytes).from_host(f[0]) for f in local]
        bufs += db
        if rep == 0:
            gv.integrate_depth_group([b.ptr for b in db], np.stack([f[3].reshape(12) for f in local]), gids, gn, 1)
        else:  # the host-image entry point
            gv.integrate_depth_group_host([f[0] for f in local], np.stack([f[3].reshape(12) for f in local]), gids, gn, 1)
        assert np.array_equal(on, gn), "needsUpdate flags after the group"
        ovalid = ov.finalize(oids, on, onew)
        gvalid = gv.finalize(gids, gn, gnew)
        assert np.array_equal(ovalid, gvalid)
        assert_chunks_equal(ov, gv, ovalid[::5], "keyframe group %d" % rep)
    ids = sorted_ids(ov.list_chunks())
    assert np.array_equal(ids, sorted_ids(gv.list_chunks()))
    assert_chunks_equal(ov, gv, ids[::3], "after both groups")
    # ---- retract the last group again (ReIntegrateKeyframe with integrateFlag = 0 runs the same loop)
    dn_o, dn_g = np.ones(len(ovalid), np.uint8), np.ones(len(ovalid), np.uint8)
    for f in local:
        ov.integrate(f[0], None, None, f[3], ovalid, dn_o, 0, -1)
    gv.integrate_depth_group([b.ptr for b in db], np.stack([f[3].reshape(12) for f in local]), gvalid, dn_g, 0)
    assert np.array_equal(dn_o, dn_g)
    assert_chunks_equal(ov, gv, ovalid[::4], "after de-integration")
    for b in bufs:
        b.free()
    gv.close()


def test_group_argument_checks(gpu_required):
    cam = synth.Camera()
    gv = capi.Volume(RES5, cam, max_chunks=1 << 12, max_list=1 << 12, max_coarse=1 << 14)
    ids = np.zeros((1, 3), np.int32)
    needs = np.zeros(1, np.uint8)
    with pytest.raises(capi.TFError):
        gv.integrate_depth_group([1] * 7, np.zeros((7, 12), np.float32), ids, needs)
    gv.close()
