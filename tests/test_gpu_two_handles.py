"""Two handles of one process, each driven by a thread of its own (the library keeps per-process state: the page-lock
registry, the keyframe units' table, k_scan's launch stamps, lazily created copy pools): one streams textured host frames,
the other runs the reference's call-by-call sequence and keyframe-unit calls on a different scene at the same time.  Each
volume must equal the oracle's, bit for bit, as if it had the GPU to itself."""
import threading

import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import capi, synth
from tests.util import assert_chunks_equal, sorted_ids, HipBuffer
from tests.test_gpu_neighbours import _compare_all

pytestmark = pytest.mark.gpu


def test_two_threads_two_volumes(gpu_required):
    cam = synth.Camera(480, 360, 393.75, 393.75, 239.5, 179.5, 0.01, 5.0)
    res = np.float32(0.006)
    fa = [synth.room_frame(3 * k, cam, with_quality=False, wobble=0.04) for k in range(30)]
    fb = [synth.room_frame(100 + 2 * k, cam, with_quality=False, wobble=0.02, radius=0.6) for k in range(28)]
    ga = capi.Volume(res, cam, max_chunks=1 << 17)
    gb = capi.Volume(res, cam, max_chunks=1 << 17)
    errors = []

    def stream_a():
        try:
            for k, f in enumerate(fa):
                ga.integrate_frame_host(f[0], f[1], f[3].reshape(12), synth.pose_inverse16(f[3]), k)
            ga.sync()
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    bufs = [(HipBuffer(f[0].nbytes).from_host(f[0]), HipBuffer(f[1].nbytes).from_host(f[1])) for f in fb]

    def calls_b():
        try:
            for g in range(4):  # keyframe-unit calls: 1 colour + 6 depth frames each, textured
                k0 = 7 * g
                grp = capi.Volume.unit_group(500 + g, (bufs[k0][0].ptr, bufs[k0][1].ptr, None, fb[k0][3]),
                                             [(bufs[k0 + 1 + i][0].ptr, fb[k0 + 1 + i][3]) for i in range(6)])
                gb.keyframe_unit(fresh=grp, moved=[], texture=True, pose_inv16=synth.pose_inverse16(fb[k0][3]))
                if g % 2:
                    gb.sync()
            gb.sync()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ta, tb = threading.Thread(target=stream_a), threading.Thread(target=calls_b)
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errors, errors

    oa_v = O.Volume(res, O.camera_from(cam), O.default_integrator()); oa_a = O.Atlas(res)
    for k, f in enumerate(fa):
        oa_v.frame_textured(oa_a, f[0], f[1], f[3], synth.pose_inverse16(f[3]), k)
    assert _compare_all(oa_v, ga, "thread A: textured host frames") > 500

    ob_v = O.Volume(res, O.camera_from(cam), O.default_integrator()); ob_a = O.Atlas(res)
    kfs = {}
    for g in range(4):
        k0 = 7 * g
        ids, new = ob_v.prepare(fb[k0][0], fb[k0][3])
        needs = np.zeros(len(ids), np.uint8)
        ob_v.integrate(fb[k0][0], fb[k0][1], None, fb[k0][3], ids, needs, 1, 500 + g)
        for i in range(6):
            ob_v.integrate(fb[k0 + 1 + i][0], None, None, fb[k0 + 1 + i][3], ids, needs, 1, -1)
        ob_v.finalize(ids, needs, new)
        ob_v.update_meshes()
        upd = ob_v.compress_meshes()
        T = synth.pose_inverse16(fb[k0][3])
        kfs[500 + g] = (np.ascontiguousarray(fb[k0][1][..., :3]), fb[k0][0], T)
        ob_v.generate_patches(ob_a, upd, np.full(len(upd), 500 + g, np.int32), kfs)
        ob_v.update_atlas(ob_a, upd)
    oids = sorted_ids(ob_v.list_chunks())
    assert np.array_equal(oids, sorted_ids(gb.list_chunks())) and len(oids) > 500
    assert_chunks_equal(ob_v, gb, oids[::3], "thread B: keyframe units")
    mids = sorted_ids(ob_v.list_meshes())
    assert np.array_equal(mids, sorted_ids(gb.list_meshes()))
    voff, ioff, V, N, Cc, I, adj, simp = gb.get_meshes(mids)
    for i, cid in enumerate(mids):
        m = ob_v.get_mesh(cid)
        assert np.array_equal(V[voff[i]:voff[i + 1]].view(np.uint32), m["verts"].view(np.uint32)), cid
        assert np.array_equal(I[ioff[i]:ioff[i + 1]], m["indices"]), cid
    assert gb.atlas_loc_next() == ob_a.loc_next()
    for b in bufs:
        b[0].free(); b[1].free()
    ga.close(); gb.close()
