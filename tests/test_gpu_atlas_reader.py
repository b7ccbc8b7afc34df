"""A second thread reads the atlas while the map thread streams textured frames -- the reference's GUI thread reads
`atlas.texture_buffer` while the map thread writes it (GCFusion/MobileFusion.h:404-421; SURVEY.md s.8(b) "Threading").
`tf_atlas_snapshot_rows` is the entry point for that thread: it touches no pipeline state, and what it returns is the
atlas of ONE moment of the handle's stream, labelled with the frame whose patches were the newest on the stream.  Every
snapshot must therefore equal the oracle's atlas after exactly that frame, whatever the map thread did meanwhile."""
import hashlib
import threading
import time

import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import capi, synth

pytestmark = pytest.mark.gpu

AW = 13824


def test_reader_thread_gets_consistent_rows_while_the_map_thread_streams(gpu_required):
    cam = synth.Camera()
    res = np.float32(0.005)
    n_frames = 50
    frames = []
    for k in range(n_frames):  # a wall seen from a camera that drifts: patches are re-projected and re-blitted every frame
        depth, rgba, _, pose = synth.wall_frame(1.2 + 0.004 * k, cam, seed=k)
        frames.append((depth, rgba, pose, synth.pose_inverse16(pose)))
    # the oracle's atlas after every frame, as a digest of the rows the stream's slots occupy (fifteen bands of 18 rows hold
    # 8640 slots; a digest per moment instead of 11 MB)
    rows = 270

    def digest(a):
        return hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=16).digest()

    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    oa = O.Atlas(res)
    states = []
    for k, (d, c, P, T) in enumerate(frames):
        ov.frame_textured(oa, d, c, P, T, k)
        states.append(digest(oa.buffer()[:rows]))
    assert int(oa.loc_next()) // AW + 18 <= rows, "the stream needs more atlas rows than the test compares"
    assert len(set(states)) > n_frames // 2, "the atlas should change with (nearly) every frame"

    gv = capi.Volume(res, cam, max_chunks=1 << 15)
    samples, errors, stop = [], [], threading.Event()

    def reader():
        try:
            while not stop.is_set():
                got, seq, fid = gv.atlas_snapshot_rows(0, rows, AW)
                samples.append((seq, fid, digest(got), bool(got.any())))
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    th = threading.Thread(target=reader)
    th.start()
    try:
        for k, (d, c, P, T) in enumerate(frames):
            gv.integrate_frame_host(d, c, P, T, k)
            if k % 7 == 3:
                gv.sync()  # (the map thread's own synchronising calls run next to the reader as well)
            time.sleep(0.001)  # (50 frames take 5 ms otherwise: the reader should see many different moments)
        gv.sync()
    finally:
        stop.set()
        th.join()
    assert not errors, errors
    # one more, from the map thread's side of things: everything is on the stream and done
    assert digest(gv.atlas_rows(0, rows, AW)) == states[-1]
    final, seq_end, fid_end = gv.atlas_snapshot_rows(0, rows, AW)
    assert fid_end == n_frames - 1 and digest(final) == states[-1]
    seen = set()
    last_seq = -1
    for seq, fid, got, nonzero in samples:
        assert seq >= last_seq, "write sequence went backwards"
        last_seq = seq
        if fid < 0:
            assert not nonzero, "texels before any patch stage"
            continue
        assert got == states[fid], "snapshot labelled frame %d differs from the oracle's atlas after that frame" % fid
        seen.add(fid)
    assert len(samples) >= 5 and len(seen) >= 3, (len(samples), sorted(seen))
    gv.close()
