"""BASELINE.json configs[0] plumbing: the S-room stream written in the layout DatasetWrapper reads
(Tools/DatasetWrapper.hpp:55-263), read back, and pushed through the path at 0.01 m voxels."""
import os

import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import dataset, synth

RES10 = np.float32(0.01)


def _write(tmp_path, n=4, step=5):
    cam = synth.Camera()
    frames = [synth.room_frame(step * k, cam, with_quality=False) for k in range(n)]
    dataset.write_sequence(str(tmp_path), frames, cam, depth_scale=5000.0, maximum_depth=3.0)
    return cam, frames


def test_round_trip_and_preprocess_rules(tmp_path):
    cam, frames = _write(tmp_path)
    seq = dataset.Sequence(str(tmp_path))
    assert len(seq) == 4 and (seq.width, seq.height) == (640, 480) and seq.depth_scale == 5000.0
    assert (seq.fx, seq.fy, seq.cx, seq.cy) == (525.0, 525.0, 319.5, 239.5)
    for k, f in enumerate(frames):
        d, rgba, w, pose = seq.load_frame(k)
        assert d.dtype == np.float32 and rgba.shape == (480, 640, 4) and (rgba[..., 3] == 1).all() and not w.any()
        assert np.array_equal(rgba[..., :3], f[1][..., :3])
        q = np.rint(f[0].astype(np.float64) * 5000.0)
        want = np.where(q > 3.0 * 5000.0, 0.0, q).astype(np.float32) / np.float32(5000.0)
        assert np.array_equal(d, want)            # 16-bit quantisation, then depth above maximum_depth -> 0
        assert (d[f[0] == 0] == 0).all()          # holes stay holes
        assert np.abs(pose - f[3]).max() < 1e-6   # quaternion round trip of the pose
    assert abs(seq.time_stamp[1] - 1 / 30) < 1e-6


def test_png_codec_against_pil(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.Generator(np.random.PCG64(3))
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    d16 = rng.integers(0, 65536, (37, 53), dtype=np.uint16)
    dataset.write_png(str(tmp_path / "a.png"), rgb)
    dataset.write_png(str(tmp_path / "b.png"), d16)
    assert np.array_equal(np.asarray(PIL.open(str(tmp_path / "a.png"))), rgb)
    assert np.array_equal(np.asarray(PIL.open(str(tmp_path / "b.png"))).astype(np.uint16), d16)
    # files written by another encoder (adaptive filters, all five types) decode the same
    PIL.fromarray(rgb).save(str(tmp_path / "c.png"), optimize=True)
    PIL.fromarray(d16).save(str(tmp_path / "d.png"))
    smooth = (np.add.outer(np.arange(64), np.arange(80)) % 256).astype(np.uint8)
    PIL.fromarray(np.stack([smooth, smooth.T[:64, :80] if False else smooth, 255 - smooth], -1)).save(str(tmp_path / "e.png"))
    assert np.array_equal(dataset.read_png(str(tmp_path / "c.png")), rgb)
    assert np.array_equal(dataset.read_png(str(tmp_path / "d.png")), d16)
    assert np.array_equal(dataset.read_png(str(tmp_path / "e.png")), np.asarray(PIL.open(str(tmp_path / "e.png"))))


def test_sequence_through_the_oracle_at_10mm(tmp_path):
    """configs[0] proper: the CPU-runnable case -- the sequence integrates and meshes on the CPU restatement."""
    _write(tmp_path, n=14, step=1)  # a wall at 2 m gains ~5 weight per frame; meshes need more than 50
    seq = dataset.Sequence(str(tmp_path))
    ov = O.Volume(RES10, O.camera_from(seq.camera()), O.default_integrator())
    for k in range(len(seq)):
        d, rgba, _, pose = seq.load_frame(k)
        ov.integrate_frame(d, rgba, pose)
    assert ov.num_chunks() > 1500
    ov.update_meshes()
    assert len(ov.list_meshes()) > 300


@pytest.mark.gpu
def test_sequence_gpu_equals_oracle_at_10mm(gpu_required, tmp_path):
    from texturefusion_amd import capi
    from tests.util import assert_chunks_equal, sorted_ids
    _write(tmp_path, n=14, step=1)
    seq = dataset.Sequence(str(tmp_path))
    cam = seq.camera()
    ov = O.Volume(RES10, O.camera_from(cam), O.default_integrator())
    gv = capi.Volume(RES10, cam, max_chunks=1 << 16)
    for k in range(len(seq)):
        d, rgba, _, pose = seq.load_frame(k)
        ov.integrate_frame(d, rgba, pose)
        gv.integrate_frame_host(d, rgba, pose.reshape(12), None, k)
    gv.sync()
    ids = sorted_ids(gv.list_chunks())
    assert np.array_equal(ids, sorted_ids(ov.list_chunks()))
    assert_chunks_equal(ov, gv, ids, "offline sequence")
    ov.update_meshes()
    gv.update_meshes()
    assert np.array_equal(sorted_ids(gv.list_meshes()), sorted_ids(ov.list_meshes())) and len(ov.list_meshes()) > 300
    gv.close()
