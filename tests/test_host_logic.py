"""Host-side logic that needs no GPU: synthetic streams, partition planner, host constants."""
import os

import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import partition as part
from texturefusion_amd import synth

RES5 = np.float32(0.005)


def test_room_frames_are_deterministic_and_have_holes():
    a = synth.room_frame(7)
    b = synth.room_frame(7)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    depth, rgba, q, pose = a
    assert depth.dtype == np.float32 and depth.shape == (480, 640)
    assert rgba.shape == (480, 640, 4) and np.all(rgba[..., 3] == 1)
    holes = (depth == 0).mean()
    assert 0.01 < holes < 0.03
    assert 0.7 < depth[depth > 0].min() and depth.max() < 3.5
    assert q.min() >= 0 and q.max() < 1


def test_pose_inverse():
    p = synth.pose_euler(0.3, -0.2, 0.1, (0.5, -0.25, 1.0))
    T = synth.pose_inverse16(p).reshape(4, 4).astype(np.float64)
    M = np.eye(4)
    M[:3] = p
    assert np.allclose(T @ M, np.eye(4), atol=1e-6)


def test_slabs_cover_every_id_exactly_once():
    ext = part.room_extent_chunks(RES5)
    for world in (1, 2, 4, 8):
        e = part.slab_bounds(ext, world)
        assert e[0] == part.INT_MIN and e[-1] == part.INT_MAX and len(e) == world + 1
        assert all(e[i] < e[i + 1] for i in range(world))
        ids = np.stack([np.arange(-200, 200), np.zeros(400, int), np.zeros(400, int)], 1).astype(np.int32)
        own = part.owner_of(ids, ext, world)
        assert own.min() == 0 and own.max() == world - 1
        for r in range(world):
            lo, hi = part.slab_for_rank(ext, r, world)
            assert np.array_equal(own == r, (ids[:, 0] >= lo) & (ids[:, 0] < hi))


def test_merge_needs_reproduces_single_process_flags():
    cam = synth.Camera()
    C, ig = O.camera_from(cam), O.default_integrator()
    depth, rgba, q, pose = synth.room_frame(2, cam)
    ref = O.Volume(RES5, C, ig)
    ids, new = ref.prepare(depth, pose)
    needs = np.zeros(len(ids), np.uint8)
    ref.integrate(depth, rgba, None, pose, ids, needs, 1, -1)
    ext = part.room_extent_chunks(RES5)
    world = 4
    own = part.owner_of(ids, ext, world)
    per_rank = []
    for r in range(world):
        v = O.Volume(RES5, C, ig)
        ids_r, _ = v.prepare(depth, pose)          # selection runs in full on every rank
        assert np.array_equal(ids_r, ids)
        mine = ids[own == r]
        nd = np.zeros(len(mine), np.uint8)
        v.integrate(depth, rgba, None, pose, mine, nd, 1, -1)
        full = np.zeros(len(ids), np.uint8)
        full[own == r] = nd
        per_rank.append(full)
    merged = part.merge_needs(per_rank, ids, ext, world)
    assert np.array_equal(merged, needs)


def test_boundary_mask():
    # ghost band of the slab [5, 10) of ChunkID.x: key 9 (read by the rank above) and keys 5..6 (the
    # mesher of the rank below reads c + {0,1}^3 and the face neighbours of those)
    ids = np.array([[4, 0, 0], [5, 0, 0], [6, 3, 3], [7, 0, 0], [9, 1, 1], [10, 0, 0]], np.int32)
    assert list(part.boundary_mask(ids, 5, 10)) == [False, True, True, False, True, False]
    # key x + y + z: four layers above the lower face
    ids = np.array([[5, 0, 0], [5, 1, 2], [5, 2, 2], [0, 0, 19], [4, 0, 0]], np.int32)
    assert list(part.boundary_mask(ids, 5, 20, (1, 1, 1))) == [True, True, False, True, False]


def test_balanced_edges_split_a_sample_evenly_and_cover_everything():
    rng = np.random.default_rng(7)
    ids = np.concatenate([rng.integers(-50, 50, (4000, 3)), np.tile([49, 0, 0], (3000, 1)) + rng.integers(0, 2, (3000, 3)) * [0, 30, 40]])
    for axis in ((1, 0, 0), (1, 1, 1), (1, 0, 1)):
        for world in (2, 3, 8):
            edges = part.balanced_edges(part.key_of(ids, axis), world)
            assert len(edges) == world + 1 and edges[0] == part.INT_MIN and edges[-1] == part.INT_MAX
            assert all(edges[i] < edges[i + 1] for i in range(world))
            own = part.owner_of_key(ids, edges, axis)
            assert own.min() >= 0 and own.max() < world
            k = part.key_of(ids, axis)
            for r in range(world):
                assert np.all((k[own == r] >= edges[r]) & (k[own == r] < edges[r + 1]))
    # the diagonal key spreads a wall that one x slab would hold alone
    wall = np.array([[49, y, z] for y in range(-30, 30) for z in range(-40, 40)])
    ex = part.balanced_edges(part.key_of(wall, (1, 0, 0)), 4)
    ed = part.balanced_edges(part.key_of(wall, (1, 1, 1)), 4)
    assert np.bincount(part.owner_of_key(wall, ex, (1, 0, 0)), minlength=4).max() == len(wall)
    assert np.bincount(part.owner_of_key(wall, ed, (1, 1, 1)), minlength=4).max() < 0.3 * len(wall)
    assert part.balanced_edges([], 3)[1:-1] == [1, 2]


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus 8` with no launcher around it and fewer than eight devices visible (none in the CPU container)
    exits non-zero without printing a line -- never an N = 1 number under an N = 8 flag (missing 2)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
    if r.returncode == 0 and r.stdout.strip().isdigit() and int(r.stdout.strip()) >= 8:
        pytest.skip("eight devices visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TF_BENCH_DEVICE", "TF_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1"], cwd=root,
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "device" in r.stderr


def test_bench_rejects_a_world_size_that_contradicts_the_flag():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1"], cwd=root,
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr
