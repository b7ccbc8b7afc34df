"""bench.py's N > 1 path end to end with two ranks sharing the one GPU of the test box (the blocks travel over
gloo instead of RCCL, which refuses two ranks on one device): slab partition, slab-restricted selection,
per-frame voxel update -> sized neighbour blocks -> texture stage, max-over-ranks timing, one JSON line.
Both ways of starting the ranks: the driver's (torch.distributed.run) and bench.py's own (`python bench.py --gpus 2`
with no launcher around it: it spawns its ranks itself before it touches the GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "12", "--warmup", "4", "--unique-frames", "48", "--cpu-frames", "0", "--no-roofline"]


def _one_line(r):
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_under_torchrun(gpu_required):
    env = dict(os.environ, TF_BENCH_DEVICE="0", TF_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-independent"] + SMALL
    d = _one_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["steps"] == 12
    assert len(d["per_rank"]) == 2


def test_bench_spawns_its_own_ranks(gpu_required):
    """`python bench.py --gpus 2` without a launcher: two ranks, n_gpus == 2 in the line, the independent-streams figure
    (two whole volumes, no exchange) next to the strong-scaling value."""
    env = dict(os.environ, TF_BENCH_DEVICE="0", TF_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    d = _one_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    ind = d["independent_streams"]
    assert "error" not in ind, ind
    assert ind["n_gpus"] == 2 and ind["scaling"] == "weak" and ind["value"] > 0 and len(ind["per_rank_ms_per_step"]) == 2


def test_more_ranks_than_devices_is_refused(gpu_required):
    """--gpus 8 on a box with fewer devices must fail, not print a smaller job's number under that flag."""
    if gpu_required >= 8:  # (the fixture's device count; torch is not imported into this process: it carries a HIP runtime of its own)
        pytest.skip("eight devices visible")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TF_BENCH_DEVICE", "TF_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + SMALL, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "only" in r.stderr and "device" in r.stderr
