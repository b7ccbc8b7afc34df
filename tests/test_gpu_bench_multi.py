"""bench.py's N > 1 path end to end with two ranks sharing the one GPU of the test box (the blocks travel over
gloo instead of RCCL, which refuses two ranks on one device): slab partition, slab-restricted selection,
per-frame voxel update -> fixed-capacity block all-gather -> texture stage, max-over-ranks timing, one JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu(gpu_required):
    env = dict(os.environ, TF_BENCH_DEVICE="0", TF_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "24",
           "--warmup", "6", "--cpu-frames", "0", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["steps"] == 24
