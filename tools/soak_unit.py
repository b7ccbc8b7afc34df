#!/usr/bin/env python3
"""Randomised parity soak of the keyframe unit (tf_keyframe_unit_device vs the oracle driven call by call, bit for bit):
random image sizes, voxel sizes, group sizes (0-6 local frames), moved keyframes, with and without quality images.

    python tools/soak_unit.py <seed> <cases>
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from texturefusion_amd import synth
from tests.test_gpu_unit import run_unit_sequence
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time()
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    W = int(rng.choice([320, 400, 480])); H = int(rng.choice([240, 304, 360]))
    f = float(rng.uniform(0.7, 1.0) * W * 525.0 / 640.0)
    res = float(rng.choice([0.006, 0.008, 0.01]))
    cam = synth.Camera(W, H, f, f, W / 2 - 0.5, H / 2 - 0.5, 0.01, 5.0)
    k0 = int(rng.integers(0, 150)); step = int(rng.integers(1, 4))
    wob = float(rng.uniform(0, 0.1)); radius = float(rng.uniform(0.3, 1.2))
    n_groups = int(rng.integers(2, 5))
    plan, k, live = [], 0, []
    for g in range(n_groups):
        nl = int(rng.integers(0, 7))
        moves = [(int(m), int(rng.integers(1, 3))) for m in live if rng.random() < 0.5]
        plan.append((10 + 3 * g, k, list(range(k + 1, k + 1 + nl)), moves))
        live.append(10 + 3 * g)
        k += 1 + nl
    frames = [synth.room_frame(k0 + step * i, cam, with_quality=True, wobble=wob, radius=radius) for i in range(k + 3)]
    with_q = bool(rng.integers(0, 2))
    nm = run_unit_sequence(cam, np.float32(res), frames, plan, with_q, max_chunks=1 << 18, stride=int(rng.integers(1, 6)))
    print("case %d: %dx%d f %.0f res %.3f groups %s quality %s wobble %.2f radius %.2f -> %d meshes OK (%.0f s)"
          % (case, W, H, f, res, [(len(p[2]), len(p[3])) for p in plan], with_q, wob, radius, nm, time.time() - t0), flush=True)
