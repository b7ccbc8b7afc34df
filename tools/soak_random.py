#!/usr/bin/env python3
"""Randomised parity soak of the textured per-frame unit (GPU vs oracle, bit for bit): random image sizes, focal lengths,
voxel sizes, orbit stretches, hand-held wobble, either entry point.

    python tools/soak_random.py <seed> <cases>      (46 cases over seeds 11, 21, 22, 23 passed at the end of round 2)
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from texturefusion_amd import synth
from tests.test_gpu_textured_soak import _run
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time()
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    W = int(rng.choice([320, 400, 480, 640])); H = int(rng.choice([240, 304, 360, 480]))
    f = float(rng.uniform(0.7, 1.0) * W * 525.0 / 640.0)
    res = float(rng.choice([0.005, 0.006, 0.008, 0.01]))
    cam = synth.Camera(W, H, f, f, W / 2 - 0.5, H / 2 - 0.5, 0.01, 5.0)
    k0 = int(rng.integers(0, 180)); step = int(rng.integers(1, 4)); n = int(rng.integers(10, 26))
    wob = float(rng.uniform(0, 0.1)); radius = float(rng.uniform(0.2, 1.2))
    frames = [synth.room_frame(k0 + step * i, cam, with_quality=False, wobble=wob, radius=radius) for i in range(n)]
    host = [False, True, "registered", "registered_async", "no_deferral", "rgb"][int(rng.integers(0, 6))]  # entry point / host-frame path
    nm = _run(cam, np.float32(res), frames, host_frames=host, max_chunks=1 << 18, stride=int(rng.integers(1, 6)))
    print("case %d: %dx%d f %.0f res %.3f frames %d step %d wobble %.2f radius %.2f host %s -> %d meshes OK (%.0f s)" % (case, W, H, f, res, n, step, wob, radius, host, nm, time.time() - t0), flush=True)
