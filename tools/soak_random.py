#!/usr/bin/env python3
"""Randomised parity soak of the textured per-frame unit (GPU vs oracle, bit for bit): random image sizes, focal lengths,
voxel sizes, orbit stretches, hand-held wobble, either entry point.

    python tools/soak_random.py <seed> <cases>      (46 cases over seeds 11, 21, 22, 23 passed at the end of round 2)
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from texturefusion_amd import synth
from tests.test_gpu_textured_soak import random_case
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time()
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    what, nm = random_case(rng)
    print("case %d: %s -> %d meshes OK (%.0f s)" % (case, what, nm, time.time() - t0), flush=True)
