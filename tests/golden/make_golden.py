#!/usr/bin/env python3
"""Regenerates tests/golden/wall_regression.npz.

NOT reference outputs: the reference cannot be built here (see oracle/tf_oracle.h).  These are
regression vectors produced by this repo's own oracle (scalar kernel) so that (a) the oracle cannot
drift unnoticed and (b) the GPU suite has a committed, input-independent target.  The inputs are
regenerated from seeds by texturefusion_amd.synth, the expected outputs are stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import api as O  # noqa: E402
from texturefusion_amd import synth  # noqa: E402


def scene():
    cam = synth.Camera()
    poses = [synth.pose_euler(0.2, -0.1, 0.05, (0.03, -0.02, 0.0)), synth.pose_euler(0.25, -0.08, 0.02, (0.05, 0.0, 0.01))]
    frames = [synth.wall_frame(1.3, cam, pose=p, seed=i, rgba_value=(180 - 40 * i, 90 + 30 * i, 60, 1),
                               quality_value=0.5 - 0.25 * i) for i, p in enumerate(poses)]
    return cam, frames


def main():
    cam, frames = scene()
    res = np.float32(0.005)
    v = O.Volume(res, O.camera_from(cam), O.default_integrator())
    out = {}
    for i, (depth, rgba, q, pose) in enumerate(frames):
        ids, new = v.prepare(depth, pose)
        needs = np.zeros(len(ids), np.uint8)
        qual = v.integrate(depth, rgba, q, pose, ids, needs, 1, i)
        valid = v.finalize(ids, needs, new)
        out["ids%d" % i] = ids
        out["new%d" % i] = new
        out["needs%d" % i] = needs
        out["quality%d" % i] = qual
        out["n_valid%d" % i] = np.int64(len(valid))
    allc = v.list_chunks()
    order = np.lexsort((allc[:, 2], allc[:, 1], allc[:, 0]))
    pick = allc[order][:: max(1, len(allc) // 12)][:12]
    out["pick"] = pick
    out["sdf"] = np.stack([v.get_chunk(c)[0] for c in pick])
    out["weight"] = np.stack([v.get_chunk(c)[1] for c in pick])
    out["color"] = np.stack([v.get_chunk(c)[2] for c in pick])
    out["n_chunks"] = np.int64(v.num_chunks())
    out["n_dirty"] = np.int64(len(v.dirty()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "wall_regression.npz"), **out)
    print("wrote wall_regression.npz:", {k: getattr(x, "shape", x) for k, x in out.items()})


if __name__ == "__main__":
    main()
