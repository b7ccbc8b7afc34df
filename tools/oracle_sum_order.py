#!/usr/bin/env python3
"""How much of the oracle's output hangs on the summation orders nobody can pin here?

Eigen is not in the image, so oracle/tf_oracle.c restates its reductions in the order its source suggests
(fixed-size 3-vector dots as a0*b0 + (a1*b1 + a2*b2), dynamic mat-vec sequentially, norm as x*x + (y*y + z*z),
normalize() as a division): "parity unpinned" (DESIGN.md s.5).  This script bounds the exposure: it runs the
S-room stream through the oracle once per alternative (tfo_set_sum_order) and reports, against the default,
  * per frame: visible-list length and order, needsUpdate flags,
  * at the end: chunk set, voxels whose sdf / weight / colour differ (count, max |d sdf|), quality sums,
  * for the meshing alternatives: meshes of the last frames -- vertex counts, max |d position|, max |d normal|.

    python tools/oracle_sum_order.py [--frames 200] [--step 1] [--res 0.005] [--out profiles/r2/oracle_sum_order.json]

CPU only (no GPU, no reference): the oracle against itself.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import api as O  # noqa: E402
from texturefusion_amd import synth  # noqa: E402

VARIANTS = [
    (0, "default: fixed a0b0+(a1b1+a2b2), dynamic sequential, norm x2+(y2+z2), normalize divides"),
    (1, "fixed-size dots sequential"),
    (2, "dynamic mat-vec in tree order"),
    (3, "both dot orders swapped"),
    (4, "gradient norm (x2+y2)+z2"),
    (8, "normalize() multiplies by the reciprocal (Eigen 3.2)"),
]


def run(bits, frames, cam, res, threads, mesh_last):
    L = O.lib()
    L.tfo_set_sum_order(bits)
    vol = O.Volume(res, O.camera_from(cam), O.default_integrator())
    vol.set_kernel(1 if L.tfo_have_avx2() else 0)
    vol.set_threads(threads)
    per_frame = []
    for k, f in enumerate(frames):
        ids, new = vol.prepare(f[0], f[3])
        needs = np.zeros(len(ids), np.uint8)
        q = vol.integrate(f[0], f[1], None, f[3], ids, needs, 1, k)
        vol.finalize(ids, needs, new)
        per_frame.append((ids.copy(), needs.copy(), q.copy()))
        if k >= len(frames) - mesh_last:
            vol.update_meshes()
    ids = vol.list_chunks()
    order = np.lexsort((ids[:, 2], ids[:, 1], ids[:, 0]))
    ids = ids[order]
    vox = {}
    for cid in ids:
        s, w, c = vol.get_chunk(cid)
        vox[tuple(int(x) for x in cid)] = (s.copy(), w.copy(), c.copy())
    meshes = {}
    for cid in vol.list_meshes():
        meshes[tuple(int(x) for x in cid)] = vol.get_mesh(cid)
    vol.close()
    L.tfo_set_sum_order(0)
    return per_frame, vox, meshes


def compare(base, alt):
    pf0, vox0, m0 = base
    pf1, vox1, m1 = alt
    out = {"frames_with_different_list": 0, "list_entries_different": 0, "frames_with_different_needs": 0,
           "needs_flags_different": 0, "quality_sums_different": 0}
    for (i0, n0, q0), (i1, n1, q1) in zip(pf0, pf1):
        if len(i0) != len(i1) or not np.array_equal(i0, i1):
            out["frames_with_different_list"] += 1
            s0 = {tuple(x) for x in i0.tolist()}
            s1 = {tuple(x) for x in i1.tolist()}
            out["list_entries_different"] += len(s0 ^ s1) if s0 != s1 else int(np.sum(np.any(i0 != i1, axis=1)))
        else:
            d = int(np.sum(n0 != n1))
            if d:
                out["frames_with_different_needs"] += 1
                out["needs_flags_different"] += d
            out["quality_sums_different"] += int(np.sum(q0.view(np.uint32) != q1.view(np.uint32)))
    k0, k1 = set(vox0), set(vox1)
    out["chunks"] = len(k0)
    out["chunks_only_in_one"] = len(k0 ^ k1)
    nv = nd_s = nd_w = nd_c = 0
    max_ds = 0.0
    for key in k0 & k1:
        s0, w0, c0 = vox0[key]
        s1, w1, c1 = vox1[key]
        nv += s0.size
        ds = s0.view(np.uint32) != s1.view(np.uint32)
        if ds.any():
            nd_s += int(ds.sum())
            both = ds & (np.abs(s0) < 100) & (np.abs(s1) < 100)
            if both.any():
                max_ds = max(max_ds, float(np.max(np.abs(s0[both] - s1[both]))))
        nd_w += int(np.sum(w0.view(np.uint32) != w1.view(np.uint32)))
        nd_c += int(np.sum(np.any(c0.reshape(-1, 4) != c1.reshape(-1, 4), axis=1)))
    out.update(voxels=nv, voxels_sdf_different=nd_s, voxels_weight_different=nd_w, voxels_colour_different=nd_c,
               max_abs_sdf_difference=max_ds)
    mk0, mk1 = set(m0), set(m1)
    out["meshes"] = len(mk0)
    out["meshes_only_in_one"] = len(mk0 ^ mk1)
    dcount = 0
    maxdp = maxdn = 0.0
    for key in mk0 & mk1:
        a, b = m0[key], m1[key]
        if a["verts"].shape != b["verts"].shape or not np.array_equal(a["indices"], b["indices"]):
            dcount += 1
            continue
        if a["verts"].size:
            maxdp = max(maxdp, float(np.max(np.abs(a["verts"] - b["verts"]))))
            maxdn = max(maxdn, float(np.max(np.abs(a["normals"] - b["normals"]))))
    out.update(meshes_with_different_topology=dcount, max_abs_vertex_difference=maxdp, max_abs_normal_difference=maxdn)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--step", type=int, default=1, help="take every step-th frame of the orbit")
    ap.add_argument("--res", type=float, default=0.005)
    ap.add_argument("--threads", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--mesh-last", type=int, default=1, help="run UpdateMeshes after each of the last N frames")
    ap.add_argument("--wobble", type=float, default=0.1,
                    help="hand-held pitch / roll amplitude in radians on top of the yaw orbit (0 = the bench stream, whose "
                         "rotations have four exact zeros and cannot show an order dependence)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    cam = synth.Camera()
    res = np.float32(args.res)
    frames = [synth.room_frame(k * args.step, cam, with_quality=False, wobble=args.wobble) for k in range(args.frames)]
    t0 = time.time()
    base = run(0, frames, cam, res, args.threads, args.mesh_last)
    report = {"stream": "S-room orbit, %d frames (every %d-th), 640x480, %.0f mm voxels, pitch/roll wobble %.2f rad"
                        % (args.frames, args.step, 1e3 * args.res, args.wobble),
              "variants": []}
    for bits, name in VARIANTS[1:]:
        alt = run(bits, frames, cam, res, args.threads, args.mesh_last)
        r = compare(base, alt)
        r["bits"] = bits
        r["alternative"] = name
        report["variants"].append(r)
        print(json.dumps(r))
    report["seconds"] = round(time.time() - t0, 1)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(report, fh, indent=1)


if __name__ == "__main__":
    main()
