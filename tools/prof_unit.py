#!/usr/bin/env python3
"""Profile of the keyframe unit -- tf_keyframe_unit_device = MobileFusion::tsdfFusion (GCFusion/MobileFusion.cpp:274-406),
the product's real caller -- on the bench's S-room orbit: every 7th frame a keyframe (depth + colour), the six behind it
its local frames (depth only), meshes / patches / atlas per keyframe; optionally one MOVED keyframe in every other call
(de-integrated over its stored validChunks and re-integrated -- at the SAME poses: a full orbit of groups re-integrated one
frame off builds double walls whose meshes exhaust the overflow pool; the work per moved group is the same).

  --run                 the workload (under rocprofv3: the program itself behind "--"); the timed window of keyframes is
                        bracketed by two launches of the library's empty kernel (tf_profile_calibrate(1) = k_null), which
                        is how --summarize finds it in the trace
  --count               the same groups call by call on a second volume (tf_prepare / tf_integrate / tf_finalize, then the
                        unit's texture stage) and the exact integer counts behind the algorithmic bytes -> JSON line
  --summarize T F W C   kernel-trace csv, FETCH_SIZE csv, WRITE_SIZE csv, counts json -> per-kernel us / bytes per keyframe,
                        roofline fraction -> JSON line (profiles/r4/)
"""
import argparse
import csv
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_LOCAL = int(os.environ.get("UNIT_N_LOCAL", "6"))  # (experiments: fewer local frames per keyframe, same keyframes)
STRIDE = 7
ORBIT = 200
WARM = 6
N_KF = 20
QUALITY_PTR = [0]  # --quality: every keyframe carries this quality image (one device image: the figure is the pass's time)


def frames_and_volume(moved):
    import numpy as np
    import torch
    from texturefusion_amd import capi, synth
    cam = synth.Camera()
    res = np.float32(0.005)
    dev = torch.device("cuda", 0)
    fr = [synth.room_frame(k, cam, with_quality=False) for k in range(ORBIT)]
    dd = [torch.from_numpy(f[0]).to(dev) for f in fr]
    dc = [torch.from_numpy(f[1]).to(dev) for f in fr]
    poses = np.stack([f[3].reshape(12) for f in fr]).astype(np.float32)
    pinv = np.stack([synth.pose_inverse16(f[3]) for f in fr]).astype(np.float32)
    vol = capi.Volume(res, cam, max_chunks=1 << 19, max_list=1 << 18, max_coarse=1 << 20)
    return cam, res, fr, dd, dc, poses, pinv, vol


def group_of(capi, dd, dc, poses, g, shift=0, old=False):
    k0 = (STRIDE * g) % ORBIT
    loc = [(k0 + 1 + i) % ORBIT for i in range(N_LOCAL)]
    P = lambda k: poses[(k + shift) % ORBIT]
    kw = {}
    if old:
        kw = dict(old_keyframe_pose=poses[k0], old_local_poses=[poses[k] for k in loc])
    return capi.Volume.unit_group(1000 + g, (dd[k0].data_ptr(), dc[k0].data_ptr(), QUALITY_PTR[0], P(k0)),
                                  [(dd[k].data_ptr(), P(k)) for k in loc], **kw), k0, loc


def run(moved, quality=False):
    from texturefusion_amd import capi
    cam, res, fr, dd, dc, poses, pinv, vol = frames_and_volume(moved)
    if quality:
        import numpy as np
        import torch
        q = torch.from_numpy(np.random.default_rng(3).uniform(0.05, 1.0, fr[0][0].shape).astype(np.float32)).to(dd[0].device)
        QUALITY_PTR[0] = q.data_ptr()

    def call(g):
        fresh, k0, _ = group_of(capi, dd, dc, poses, g)
        mv = [group_of(capi, dd, dc, poses, g - 2, shift=0, old=True)[0]] if moved and g >= 2 and g % 2 == 0 else []
        vol.keyframe_unit(fresh=fresh, moved=mv, texture=True, pose_inv16=pinv[k0])
        return len(mv)

    # pre-roll: one orbit's worth of keyframes, so that the window meets a steady-state volume (meshes everywhere)
    n_pre = ORBIT // STRIDE
    for g in range(n_pre):
        call(g)
    vol.sync()
    vol.profile_calibrate(1)          # delimiter launch (k_null)
    t0 = time.perf_counter()
    n_moved = 0
    for g in range(n_pre, n_pre + N_KF):
        n_moved += call(g)
    vol.sync()
    dt = time.perf_counter() - t0
    vol.profile_calibrate(1)          # delimiter launch
    print(json.dumps({"keyframes": N_KF, "moved_groups": n_moved, "ms_per_keyframe": 1e3 * dt / N_KF,
                      "keyframes_per_s": N_KF / dt, "frame_integrations_per_s": (N_KF * STRIDE + 2 * STRIDE * n_moved) / dt,
                      "first_keyframe": n_pre, "unit_store": vol.keyframe_unit_stats()}))
    vol.close()


def count(moved):
    """exact integer counts of the window's keyframes: per frame the 8-voxel rows K-A rewrites (call-by-call replay on a
    volume that went through the same pre-roll), per keyframe what the texture stage handles"""
    import numpy as np
    from texturefusion_amd import capi
    cam, res, fr, dd, dc, poses, pinv, vol = frames_and_volume(moved)
    n_pre = ORBIT // STRIDE
    W, H = cam.width, cam.height

    def unit(g):
        fresh, k0, _ = group_of(capi, dd, dc, poses, g)
        mv = [group_of(capi, dd, dc, poses, g - 2, shift=0, old=True)[0]] if moved and g >= 2 and g % 2 == 0 else []
        vol.keyframe_unit(fresh=fresh, moved=mv, texture=True, pose_inv16=pinv[k0])

    for g in range(n_pre):
        unit(g)
    vol.sync()
    tot = dict(rows_tsdf=0, rows_color=0, frames=0, colour_frames=0, n_dirty=0, n_exact=0, n_surface=0, n_vertices=0, n_triangles=0,
               roi_pixels=0, n_patches=0, selected=0, updated=0)
    for g in range(n_pre, n_pre + N_KF):
        # the group's frames one by one through the call-by-call entry points on THIS volume (same voxel results as the
        # unit's group kernel: tests/test_gpu_group.py), reading the row counts behind every frame; then the unit's own
        # texture stage through a unit call with no frames is not possible -- so the stage runs via tf_texture_frame_device
        k0 = (STRIDE * g) % ORBIT
        loc = [(k0 + 1 + i) % ORBIT for i in range(N_LOCAL)]
        vol.frame_bind_device(dd[k0].data_ptr(), dc[k0].data_ptr(), 0)
        ids, new = vol.prepare(poses[k0])
        needs = np.zeros(len(ids), np.uint8)
        vol.integrate(poses[k0], ids, needs, 1, True, False)   # (needs is updated in place)
        st = vol.stats()
        tot["rows_tsdf"] += st.rows_tsdf; tot["rows_color"] += st.rows_color; tot["frames"] += 1; tot["colour_frames"] += 1
        tot["selected"] += len(ids)
        for k in loc:
            vol.frame_bind_device(dd[k].data_ptr(), 0, 0)
            vol.integrate(poses[k], ids, needs, 1, False, False)
            st = vol.stats()
            tot["rows_tsdf"] += st.rows_tsdf; tot["frames"] += 1
        vol.finalize(ids, needs, new)
        tot["updated"] += int(needs.sum())
        vol.frame_bind_device(dd[k0].data_ptr(), dc[k0].data_ptr(), 0)
        vol.texture_frame_device(pinv[k0], 1000 + g)
        ts = vol.texture_stats()
        for k in ("n_dirty", "n_exact", "n_surface", "n_vertices", "n_triangles", "roi_pixels", "n_patches"):
            tot[k] += getattr(ts, k)
    b_ka = 128 * tot["rows_tsdf"] + 128 * tot["rows_color"] + 4 * W * H * tot["frames"] + 4 * W * H * tot["colour_frames"]
    b_mesh = 32 * tot["n_dirty"] + 4096 * tot["n_exact"] + 6552 * tot["n_surface"] + 44 * tot["n_vertices"] + 6 * tot["n_triangles"]
    b_atlas = 44 * tot["n_vertices"] + 6 * tot["roi_pixels"]
    print(json.dumps({"keyframes": N_KF, "per_keyframe": {k: v / N_KF for k, v in tot.items()},
                      "algorithmic_bytes_per_keyframe": {"voxel_update": b_ka / N_KF, "mesh": b_mesh / N_KF, "patch_atlas": b_atlas / N_KF,
                                                         "all": (b_ka + b_mesh + b_atlas) / N_KF},
                      "formula": "128 B per rewritten 8-voxel row (sdf+weight, colour) of EVERY frame of the group + 4 B per depth / "
                                 "colour pixel read (SURVEY.md s.8d, per frame as the reference moves them: the group kernel visits a "
                                 "chunk once for the six local frames and moves less); mesh and atlas terms as in bench.py"}))
    vol.close()


def _rows(path, col, want=None):
    out = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if want and r.get("Counter_Name") != want:
                continue
            name = r["Kernel_Name"]
            name = name.split("(")[0].replace("void ", "").replace("tf::", "").strip()
            out.append((int(r["Dispatch_Id"]), name, col(r)))
    return sorted(out)


def _window(rows):
    marks = [d for d, n, _ in rows if n.startswith("k_null")]
    if len(marks) < 2:
        raise SystemExit("delimiter launches (k_null) not found in the trace")
    lo, hi = marks[-2], marks[-1]
    return [(d, n, v) for d, n, v in rows if lo < d < hi]


def summarize(trace, fetch, write, counts, run_line=None):
    cnt = json.loads(open(counts).read().strip().splitlines()[-1])
    K = cnt["keyframes"]
    per = {}
    # (round 6: a call's front end runs on a stream of its own beside the previous call's texture stage -- the SUM of the
    # kernel durations counts that time twice; the span of the window and the un-profiled wall time per keyframe do not)
    se = _window(_rows(trace, lambda r: (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))))
    span_us = 1e-3 * (max(v[1] for _, _, v in se) - min(v[0] for _, _, v in se)) / K if se else 0.0
    wall_us = None
    if run_line and os.path.exists(run_line):
        try:
            wall_us = 1e3 * json.loads(open(run_line).read().strip().splitlines()[-1])["ms_per_keyframe"]
        except Exception:
            wall_us = None
    for _, n, v in _window(_rows(trace, lambda r: 1e-3 * (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))):
        e = per.setdefault(n, {"us": 0.0, "launches": 0.0, "fetch_B": 0.0, "write_B": 0.0})
        e["us"] += v / K
        e["launches"] += 1.0 / K
    for path, key, scale, name in ((fetch, "fetch_B", 2048.0, "FETCH_SIZE"), (write, "write_B", 1024.0, "WRITE_SIZE")):
        if path and os.path.exists(path):
            for _, n, v in _window(_rows(path, lambda r: float(r["Counter_Value"]), want=name)):
                per.setdefault(n, {"us": 0.0, "launches": 0.0, "fetch_B": 0.0, "write_B": 0.0})[key] += scale * v / K
    tot_us = sum(e["us"] for e in per.values())
    tot_b = sum(e["fetch_B"] + e["write_B"] for e in per.values())
    alg = cnt["algorithmic_bytes_per_keyframe"]["all"]
    out = {"unit": "per keyframe (1 colour + 6 depth-only frames, meshes, patches, atlas)", "keyframes": K,
           "kernel_us_per_keyframe": tot_us, "algorithmic_bytes_per_keyframe": cnt["algorithmic_bytes_per_keyframe"],
           "roofline": {"bound": "hbm", "achieved_GBs": alg / tot_us / 1e3 if tot_us else 0.0, "peak_GBs": 8000.0,
                        "frac": alg / tot_us / 1e3 / 8000.0 if tot_us else 0.0, "traffic_bytes_per_keyframe": tot_b or None,
                        "span_us_per_keyframe_profiled": span_us, "frac_span": alg / span_us / 1e3 / 8000.0 if span_us else None,
                        "wall_us_per_keyframe": wall_us, "frac_wall": alg / wall_us / 1e3 / 8000.0 if wall_us else None},
           "kernels": {k: {kk: round(vv, 3) for kk, vv in e.items()} for k, e in sorted(per.items(), key=lambda kv: -kv[1]["us"])},
           "counts_per_keyframe": cnt["per_keyframe"],
           "pmc": "FETCH_SIZE x 2048 + WRITE_SIZE x 1024 bytes (profiles/r2/README.md calibration), separate passes"}
    print(json.dumps(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--count", action="store_true")
    ap.add_argument("--moved", action="store_true", help="one moved keyframe in every other call")
    ap.add_argument("--summarize", nargs=4, metavar=("TRACE", "FETCH", "WRITE", "COUNTS"))
    ap.add_argument("--quality", action="store_true", help="--run: keyframes with a quality image")
    ap.add_argument("--run-line", default=None, help="--summarize: the un-profiled --run output (wall time per keyframe)")
    a = ap.parse_args()
    if a.run:
        run(a.moved, a.quality)
    elif a.count:
        count(a.moved)
    elif a.summarize:
        summarize(*a.summarize, run_line=a.run_line)
