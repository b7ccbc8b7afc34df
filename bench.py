#!/usr/bin/env python3
"""Headline benchmark: RGB-D frames/s of the textured per-frame unit on MI355X (BASELINE.json configs[2]).

One "step" = one synthetic 640x480 RGB-D frame of the S-room stream through
  prepare -> integrate(depth+colour) -> finalize          Chisel::IntegrateDepthScanColor 5-arg, Structure/Chisel.h:453-468
  -> UpdateMeshes -> CompressMeshes                        over that frame's dirty chunks (Structure/Chisel.h:479-481, Chisel.cpp:112-147)
  -> GeneratePatches(label = this frame) -> UpdateAtlas    Structure/Chisel.cpp:149-196
at 5 mm voxels, frames already resident in HBM, nothing copied back, no host synchronisation inside the
timed region (tf_stream_frames_textured_device).  --mode tsdf runs configs[1] (atlas off).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (metric/value/... + "roofline" + "cpu_baseline").
N > 1: one process per GPU, static chunk-range partition of ONE stream (slabs of the key x + y + z) -- every
rank sees every frame, selects / integrates / meshes / textures only the chunks of its slab ("strong"
scaling); the ranks all-gather the chunks of their ghost bands over RCCL.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
STEP_KERNELS = ("integrate", "dirty", "mesh", "finalize", "patch_rank", "patch_project")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--res", type=float, default=0.005)
    ap.add_argument("--mode", choices=("textured", "tsdf"), default="textured",
                    help="textured = BASELINE configs[2] (TSDF + mesh + atlas per frame); tsdf = configs[1]")
    ap.add_argument("--hires", action="store_true", help="1280x960 camera (configs[3])")
    ap.add_argument("--scene", choices=("room", "big"), default="room",
                    help="room = S-room 4x3x4 m; big = 8x6x8 m hall, walls at 3-4 m (configs[3] HBM stress)")
    ap.add_argument("--unique-frames", type=int, default=200, help="distinct frames of the orbit kept in HBM")
    ap.add_argument("--exchange-every", type=int, default=40, help="N>1, --mode tsdf: boundary all-gather period (frames)")
    ap.add_argument("--exchange-cap", type=int, default=1024,
                    help="N>1: records per rank of the fixed-capacity boundary all-gather (8 KiB each)")
    ap.add_argument("--cpu-frames", type=int, default=48, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-warmup", type=int, default=24, help="untimed frames that build up the CPU baseline's volume")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = the reference's parallel_for policy")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) that fill roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the workload only, run under rocprofv3
    ap.add_argument("--pmc-steps", type=int, default=40, help="frames of each counter pass")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-frames (H2D per frame) measurement")
    ap.add_argument("--no-group", action="store_true", help="skip the keyframe-group (1 colour + 6 depth frames) measurement")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the N>1 code path (partition + boundary all-gather) even with one rank (smoke test)")
    return ap.parse_args()


def make_frame(k, cam, args):
    from texturefusion_amd import synth
    if args.scene == "big":
        return synth.room_frame(k, cam, half=(4.0, 3.0, 4.0), radius=0.5, with_quality=False)
    return synth.room_frame(k, cam, with_quality=False)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "TF_BENCH_DEVICE" in os.environ:  # test hook: several ranks on one GPU (with TF_BENCH_BACKEND=gloo)
        local_rank = int(os.environ["TF_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    if args.pmc_child:
        args.no_roofline = args.no_host_path = args.no_group = True
        args.cpu_frames = 0
    # HBM-traffic counters first, in child processes, before this process touches the GPU
    traffic = None
    if world == 1 and not (args.no_pmc or args.pmc_child or args.no_roofline or args.force_exchange):
        traffic = pmc_traffic(args)

    import torch  # plumbing: device memory for the frames, barrier/collectives, device sync
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_exchange
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        backend = os.environ.get("TF_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from texturefusion_amd import capi, synth
    from texturefusion_amd import partition as part

    cam = synth.Camera.hires() if args.hires else synth.Camera()
    res = np.float32(args.res)
    K, Wm = args.steps, args.warmup
    n_unique = max(1, min(args.unique_frames, 3 * K + Wm + 4))
    textured = args.mode == "textured"

    # ---- synthetic stream, generated once and parked in HBM -------------------------------
    frames = [make_frame(k, cam, args) for k in range(n_unique)]
    d_depth = [torch.from_numpy(f[0]).to(dev) for f in frames]
    d_rgba = [torch.from_numpy(f[1]).to(dev) for f in frames]
    poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
    pinv = np.stack([synth.pose_inverse16(f[3]) for f in frames]).astype(np.float32)
    torch.cuda.synchronize()

    s_main = torch.cuda.Stream(device=dev) if multi else None
    big = args.scene == "big"
    vol = capi.Volume(res, cam, max_chunks=(1 << 21) if big else (1 << 19), max_list=(1 << 20) if big else (1 << 18),
                      max_coarse=1 << 22 if big else 1 << 20, device=local_rank,
                      stream=s_main.cuda_stream if multi else None)
    if multi:
        # Ownership key x + y + z: axis-aligned walls and floors are cut diagonally, so no rank holds a
        # whole wall.  Slab edges split the chunk keys of eight sample frames spread over the orbit into
        # equally populated slabs; every rank computes them itself (selection is deterministic), so
        # nothing has to be communicated.
        axis = (1, 1, 1)
        keys = []
        for i in range(0, 200, 25):
            f = frames[i % n_unique] if i < n_unique else make_frame(i, cam, args)
            vol.frame_upload(f[0], None, None)
            ids_s, _ = vol.prepare(f[3])
            keys.append(part.key_of(ids_s, axis))
        vol.reset()
        edges = part.balanced_edges(np.concatenate(keys), max(world, 2) if args.force_exchange and world == 1 else world)
        lo, hi = edges[rank], edges[rank + 1]
        if args.force_exchange and world == 1:
            lo, hi = edges[1] - 12, edges[1] + 12  # a real interior slab so that faces exist and get packed
        vol.set_partition(lo, hi, axis)
        # The collective: ONE fixed-capacity all-gather of [count | records] blocks, no host round trip.
        # backend nccl: RCCL inside the library (tf_comm_init / tf_exchange_boundary) on the volume's stream;
        # other backends (test hook: several ranks on one GPU over gloo): the same blocks through torch.distributed.
        cap = args.exchange_cap
        use_rccl = world > 1 and os.environ.get("TF_BENCH_BACKEND", "nccl") == "nccl"
        if use_rccl:
            uid = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            vol.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()))
            if textured:
                vol.comm_exchange_every_frame(cap)
        else:
            bb = capi.boundary_block_bytes(cap)
            blk = torch.zeros(bb, dtype=torch.uint8, device=dev)
            allb = torch.zeros(bb * max(world, 1), dtype=torch.uint8, device=dev)

    def torch_exchange(join_dirty):
        """[count | records] blocks through torch.distributed on the volume's stream (no .item(), no host wait)."""
        with torch.cuda.stream(s_main):
            vol.boundary_pack_block(blk.data_ptr(), cap)
            if world > 1:
                dist.all_gather_into_tensor(allb, blk)
            else:
                allb.copy_(blk)
            vol.boundary_unpack_blocks(allb.data_ptr(), max(world, 1), rank, cap, join_dirty=join_dirty)

    def run(first, count, ahead=2):
        """Frames [first, first+count) of the stream (cyclic over the unique frames); the next `ahead` frames go
        through their selection stages too, so that a following run(first + count, ...) starts primed."""
        idx = [(first + i) % n_unique for i in range(count + ahead)]
        dd = [d_depth[i].data_ptr() for i in idx]
        dr = [d_rgba[i].data_ptr() for i in idx]
        if textured and (not multi or use_rccl):
            vol.stream_frames_textured_device(dd, dr, poses[idx], pinv[idx], first, n_ahead=ahead)  # N>1: exchange inside
        elif not multi:
            vol.stream_frames_device(dd, dr, poses[idx], n_ahead=ahead)
        elif textured:  # torch transport: voxel update, exchange, texture stage -- frame by frame
            for j in range(count):
                sub = idx[j:j + 1 + min(2, count + ahead - j - 1)]
                vol.stream_frames_device([d_depth[i].data_ptr() for i in sub], [d_rgba[i].data_ptr() for i in sub],
                                         poses[sub], n_ahead=len(sub) - 1)
                torch_exchange(True)
                vol.texture_frame_device(pinv[idx[j]], first + j)
        else:  # TSDF only: batches of --exchange-every frames, one exchange behind each
            for b in range(0, count, args.exchange_every):
                e = min(b + args.exchange_every, count)
                sub = idx[b:e + (ahead if e == count else 0)]
                vol.stream_frames_device([d_depth[i].data_ptr() for i in sub], [d_rgba[i].data_ptr() for i in sub],
                                         poses[sub], n_ahead=len(sub) - (e - b))
                if use_rccl:
                    vol.exchange_boundary(cap)
                else:
                    torch_exchange(False)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up, then the timed region (the launch pipeline stays primed across the boundary) ---
    run(0, Wm)
    vol.sync()
    barrier()
    t0 = time.perf_counter()
    run(Wm, K)
    t_enq = time.perf_counter() - t0  # host time to enqueue the timed region (launches are asynchronous)
    barrier()
    dt = time.perf_counter() - t0
    vol.sync()  # surfaces any device-side capacity error of the timed region
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- instrumented repeat of the next K frames: HIP events (on the handle's stream) around every
    # kernel of the step.  Kept out of the timed region above because the event pairs cost throughput;
    # the per-launch durations are what the roofline needs.
    prof = None
    dt_instr = None
    kinds = STEP_KERNELS if textured else ("integrate",)
    if not args.no_roofline and not multi:
        vol.profile_enable(kinds)
        barrier()
        t1 = time.perf_counter()
        run(Wm + K, K)
        barrier()
        dt_instr = time.perf_counter() - t1
        prof = vol.profile_get(reset=True)
        vol.profile_enable([])
        vol.sync()

    what = ("TSDF integrate + mesh + atlas update per frame (BASELINE.json configs[%d])" % (4 if world > 1 else (3 if args.hires or big else 2))
            if textured else "TSDF integrate, atlas off (BASELINE.json configs[%d])" % (4 if world > 1 else (3 if args.hires or big else 1)))
    out = {
        "metric": ("RGB-D frames/s (TSDF integrate + atlas update: prepare->integrate->finalize, UpdateMeshes, "
                   "GeneratePatches + UpdateAtlas over the frame's dirty chunks)" if textured else
                   "RGB-D frames/s (TSDF integrate, depth+colour, fused prepare->integrate->finalize)"),
        "value": K / dt,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": K,
        "warmup": Wm,
        "ms_per_step": 1e3 * dt / K,
        "host_enqueue_ms_per_step": 1e3 * t_enq / K,
        "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "%s orbit stream, %dx%d RGB-D, %.0f mm voxels, 8^3 chunks, %s; frames resident in HBM"
                        % ("S-hall 8x6x8 m" if big else "S-room 4x3x4 m", cam.width, cam.height, 1e3 * float(res), what),
            "frames_in_hbm": n_unique,
            "parallelism": ("1 GPU" if world == 1 else
                            "%d ranks, chunk-range slabs of the key x+y+z of one stream; one fixed-capacity RCCL all-gather "
                            "(%d records of 8 KiB per rank) of the updated ghost-band chunks %s"
                            % (world, args.exchange_cap,
                               "after every voxel update, ahead of the mesher" if textured else
                               "every %d frames" % args.exchange_every)),
        },
    }

    # ---- roofline over ALL kernels of a step ------------------------------------------------
    if rank == 0 and prof is not None and not multi:
        out["roofline"] = roofline(args, vol, cam, prof, kinds, K, Wm, n_unique, d_depth, d_rgba, poses, pinv, textured,
                                   dt_instr)
        if traffic is not None and "bytes_per_step" in traffic:
            out["roofline"]["traffic"] = traffic["bytes_per_step"]
        out["roofline"]["traffic_detail"] = traffic

    # ---- the drop-in per-frame path: host images in, one call per frame (H2D included) ---------
    if rank == 0 and not multi and not args.no_host_path:
        out["host_frames"] = host_path(args, vol, frames, poses, pinv, textured, Wm, K, n_unique)

    # ---- the keyframe-group flow of TSDFFusion: 1 colour + 6 depth-only frames over one chunk list -------
    if rank == 0 and not multi and not args.no_group and not args.no_host_path:
        out["keyframe_group"] = keyframe_group(args, cam, res, frames, d_depth, d_rgba, poses, n_unique, local_rank)

    # ---- CPU baseline: the oracle (C port of the reference path) on the host cores ------------
    if rank == 0 and world == 1 and args.cpu_frames > 0:  # (rank 0 at N = 1 only)
        out["cpu_baseline"] = cpu_baseline(args, cam, res, frames, n_unique, textured)

    if rank == 0:
        print(json.dumps(out))
    vol.close()
    if multi:
        dist.destroy_process_group()


def roofline(args, vol, cam, prof, kinds, K, Wm, n_unique, d_depth, d_rgba, poses, pinv, textured, dt_instr):
    """Algorithmic bytes of a step / summed kernel time of a step (HIP events of this run).

    Bytes: voxel update 128 B per rewritten TSDF row + 128 B per rewritten colour
    row + one read of the depth and RGBA images; meshing 4 KiB (the chunk's own sdf/weight plane) per dirty
    chunk + 6552 B (the 11^3 - 8^3 halo voxels) per chunk that yields a mesh + 8 B colour read and 36 B written
    per vertex + 6 B per triangle; atlas 44 B per projected vertex (24 read, 20 written) + 3 B read and 3 B
    written per ROI pixel.  The integer counts depend only on the stream: the instrumented frames are replayed
    one by one (untimed) and the exact integers read back after each."""
    idx = [(Wm + 2 * K + i) % n_unique for i in range(K)]
    b_tsdf = b_mesh = b_atlas = 0
    cnt = dict(sel=0, upd=0, dirty=0, meshes=0, verts=0, tris=0, roi=0, patches=0)
    first = Wm + 2 * K
    for j, i in enumerate(idx):
        i1, i2 = (first + j + 1) % n_unique, (first + j + 2) % n_unique
        sub = [i, i1, i2]
        dd = [d_depth[q].data_ptr() for q in sub]
        dr = [d_rgba[q].data_ptr() for q in sub]
        if textured:
            vol.stream_frames_textured_device(dd, dr, poses[sub], pinv[sub], first + j, n_ahead=2)
        else:
            vol.stream_frames_device(dd, dr, poses[sub], n_ahead=2)
        st = vol.stats()
        b_tsdf += 128 * st.rows_tsdf + 128 * st.rows_color + 8 * cam.width * cam.height
        cnt["sel"] += st.n_selected
        cnt["upd"] += st.n_updated
        if textured:
            ts = vol.texture_stats()
            b_mesh += 4096 * ts.n_dirty + 6552 * ts.n_meshes + 44 * ts.n_vertices + 6 * ts.n_triangles
            b_atlas += 44 * ts.n_vertices + 6 * ts.roi_pixels
            for k, v in (("dirty", ts.n_dirty), ("meshes", ts.n_meshes), ("verts", ts.n_vertices),
                         ("tris", ts.n_triangles), ("roi", ts.roi_pixels), ("patches", ts.n_patches)):
                cnt[k] += v
    per_kernel = {}
    t_step = 0.0
    for k in kinds:
        ms, n = prof[k]
        if n:
            per_kernel[k] = {"us_per_step": 1e3 * ms / K, "launches_per_step": n / K}
            t_step += 1e-3 * ms / K
    bytes_step = (b_tsdf + b_mesh + b_atlas) / K
    achieved = bytes_step / t_step / 1e9 if t_step > 0 else 0.0
    group = {"integrate": b_tsdf / K, "mesh": b_mesh / K, "patch_project": b_atlas / K}
    for k, b in group.items():
        if k in per_kernel and per_kernel[k]["us_per_step"] > 0:
            per_kernel[k]["algorithmic_bytes_per_step"] = b
            per_kernel[k]["achieved_GBs"] = b / per_kernel[k]["us_per_step"] / 1e3
    return {
        "bound": "hbm",
        "kernel": ("all kernels of a step: k_frame (K-A + K-C + K-B roles), k_dirty_frame, k_mesh_filter + k_mesh, "
                   "k_patch (adjacency exchange + slot ranks + project + blit)" if textured else
                   "k_frame<color> = K-A(f) + K-C(f+1) + K-B(f+2) block ranges"),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": None,  # filled from the --pmc child passes of this run (pmc_traffic), stays null without them
        "algorithmic_bytes_per_step": bytes_step, "kernel_us_per_step": 1e6 * t_step,
        "kernels": per_kernel,
        "instrumented_ms_per_step": 1e3 * dt_instr / K,
        "per_step": {k: v / K for k, v in cnt.items()},
    }


# rocprofv3 kernel names of one step (one launch each per frame)
PMC_STEP_KERNELS = {"textured": ("k_frame<true>", "k_dirty_frame", "k_mesh_filter", "k_mesh<128>",
                                 "k_patch<true, true, true>"),
                    "tsdf": ("k_frame<true>",)}
# profiles/r2/README.md (tools/calib_fetch on this box type): both counters are in KiB; WRITE_SIZE is exact;
# FETCH_SIZE reads exactly 1/2 of the bytes for every read shape the kernels use (4/8/16 B per lane streams,
# scattered 4-KiB blocks, 4-B gathers: 128-B requests tallied as 64 B), as MI355X_MICROARCH.md states for 16 B/lane.
PMC_BYTES = {"FETCH_SIZE": 2048.0, "WRITE_SIZE": 1024.0}


def pmc_traffic(args):
    """HBM-side bytes per step of the same workload from the L2's fabric counters: one rocprofv3 --pmc pass per
    counter (they do not fit one pass), no trace domain in the same run, the program itself behind "--"
    (MI355X_MICROARCH.md, HBM / PMC sections).  Returns None when rocprofv3 is missing, else a dict; on any
    failure the dict carries "error" and no "bytes_per_step" (roofline.traffic stays null -- never a constant)."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    want = PMC_STEP_KERNELS[args.mode]
    Wc, Kc = 10, args.pmc_steps
    per_kernel = {k: {} for k in want}
    tmp = tempfile.mkdtemp(prefix="tf_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--pmc-child", "--steps", str(Kc), "--warmup", str(Wc),
                   "--mode", args.mode, "--scene", args.scene, "--res", repr(args.res),
                   "--unique-frames", str(args.unique_frames)] + (["--hires"] if args.hires else [])
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=150)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": "rocprofv3 --pmc %s: rc %d, %d csv: %s" % (counter, r.returncode, len(files), r.stderr[-300:])}
            rows = {k: [] for k in want}
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") != counter:
                            continue
                        name = re.sub(r"^void ", "", row["Kernel_Name"])
                        name = re.sub(r"\(.*$", "", name).replace("tf::", "").strip()
                        name = re.sub(r"^k_mesh_filter<\w+>$", "k_mesh_filter", name)  # (two forms of one stage)
                        if name in rows:
                            rows[name].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            for k in want:
                v = [c for _, c in sorted(rows[k])][Wc:]  # launches behind the warm-up frames
                if not v:
                    return {"error": "no %s samples of %s" % (counter, k)}
                per_kernel[k][counter] = PMC_BYTES[counter] * sum(v) / len(v)
                per_kernel[k]["launches"] = len(v)
    except Exception as e:  # a counter pass must never take the benchmark down
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for k in want:
        per_kernel[k]["bytes_per_launch"] = per_kernel[k]["FETCH_SIZE"] + per_kernel[k]["WRITE_SIZE"]
    return {"bytes_per_step": sum(v["bytes_per_launch"] for v in per_kernel.values()),
            "kernels": per_kernel,
            "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of this command (%d frames each behind %d "
                   "warm-up frames), bytes = 2048 x FETCH_SIZE + 1024 x WRITE_SIZE per launch (KiB units; FETCH_SIZE counts "
                   "128-B requests as 64 B on gfx950 -- calibration in profiles/r2/README.md), one launch of each kernel per step; "
                   "L2-miss traffic: Infinity-Cache hits are included" % (Kc, Wc)}


def host_path(args, vol, frames, poses, pinv, textured, Wm, K, n_unique):
    """The reference's calling convention: one call per frame with HOST images (MobileFusion::IntegrateFrame,
    GCFusion/MobileFusion.cpp:223-250): double-buffered pinned staging, H2D of frame f+1 overlapped with the
    kernels of frame f.  PCIe-inclusive; never the headline value."""
    n = min(K, 100)
    first = Wm + 3 * K
    idx = [(first + i) % n_unique for i in range(n)]
    for i in idx[:4]:  # warm the staging path
        f = frames[i]
        vol.integrate_frame_host(f[0], f[1], poses[i], pinv[i] if textured else None, i)
    vol.sync()
    t0 = time.perf_counter()
    for j, i in enumerate(idx):
        f = frames[i]
        vol.integrate_frame_host(f[0], f[1], poses[i], pinv[i] if textured else None, first + j)
    t1 = time.perf_counter()
    vol.sync()
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "frames": n,
            "host_call_us": 1e6 * (t1 - t0) / n, "drain_ms": 1e3 * (dt - (t1 - t0)),
            "note": "tf_integrate_frame_host: host depth + RGBA in, staged through pinned memory (helper threads share the "
                    "copy) and copied H2D (%.1f MB per frame) inside the timed region, one call per frame, one "
                    "synchronisation at the end.  The entry point runs two frames behind the caller so that a frame's "
                    "voxel update shares its launch with the selection stages of the next two (any other entry point "
                    "flushes first).  The source frames are %d distinct host arrays (%.0f MB, not cache-resident)"
                    % (8e-6 * frames[0][0].size, len(frames), 8e-6 * frames[0][0].size * len(frames))}


def keyframe_group(args, cam, res, frames, d_depth, d_rgba, poses, n_unique, device):
    """MobileFusion::TSDFFusion's integration loop (GCFusion/MobileFusion.cpp:165-217): PrepareIntersectChunks for the
    keyframe, its depth + colour, then the six local frames depth-only over the SAME list, FinalizeIntegrateChunks.
    Two volumes get the same groups: one integrates the local frames with six tf_integrate calls, the other with one
    tf_integrate_depth_group; the kernel times come from HIP events around the launches (the calls themselves
    return lists and flags to the host, so their wall time is dominated by host round trips)."""
    from texturefusion_amd import capi
    n_groups, n_local = 8, 6
    vols = [capi.Volume(res, cam, max_chunks=1 << 18, max_list=1 << 17, max_coarse=1 << 20, device=device) for _ in range(2)]
    t_loc = [0.0, 0.0]
    t_kf = 0.0
    chunks = 0
    for g in range(n_groups + 1):  # group 0 warms up
        k0 = (7 * g) % max(1, n_unique - n_local - 1)
        loc = [k0 + 1 + i for i in range(n_local)]
        for w, vol in enumerate(vols):
            vol.profile_enable(["integrate"])
            vol.frame_bind_device(d_depth[k0].data_ptr(), d_rgba[k0].data_ptr(), 0)
            ids, new = vol.prepare(poses[k0])
            needs = np.zeros(len(ids), np.uint8)
            vol.integrate(poses[k0], ids, needs, 1, True, False)
            a = vol.profile_get(reset=True)["integrate"][0]
            if w == 0:
                for i in loc:
                    vol.frame_bind_device(d_depth[i].data_ptr(), 0, 0)
                    vol.integrate(poses[i], ids, needs, 1, False, False)
            else:
                vol.integrate_depth_group([d_depth[i].data_ptr() for i in loc], poses[loc], ids, needs, 1)
            b = vol.profile_get(reset=True)["integrate"][0]
            vol.finalize(ids, needs, new)
            if g:
                t_loc[w] += b
                if w == 0:
                    t_kf += a
                    chunks += len(ids)
    for vol in vols:
        vol.close()
    us_seq, us_grp, us_kf = 1e3 * t_loc[0] / n_groups, 1e3 * t_loc[1] / n_groups, 1e3 * t_kf / n_groups
    return {"frames_per_group": 1 + n_local, "chunks_per_list": chunks / n_groups,
            "keyframe_colour_depth_us": us_kf,
            "local_frames_one_by_one_us": us_seq, "local_frames_grouped_us": us_grp,
            "kernel_frames_per_s": {"one_by_one": (1 + n_local) / ((us_kf + us_seq) * 1e-6),
                                    "grouped": (1 + n_local) / ((us_kf + us_grp) * 1e-6)},
            "note": "HIP-event kernel time of the integration launches of a group (k_pre + k_integrate per frame, or "
                    "k_pre x 6 + k_integrate_group); selection and finalize are the same on both sides and excluded"}


def cpu_baseline(args, cam, res, frames, n_unique, textured):
    """oracle/ timed on a bounded sample of the same workload: --cpu-warmup untimed frames build up the volume
    (meshes need weight > 50), then the next --cpu-frames frames of the stream are timed."""
    from oracle import api as O
    from texturefusion_amd import synth
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    avx2 = bool(O.lib().tfo_have_avx2())
    ov.set_kernel(1 if avx2 else 0)  # AVX2 row kernel (as the reference's), bit-identical to the scalar checker
    ncpu = os.cpu_count() or 1
    oa = O.Atlas(res) if textured else None
    T = args.cpu_threads if args.cpu_threads > 0 else max(1, ncpu - 2)  # chisel::parallel_for: hardware_concurrency - 2,
    ov.set_threads(T)                                                    # groups of >= 1000 items (applied per call inside)

    def step(k):
        f = frames[k % n_unique]
        if textured:
            ov.frame_textured(oa, f[0], f[1], f[3], synth.pose_inverse16(f[3]), k)
        else:
            ov.integrate_frame(f[0], f[1], f[3])

    for k in range(args.cpu_warmup):
        step(k)
    n = args.cpu_frames
    t0 = time.perf_counter()
    for k in range(args.cpu_warmup, args.cpu_warmup + n):
        step(k)
    t_all = time.perf_counter() - t0
    # TSDF-only figure of the same port (1 thread and the reference policy) on a shorter sample
    out = {
        "value": n / t_all, "unit": "frames/s", "cores": T,
        "kind": "port",
        "sample": "oracle/ C port of the reference path (%s voxel kernel, no FMA; selection scalar, 1 thread; %s) on frames "
                  "%d..%d of the same stream after %d untimed frames; parallel stages use chisel::parallel_for's policy "
                  "(hardware_concurrency - 2 = %d threads, groups of >= 1000 items); host has %d logical cores"
                  % ("AVX2 8-lane" if avx2 else "scalar",
                     "UpdateMeshes parallel, CompressMeshes / GeneratePatches / UpdateAtlas serial as in the reference" if textured
                     else "atlas off",
                     args.cpu_warmup, args.cpu_warmup + n - 1, args.cpu_warmup, T, ncpu),
        "host_cores": ncpu,
    }
    if textured:
        ov1 = O.Volume(res, O.camera_from(cam), O.default_integrator())
        ov1.set_kernel(1 if avx2 else 0)
        ov1.set_threads(T)
        n1 = max(4, n // 4)
        for k in range(4):
            f = frames[k % n_unique]
            ov1.integrate_frame(f[0], f[1], f[3])
        t0 = time.perf_counter()
        for k in range(4, 4 + n1):
            f = frames[k % n_unique]
            ov1.integrate_frame(f[0], f[1], f[3])
        out["value_tsdf_only"] = n1 / (time.perf_counter() - t0)
    return out


if __name__ == "__main__":
    main()
