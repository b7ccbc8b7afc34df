#!/usr/bin/env python3
"""Headline benchmark: RGB-D frames/s of the fused voxel-fusion unit on MI355X.

One "step" = one synthetic 640x480 RGB-D frame of the S-room stream through
prepare -> integrate(depth+colour) -> finalize (Chisel::IntegrateDepthScanColor 5-arg,
Structure/Chisel.h:453-468) at 5 mm voxels, frames already resident in HBM.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (metric/value/... + "roofline" + "cpu_baseline").
N > 1: one process per GPU, static chunk-range partition of ONE stream (slabs of the key x + y + z) -- every
rank sees every frame, selects and integrates only the chunks of its slab ("strong" scaling), and
the ranks all-gather their updated boundary chunks over RCCL every --exchange-every frames.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--res", type=float, default=0.005)
    ap.add_argument("--hires", action="store_true", help="1280x960 camera (config 4)")
    ap.add_argument("--unique-frames", type=int, default=200, help="distinct frames of the orbit kept in HBM")
    ap.add_argument("--exchange-every", type=int, default=40, help="N>1: boundary all-gather period (frames)")
    ap.add_argument("--cpu-frames", type=int, default=40, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = the reference's parallel_for policy")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the N>1 code path (partition + boundary all-gather) even with one rank (smoke test)")
    ap.add_argument("--atlas-every", type=int, default=0,
                    help="N>0: every N frames run GeneratePatches+UpdateAtlas on that frame (BASELINE configs[2])")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "TF_BENCH_DEVICE" in os.environ:  # test hook: several ranks on one GPU (with TF_BENCH_BACKEND=gloo)
        local_rank = int(os.environ["TF_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

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

    from texturefusion_amd import capi, exchange, synth
    from texturefusion_amd import partition as part

    cam = synth.Camera.hires() if args.hires else synth.Camera()
    res = np.float32(args.res)
    K, Wm = args.steps, args.warmup
    n_unique = max(1, min(args.unique_frames, K + Wm))

    # ---- synthetic stream, generated once and parked in HBM -------------------------------
    frames = [synth.room_frame(k, cam, with_quality=False) for k in range(n_unique)]
    d_depth = [torch.from_numpy(f[0]).to(dev) for f in frames]
    d_rgba = [torch.from_numpy(f[1]).to(dev) for f in frames]
    poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
    torch.cuda.synchronize()

    # N > 1: the volume runs on a torch stream so that the boundary exchange (its own stream) can be
    # ordered against it with events and overlap the next frame batch
    s_main = torch.cuda.Stream(device=dev) if multi else None
    s_xchg = torch.cuda.Stream(device=dev) if multi else None
    vol = capi.Volume(res, cam, max_chunks=1 << 19, max_list=1 << 18, device=local_rank,
                      stream=s_main.cuda_stream if multi else None)
    # atlas leg: keyframe = every --atlas-every-th frame; its per-chunk meshes are depth-derived
    # vertex clouds (meshing is the next-stage scope), its RGB / depth are already in HBM
    atlas = {}
    if args.atlas_every > 0 and not multi:
        for i in range(0, n_unique, args.atlas_every):
            rgb = torch.from_numpy(np.ascontiguousarray(frames[i][1][..., :3])).to(dev)
            ids, voff, verts, cols = synth.mesh_from_depth(frames[i][0], frames[i][1], frames[i][3], cam, res, 4)
            # the per-chunk meshes are inputs of the atlas stage: resident in HBM like the frames, and
            # the per-vertex / per-patch results stay there (tf_patches_update_device, asynchronous)
            nv = int(voff[-1])
            atlas[i] = dict(rgb=rgb, ids=ids, voff=voff, verts=verts, cols=cols,
                            kf=np.full(len(ids), i, np.int32),
                            T=np.tile(synth.pose_inverse16(frames[i][3]), (len(ids), 1)),
                            d_verts=torch.from_numpy(np.ascontiguousarray(verts, np.float32)).to(dev),
                            d_cols=torch.from_numpy(np.ascontiguousarray(cols, np.float32)).to(dev),
                            d_tc=torch.empty(max(nv, 1) * 2, dtype=torch.float32, device=dev),
                            d_tcol=torch.empty(max(nv, 1) * 3, dtype=torch.float32, device=dev),
                            d_po=torch.empty(max(len(ids), 1) * 8, dtype=torch.int32, device=dev))
            vol.keyframe_cache_device(i, rgb.data_ptr(), d_depth[i].data_ptr())
        torch.cuda.synchronize()
    if multi:
        # Ownership key x + y + z: axis-aligned walls and floors are cut diagonally, so no rank holds a
        # whole wall.  Slab edges split the chunk keys of eight sample frames spread over the orbit into
        # equally populated slabs; every rank computes them itself (selection is deterministic), so
        # nothing has to be communicated.
        axis = (1, 1, 1)
        keys = []
        for i in range(0, 200, 25):
            f = frames[i % n_unique] if i < n_unique else synth.room_frame(i, cam, with_quality=False)
            vol.frame_upload(f[0], None, None)
            ids_s, _ = vol.prepare(f[3])
            keys.append(part.key_of(ids_s, axis))
        vol.reset()
        edges = part.balanced_edges(np.concatenate(keys), max(world, 2) if args.force_exchange and world == 1 else world)
        lo, hi = edges[rank], edges[rank + 1]
        if args.force_exchange and world == 1:
            lo, hi = edges[1] - 12, edges[1] + 12  # a real interior slab so that faces exist and get packed
        vol.set_partition(lo, hi, axis)
        rec_cap = 1 << 14
        send = [torch.empty(rec_cap * capi.TF_BOUNDARY_RECORD_BYTES, dtype=torch.uint8, device=dev) for _ in range(2)]
        cnt = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2)]
        keep = []  # received buffers stay referenced until the unpack kernels that read them have run

    def start_exchange(slot):
        """Pack the boundary chunks updated since the last exchange (asynchronous, on the volume's stream)."""
        vol.boundary_pack_async(send[slot].data_ptr(), rec_cap, cnt[slot].data_ptr())
        ev = torch.cuda.Event()
        ev.record(s_main)
        return slot, ev

    def finish_exchange(pending):
        """All-gather over RCCL (xGMI) on the exchange stream: counts first, then payloads; host waits of
        this step overlap the frame batch that is already enqueued on the volume's stream."""
        slot, ev = pending
        with torch.cuda.stream(s_xchg):
            s_xchg.wait_event(ev)
            got = exchange.allgather_records(send[slot], cnt[slot], synchronize=False)
            done = torch.cuda.Event()
            done.record(s_xchg)
        s_main.wait_event(done)
        for r, (buf, m) in enumerate(got):
            if r != rank and m:
                vol.boundary_unpack(buf.data_ptr(), m)
        keep.append(got)
        if len(keep) > 2:
            keep.pop(0)

    def run(first, count, timed):
        """Frames [first, first+count) of the stream (cyclic over the unique frames)."""
        idx = [(first + i) % n_unique for i in range(count)]
        if not multi and atlas:
            b0 = 0
            for j, i in enumerate(idx):
                if i in atlas:  # flush the frames up to and including the keyframe, then texture it
                    sub = idx[b0:j + 1]
                    vol.integrate_frames_device([d_depth[k].data_ptr() for k in sub],
                                                [d_rgba[k].data_ptr() for k in sub], poses[sub])
                    a = atlas[i]
                    rc, _, _ = vol.patches_update_device(a["ids"], a["kf"], a["T"], a["voff"], a["d_verts"].data_ptr(),
                                                         a["d_cols"].data_ptr(), a["d_tc"].data_ptr(),
                                                         a["d_tcol"].data_ptr(), a["d_po"].data_ptr())
                    if rc != 0:
                        raise SystemExit("atlas full")
                    b0 = j + 1
            sub = idx[b0:]
            if sub:
                vol.integrate_frames_device([d_depth[k].data_ptr() for k in sub],
                                            [d_rgba[k].data_ptr() for k in sub], poses[sub])
        elif not multi:
            vol.integrate_frames_device([d_depth[i].data_ptr() for i in idx],
                                        [d_rgba[i].data_ptr() for i in idx], poses[idx])
        else:
            pending = None
            for k, b in enumerate(range(0, count, args.exchange_every)):
                sub = idx[b:b + args.exchange_every]
                vol.integrate_frames_device([d_depth[i].data_ptr() for i in sub],
                                            [d_rgba[i].data_ptr() for i in sub], poses[sub])
                nxt = start_exchange(k & 1)
                if pending is not None:
                    finish_exchange(pending)  # exchange of the previous batch, hidden behind this one
                pending = nxt
            if pending is not None:
                finish_exchange(pending)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up, then the timed region ---------------------------------------------------
    run(0, Wm, False)
    vol.sync()
    barrier()
    t0 = time.perf_counter()
    run(Wm, K, True)
    t_enq = time.perf_counter() - t0  # host time to enqueue the timed region (launches are asynchronous)
    barrier()
    dt = time.perf_counter() - t0
    vol.sync()  # surfaces any device-side capacity error of the timed region
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- instrumented repeat of the same K frames: HIP events (on the handle's stream) around
    # every launch of the dominant kernel.  Kept out of the timed region above because the event
    # pairs cost ~10 % throughput; the per-launch duration is what the roofline needs.
    prof = None
    dt_instr = None
    if not args.no_roofline:
        vol.profile_enable(["integrate"])
        barrier()
        t1 = time.perf_counter()
        run(Wm + K, K, False)
        barrier()
        dt_instr = time.perf_counter() - t1
        prof = vol.profile_get(reset=True)
        vol.profile_enable([])
        vol.sync()

    out = {
        "metric": "RGB-D frames/s (TSDF integrate, depth+colour, fused prepare->integrate->finalize)",
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
            "workload": "S-room orbit stream, %dx%d RGB-D, %.0f mm voxels, 8^3 chunks, TSDF+colour integrate, %s; "
                        "frames resident in HBM"
                        % (cam.width, cam.height, 1e3 * float(res),
                           ("atlas patch update every %d frames (BASELINE.json configs[2])" % args.atlas_every)
                           if atlas else "atlas off (BASELINE.json configs[1])"),
            "frames_in_hbm": n_unique,
            "parallelism": ("1 GPU" if world == 1 else
                            "%d ranks, ChunkID.x slab partition of one stream, boundary all-gather every %d frames"
                            % (world, args.exchange_every)),
        },
    }

    # ---- roofline of the dominant kernel (k_integrate) ------------------------------------
    if rank == 0 and prof is not None:
        # Algorithmic bytes per launch (SURVEY.md s.8d / DESIGN.md): 128 B per rewritten TSDF row
        # (8 voxels x {sdf,weight} read+write), 128 B per rewritten colour row, plus one read of
        # the depth and RGBA images.  Row counts depend only on (depth, pose): replay the timed
        # frames untimed and read the exact integers back.
        ka_ms, ka_n = prof["integrate"]
        algo = 0
        idx = [(Wm + K + i) % n_unique for i in range(K)]
        rows_cache = {}
        for i in sorted(set(idx)):
            vol.frame_bind_device(d_depth[i].data_ptr(), d_rgba[i].data_ptr(), 0)
            vol.integrate_frame(poses[i], True)
            st = vol.stats()
            rows_cache[i] = (st.rows_tsdf, st.rows_color, st.n_selected, st.n_updated)
        for i in idx:
            rt, rc, _, _ = rows_cache[i]
            algo += 128 * rt + 128 * rc + 8 * cam.width * cam.height
        per_launch = algo / max(ka_n, 1)
        avg_s = 1e-3 * ka_ms / max(ka_n, 1)
        achieved = per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
        traffic, traffic_src = pmc_traffic(args)
        out["roofline"] = {
            "bound": "hbm", "kernel": "k_frame<color> = K-A(f) + K-C(f+1) + K-B(f+2) block ranges; bytes counted for K-A only", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": per_launch, "avg_launch_us": 1e6 * avg_s, "launches": ka_n,
            "instrumented_ms_per_step": 1e3 * dt_instr / K,
            "chunks_selected_avg": float(np.mean([rows_cache[i][2] for i in idx])),
            "chunks_updated_avg": float(np.mean([rows_cache[i][3] for i in idx])),
        }

    # ---- CPU baseline: the oracle (scalar port of the reference path) on the host cores ------
    if rank == 0 and args.cpu_frames > 0:
        out["cpu_baseline"] = cpu_baseline(args, cam, res, frames, Wm, n_unique, atlas)

    if rank == 0:
        print(json.dumps(out))
    vol.close()
    if multi:
        dist.destroy_process_group()


def pmc_traffic(args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of this very
    command (FETCH_SIZE and WRITE_SIZE need separate --pmc passes, so they cannot be collected inside a
    timed run): raw FETCH_SIZE + WRITE_SIZE, KiB -> bytes.  The gfx950 x2 correction of FETCH_SIZE applies
    to 16-B-per-lane streaming reads only; this kernel reads 8 B and 4 B per lane, so the raw value is
    reported (a lower bound).  None when the workload is not the profiled one."""
    if args.hires or args.atlas_every or args.gpus > 1 or abs(args.res - 0.005) > 1e-9:
        return None, None
    path = os.path.join(ROOT, "profiles", "r1", "05_pmc_k_frame.csv")
    try:
        vals = {}
        for line in open(path).read().splitlines()[1:]:
            k, v, _ = line.split(",")
            vals[k] = float(v)
        return (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "profiles/r1/05_pmc_k_frame.csv (raw FETCH_SIZE + WRITE_SIZE, separate --pmc passes)"
    except Exception:
        return None, None


def cpu_baseline(args, cam, res, frames, Wm, n_unique, atlas=None):
    """oracle/ timed on a bounded sample of the same workload: same warm-up frames (untimed),
    then the next --cpu-frames frames of the stream."""
    from oracle import api as O
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    avx2 = bool(O.lib().tfo_have_avx2())
    ov.set_kernel(1 if avx2 else 0)  # AVX2 row kernel (as the reference's), bit-identical to the scalar checker
    ncpu = os.cpu_count() or 1
    for i in range(min(Wm, 5)):  # a short warm-up bounds the CPU time; state is first-touch either way
        f = frames[i % n_unique]
        ov.integrate_frame(f[0], f[1], f[3])
    n = args.cpu_frames
    oa = O.Atlas(res) if atlas else None   # --atlas-every: GeneratePatches + UpdateAtlas on the keyframes, serial as in the reference
    seen = {}
    n_kf = 0
    sel = []
    # thread policy of chisel::parallel_for (threading/Threading.h:36-54)
    t_all = 0.0
    threads_used = []
    for i in range(n):
        f = frames[(Wm + i) % n_unique]
        if args.cpu_threads > 0:
            T = args.cpu_threads
        else:
            nsel = sel[-1] if sel else 8000
            T = max(1, min(ncpu - 2, -(-nsel // 1000)))
        ov.set_threads(T)
        threads_used.append(T)
        t0 = time.perf_counter()
        _, ns = ov.integrate_frame(f[0], f[1], f[3])
        fi = (Wm + i) % n_unique
        if atlas and fi in atlas:
            a = atlas[fi]
            rgb = np.ascontiguousarray(f[1][..., :3])
            Cc = O.camera_from(cam)
            tls = []
            for p in range(len(a["ids"])):
                key = tuple(int(x) for x in a["ids"][p])
                if key not in seen:
                    rc, seen[key] = oa.alloc()
                tls.append(seen[key])
            oa.patches_batch(tls, a["voff"], a["verts"], a["cols"], a["T"], rgb, f[0], Cc)
            n_kf += 1
        t_all += time.perf_counter() - t0
        sel.append(ns)
    # single-thread figure on a shorter sample
    ov1 = O.Volume(res, O.camera_from(cam), O.default_integrator())
    ov1.set_kernel(1 if avx2 else 0)
    ov1.set_threads(1)
    n1 = max(1, n // 4)
    t1 = 0.0
    for i in range(n1):
        f = frames[(Wm + i) % n_unique]
        t0 = time.perf_counter()
        ov1.integrate_frame(f[0], f[1], f[3])
        t1 += time.perf_counter() - t0
    return {
        "value": n / t_all, "unit": "frames/s", "cores": int(round(float(np.mean(threads_used)))),
        "kind": "port",
        "sample": "oracle/ C port of the reference path (%s voxel kernel, no FMA; selection scalar, 1 thread) on "
                  "frames %d..%d of the same S-room stream; integrate threads = reference parallel_for policy "
                  "min(hw-2, ceil(N/1000)); host has %d logical cores%s"
                  % ("AVX2 8-lane" if avx2 else "scalar", Wm, Wm + n - 1, ncpu,
                     ("; %d keyframes of the sample also ran GeneratePatches + UpdateAtlas (1 thread, as the reference)" % n_kf)
                     if atlas else ""),
        "value_1thread": n1 / t1,
        "host_cores": ncpu,
    }


if __name__ == "__main__":
    main()
