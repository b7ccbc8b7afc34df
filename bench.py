#!/usr/bin/env python3
"""Headline benchmark: RGB-D frames/s of the textured per-frame unit on MI355X (BASELINE.json configs[2]).

One "step" = one synthetic 640x480 RGB-D frame of the S-room stream, handed over as HOST images
(the reference's calling convention, GCFusion/MobileFusion.cpp:223-250), through
  H2D copy of depth + RGBA out of the caller's arrays (inside the timed region; the two arrays that hold the orbit's images are
  registered once with tf_host_register before the pre-roll, as a caller with a fixed set of frame buffers does at start-up:
  no staging copy, one host thread, the call returns when its upload is through; "staged_host_frames" = the same frames
  through the library's pinned staging slots -- a CPU copy by helper threads + an asynchronous upload; --staged-host-frames
  makes that the timed path)
  prepare -> integrate(depth+colour) -> finalize          Chisel::IntegrateDepthScanColor 5-arg, Structure/Chisel.h:453-468
  -> UpdateMeshes -> CompressMeshes                        over that frame's dirty chunks (Structure/Chisel.h:479-481, Chisel.cpp:112-147)
  -> GeneratePatches(label = this frame) -> UpdateAtlas    Structure/Chisel.cpp:149-196
at 5 mm voxels (tf_integrate_frame_host), nothing copied back, no host synchronisation inside the timed region.
--mode tsdf runs configs[1] (atlas off).

Frame windows (ORBIT = 200 frames = one turn of the camera; every window starts at the same orbit position):
  pre-roll   one full orbit, untimed: every later frame meets a steady-state volume (~2.9 k meshes per frame)
  warm-up    W frames through the timed entry point
  timed      K frames -> "value", "ms_per_step" (wall clock, barrier + device synchronisation on both sides)
  resident   the same K orbit positions one turn later, frames already in HBM (no H2D) -> "resident"
  staged     the same positions as host frames from unregistered arrays (staging copy by helper threads) -> "staged_host_frames"
  rgb host   the same positions as host frames with Frame::rgb (3 B per pixel) instead of the RGBA staging image -> "rgb_host_frames"
  events     the same positions again with HIP events around every launch -> per-kernel times
  replay     the same positions again, frame by frame, reading back the exact integer counts -> algorithmic bytes
A rocprofv3 --kernel-trace child pass and two --pmc child passes of the same command (same pre-roll / warm-up /
steps: the same frames, the launches of the timed window sliced out) give the profiler's kernel durations and
the HBM-side traffic.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (metric/value/... + "roofline" + "cpu_baseline").
N > 1: one process per GPU, static chunk-range partition of ONE stream (slabs of the key x + y + z) -- every
rank sees every frame, selects / integrates / meshes / textures only the chunks of its slab ("strong"
scaling); the ranks exchange the chunks of their ghost bands over RCCL.
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
    ap.add_argument("--unique-frames", type=int, default=200, help="frames of one camera orbit (distinct host / HBM images)")
    ap.add_argument("--no-preroll", action="store_true", help="start the warm-up on an empty volume (lighter frames)")
    ap.add_argument("--resident-headline", action="store_true",
                    help="report the HBM-resident rate (no H2D) as `value` instead of the host-frames rate")
    ap.add_argument("--exchange-every", type=int, default=40, help="N>1, --mode tsdf: boundary exchange period (frames)")
    ap.add_argument("--exchange-cap", type=int, default=1024,
                    help="N>1: records per rank of the fixed-capacity boundary exchange (8 KiB each)")
    ap.add_argument("--cpu-frames", type=int, default=48, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-warmup", type=int, default=24, help="untimed frames that build up the CPU baseline's volume")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = the reference's parallel_for policy")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the rocprofv3 child passes (kernel trace, FETCH_SIZE, WRITE_SIZE)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="of the rocprofv3 child passes run only the kernel trace (no FETCH_SIZE / WRITE_SIZE passes)")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side runs of the default line: configs[1] (--mode tsdf) and configs[3] (--scene big --hires) as "
                         "child processes of their own, reported as scalars (side.*, roofline.side_*)")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)  # the timed workload only, run under rocprofv3
    ap.add_argument("--repeats", type=int, default=5, help="N=1: further timed windows on the same orbit positions (median / min / max next to value)")
    ap.add_argument("--no-group", action="store_true", help="skip the keyframe-group (1 colour + 6 depth frames) measurement")
    ap.add_argument("--staged-host-frames", action="store_true",
                    help="the timed host frames go through the library's pinned staging slots (a CPU copy per frame by a pool of "
                         "helper threads) instead of straight out of the caller's arrays, registered once with tf_host_register")
    ap.add_argument("--max-chunks-log2", type=int, default=0, help="chunk pool of 2^n slots (default: 19 room / 21 hall)")
    ap.add_argument("--no-independent", action="store_true",
                    help="N>1: skip the independent-streams (one whole volume per GPU) and sharded keyframe-unit figures")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N>1: the boundary exchange serially in the stream (round 4's order) instead of next to the interior mesh pass")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the N>1 code path (partition + boundary exchange) even with one rank (smoke test)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------
# the synthetic stream: generated once per parameter set, kept in /tmp so that the profiler child passes (and a
# second run on the same box) load it instead of generating it again
# ---------------------------------------------------------------------------------------------------------
def cgroup_cpu_quota():
    """CPUs' worth of CFS quota of this container (cgroup v2 cpu.max / v1 cfs_quota_us), or None when unlimited"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def load_stream(args, cam):
    from texturefusion_amd import synth
    n = args.unique_frames
    key = "%s_%dx%d_%d" % (args.scene, cam.width, cam.height, n)
    base = os.path.join("/tmp", "tf_bench_stream_" + key)
    names = [base + suf for suf in ("_depth.npy", "_rgba.npy", "_pose.npy")]
    try:
        if all(os.path.exists(f) for f in names):
            depth, rgba, pose = (np.load(f) for f in names)
            if depth.shape == (n, cam.height, cam.width) and rgba.shape == (n, cam.height, cam.width, 4):
                return depth, rgba, pose
    except Exception:
        pass
    depth = np.empty((n, cam.height, cam.width), np.float32)
    rgba = np.empty((n, cam.height, cam.width, 4), np.uint8)
    pose = np.empty((n, 3, 4), np.float32)
    for k in range(n):
        if args.scene == "big":
            f = synth.room_frame(k, cam, half=(4.0, 3.0, 4.0), radius=0.5, with_quality=False)
        else:
            f = synth.room_frame(k, cam, with_quality=False)
        depth[k], rgba[k], pose[k] = f[0], f[1], np.asarray(f[3], np.float32).reshape(3, 4)
    try:
        for f, a in zip(names, (depth, rgba, pose)):
            tmp = f + ".%d.tmp.npy" % os.getpid()
            np.save(tmp, a)
            os.replace(tmp, f)
    except Exception:
        pass
    return depth, rgba, pose


def pin_to_numa_node(gpu_index=0):
    """Keep this process (frame arrays, pinned staging buffers, the copy helper threads of the library) on ONE NUMA node
    -- the GPU's, when sysfs tells, else the one the process is running on: the GPU boxes have two sockets, and a main
    thread the scheduler moves to the other socket halves the rate of the staging copy of tf_integrate_frame_host
    (measured: 37 vs 123 us per frame from run to run).  Returns the original affinity (restored for the CPU baseline,
    which uses every core) and a description."""
    import glob
    try:
        orig = os.sched_getaffinity(0)
        nodes = {}
        for d in glob.glob("/sys/devices/system/node/node[0-9]*"):
            cpus = set()
            for part in open(os.path.join(d, "cpulist")).read().strip().split(","):
                a, _, b = part.partition("-")
                if a:
                    cpus.update(range(int(a), int(b or a) + 1))
            nodes[int(os.path.basename(d)[4:])] = cpus
        if len(nodes) < 2:
            return orig, None
        node, why = None, ""
        try:  # the NUMA node of the gpu_index-th AMD display device
            cards = sorted(c for c in glob.glob("/sys/class/drm/card[0-9]*") if "-" not in os.path.basename(c)
                           and open(os.path.join(c, "device", "vendor")).read().strip() == "0x1002")
            if cards:
                n = int(open(os.path.join(cards[min(gpu_index, len(cards) - 1)], "device", "numa_node")).read())
                if n in nodes:
                    node, why = n, "the GPU's "
        except Exception:
            pass
        if node is None:
            with open("/proc/self/stat") as fh:  # field 39: the CPU this thread last ran on
                cpu = int(fh.read().rsplit(")", 1)[1].split()[36])
            node = next((n for n, c in nodes.items() if cpu in c), None)
        want = nodes.get(node, set()) & orig
        if len(want) >= 4 and want != orig:
            os.sched_setaffinity(0, want)
            return orig, "%sNUMA node %d (%d of %d cpus)" % (why, node, len(want), len(orig))
        return orig, None
    except Exception:
        return None, None


def under_profiler():
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def visible_gpus():
    """HIP devices this process could use, WITHOUT initialising the GPU (the launcher below must stay a process that
    never touched it: its children are started with fork + exec).  The census of KFD topology nodes in sysfs needs no
    runtime and no torch import; HIP_/ROCR_VISIBLE_DEVICES narrow it the way the runtime would.  Only when sysfs has no
    KFD tree at all (not a ROCm host) is torch asked -- device_count() reads the driver's list and makes no context on
    this image, but that is a property of the image, not a contract."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if nodes:
        n = 0
        for p in nodes:
            try:
                props = dict(l.split()[:2] for l in open(p).read().splitlines() if len(l.split()) >= 2)
            except OSError:
                continue
            n += 1 if int(props.get("simd_count", "0")) > 0 else 0
        for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            lst = os.environ.get(var)
            if lst is not None:
                n = min(n, len([x for x in lst.split(",") if x.strip() != ""]))
        return n
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): this process becomes the
    launcher.  It starts N fresh child processes of this very command -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one
    per GPU -- BEFORE it makes any GPU call itself (it never does), passes rank 0's JSON line through, and fails when
    fewer than N devices are visible or any rank fails: an N = 1 number is never printed under an N > 1 flag."""
    import socket
    import subprocess
    n = args.gpus
    one_gpu_hook = "TF_BENCH_DEVICE" in os.environ  # (tests: several ranks on ONE device, blocks over gloo)
    have = visible_gpus()
    if not one_gpu_hook and have < n:
        sys.stderr.write("bench.py: --gpus %d but only %d HIP device(s) visible -- not running (a smaller job under this flag "
                         "would be a wrong number)\n" % (n, have))
        return 3
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:  # a free rendezvous port on the loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    procs = []
    rc = 0
    with tempfile.TemporaryFile(mode="w+") as out0:  # rank 0's stdout (the other ranks print nothing there)
        try:
            for r in range(n):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", str(port)), TF_BENCH_SPAWNED="1")
                env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL between processes needs it on this pool)
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=ROOT,
                                              stdout=out0 if r == 0 else subprocess.DEVNULL))
            deadline = time.time() + float(os.environ.get("TF_BENCH_SPAWN_TIMEOUT", "1500"))
            while any(p.poll() is None for p in procs):
                bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
                if bad or time.time() > deadline:  # one rank gone (or the job stuck): the others would wait in a collective
                    rc = (bad[0][1] if bad else 4) or 1
                    break
                time.sleep(0.05)
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    sys.stderr.write("bench.py: rank %d exited with %s\n" % (r, p.returncode))
                    rc = rc or int(p.returncode) or 1
        finally:
            for p in procs:  # (exactly the processes started here)
                if p.poll() is None:
                    p.kill()
                    p.wait()
        out0.seek(0)
        text = out0.read()
    lines = [l for l in text.splitlines() if l.startswith("{")]
    if rc == 0 and len(lines) != 1:
        sys.stderr.write("bench.py: rank 0 printed %d JSON lines\n" % len(lines))
        rc = 5
    if rc == 0:
        d = json.loads(lines[0])
        if d.get("n_gpus") != n:
            sys.stderr.write("bench.py: rank 0 reports n_gpus = %r under --gpus %d\n" % (d.get("n_gpus"), n))
            rc = 6
    if rc == 0:
        print(lines[0])
    return rc


class Job:
    """One rank's run of the benchmark: the synthetic stream (host + HBM copies), the volume, the transport of an N > 1 run, and
    `pos`, the position in the stream.  main() calls the phases in order; every window of the line (timed, repeats, resident,
    the other host-frame entry points, events) is one method that starts at the timed window's orbit position."""

    # ---- phase 0: environment, affinity, the stream on the host (CPU only) ------------------------------------------
    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if "TF_BENCH_DEVICE" in os.environ:  # test hook: several ranks on one GPU (with TF_BENCH_BACKEND=gloo)
            self.local_rank = int(os.environ["TF_BENCH_DEVICE"])
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, self.world))
        if self.world > 1:
            # N ranks share the container's CPU quota (16 CPUs on the 1-GPU boxes): a rank's host thread sleeps between polls of
            # "is my upload / my staging slot through" instead of spinning, so that eight ranks and their RCCL proxy threads do
            # not exhaust a period's budget and get the whole job frozen (profiles/r4/README.md, "Host frames")
            os.environ.setdefault("TF_HOST_POLL_SLEEP_US", "20")
        self.orig_affinity, self.numa_note = pin_to_numa_node(self.local_rank)  # before any buffer is allocated
        from texturefusion_amd import synth
        self.synth = synth
        self.cam = synth.Camera.hires() if args.hires else synth.Camera()
        self.h_depth, self.h_rgba, self.h_pose = load_stream(args, self.cam)  # CPU only: before the child passes, before any GPU use
        if args.child:
            args.no_roofline = args.no_group = True
            args.cpu_frames = 0
        self.side = self.prof_child = None
        self.pos = 0

    # ---- phase 1: the processes that must run BEFORE this one touches the GPU ------------------------------------------
    def children(self):
        # The other single-GPU configurations of BASELINE.json come FIRST, each a bench.py process of its own (the default line
        # carries their headline figures as scalars); then the profiler child passes of this line.  All of them before this process
        # touches the GPU -- and in this order because VRAM a process leaves behind is wiped in the background with the DMA
        # engines the host-frame uploads use: the hall's tens of GB are wiped while the profiler passes run,
        # and what is left ahead of this line's own windows is what every earlier round's line had.
        args, world = self.args, self.world
        default_line = (world == 1 and args.mode == "textured" and args.scene == "room" and not args.hires and not args.child
                        and not args.force_exchange and not args.resident_headline and not args.no_preroll)
        if default_line and not args.no_side and not args.no_roofline and not under_profiler():
            self.side = side_runs(args)
        if world == 1 and not (args.no_pmc or args.child or args.no_roofline or args.force_exchange):
            self.prof_child = {"error": "running under a profiler"} if under_profiler() else child_passes(args)

    # ---- phase 2: the device, the process group, the stream once more in HBM -------------------------------------------
    def device(self):
        args = self.args
        import torch  # plumbing: device memory for the frames, barrier/collectives, device sync
        import torch.distributed as dist
        torch.set_num_threads(1)  # (no CPU tensor work here; an idle OpenMP pool would only burn the container's CPU quota)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
        torch.cuda.set_device(self.local_rank)
        self.torch, self.dist = torch, dist
        self.dev = dev = torch.device("cuda", self.local_rank)
        self.multi = self.world > 1 or args.force_exchange
        if self.multi:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29513")
            backend = os.environ.get("TF_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
            if backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=self.rank, world_size=self.world)
        from texturefusion_amd import capi
        from texturefusion_amd import partition as part
        self.capi, self.part = capi, part
        self.res = np.float32(args.res)
        self.K, self.Wm = args.steps, args.warmup
        self.ORBIT = self.n_unique = n_unique = args.unique_frames
        self.textured = args.mode == "textured"
        # the stream once more in HBM (pre-roll, resident / event / replay passes)
        self.d_depth = [torch.from_numpy(self.h_depth[k]).to(dev) for k in range(n_unique)]
        self.d_rgba = [torch.from_numpy(self.h_rgba[k]).to(dev) for k in range(n_unique)]
        self.poses = self.h_pose.reshape(n_unique, 12).astype(np.float32)
        self.pinv = np.stack([self.synth.pose_inverse16(self.h_pose[k]) for k in range(n_unique)]).astype(np.float32)
        torch.cuda.synchronize()
        # (addresses of the frames' arrays, taken once: the driver loop is Python, and turning four numpy arrays into ctypes
        # pointers costs ~10 us per call -- a tenth of a step -- that a C++ caller does not pay)
        self.a_depth = [self.h_depth[i].ctypes.data for i in range(n_unique)]
        self.a_rgba = [self.h_rgba[i].ctypes.data for i in range(n_unique)]
        self.a_pose = [self.poses[i].ctypes.data for i in range(n_unique)]
        self.a_pinv = [self.pinv[i].ctypes.data for i in range(n_unique)]
        self.quota = cgroup_cpu_quota()

    # ---- phase 3: the volume, the caller's buffers ------------------------------------------------------------------
    def volume(self):
        args, torch = self.args, self.torch
        self.s_main = torch.cuda.Stream(device=self.dev) if self.multi else None
        self.big = big = args.scene == "big"
        pool = (1 << args.max_chunks_log2) if args.max_chunks_log2 > 0 else ((1 << 21) if big else (1 << 19))
        # (mesh_blocks: the library's default gives every pool slot a mesh block, as the reference's allMeshes can hold one per
        # chunk; the bench knows its scenes -- under a third of a scanned scene's chunks lie on a surface -- and takes a quarter,
        # which keeps the volumes at the footprint every earlier round measured: 2.7 GB instead of 10.7 GB of store for the room)
        self.vol = self.capi.Volume(self.res, self.cam, max_chunks=pool, max_list=(1 << 20) if big else (1 << 18), mesh_blocks=pool // 4,
                                    max_coarse=1 << 22 if big else 1 << 20, device=self.local_rank,
                                    stream=self.s_main.cuda_stream if self.multi else None)
        # The caller's frame buffers are registered once, as a caller with a fixed set of image buffers does at start-up; host
        # frames then go up straight out of them (no staging copy, ONE host thread, the call returns when the upload is through).
        # --staged-host-frames: the copy through the library's pinned slots (any caller buffer, nothing registered) -- faster on
        # a quiet host (its upload is asynchronous), but its eight copy threads must all be scheduled promptly: on the shared
        # hosts of the GPU boxes (load average 20-40, a CFS quota of 16 CPUs) the same build measured 59-116 us per TSDF-only
        # frame from box to box, against 80-82 us with registered buffers on every one of them (profiles/r4/README.md).
        self.host_registered = False
        if not args.staged_host_frames:
            try:
                self.vol.host_register(self.h_depth)  # (the two arrays that hold the orbit's images)
                self.vol.host_register(self.h_rgba)
                self.host_registered = True
            except Exception as e:  # (e.g. a locked-memory limit: the staging path works everywhere)
                print("bench: tf_host_register failed (%r): host frames take the staging path" % (e,), file=sys.stderr)
        self.use_rccl = False
        self.cap = args.exchange_cap
        self.torch_wire = [0, 0]  # bytes sent / received through the torch transport
        self.part_spec = None

    # ---- phase 4 (N > 1, --force-exchange): the partition and the transport ----------------------------------------
    def partition(self):
        args, torch, dist, vol, capi, part = self.args, self.torch, self.dist, self.vol, self.capi, self.part
        rank, world, dev, cap = self.rank, self.world, self.dev, self.cap
        # Ownership key x + y + z: axis-aligned walls and floors are cut diagonally, so no rank holds a
        # whole wall.  Slab edges split the chunk keys of eight sample frames spread over the orbit into
        # equally populated slabs; every rank computes them itself (selection is deterministic), so
        # nothing has to be communicated.
        axis = (1, 1, 1)
        keys = []
        for i in range(0, self.ORBIT, max(1, self.ORBIT // 8)):
            vol.frame_upload(self.h_depth[i], None, None)
            ids_s, _ = vol.prepare(self.h_pose[i])
            keys.append(part.key_of(ids_s, axis))
        vol.reset()
        edges = part.balanced_edges(np.concatenate(keys), max(world, 2) if args.force_exchange and world == 1 else world)
        lo, hi = edges[rank], edges[rank + 1]
        if args.force_exchange and world == 1:
            lo, hi = edges[1] - 12, edges[1] + 12  # a real interior slab so that faces exist and get packed
        vol.set_partition(lo, hi, axis)
        self.part_spec = (lo, hi, axis)
        # The exchange: fixed-capacity [count | records] blocks, no host round trip.
        # backend nccl: RCCL inside the library (tf_comm_init / tf_exchange_boundary) on the volume's stream;
        # other backends (test hook: several ranks on one GPU over gloo): the same blocks through torch.distributed.
        # (--force-exchange with one rank: the in-library path all the same -- RCCL with a single rank: pack, the exchange's
        # second stream, unpack, the interior / boundary mesh passes; nothing travels)
        self.use_rccl = (world > 1 or args.force_exchange) and os.environ.get("TF_BENCH_BACKEND", "nccl") == "nccl"
        if self.use_rccl:
            uid = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
            if world > 1:
                dist.broadcast(uid, 0)
            vol.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()))
            # (neighbour send / receive pairs need every slab above the lowest to hold its own ghost band; the library
            # checks that itself at the first exchange and falls back to the all-gather form on every rank otherwise)
            if self.textured:
                vol.comm_exchange_every_frame(cap)
                if args.no_overlap:
                    vol.comm_exchange_overlap(False)  # (A/B: the exchange in the stream between the voxel update and the mesher)
        else:
            bb = capi.boundary_block_bytes(cap)
            self.blk = torch.zeros(bb, dtype=torch.uint8, device=dev)
            self.allb = torch.zeros(bb * max(world, 1), dtype=torch.uint8, device=dev)
            self.blk_dn = torch.zeros(bb, dtype=torch.uint8, device=dev)
            self.blk_up = torch.zeros(bb, dtype=torch.uint8, device=dev)

    # ---- the stream's moving parts ------------------------------------------------------------------------------------
    def torch_exchange(self, join_dirty):
        """[count | records] blocks through torch.distributed on the volume's stream (no .item(), no host wait)."""
        torch, dist, vol = self.torch, self.dist, self.vol
        with torch.cuda.stream(self.s_main):
            vol.boundary_pack_block(self.blk.data_ptr(), self.cap)
            if self.world > 1:
                dist.all_gather_into_tensor(self.allb, self.blk)
            else:
                self.allb.copy_(self.blk)
            vol.boundary_unpack_blocks(self.allb.data_ptr(), max(self.world, 1), self.rank, self.cap, join_dirty=join_dirty)

    def torch_exchange_sized(self):
        """The per-frame exchange of the textured unit through torch.distributed in its SIZED neighbour form: the four
        block capacities come from the frame's own selection (tf_boundary_band_bounds: the same numbers on both ends of
        every transfer), so a block is as long as what the frame can have changed."""
        from texturefusion_amd import exchange
        torch, vol, capi, rank, world, dev = self.torch, self.vol, self.capi, self.rank, self.world, self.dev
        sd, su, rb, ra = vol.boundary_band_bounds(self.cap)
        with torch.cuda.stream(self.s_main):
            vol.boundary_pack_bands2(self.blk_dn.data_ptr(), sd, self.blk_up.data_ptr(), su)
            nd, nu = capi.boundary_block_bytes(sd), capi.boundary_block_bytes(su)
            if world > 1:
                self.s_main.synchronize()  # (gloo moves host copies; RCCL inside the library needs none of this)
                below, above = exchange.neighbour_exchange_sized(self.blk_dn[:nd].cpu(), self.blk_up[:nu].cpu(),
                                                                 capi.boundary_block_bytes(rb), capi.boundary_block_bytes(ra))
                below, above = below.to(dev), above.to(dev)
                self.torch_wire[0] += (nd if rank > 0 else 0) + (nu if rank + 1 < world else 0)
                self.torch_wire[1] += (below.numel() if rank > 0 else 0) + (above.numel() if rank + 1 < world else 0)
            else:
                below = torch.zeros(16, dtype=torch.uint8, device=dev)
                above = torch.zeros(16, dtype=torch.uint8, device=dev)
            vol.boundary_unpack_pair(below.data_ptr(), rb, above.data_ptr(), ra, join_dirty=True)
            self.s_main.synchronize()  # (below / above are temporaries)

    def run(self, first, count, ahead=2):
        """Frames [first, first+count) of the stream (cyclic over the orbit), images already in HBM; the next `ahead`
        frames go through their selection stages too, so that a following run(first + count, ...) starts primed."""
        vol, d_depth, d_rgba, poses, pinv = self.vol, self.d_depth, self.d_rgba, self.poses, self.pinv
        idx = [(first + i) % self.n_unique for i in range(count + ahead)]
        dd = [d_depth[i].data_ptr() for i in idx]
        dr = [d_rgba[i].data_ptr() for i in idx]
        if self.textured and (not self.multi or self.use_rccl):
            vol.stream_frames_textured_device(dd, dr, poses[idx], pinv[idx], first, n_ahead=ahead)  # N>1: exchange inside
        elif not self.multi:
            vol.stream_frames_device(dd, dr, poses[idx], n_ahead=ahead)
        elif self.textured:  # torch transport: voxel update, exchange, texture stage -- frame by frame
            for j in range(count):
                sub = idx[j:j + 1 + min(2, count + ahead - j - 1)]
                vol.stream_frames_device([d_depth[i].data_ptr() for i in sub], [d_rgba[i].data_ptr() for i in sub],
                                         poses[sub], n_ahead=len(sub) - 1)
                self.torch_exchange_sized()
                vol.texture_frame_device(pinv[idx[j]], first + j)
        else:  # TSDF only: batches of --exchange-every frames, one exchange behind each
            every = self.args.exchange_every
            for b in range(0, count, every):
                e = min(b + every, count)
                sub = idx[b:e + (ahead if e == count else 0)]
                vol.stream_frames_device([d_depth[i].data_ptr() for i in sub], [d_rgba[i].data_ptr() for i in sub],
                                         poses[sub], n_ahead=len(sub) - (e - b))
                if self.use_rccl:
                    vol.exchange_boundary(self.cap)
                else:
                    self.torch_exchange(False)

    def run_host(self, first, count):
        """Frames [first, first+count) as HOST images, one tf_integrate_frame_host call per frame (staging copy into
        pinned memory + H2D inside).  The entry point runs tf_host_frame_deferral() = four frames behind the caller (its
        launch for frame f carries the voxel update of f - 4 next to the selection stages of f - 3 and f - 2), so `count`
        calls put `count` frames' H2D copies and `count` frames' kernels on the device."""
        vol, a_depth, a_rgba, a_pose, a_pinv, textured, n_unique = (self.vol, self.a_depth, self.a_rgba, self.a_pose, self.a_pinv,
                                                                    self.textured, self.n_unique)
        for j in range(count):
            i = (first + j) % n_unique
            vol.integrate_frame_host_addr(a_depth[i], a_rgba[i], a_pose[i], a_pinv[i] if textured else 0, first + j)

    def barrier(self):
        if self.multi:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def fresh_period(self):
        """The GPU boxes run this process under a CFS quota (cpu.max: e.g. 16 CPUs per 100 ms on a 256-CPU host): a burst of
        many runnable threads ahead of a timed window (image upload, registration, the profiler child passes, the runtime's
        helper threads) can exhaust the period's budget, and the kernel then freezes EVERY thread of the container until the
        period ends -- tens of milliseconds inside a 2-ms window.  A window's own work needs well under the quota; waiting
        out one period before the warm-up frames lets it start with a full budget."""
        if self.quota is not None:
            self.torch.cuda.synchronize()
            time.sleep(0.12)

    def skip_to_window(self):
        """advance the stream (untimed, resident frames) to the timed window's orbit position in the next turn"""
        nxt = self.pos + ((self.p0 - self.pos) % self.ORBIT)
        if nxt > self.pos:
            self.run(self.pos, nxt - self.pos)
        self.pos = nxt

    def land_before_window(self):
        """... to Wm frames ahead of it: a window that warms its own entry point up over those frames"""
        nxt = self.pos + ((self.p0 - self.Wm - self.pos) % self.ORBIT)
        if nxt > self.pos:
            self.run(self.pos, nxt - self.pos)
        self.pos = nxt

    # ---- phase 5: pre-roll, (N > 1) the order of the exchange, the link --------------------------------------------
    def preroll(self):
        # (the driver loop is Python: a cyclic-GC pass over torch's and numpy's objects takes milliseconds -- longer than the
        # whole window at the driver's --steps 20 -- and has nothing to do with the path; nothing below builds cycles)
        import gc
        gc.collect()
        gc.freeze()
        gc.disable()
        self.pos = 0
        if not self.args.no_preroll:
            self.run(0, self.ORBIT)
            self.pos = self.ORBIT

    def pick_exchange_order(self):
        """N > 1: the order of the per-frame exchange -- on the library's second stream next to the interior mesh pass, or in the
        main stream between the voxel update and the mesher -- is picked by measurement: the overlapped order costs a second
        filter + mesher pass and a stream fork / join per frame and pays only when the wire time exceeds that"""
        args, torch, dist, vol = self.args, self.torch, self.dist, self.vol
        if not (self.use_rccl and self.textured and not args.no_overlap and not args.child):
            return None
        t_ord = {}
        for on in (True, False):
            vol.comm_exchange_overlap(on)
            self.run(self.pos, 8)
            vol.sync()
            self.barrier()
            tq = time.perf_counter()
            self.run(self.pos + 8, 24)
            vol.sync()
            self.barrier()
            tt = torch.tensor([time.perf_counter() - tq], dtype=torch.float64, device=self.dev)
            if self.world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_ord[on] = float(tt.item()) / 24
            self.pos += 32
        pick = t_ord[True] <= t_ord[False]
        vol.comm_exchange_overlap(pick)
        return {"picked": "overlapped with the interior mesh pass" if pick else "serial, between voxel update and mesher",
                "us_per_frame_overlapped": 1e6 * t_ord[True], "us_per_frame_serial": 1e6 * t_ord[False],
                "how": "24 resident frames each way behind 8 warm-up frames, max over ranks; the faster order runs the timed window"}

    def settle_link(self):
        """VRAM that earlier processes of this job left behind (the profiler child passes, the side runs) is wiped in the background
        with the DMA engines the host-frame uploads use: for a while after such a process ends, a 2.46 MB upload takes 250 us
        instead of 53 (a window of r6 measured exactly that).  One frame's worth of bytes is uploaded until
        the link is back at its rate (at most 3 s), before -- not inside -- the timed region."""
        torch = self.torch
        if not (self.use_host and self.world == 1):
            return None
        probe_h = torch.empty(self.h_depth[0].nbytes + self.h_rgba[0].nbytes, dtype=torch.uint8).pin_memory()
        probe_d = torch.empty_like(probe_h, device=self.dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best, tries, last = None, 0, None
        for tries in range(1, 31):
            e0.record(); probe_d.copy_(probe_h, non_blocking=True); e1.record(); e1.synchronize()
            last = 1e3 * e0.elapsed_time(e1)
            best = last if best is None else min(best, last)
            ok = last <= probe_h.numel() / 25e3  # microseconds at 25 GB/s (the link moves 46 GB/s when nothing else uses it)
            if tries >= 3 and ok:
                break
            time.sleep(0.0 if ok else 0.1)
        return {"probes": tries, "last_upload_us": last, "best_upload_us": best, "bytes": int(probe_h.numel())}

    # ---- phase 6: warm-up and THE timed region ----------------------------------------------------------------------
    def timed_window(self):
        """W untimed warm-up steps, then exactly K steps between two barrier + device-synchronise pairs -> self.dt (seconds),
        self.t_enq (host time to enqueue them), self.host_phases."""
        vol, K, Wm = self.vol, self.K, self.Wm
        self.p0 = self.pos + Wm  # stream position of the timed window; p0 % ORBIT = its orbit position
        # host frames: the reference's calling convention.  Every rank of an N > 1 run is handed every frame.
        host_ok = (not self.multi) or (self.use_rccl and self.textured)
        self.use_host = host_ok and not self.args.resident_headline
        self.host_phases = None
        self.link_settle = self.settle_link()
        self.fresh_period()
        if self.use_host:
            self.run_host(self.pos, Wm)  # (leaves the entry point's four-frame pipeline primed)
            self.barrier()
            vol.host_frame_times(reset=True)
            t0 = time.perf_counter()
            self.run_host(self.p0, K)
            self.t_enq = time.perf_counter() - t0
            self.barrier()
            self.dt = time.perf_counter() - t0
            self.host_phases = vol.host_frame_times(reset=True)
        else:
            self.run(self.pos, Wm)
            vol.sync()
            self.barrier()
            t0 = time.perf_counter()
            self.run(self.p0, K)
            self.t_enq = time.perf_counter() - t0  # host time to enqueue the timed region (launches are asynchronous)
            self.barrier()
            self.dt = time.perf_counter() - t0
        vol.sync()  # brings the deferred frames onto the stream; surfaces any device-side capacity error
        self.pos = self.p0 + K

    def rank_stats(self):
        """N > 1: per rank its own wall time for the K frames and what the exchange moved (tf_comm_stats_ex: counted since the
        volume was made, i.e. over pre-roll + warm-up + timed frames); self.dt becomes the MAX over ranks."""
        args, torch, dist, vol, K = self.args, self.torch, self.dist, self.vol, self.K
        st = vol.comm_stats_ex()
        if not self.use_rccl and self.textured:
            st["bytes_sent"], st["bytes_received"] = self.torch_wire
        mine = {"rank": self.rank, "ms_per_step": 1e3 * self.dt / K, "exchanges": st["exchanges"],
                "exchange_bytes_sent": st["bytes_sent"], "exchange_bytes_received": st["bytes_received"],
                "ghost_records_packed": st["records_sent"], "ghost_records_received": st["records_received"],
                "bytes_received_per_record_received": (st["bytes_received"] / st["records_received"]) if st["records_received"] else None,
                "exchange_form": "neighbours (sized by the frame's selection)" if st["mode"] == 0 else "all-gather (fixed capacity)",
                "transport": "RCCL inside the library" if self.use_rccl else "torch.distributed test hook"}
        if self.use_rccl and self.textured and not args.child:
            # the same window once more with HIP events around every launch and around the exchange (untimed diagnostic)
            vol.sync()
            vol.profile_enable(("integrate", "dirty", "mesh", "xchg", "xchg_wait"))
            self.run(self.pos, K)
            vol.sync()
            pr = vol.profile_get(reset=True)
            vol.profile_enable(())
            self.pos += K
            mine["event_us_per_step"] = {k: 1e3 * pr[k][0] / K for k in ("integrate", "dirty", "mesh", "xchg", "xchg_wait") if pr[k][1]}
            # the exchange runs on the library's second stream next to the interior mesh pass: what the main stream still
            # waits for it is exposed, the rest hidden (an event pair costs ~6 us by itself: small values are that floor)
            x_all = mine["event_us_per_step"].get("xchg")
            x_wait = mine["event_us_per_step"].get("xchg_wait")
            if x_all is not None and x_wait is not None:
                mine["exchange_us_exposed"] = x_wait
                mine["exchange_us_hidden"] = max(0.0, x_all - x_wait)
        mine["exchanges_overlapped_with_interior_meshes"] = st.get("overlapped", 0)
        gathered = [None] * self.world if self.world > 1 else [mine]
        if self.world > 1:
            dist.all_gather_object(gathered, mine)
        t = torch.tensor([self.dt], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        self.dt = float(t.item())
        return gathered

    def replicas(self):
        """N > 1: the same N GPUs as N INDEPENDENT streams (one whole volume per GPU, no partition, no exchange: SURVEY.md
        s.8(e)'s stated fallback, "replicas") and the sharded keyframe unit -- printed next to the strong-scaling `value`.
        Closes the partitioned volume."""
        args = self.args
        indep = sharded_unit = None
        # (the partitioned volume stays open through the replica's run: the library counts registrations of the caller's
        # arrays per process -- and on this ROCm a volume destroyed BEFORE another one streams host frames leaves that one's
        # uploads at ~10 GB/s instead of ~40 for its first ~100 ms: the freed device pool is wiped in the background by the
        # DMA engines the uploads use, tools/two_volumes_probe.py)
        try:
            indep = independent_streams(args, self.cam, self.res, self.h_depth, self.h_rgba, self.d_depth, self.d_rgba, self.poses, self.pinv,
                                        self.a_depth, self.a_rgba, self.a_pose, self.a_pinv, self.n_unique, self.local_rank, self.world,
                                        self.dist if self.world > 1 else None, self.dev, self.textured, self.fresh_period)
        except Exception as e:  # (a side figure: never fail the bench line for it)
            indep = {"error": repr(e)[:300]}
        self.vol.close()
        if self.use_rccl and self.textured and not args.no_group:
            try:
                sharded_unit = sharded_keyframe_unit(args, self.cam, self.res, self.d_depth, self.d_rgba, self.poses, self.pinv, self.n_unique,
                                                     self.local_rank, self.rank, self.world, self.dist, self.dev, self.part_spec, args.exchange_cap)
            except Exception as e:
                sharded_unit = {"error": repr(e)[:300]}
        return indep, sharded_unit

    # ---- phase 7 (N = 1): the timed window's orbit positions again, other ways ----------------------------------------
    def _time_host_window(self, fn):
        """Wm frames through `fn` (untimed) from Wm frames ahead of the window, then the K frames timed -> seconds"""
        self.land_before_window()
        self.fresh_period()
        fn(self.pos, self.Wm)
        self.barrier()
        self.vol.host_frame_times(reset=True)
        t1 = time.perf_counter()
        fn(self.pos + self.Wm, self.K)
        self.barrier()
        return time.perf_counter() - t1

    def repeat_windows(self):
        """the timed window again (same orbit positions, later turns of the steady-state volume): spread of `value`"""
        K, Wm = self.K, self.Wm
        rep = []
        fn = self.run_host if self.use_host else self.run
        for _ in range(self.args.repeats):
            # land Wm frames ahead of the window, warm the entry point up over them, time the K frames
            self.land_before_window()
            self.fresh_period()
            fn(self.pos, Wm)
            if not self.use_host:
                self.vol.sync()
            self.barrier()
            t1 = time.perf_counter()
            fn(self.pos + Wm, K)
            self.barrier()
            rep.append(1e3 * (time.perf_counter() - t1) / K)
            self.vol.sync()
            self.pos += Wm + K
        allw = sorted(rep + [1e3 * self.dt / K])
        med = allw[len(allw) // 2] if len(allw) % 2 else 0.5 * (allw[len(allw) // 2 - 1] + allw[len(allw) // 2])
        return {"windows": len(allw), "ms_per_step_median": med, "ms_per_step_min": allw[0], "ms_per_step_max": allw[-1],
                "value_median": 1e3 / med, "value_min": 1e3 / allw[-1], "value_max": 1e3 / allw[0],
                "note": "`value` is the FIRST window (the contract's K timed steps); the others time the same %d orbit positions "
                        "in later turns, each behind its own %d warm-up frames" % (K, Wm)}

    def resident_window(self):
        """the same orbit positions with the frames already in HBM"""
        K = self.K
        self.skip_to_window()
        self.vol.sync()
        self.barrier()
        t1 = time.perf_counter()
        self.run(self.pos, K)
        self.barrier()
        dt_res = time.perf_counter() - t1
        self.vol.sync()
        self.pos += K
        return {"value": K / dt_res, "unit": "frames/s", "ms_per_step": 1e3 * dt_res / K,
                "note": "tf_stream_frames_textured_device on the same %d orbit positions one turn later: images "
                        "already in HBM, no H2D, one call for all frames" % K}

    def other_host_window(self):
        """the same positions as HOST frames by the OTHER way in: registered caller buffers <-> the library's staging slots"""
        vol, K, Wm, n_unique = self.vol, self.K, self.Wm, self.n_unique
        o_depth = self.h_depth[:n_unique].copy()   # (a second set of caller arrays: outside the registered ranges)
        o_rgba = self.h_rgba[:n_unique].copy()
        if not self.host_registered:
            vol.host_register(o_depth)
            vol.host_register(o_rgba)
        ao_depth = [o_depth[i].ctypes.data for i in range(n_unique)]
        ao_rgba = [o_rgba[i].ctypes.data for i in range(n_unique)]
        a_pose, a_pinv, textured = self.a_pose, self.a_pinv, self.textured

        def run_host_other(first, count):
            for j in range(count):
                i = (first + j) % n_unique
                vol.integrate_frame_host_addr(ao_depth[i], ao_rgba[i], a_pose[i], a_pinv[i] if textured else 0, first + j)

        dt_o = self._time_host_window(run_host_other)
        o_ph = vol.host_frame_times(reset=True)
        vol.sync()
        self.pos += Wm + K
        if not self.host_registered:
            vol.host_unregister(o_rgba)
            vol.host_unregister(o_depth)
        return {"value": K / dt_o, "unit": "frames/s", "ms_per_step": 1e3 * dt_o / K,
                "host_phases_us_per_step": {k: v for k, v in o_ph.items()},
                "note": ("tf_integrate_frame_host on the same %d orbit positions " % K) +
                        ("from arrays that were never registered: a copy into the library's pinned slots by a pool of helper "
                         "threads, asynchronous upload" if self.host_registered else
                         "out of caller arrays registered once with tf_host_register: no staging copy, no helper threads, the "
                         "call returns when its upload is through")}

    def async_host_window(self):
        """the same positions out of the registered arrays WITHOUT waiting for every upload (tf_host_frame_set_async: the
        caller keeps the "do not touch a buffer before the fence" contract itself -- here: 200 distinct frames, one fence at the end)"""
        vol, K, Wm = self.vol, self.K, self.Wm
        vol.host_frame_set_async(True)
        # (the first ~40 calls that run AHEAD of their uploads make the runtime grow its pool of copy resources: four of them
        # take 8-11 ms each, once per process -- tools/async_probe.py, profiles/r6/README.md; not part of any window)
        self.run_host(self.pos, 64)
        vol.host_frame_fence()
        self.pos += 64
        self.land_before_window()
        self.fresh_period()
        self.run_host(self.pos, Wm)
        vol.host_frame_fence()
        self.barrier()
        t1 = time.perf_counter()
        self.run_host(self.pos + Wm, K)
        vol.host_frame_fence()
        self.barrier()
        dt_a = time.perf_counter() - t1
        vol.sync()
        vol.host_frame_set_async(False)
        self.pos += Wm + K
        return {"value": K / dt_a, "unit": "frames/s", "ms_per_step": 1e3 * dt_a / K,
                "note": "the same %d orbit positions out of the registered arrays with tf_host_frame_set_async(1): a call returns "
                        "when its upload is QUEUED, one tf_host_frame_fence at the end (a caller with a ring of frame buffers); "
                        "the upload of frame f overlaps the call for f + 1" % K}

    def rgb_host_window(self):
        """the same positions as HOST frames with the colour image as the caller holds it (Frame::rgb, 3 B per pixel)"""
        vol, K, Wm, n_unique, h_depth, h_rgba, poses, pinv, textured = (self.vol, self.K, self.Wm, self.n_unique, self.h_depth, self.h_rgba,
                                                                        self.poses, self.pinv, self.textured)
        h_rgb = [np.ascontiguousarray(h_rgba[k][..., :3]) for k in range(n_unique)]
        all_valid = all(bool(h_rgba[k][..., 3].all()) for k in range(n_unique))
        h_valid = None if all_valid else [np.ascontiguousarray(h_rgba[k][..., 3]) for k in range(n_unique)]

        def run_host_rgb(first, count):
            for j in range(count):
                i = (first + j) % n_unique
                vol.integrate_frame_host_rgb(h_depth[i], h_rgb[i], None if h_valid is None else h_valid[i], poses[i],
                                             pinv[i] if textured else None, first + j)

        dt_rgb = self._time_host_window(run_host_rgb)
        vol.sync()
        self.pos += Wm + K
        return {"value": K / dt_rgb, "unit": "frames/s", "ms_per_step": 1e3 * dt_rgb / K,
                "bytes_uploaded_per_frame": self.cam.width * self.cam.height * (7 if h_valid is None else 8),
                "note": "tf_integrate_frame_host_rgb on the same %d orbit positions: depth + Frame::rgb (3 B per pixel%s) "
                        "as the reference's caller holds them; its RGBA staging loop (MobileFusion.cpp:232-243) runs on "
                        "the device behind the upload" % (K, "" if h_valid is None else " + colorValidFlag")}

    def event_window(self, kinds):
        """the same positions again: HIP events (on the handle's stream) around every launch of a step -> (prof, seconds, pair_us)"""
        vol, K = self.vol, self.K
        pair_us = vol.profile_calibrate(200)
        self.skip_to_window()
        vol.sync()
        vol.profile_enable(kinds)
        self.barrier()
        t1 = time.perf_counter()
        self.run(self.pos, K)
        self.barrier()
        dt_instr = time.perf_counter() - t1
        prof = vol.profile_get(reset=True)
        vol.profile_enable([])
        vol.sync()
        self.pos += K
        return prof, dt_instr, pair_us

    # ---- phase 8: the line ------------------------------------------------------------------------------------------
    def line(self):
        """the contract's keys (metric, value, unit, ... config) from the timed window"""
        args, world, K, Wm, cam, big, textured = self.args, self.world, self.K, self.Wm, self.cam, self.big, self.textured
        what = ("TSDF integrate + mesh + atlas update per frame (BASELINE.json configs[%d])" % (4 if world > 1 else (3 if args.hires or big else 2))
                if textured else "TSDF integrate, atlas off (BASELINE.json configs[%d])" % (4 if world > 1 else (3 if args.hires or big else 1)))
        out = {
            "metric": ("RGB-D frames/s (TSDF integrate + atlas update: prepare->integrate->finalize, UpdateMeshes, "
                       "GeneratePatches + UpdateAtlas over the frame's dirty chunks)" if textured else
                       "RGB-D frames/s (TSDF integrate, depth+colour, fused prepare->integrate->finalize)"),
            "value": K / self.dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": 1e3 * self.dt / K,
            "host_enqueue_ms_per_step": 1e3 * self.t_enq / K,
            "higher_is_better": True,
            "scaling": "strong",  # N ranks partition ONE stream by chunk range (total work fixed); the N = 1 line is that series' first point
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s orbit stream, %dx%d RGB-D, %.0f mm voxels, 8^3 chunks, %s; %s"
                            % ("S-hall 8x6x8 m" if big else "S-room 4x3x4 m", cam.width, cam.height, 1e3 * float(self.res), what,
                               "frames handed over as host images, staging + H2D copy (%.2f MB per frame) inside the timed region"
                               % (8e-6 * cam.width * cam.height) if self.use_host else "frames resident in HBM"),
                "h2d_in_timed_region": bool(self.use_host),
                "host_affinity": self.numa_note or "unchanged",
                "frames_per_orbit": self.ORBIT,
                "preroll_frames": 0 if args.no_preroll else self.ORBIT,
                "timed_window": {"first_frame": self.p0, "orbit_position": self.p0 % self.ORBIT, "frames": K},
                "parallelism": ("1 GPU" if world == 1 else
                                "%d ranks, chunk-range slabs of the key x+y+z of one stream; one RCCL exchange (neighbour send / "
                                "receive pairs: the band below to rank - 1, the band above to rank + 1) of the updated ghost-band "
                                "chunks %s"
                                % (world, "after every voxel update, ahead of the mesher, each block sized by the frame's own "
                                   "selection (8-record buckets, at most %d records of 8 KiB)" % args.exchange_cap if textured else
                                   "every %d frames, fixed blocks of %d records" % (args.exchange_every, args.exchange_cap))),
            },
        }
        if self.link_settle is not None:
            out["config"]["link_settle"] = self.link_settle
        return out


def _side_figure(fn):
    """a side figure never fails the bench line"""
    try:
        return fn()
    except Exception as e:
        return {"error": repr(e)[:300]}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.child:
        raise SystemExit(spawn_ranks(args))
    job = Job(args)            # environment, affinity, the stream on the host
    job.children()             # side runs + profiler child passes: before this process touches the GPU
    job.device()               # torch, process group, the stream in HBM
    job.volume()
    if job.multi:
        job.partition()
    rank, world, multi, vol = job.rank, job.world, job.multi, job.vol
    job.preroll()
    exchange_order = job.pick_exchange_order()
    job.timed_window()         # warm-up, then THE K timed steps -> job.dt, job.t_enq
    per_rank = job.rank_stats() if multi else None
    if args.child:  # profiler child pass: the workload above is all there is
        vol.close()
        return
    indep = sharded_unit = None
    if multi and not args.no_independent:
        indep, sharded_unit = job.replicas()

    # ---- N = 1: the timed window's orbit positions again -- its spread, and the other ways a frame can come in ----
    single_host = job.use_host and not multi
    repeats = job.repeat_windows() if (not multi and args.repeats > 0) else None
    resident = job.resident_window() if single_host else None
    other_host = _side_figure(job.other_host_window) if single_host else None
    async_host = _side_figure(job.async_host_window) if (single_host and job.host_registered) else None
    rgb_host = _side_figure(job.rgb_host_window) if single_host else None
    prof = dt_instr = pair_us = None
    kinds = STEP_KERNELS if job.textured else ("integrate",)
    if not args.no_roofline and not multi:
        prof, dt_instr, pair_us = job.event_window(kinds)

    out = job.line()
    if repeats is not None:
        out["repeats"] = repeats
    if resident is not None:
        out["resident"] = resident
    if other_host is not None:
        out["staged_host_frames" if job.host_registered else "registered_host_frames"] = other_host
    if rgb_host is not None:
        out["rgb_host_frames"] = rgb_host
    if async_host is not None:
        out["registered_async_host_frames"] = async_host
    if job.use_host and job.host_phases:
        host_phases = job.host_phases
        host_phases["host_buffers"] = ("registered with tf_host_register: uploaded in place, the call returns when the upload is through "
                                       "(wait_for_upload_us)" if job.host_registered else "copied into the library's pinned staging slots (staging_copy_us)")
        host_phases["note"] = ("host microseconds per call inside the timed window: host_enqueue_ms_per_step includes wait_for_device_us "
                               "(the entry point blocks until the device frees a staging slot -- back-pressure, not host work)")
        out["host_phases_us_per_step"] = host_phases
    if per_rank is not None:
        out["per_rank"] = per_rank
    if exchange_order is not None:
        out["exchange_order"] = exchange_order
    if indep is not None:
        out["independent_streams"] = indep
    if sharded_unit is not None:
        out["keyframe_unit_sharded"] = sharded_unit

    cam, res, n_unique = job.cam, job.res, job.n_unique
    # ---- roofline over ALL kernels of a step ------------------------------------------------
    if rank == 0 and prof is not None and not multi:
        job.skip_to_window()
        out["roofline"] = roofline(args, vol, cam, prof, kinds, job.K, job.pos, n_unique, job.d_depth, job.d_rgba, job.poses, job.pinv, job.textured,
                                   dt_instr, pair_us, job.prof_child, 1e6 * job.dt / job.K)
        job.pos += job.K

    # ---- the keyframe-group flow of TSDFFusion: 1 colour + 6 depth-only frames over one chunk list -------
    if rank == 0 and not multi and not args.no_group and not args.no_roofline:
        out["keyframe_group"] = keyframe_group(args, cam, res, job.d_depth, job.d_rgba, job.poses, n_unique, job.local_rank)

    # ---- the keyframe unit: tsdfFusion as one asynchronous call per keyframe -------------------------------
    if rank == 0 and not multi and not args.no_group and not args.no_roofline:
        out["keyframe_unit"] = keyframe_unit(args, cam, res, job.d_depth, job.d_rgba, job.poses, job.pinv, n_unique, job.local_rank)

    # ---- CPU baseline: the oracle (C port of the reference path) on the host cores ------------
    if rank == 0 and world == 1 and args.cpu_frames > 0:  # (rank 0 at N = 1 only)
        if job.orig_affinity:
            try:
                os.sched_setaffinity(0, job.orig_affinity)  # the CPU port gets every core of the host
            except Exception:
                pass
        out["cpu_baseline"] = cpu_baseline(args, cam, res, job.h_depth, job.h_rgba, job.h_pose, n_unique, job.textured)

    side = job.side
    if rank == 0 and side is not None:
        out["side"] = side
    if rank == 0 and "roofline" in out:
        # scalar leaves under `roofline` (a consumer that keeps only that dictionary's scalars keeps these): the other
        # single-GPU configurations and the spread of this line's own repeat windows
        for k, v in (side or {}).items():
            if isinstance(v, (int, float)) or v is None:
                out["roofline"]["side_" + k] = v
        if repeats is not None:
            for k in ("ms_per_step_median", "ms_per_step_min", "ms_per_step_max"):
                out["roofline"]["repeats_" + k] = repeats.get(k)
    vol.close()
    # the line is the LAST thing on stdout: what C libraries have printed so far (RCCL's version banner sits in the C runtime's
    # buffer when stdout is a file or a pipe) goes out first -- on every rank, ahead of a last barrier
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if multi:
        if world > 1:
            job.dist.barrier()
        job.dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def all_ranks_ok(dist, dev, ok):
    """True when `ok` holds on every rank (one small all-reduce): a rank whose local set-up failed must not leave the
    others waiting in the collectives that follow"""
    if dist is None:
        return bool(ok)
    import torch
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def independent_streams(args, cam, res, h_depth, h_rgba, d_depth, d_rgba, poses, pinv, a_depth, a_rgba, a_pose, a_pinv,
                        n_unique, device, world, dist, dev, textured, fresh_period):
    """N GPUs as N independent streams: every rank integrates the WHOLE stream into a volume of its own (no partition, no
    exchange) -- the reference scaled out by running one sequence per GPU.  Same method as the N = 1 headline: pre-roll
    of one orbit, warm-up, K host frames (H2D inside), barrier + device synchronisation on both sides, MAX over ranks;
    value = N x K / that time ("weak": per-GPU work is fixed)."""
    import torch
    from texturefusion_amd import capi
    K, Wm = args.steps, args.warmup
    big = args.scene == "big"
    vol, err = None, None
    try:
        vol = capi.Volume(res, cam, max_chunks=(1 << 21) if big else (1 << 19), mesh_blocks=(1 << 19) if big else (1 << 17), max_list=(1 << 20) if big else (1 << 18),
                          max_coarse=1 << 22 if big else 1 << 20, device=device)
    except Exception as e:
        err = repr(e)[:200]
    if not all_ranks_ok(dist, dev, vol is not None):
        if vol is not None:
            vol.close()
        return {"error": err or "another rank could not create its volume"}
    registered = False
    if not args.staged_host_frames:
        try:
            vol.host_register(h_depth)
            vol.host_register(h_rgba)
            registered = True
        except Exception:
            pass

    def run(first, count, ahead=2):
        idx = [(first + i) % n_unique for i in range(count + ahead)]
        dd = [d_depth[i].data_ptr() for i in idx]
        dr = [d_rgba[i].data_ptr() for i in idx]
        if textured:
            vol.stream_frames_textured_device(dd, dr, poses[idx], pinv[idx], first, n_ahead=ahead)
        else:
            vol.stream_frames_device(dd, dr, poses[idx], n_ahead=ahead)

    def run_host(first, count):
        for j in range(count):
            i = (first + j) % n_unique
            vol.integrate_frame_host_addr(a_depth[i], a_rgba[i], a_pose[i], a_pinv[i] if textured else 0, first + j)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    pos = 0
    ok = True
    try:
        if not args.no_preroll:
            run(0, n_unique)
            pos = n_unique
        fresh_period()
        run_host(pos, Wm)
    except Exception as e:
        ok, err = False, repr(e)[:200]
    if not all_ranks_ok(dist, dev, ok):  # (before the first barrier: nobody waits for a rank that failed)
        vol.close()
        return {"error": err or "another rank failed ahead of the timed window"}
    barrier()
    vol.host_frame_times(reset=True)
    t0 = time.perf_counter()
    try:
        run_host(pos + Wm, K)
    except Exception as e:
        ok, err = False, repr(e)[:200]
    barrier()
    dt = time.perf_counter() - t0
    if not all_ranks_ok(dist, dev, ok):
        vol.close()
        return {"error": err or "another rank failed inside the timed window"}
    phases = vol.host_frame_times(reset=True)
    vol.sync()
    mine = 1e3 * dt / K
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        allms = [None] * world
        dist.all_gather_object(allms, mine)
    else:
        allms = [mine]
    vol.close()
    return {"value": world * K / dt, "unit": "frames/s", "scaling": "weak", "n_gpus": world, "ms_per_step_per_stream": 1e3 * dt / K,
            "per_rank_ms_per_step": allms, "host_buffers": "registered" if registered else "staged",
            "host_phases_us_per_step_rank0": phases,
            "note": "%d independent streams, one whole (unpartitioned) volume per GPU, no exchange: every rank runs the N = 1 "
                    "headline workload (host frames, H2D inside) on its own GPU at the same time; value = N x K / max-over-ranks "
                    "time" % world}


def sharded_keyframe_unit(args, cam, res, d_depth, d_rgba, poses, pinv, n_unique, device, rank, world, dist, dev, part_spec, cap):
    """tf_keyframe_unit_device (one call per keyframe: 1 colour + 6 depth frames, meshes, patches, atlas) on the
    chunk-range partition of the strong-scaling run: every rank runs every keyframe over its slab, the ghost band is
    exchanged (fixed-capacity neighbour blocks: the unit's lists are not sized by a fused selection) once per keyframe
    ahead of the mesher.  Seven frames of voxel work per exchange instead of one."""
    import torch
    from texturefusion_amd import capi
    lo, hi, axis = part_spec
    n_local = 6
    stride = 1 + n_local
    n_kf = max(4, min(24, n_unique // stride - 1))
    big = args.scene == "big"
    s_main = torch.cuda.Stream(device=dev)
    vol, err = None, None
    try:
        vol = capi.Volume(res, cam, max_chunks=(1 << 21) if big else (1 << 19), mesh_blocks=(1 << 19) if big else (1 << 17), max_list=(1 << 20) if big else (1 << 18),
                          max_coarse=(1 << 22) if big else (1 << 20), device=device, stream=s_main.cuda_stream)
        vol.set_partition(lo, hi, axis)
    except Exception as e:
        err = repr(e)[:200]
    if not all_ranks_ok(dist if world > 1 else None, dev, err is None):  # (ncclCommInitRank below is a collective)
        if vol is not None:
            vol.close()
        return {"error": err or "another rank could not create its volume"}
    uid = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
    if world > 1:
        dist.broadcast(uid, 0)
    vol.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()))
    vol.comm_exchange_every_frame(cap)

    def call(g):
        k0 = (stride * g) % n_unique
        loc = [(k0 + 1 + i) % n_unique for i in range(n_local)]
        fresh = capi.Volume.unit_group(1000 + g, (d_depth[k0].data_ptr(), d_rgba[k0].data_ptr(), 0, poses[k0]),
                                       [(d_depth[k].data_ptr(), poses[k]) for k in loc])
        vol.keyframe_unit(fresh=fresh, moved=[], texture=True, pose_inv16=pinv[k0])

    warm = 4
    ok = True
    try:
        for g in range(warm):
            call(g)
        vol.sync()
    except Exception as e:
        ok, err = False, repr(e)[:200]
    if not all_ranks_ok(dist if world > 1 else None, dev, ok):
        vol.close()
        return {"error": err or "another rank failed in the warm-up keyframes"}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        for g in range(warm, warm + n_kf):
            call(g)
        vol.sync()
    except Exception as e:
        ok, err = False, repr(e)[:200]
    if not all_ranks_ok(dist if world > 1 else None, dev, ok):
        vol.close()
        return {"error": err or "another rank failed in the timed keyframes"}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    st = vol.comm_stats_ex()
    vol.close()
    return {"keyframes_per_s": n_kf / dt, "ms_per_keyframe": 1e3 * dt / n_kf, "frame_integrations_per_s": n_kf * stride / dt,
            "keyframes": n_kf, "n_gpus": world, "scaling": "strong", "exchanges": st["exchanges"],
            "exchange_bytes_sent_rank0": st["bytes_sent"],
            "note": "one tf_keyframe_unit_device call per keyframe on every rank over its slab of the chunk-range partition; one "
                    "fixed-capacity neighbour exchange (%d records per block) per keyframe between the group's voxel updates and "
                    "the mesher" % cap}


def roofline(args, vol, cam, prof, kinds, K, first, n_unique, d_depth, d_rgba, poses, pinv, textured, dt_instr, pair_us,
             prof_child, wall_us):
    """Algorithmic bytes of a step / summed kernel time of a step.

    Bytes = what the DEVICE algorithm has to move: voxel update 128 B per rewritten
    TSDF row + 128 B per rewritten colour row + one read of the depth and RGBA images; meshing 32 B of class summaries
    per dirty chunk (its own and its seven +x/+y/+z neighbours' words) + 4 KiB (the chunk's own sdf/weight plane) per
    chunk the summaries cannot rule out + 6552 B (the 11^3 - 8^3 halo voxels) per chunk the surface passes through + 8 B
    colour read and 36 B written per vertex + 6 B per triangle; atlas 44 B per projected vertex (24 read, 20 written)
    + 3 B read and 3 B written per ROI pixel.  The integer counts are read back frame by frame in an untimed replay of
    the timed window's orbit positions (steady state: the same work as the timed frames one turn earlier).

    Kernel time: the profiler's own durations of the timed window's launches when the rocprofv3 --kernel-trace child
    pass ran (the same command, the same frames); otherwise the HIP-event times of the event pass minus the calibrated
    cost of an event pair around an empty launch."""
    b_tsdf = b_mesh = b_atlas = 0
    cnt = dict(sel=0, upd=0, dirty=0, exact=0, survivors=0, surface=0, meshes=0, verts=0, tris=0, roi=0, patches=0)
    for j in range(K):
        sub = [(first + j + q) % n_unique for q in range(3)]
        dd = [d_depth[q].data_ptr() for q in sub]
        dr = [d_rgba[q].data_ptr() for q in sub]
        if textured:
            vol.stream_frames_textured_device(dd, dr, poses[sub], pinv[sub], first + j, n_ahead=2)
        else:
            vol.stream_frames_device(dd, dr, poses[sub], n_ahead=2)
        st = vol.stats()
        b_tsdf += 128 * st.rows_tsdf + 128 * st.rows_color + 8 * cam.width * cam.height
        cnt["sel"] += st.n_selected
        cnt["listed"] = cnt.get("listed", 0) + st.n_listed
        cnt["upd"] += st.n_updated
        if textured:
            ts = vol.texture_stats()
            b_mesh += 32 * ts.n_dirty + 4096 * ts.n_exact + 6552 * ts.n_surface + 44 * ts.n_vertices + 6 * ts.n_triangles
            b_atlas += 44 * ts.n_vertices + 6 * ts.roi_pixels
            for k, v in (("dirty", ts.n_dirty), ("exact", ts.n_exact), ("survivors", ts.n_survivors), ("surface", ts.n_surface),
                         ("meshes", ts.n_meshes),
                         ("verts", ts.n_vertices), ("tris", ts.n_triangles), ("roi", ts.roi_pixels), ("patches", ts.n_patches)):
                cnt[k] += v
    # event pass: per kind, raw and with the empty-pair cost taken off every launch
    ev = {}
    for k in kinds:
        ms, n = prof[k]
        if n:
            ev[k] = {"event_us_per_step": 1e3 * ms / K, "launches_per_step": n / K,
                     "event_us_minus_pair": max(0.0, 1e3 * ms / K - pair_us * n / K)}
    t_events = sum(v["event_us_minus_pair"] for v in ev.values())
    bytes_step = (b_tsdf + b_mesh + b_atlas) / K
    # the patch stage of frame f - 1 runs inside the launch of frame f's voxel update: their bytes share its time
    groups = {"k_frame": (b_tsdf + b_atlas) / K, "mesh": b_mesh / K} if textured else {"k_frame": b_tsdf / K}
    trace = (prof_child or {}).get("trace")
    kern = {}
    if trace and "kernels" in trace:
        t_step = trace["us_per_step"]
        source = "rocprofv3 --kernel-trace child pass of this command: durations of the timed window's launches"
        for name, v in trace["kernels"].items():
            kern[name] = dict(v)
        t_kframe = sum(v["us_per_step"] for n, v in trace["kernels"].items() if n.startswith("k_frame") or n.startswith("k_patch"))
        t_mesh = sum(v["us_per_step"] for n, v in trace["kernels"].items() if n.startswith("k_mesh") or n.startswith("k_dirty"))
    else:
        t_step = t_events
        source = ("HIP events around every launch of the event pass minus %.2f us per launch (an event pair around an "
                  "empty launch)" % pair_us)
        t_kframe = sum(ev[k]["event_us_minus_pair"] for k in ("integrate", "patch_project", "patch_rank") if k in ev)
        t_mesh = sum(ev[k]["event_us_minus_pair"] for k in ("dirty", "mesh") if k in ev)
    achieved = bytes_step / t_step / 1e3 if t_step > 0 else 0.0
    per_group = {"k_frame (+ k_patch launches)": {"us_per_step": t_kframe, "algorithmic_bytes_per_step": groups["k_frame"],
                                                  "achieved_GBs": groups["k_frame"] / t_kframe / 1e3 if t_kframe else 0.0}}
    if textured:
        per_group["k_dirty_frame + k_mesh_filter + k_mesh"] = {
            "us_per_step": t_mesh, "algorithmic_bytes_per_step": groups["mesh"],
            "achieved_GBs": groups["mesh"] / t_mesh / 1e3 if t_mesh else 0.0}
    r = {
        "bound": "hbm",
        "kernel": ("all kernels of a step: k_frame (voxel update of frame f + patch stage of frame f-1 + selection of f+1, f+2), "
                   "k_dirty_frame, k_mesh_filter, k_mesh" if textured else
                   "k_frame<color> = K-A(f) + K-C(f+1) + K-B(f+2) block ranges"),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": None,  # filled from the --pmc child passes of this run, stays null without them
        "algorithmic_bytes_per_step": bytes_step, "kernel_us_per_step": t_step, "kernel_time_source": source,
        "wall_us_per_step": wall_us,
        "groups": per_group,
        "kernels": kern,
        "events": {"pair_us": pair_us, "kernel_us_per_step_minus_pair": t_events, "instrumented_ms_per_step": 1e3 * dt_instr / K,
                   "kinds": ev},
        "per_step": {k: v / K for k, v in cnt.items()},
        "windows": "timed, resident, event and replay passes cover the same %d orbit positions in consecutive turns of a "
                   "steady-state volume; the profiler child passes run the timed window itself" % K,
    }
    if wall_us and t_step > wall_us:
        # the profiled pass is another process: with ~1 us of idle device per frame its kernel durations (which carry the
        # profiler's per-dispatch overhead) can exceed this process's unprofiled wall time of the same frames
        r["kernel_vs_wall"] = ("kernel_us_per_step (%.1f, %s) exceeds wall_us_per_step (%.1f, this process, unprofiled) by %.1f %%: "
                               "two runs of the same frames; `frac` uses the larger one" % (t_step, "profiled child pass" if (trace and "kernels" in trace) else "events", wall_us, 100.0 * (t_step / wall_us - 1.0)))
    pmc = (prof_child or {}).get("pmc")
    if pmc and "bytes_per_step" in pmc:
        r["traffic"] = pmc["bytes_per_step"]
    # the per-kernel split once more as SCALAR leaves (a consumer that keeps only scalars -- the driver's BENCH file -- keeps these)
    def _k(prefix):
        return sum(v["us_per_step"] for n, v in kern.items() if n.startswith(prefix)) if kern else None
    r["k_frame_us"] = _k("k_frame") if kern else t_kframe
    r["k_patch_us"] = _k("k_patch") if kern else None
    r["k_mesh_filter_us"] = _k("k_mesh_filter") if kern else None
    r["k_mesh_us"] = (_k("k_mesh<") if kern else None)
    r["k_dirty_frame_us"] = _k("k_dirty") if kern else None
    r["mesh_group_us"] = t_mesh if textured else None
    r["k_frame_frac"] = (groups["k_frame"] / t_kframe / 1e3 / HBM_PEAK_GBS) if t_kframe else None
    r["mesh_group_frac"] = (groups["mesh"] / t_mesh / 1e3 / HBM_PEAK_GBS) if textured and t_mesh else None
    r["traffic_ratio"] = (r["traffic"] / bytes_step) if r["traffic"] and bytes_step else None
    # SURVEY.md s.8(d)'s B to the letter -- voxel rows + images + the atlas term, WITHOUT the meshing term this line's
    # numerator adds for the f-1 row (the mesher's kernels stay in the denominator: they are part of the step)
    r["algorithmic_bytes_survey_B"] = (b_tsdf + b_atlas) / K
    r["frac_survey_B"] = ((b_tsdf + b_atlas) / K / t_step / 1e3 / HBM_PEAK_GBS) if t_step > 0 else None
    r["traffic_detail"] = pmc if pmc else ({"error": prof_child["error"]} if prof_child and "error" in prof_child else None)
    return r


# profiles/r2/README.md (tools/calib_fetch on this box type): both counters are in KiB; WRITE_SIZE is exact;
# FETCH_SIZE reads exactly 1/2 of the bytes for every read shape the kernels use (4/8/16 B per lane streams,
# scattered 4-KiB blocks, 4-B gathers: 128-B requests tallied as 64 B), as MI355X_MICROARCH.md states for 16 B/lane.
PMC_BYTES = {"FETCH_SIZE": 2048.0, "WRITE_SIZE": 1024.0}


def _kernel_name(raw):
    import re
    name = re.sub(r"^void ", "", raw)
    name = re.sub(r"\(.*$", "", name).replace("tf::", "").strip()
    return name


def _window(rows, first_frame, K):
    """rows: (dispatch id, kernel name, value) of one child pass.  The step launches of frame f start at the f-th
    launch of a k_frame instance that carries a colour voxel update (k_frame<true, *>; the selection-only launches of
    a pipeline fill are the depth-only instance).  Returns the rows between the launch of frame `first_frame` and the
    launch of frame `first_frame + K`."""
    rows = sorted(rows)
    starts = [d for d, n, _ in rows if n.startswith("k_frame<true")]
    if len(starts) < first_frame + K:
        return None
    lo = starts[first_frame]
    hi = starts[first_frame + K] if len(starts) > first_frame + K else rows[-1][0] + 1
    return [(d, n, v) for d, n, v in rows if lo <= d < hi]


def side_runs(args):
    """configs[1] (TSDF only, atlas off) and configs[3] (S-hall, 1280x960) with the flags of this run, each as a fresh
    `bench.py` process started before this one touches the GPU (one process per configuration: each brings its own volume,
    its own steady-state pre-roll and its own rocprofv3 --kernel-trace child pass).  CPU baseline, keyframe flows and the
    PMC traffic passes are left out; what comes back are scalars."""
    import subprocess
    out = {}
    hall_steps = min(args.steps, 20)
    runs = (("tsdf_only", ["--mode", "tsdf", "--steps", str(args.steps), "--warmup", str(args.warmup)],
             "configs[1]: S-room, 640x480, 5 mm, voxel update only (host frames, H2D inside)"),
            ("hall", ["--scene", "big", "--hires", "--steps", str(hall_steps), "--warmup", str(min(args.warmup, 5))],
             "configs[3]: S-hall 8x6x8 m, 1280x960, 5 mm, textured unit (host frames, H2D inside)"))
    for tag, flags, what in runs:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + flags + ["--no-side", "--no-traffic", "--cpu-frames", "0", "--no-group",
                                                                         "--repeats", "2", "--res", repr(args.res)]
        t0 = time.time()
        try:
            r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=420)
            line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                out[tag + "_error"] = "rc %d: %s" % (r.returncode, r.stderr[-300:])
                continue
            d = json.loads(line[-1])
        except Exception as e:  # a side run must never take the headline down
            out[tag + "_error"] = "%s: %s" % (type(e).__name__, e)
            continue
        rf = d.get("roofline") or {}
        out[tag + "_workload"] = what
        out[tag + "_value"] = d.get("value")
        out[tag + "_ms_per_step"] = d.get("ms_per_step")
        out[tag + "_steps"] = d.get("steps")
        out[tag + "_resident_value"] = (d.get("resident") or {}).get("value")
        out[tag + "_frac"] = rf.get("frac")
        out[tag + "_kernel_us"] = rf.get("kernel_us_per_step")
        out[tag + "_algorithmic_bytes"] = rf.get("algorithmic_bytes_per_step")
        for k in ("k_frame_us", "k_dirty_frame_us", "k_mesh_filter_us", "k_mesh_us"):
            if rf.get(k) is not None:
                out[tag + "_" + k] = rf.get(k)
        out[tag + "_run_s"] = round(time.time() - t0, 1)
    return out


def child_passes(args):
    """The same command (pre-roll, warm-up, K host frames) three more times as child processes under rocprofv3, before
    this process touches the GPU: one --kernel-trace pass (kernel durations) and one --pmc pass each for FETCH_SIZE and
    WRITE_SIZE (they do not fit one pass; no trace domain in a counter pass; the program itself behind "--":
    MI355X_MICROARCH.md, HBM / PMC sections).  From each the launches of the timed window are sliced out.  Returns None
    when rocprofv3 is missing; a failing pass leaves "error" in its part -- never a constant."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    K, Wm = args.steps, args.warmup
    pre = 0 if args.no_preroll else args.unique_frames
    # the host entry point runs `behind` frames behind its caller (tf_host_frame_deferral: kHostDefer of the build): the
    # launches inside the parent's timed region are those of the frames [pre + Wm - behind, pre + Wm + K - behind)
    # (asked in a process of its own: this one must not map the HIP runtime before torch brings its copy)
    behind = 0
    if not args.resident_headline:
        q = subprocess.run([sys.executable, "-c", "from texturefusion_amd import capi; print(capi.host_frame_deferral()[0])"],
                           cwd=ROOT, capture_output=True, text=True, timeout=120)
        behind = int(q.stdout.strip().splitlines()[-1]) if q.returncode == 0 and q.stdout.strip() else 4
    first = max(0, pre + Wm - behind)
    base_cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--child", "--steps", str(K), "--warmup", str(Wm),
                "--mode", args.mode, "--scene", args.scene, "--res", repr(args.res), "--unique-frames", str(args.unique_frames)]
    base_cmd += (["--hires"] if args.hires else []) + (["--no-preroll"] if args.no_preroll else [])
    base_cmd += ["--max-chunks-log2", str(args.max_chunks_log2)] if args.max_chunks_log2 > 0 else []
    base_cmd += ["--resident-headline"] if args.resident_headline else []
    base_cmd += ["--staged-host-frames"] if args.staged_host_frames else []
    tmp = tempfile.mkdtemp(prefix="tf_prof_", dir="/tmp")
    out = {}

    def run_pass(tag, flags, pattern, col_name, col_val, want_counter=None):
        d = os.path.join(tmp, tag)
        cmd = [exe] + flags + ["--output-format", "csv", "-d", d, "-o", "t", "--"] + base_cmd
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=400)
        files = glob.glob(os.path.join(d, "**", pattern), recursive=True)
        if r.returncode != 0 or not files:
            return "rocprofv3 %s: rc %d, %d csv: %s" % (tag, r.returncode, len(files), r.stderr[-300:])
        rows = []
        for f in files:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if want_counter and row.get("Counter_Name") != want_counter:
                        continue
                    rows.append((int(row["Dispatch_Id"]), _kernel_name(row[col_name]), col_val(row)))
        return rows

    try:
        rows = run_pass("trace", ["--kernel-trace"], "*kernel_trace.csv", "Kernel_Name",
                        lambda r: 1e-3 * (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        if isinstance(rows, str):
            out["trace"] = {"error": rows}
        else:
            win = _window(rows, first, K)
            if win is None:
                out["trace"] = {"error": "fewer k_frame launches than frames in the kernel trace"}
            else:
                per = {}
                for _, n, v in win:
                    e = per.setdefault(n, {"us_per_step": 0.0, "launches_per_step": 0.0})
                    e["us_per_step"] += v / K
                    e["launches_per_step"] += 1.0 / K
                out["trace"] = {"kernels": per, "us_per_step": sum(e["us_per_step"] for e in per.values()),
                                "frames": [first, first + K]}
        pmc = {"kernels": {}}
        for counter in (() if args.no_traffic else ("FETCH_SIZE", "WRITE_SIZE")):
            rows = run_pass(counter, ["--pmc", counter], "*counter_collection.csv", "Kernel_Name",
                            lambda r: float(r["Counter_Value"]), want_counter=counter)
            if isinstance(rows, str):
                pmc = {"error": rows}
                break
            win = _window(rows, first, K)
            if win is None:
                pmc = {"error": "fewer k_frame launches than frames in the %s pass" % counter}
                break
            for _, n, v in win:
                e = pmc["kernels"].setdefault(n, {})
                e[counter] = e.get(counter, 0.0) + PMC_BYTES[counter] * v / K
        if args.no_traffic:
            pmc = {"error": "--no-traffic: kernel trace only"}
        if "kernels" in pmc:
            for e in pmc["kernels"].values():
                e["bytes_per_step"] = e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0)
            pmc["bytes_per_step"] = sum(e["bytes_per_step"] for e in pmc["kernels"].values())
            pmc["frames"] = [first, first + K]
            pmc["how"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of this command; the launches of the "
                          "timed window (frames %d..%d of the stream) summed per kernel and divided by %d steps; bytes = "
                          "2048 x FETCH_SIZE + 1024 x WRITE_SIZE (KiB units; FETCH_SIZE counts 128-B requests as 64 B on "
                          "gfx950 -- calibration in profiles/r2/README.md); L2-miss traffic: Infinity-Cache hits are included"
                          % (first, first + K - 1, K))
        out["pmc"] = pmc
    except Exception as e:  # a profiler pass must never take the benchmark down
        out.setdefault("pmc", {"error": "%s: %s" % (type(e).__name__, e)})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def keyframe_group(args, cam, res, d_depth, d_rgba, poses, n_unique, device):
    """MobileFusion::TSDFFusion's integration loop (GCFusion/MobileFusion.cpp:165-217): PrepareIntersectChunks for the
    keyframe, its depth + colour, then the six local frames depth-only over the SAME list, FinalizeIntegrateChunks.
    Two volumes get the same groups: one integrates the local frames with six tf_integrate calls, the other with one
    tf_integrate_depth_group; the kernel times come from HIP events around the launches (the calls themselves
    return lists and flags to the host, so their wall time is dominated by host round trips)."""
    from texturefusion_amd import capi
    n_groups, n_local = 8, 6
    big = args.scene == "big"
    vols = [capi.Volume(res, cam, max_chunks=(1 << 20) if big else (1 << 18), mesh_blocks=(1 << 18) if big else (1 << 16), max_list=(1 << 19) if big else (1 << 17),
                        max_coarse=(1 << 22) if big else (1 << 20), device=device) for _ in range(2)]
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


def keyframe_unit(args, cam, res, d_depth, d_rgba, poses, pinv, n_unique, device):
    """tf_keyframe_unit_device (MobileFusion::tsdfFusion, GCFusion/MobileFusion.cpp:274-406) along the orbit: every
    7th frame is a keyframe (depth + colour), the six frames behind it are its local frames (depth-only, one visit
    per chunk); per keyframe: integrate the group, UpdateMeshes, CompressMeshes, GeneratePatches (label = the
    keyframe), UpdateAtlas -- one call, nothing synchronised.  Second figure: every call also carries ONE moved
    keyframe (the group integrated two calls earlier, de-integrated over its stored validChunks and re-integrated
    with the poses of the neighbouring frames)."""
    from texturefusion_amd import capi
    n_local = 6
    stride = 1 + n_local
    n_kf = max(4, min(24, n_unique // stride - 1))
    res_out = {}
    big = args.scene == "big"
    for with_moved in (False, True):
        vol = capi.Volume(res, cam, max_chunks=(1 << 21) if big else (1 << 19), mesh_blocks=(1 << 19) if big else (1 << 17), max_list=(1 << 20) if big else (1 << 18),
                          max_coarse=(1 << 22) if big else (1 << 20), device=device)

        def group(g, shift=0, old=False):
            k0 = (stride * g) % n_unique
            loc = [(k0 + 1 + i) % n_unique for i in range(n_local)]
            P = lambda k: poses[(k + shift) % n_unique]
            kw = {}
            if old:
                kw = dict(old_keyframe_pose=poses[k0], old_local_poses=[poses[k] for k in loc])
            return capi.Volume.unit_group(1000 + g, (d_depth[k0].data_ptr(), d_rgba[k0].data_ptr(), 0, P(k0)),
                                          [(d_depth[k].data_ptr(), P(k)) for k in loc], **kw), k0

        def call(g):
            fresh, k0 = group(g)
            moved = [group(g - 2, shift=1, old=True)[0]] if with_moved and g >= 2 and g % 2 == 0 else []
            # (a group is moved once: its second move would need the shifted poses as the old ones)
            vol.keyframe_unit(fresh=fresh, moved=moved, texture=True, pose_inv16=pinv[k0])
            return len(moved)

        # (one turn of the orbit first, untimed -- the figure is the steady state, like the headline behind its pre-roll; timed
        # from an empty volume the same keyframes cost 196-201 us instead of 176-178: chunk creation)
        warm = max(4, n_unique // stride)
        for g in range(warm):
            call(g)
        vol.sync()
        if cgroup_cpu_quota() is not None:
            time.sleep(0.12)  # (a fresh CFS period: see fresh_period() in main)
        t0 = time.perf_counter()
        n_moved = 0
        for g in range(warm, warm + n_kf):
            n_moved += call(g)
        vol.sync()
        dt = time.perf_counter() - t0
        frames = n_kf * stride + n_moved * 2 * stride  # a moved group is integrated twice (flag 0, flag 1)
        res_out["with_one_moved_keyframe_every_other_call" if with_moved else "new_keyframes_only"] = {
            "keyframes_per_s": n_kf / dt, "ms_per_keyframe": 1e3 * dt / n_kf, "frame_integrations_per_s": frames / dt,
            "keyframes": n_kf, "moved_groups": n_moved, "untimed_keyframes_ahead": warm}
        vol.close()
        time.sleep(0.3)  # (the freed pool is wiped in the background for tens of ms: the next volume's figure should not run into it)
    # ---- the product's own order (MobileFusion::tsdfFusion with its view selection on the host): the unit WITHOUT its texture
    # stage (asynchronous), then the caller's CompressMeshes (returns chunksToUpdate: the one synchronisation), its view
    # selection (here: every chunk labelled with the new keyframe), GeneratePatches and UpdateAtlas through the entry points
    try:
        import numpy as np
        vol = capi.Volume(res, cam, max_chunks=(1 << 21) if big else (1 << 19), mesh_blocks=(1 << 19) if big else (1 << 17), max_list=(1 << 20) if big else (1 << 18),
                          max_coarse=(1 << 22) if big else (1 << 20), device=device)

        def unit_then_caller(g):
            k0 = (stride * g) % n_unique
            loc = [(k0 + 1 + i) % n_unique for i in range(n_local)]
            grp = capi.Volume.unit_group(1000 + g, (d_depth[k0].data_ptr(), d_rgba[k0].data_ptr(), 0, poses[k0]),
                                         [(d_depth[k].data_ptr(), poses[k]) for k in loc])
            vol.keyframe_unit(fresh=grp, moved=[], texture=False)
            upd = vol.compress_meshes()
            vol.keyframe_cache_device(1000 + g, d_rgba[k0].data_ptr(), d_depth[k0].data_ptr(), stride=4, pose_inv16=pinv[k0])
            vol.generate_patches(upd, np.full(len(upd), 1000 + g, np.int32))
            vol.update_atlas(upd)
            if g >= 8:
                vol.keyframe_release(1000 + g - 8)

        for g in range(4):
            unit_then_caller(g)
        vol.sync()
        if cgroup_cpu_quota() is not None:
            time.sleep(0.12)
        t0 = time.perf_counter()
        for g in range(4, 4 + n_kf):
            unit_then_caller(g)
        vol.sync()
        dt = time.perf_counter() - t0
        res_out["unit_then_callers_view_selection_calls"] = {
            "keyframes_per_s": n_kf / dt, "ms_per_keyframe": 1e3 * dt / n_kf, "frame_integrations_per_s": n_kf * stride / dt,
            "keyframes": n_kf,
            "note": "tf_keyframe_unit_device(texture = 0), then tf_compress_meshes / tf_generate_patches (labels from the caller) / "
                    "tf_update_atlas: the reference's own order when the view selection stays on the host"}
        vol.close()
        time.sleep(0.3)
    except Exception as e:  # (a diagnostic figure: never fail the bench line for it)
        res_out["unit_then_callers_view_selection_calls"] = {"error": repr(e)[:300]}
    # ---- the same keyframes through the reference's OWN call sequence (what a caller that only swaps the headers gets,
    # INTEGRATION.md approach A): PrepareIntersectChunks -> IntegrateDepthScanColor (keyframe) -> 6 x IntegrateDepthScanColor
    # (depth only) -> FinalizeIntegrateChunks -> UpdateMeshes -> CompressMeshes -> GeneratePatches -> UpdateAtlas, every call
    # synchronous with its lists and flags on the host, images device-resident
    try:
        import numpy as np
        vol = capi.Volume(res, cam, max_chunks=(1 << 21) if big else (1 << 19), mesh_blocks=(1 << 19) if big else (1 << 17), max_list=(1 << 20) if big else (1 << 18),
                          max_coarse=(1 << 22) if big else (1 << 20), device=device)
        rgb3 = {}

        def call_by_call(g):
            k0 = (stride * g) % n_unique
            loc = [(k0 + 1 + i) % n_unique for i in range(n_local)]
            vol.frame_bind_device(d_depth[k0].data_ptr(), d_rgba[k0].data_ptr(), 0)
            ids, new = vol.prepare(poses[k0])
            needs = np.zeros(len(ids), np.uint8)
            vol.integrate(poses[k0], ids, needs, 1, True, False)
            for k in loc:
                vol.frame_bind_device(d_depth[k].data_ptr(), 0, 0)
                vol.integrate(poses[k], ids, needs, 1, False, False)
            vol.finalize(ids, needs, new)
            vol.update_meshes()
            upd = vol.compress_meshes()
            vol.keyframe_cache_device(1000 + g, d_rgba[k0].data_ptr(), d_depth[k0].data_ptr(), stride=4, pose_inv16=pinv[k0])
            vol.generate_patches(upd, np.full(len(upd), 1000 + g, np.int32))
            vol.update_atlas(upd)
            if g >= 8:
                vol.keyframe_release(1000 + g - 8)

        for g in range(4):
            call_by_call(g)
        vol.sync()
        if cgroup_cpu_quota() is not None:
            time.sleep(0.12)
        n_cc = max(4, n_kf // 2)
        t0 = time.perf_counter()
        for g in range(4, 4 + n_cc):
            call_by_call(g)
        vol.sync()
        dt = time.perf_counter() - t0
        res_out["call_by_call_reference_sequence"] = {
            "keyframes_per_s": n_cc / dt, "ms_per_keyframe": 1e3 * dt / n_cc, "frame_integrations_per_s": n_cc * stride / dt,
            "keyframes": n_cc,
            "note": "the reference's own sequence of entry points (tf_prepare / tf_integrate x 7 / tf_finalize / tf_update_meshes / "
                    "tf_compress_meshes / tf_generate_patches / tf_update_atlas), each synchronising with lists and flags on the host: "
                    "what swapping the headers alone buys; the one-call unit above keeps everything on the device"}
        vol.close()
    except Exception as e:  # (a diagnostic figure: never fail the bench line for it)
        res_out["call_by_call_reference_sequence"] = {"error": repr(e)[:300]}
    res_out["note"] = ("one tf_keyframe_unit_device call per keyframe: 1 colour + %d depth-only frames over one chunk list, "
                       "meshes of everything marked, CompressMeshes, GeneratePatches (label = the keyframe), UpdateAtlas; "
                       "asynchronous, device-resident images, wall time over the calls + one final synchronisation" % n_local)
    return res_out


def cpu_baseline(args, cam, res, h_depth, h_rgba, h_pose, n_unique, textured):
    """oracle/ timed on a bounded sample of the same workload: --cpu-warmup untimed frames build up the volume
    (meshes need weight > 50), then the next --cpu-frames frames of the stream are timed."""
    from oracle import api as O
    from texturefusion_amd import synth
    ov = O.Volume(res, O.camera_from(cam), O.default_integrator())
    avx2 = bool(O.lib().tfo_have_avx2())
    ov.set_kernel(1 if avx2 else 0)  # AVX2 row kernel (as the reference's), bit-identical to the scalar checker
    O.lib().tfo_set_select_kernel(1 if avx2 else 0)  # ... and the AVX2 selection (ChunkManager.h:303-364,561-636)
    ncpu = os.cpu_count() or 1
    oa = O.Atlas(res) if textured else None
    T = args.cpu_threads if args.cpu_threads > 0 else max(1, ncpu - 2)  # chisel::parallel_for: hardware_concurrency - 2,
    ov.set_threads(T)                                                    # groups of >= 1000 items (applied per call inside)
    quota = cgroup_cpu_quota()  # (std::thread::hardware_concurrency() does not see a CFS quota: the policy above is the reference's)

    def step(k):
        i = k % n_unique
        if textured:
            ov.frame_textured(oa, h_depth[i], h_rgba[i], h_pose[i], synth.pose_inverse16(h_pose[i]), k)
        else:
            ov.integrate_frame(h_depth[i], h_rgba[i], h_pose[i])

    for k in range(args.cpu_warmup):
        step(k)
    n = args.cpu_frames
    t0 = time.perf_counter()
    for k in range(args.cpu_warmup, args.cpu_warmup + n):
        step(k)
    t_all = time.perf_counter() - t0
    out = {
        "value": n / t_all, "unit": "frames/s", "cores": T,
        "kind": "port",
        "sample": "oracle/ C port of the reference path (%s voxel kernel and selection, no FMA; selection on 1 thread as in the reference; %s) on frames "
                  "%d..%d of the same stream after %d untimed frames on an empty volume (lighter than the GPU's steady-state "
                  "frames: the GPU / CPU ratio is understated); parallel stages use chisel::parallel_for's policy "
                  "(hardware_concurrency - 2 = %d threads, groups of >= 1000 items); host has %d logical cores"
                  % ("AVX2 8-lane" if avx2 else "scalar",
                     "UpdateMeshes parallel, CompressMeshes / GeneratePatches / UpdateAtlas serial as in the reference" if textured
                     else "atlas off",
                     args.cpu_warmup, args.cpu_warmup + n - 1, args.cpu_warmup, T, ncpu),
        "host_cores": ncpu,
        "cpu_quota_cpus": quota,
    }
    if quota is not None and quota < T and args.cpu_threads <= 0:
        # the container may only use `quota` CPUs' worth of time: the same frames once more with that many threads (the
        # reference's policy oversubscribes a throttled container); `value` / `cores` report the faster of the two
        Tq = max(1, int(quota))
        ov.set_threads(Tq)
        t0 = time.perf_counter()
        for k in range(args.cpu_warmup + n, args.cpu_warmup + 2 * n):
            step(k)
        vq = n / (time.perf_counter() - t0)
        out["value_reference_thread_policy"] = out["value"]
        out["value_quota_threads"] = vq
        out["sample"] += ("; the container's CFS quota is %.0f CPUs: %d threads give %.1f frames/s on the next %d frames, the "
                          "reference's %d threads %.1f" % (quota, Tq, vq, n, T, out["value"]))
        if vq > out["value"]:
            out["value"], out["cores"] = vq, Tq
        ov.set_threads(T)
    if textured:  # TSDF-only figure of the same port on a shorter sample
        ov1 = O.Volume(res, O.camera_from(cam), O.default_integrator())
        ov1.set_kernel(1 if avx2 else 0)
        ov1.set_threads(T)
        n1 = max(4, n // 4)
        for k in range(4):
            ov1.integrate_frame(h_depth[k % n_unique], h_rgba[k % n_unique], h_pose[k % n_unique])
        t0 = time.perf_counter()
        for k in range(4, 4 + n1):
            ov1.integrate_frame(h_depth[k % n_unique], h_rgba[k % n_unique], h_pose[k % n_unique])
        out["value_tsdf_only"] = n1 / (time.perf_counter() - t0)
    return out


if __name__ == "__main__":
    main()
