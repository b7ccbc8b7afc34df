"""Probe: host frames out of registered caller buffers on a SECOND volume of the same process.
    python tools/two_volumes_probe.py one | two | two_torchstream      (KEEP_A=1: the first volume stays open; CLOSE_FIRST=1: closed before the second is made)
Measured (MI355X, ROCm 7.2): one volume 65 us of upload wait per 640x480 frame; a second volume behind a DESTROYED first one 230 us;
with the first one kept open 64 us.  GPU_MAX_HW_QUEUES and HSA_ENABLE_SDMA=0 do not change it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from texturefusion_amd import capi, synth
mode = sys.argv[1]
cam = synth.Camera()
n = 40
fr = [synth.room_frame(k, cam, with_quality=False) for k in range(n)]
h_depth = np.stack([f[0] for f in fr]); h_rgba = np.stack([f[1] for f in fr])
poses = np.stack([np.asarray(f[3], np.float32).reshape(12) for f in fr]); pinv = np.stack([synth.pose_inverse16(f[3]) for f in fr]).astype(np.float32)
dev = torch.device("cuda", 0)
res = np.float32(0.005)
def mk(stream=None):
    return capi.Volume(res, cam, max_chunks=1 << 17, max_list=1 << 17, max_coarse=1 << 20, device=0, stream=stream)
if mode in ("two", "two_torchstream"):
    s_main = torch.cuda.Stream(device=dev) if mode == "two_torchstream" else None
    a = mk(s_main.cuda_stream if s_main else None)
    a.host_register(h_depth); a.host_register(h_rgba)
    for k in range(10):
        a.integrate_frame_host(h_depth[k], h_rgba[k], poses[k], pinv[k], k)
    a.sync()
    if os.environ.get("CLOSE_FIRST"):
        a.close()
    b = mk()
    b.host_register(h_depth); b.host_register(h_rgba)
    if not os.environ.get("CLOSE_FIRST") and not os.environ.get("KEEP_A"):
        a.close()
else:
    b = mk()
    b.host_register(h_depth); b.host_register(h_rgba)
if os.environ.get("SLEEP_AFTER"):
    time.sleep(float(os.environ["SLEEP_AFTER"]))  # (does the slow-down pass with time?)
for rep in range(int(os.environ.get("REPS", "3"))):
    b.host_frame_times(reset=True)
    t0 = time.perf_counter()
    for k in range(n):
        b.integrate_frame_host(h_depth[k], h_rgba[k], poses[k], pinv[k], k)
    b.sync()
    dt = time.perf_counter() - t0
    ph = b.host_frame_times(reset=True)
    print(mode, "rep", rep, "us/frame %.1f" % (1e6 * dt / n), "wait_upload %.1f" % ph["wait_for_upload_us"])
b.close()
