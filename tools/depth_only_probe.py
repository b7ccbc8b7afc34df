import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from texturefusion_amd import capi, synth
cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
frames = [synth.room_frame(k, cam, with_quality=False) for k in range(120)]
dd = [torch.from_numpy(f[0]).to(dev) for f in frames]; dc = [torch.from_numpy(f[1]).to(dev) for f in frames]
poses = np.stack([f[3].reshape(12) for f in frames]).astype(np.float32)
for colour in (True, False):
    vol = capi.Volume(res, cam, max_chunks=1 << 19, max_list=1 << 18, max_coarse=1 << 20)
    ptr_c = [x.data_ptr() if colour else 0 for x in dc]
    vol.stream_frames_device([x.data_ptr() for x in dd[:40]], ptr_c[:40], poses[:40])
    vol.sync(); torch.cuda.synchronize(); t = time.perf_counter()
    vol.stream_frames_device([x.data_ptr() for x in dd[40:120]], ptr_c[40:120], poses[40:120])
    vol.sync(); dt = time.perf_counter() - t
    print("colour" if colour else "depth only", "%.1f us per frame" % (1e6 * dt / 80))
    vol.close()
