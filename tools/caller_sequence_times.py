"""Host time per call of the product's own order with the view selection on the host (MobileFusion::tsdfFusion): the unit
without its texture stage, then tf_compress_meshes / tf_keyframe_cache_device / tf_generate_patches / tf_update_atlas.
   PYTHONPATH=. python tools/caller_sequence_times.py"""
import time
import numpy as np
import torch
from texturefusion_amd import capi, synth

cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
ORBIT, stride, n_local = 200, 7, 6
fr = [synth.room_frame(k, cam, with_quality=False) for k in range(ORBIT)]
dd = [torch.from_numpy(f[0]).to(dev) for f in fr]; dc = [torch.from_numpy(f[1]).to(dev) for f in fr]
poses = np.stack([f[3].reshape(12) for f in fr]).astype(np.float32)
pinv = np.stack([synth.pose_inverse16(f[3]) for f in fr]).astype(np.float32)
vol = capi.Volume(res, cam, max_chunks=1 << 19, mesh_blocks=1 << 17, max_list=1 << 18, max_coarse=1 << 20)
acc = {}

def lap(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1

def seq(g, timed):
    k0 = (stride * g) % ORBIT
    loc = [(k0 + 1 + i) % ORBIT for i in range(n_local)]
    grp = capi.Volume.unit_group(1000 + g, (dd[k0].data_ptr(), dc[k0].data_ptr(), 0, poses[k0]), [(dd[k].data_ptr(), poses[k]) for k in loc])
    t = time.perf_counter()
    vol.keyframe_unit(fresh=grp, moved=[], texture=False); t = lap("unit(texture=0) enqueue", t) if timed else time.perf_counter()
    upd = vol.compress_meshes(); t = lap("compress_meshes (waits for the unit)", t) if timed else time.perf_counter()
    vol.keyframe_cache_device(1000 + g, dc[k0].data_ptr(), dd[k0].data_ptr(), stride=4, pose_inv16=pinv[k0]); t = lap("keyframe_cache_device", t) if timed else time.perf_counter()
    lab = np.full(len(upd), 1000 + g, np.int32)
    vol.generate_patches(upd, lab); t = lap("generate_patches", t) if timed else time.perf_counter()
    vol.update_atlas(upd); t = lap("update_atlas", t) if timed else time.perf_counter()
    if g >= 8:
        vol.keyframe_release(1000 + g - 8)
    return len(upd)

for g in range(28):
    seq(g, False)
vol.sync()
N = 20
t0 = time.perf_counter(); n = 0
for g in range(28, 28 + N):
    n += seq(g, True)
vol.sync(); dt = time.perf_counter() - t0
print("per keyframe %.1f us, chunksToUpdate %.0f" % (1e6 * dt / N, n / N), {k: round(1e6 * v / N, 1) for k, v in acc.items()})
vol.close()
