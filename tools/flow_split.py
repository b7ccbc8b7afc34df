import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from texturefusion_amd import capi, synth
cam = synth.Camera(); res = np.float32(0.005); dev = torch.device("cuda", 0)
N = 200
fr = [synth.room_frame(k, cam, with_quality=False) for k in range(N)]
dd = [torch.from_numpy(f[0]).to(dev) for f in fr]; dc = [torch.from_numpy(f[1]).to(dev) for f in fr]
poses = np.stack([f[3].reshape(12) for f in fr]).astype(np.float32)
pinv = np.stack([synth.pose_inverse16(f[3]) for f in fr]).astype(np.float32)
vol = capi.Volume(res, cam, max_chunks=1 << 19, max_list=1 << 18, max_coarse=1 << 20)
T = dict(unit=0.0, compress=0.0, cache=0.0, gen=0.0, atlas=0.0)
def one(g, acc):
    k0 = (7 * g) % N; loc = [(k0 + 1 + i) % N for i in range(6)]
    grp = capi.Volume.unit_group(1000 + g, (dd[k0].data_ptr(), dc[k0].data_ptr(), 0, poses[k0]), [(dd[k].data_ptr(), poses[k]) for k in loc])
    t0 = time.perf_counter(); vol.keyframe_unit(fresh=grp, moved=[], texture=False); t1 = time.perf_counter()
    upd = vol.compress_meshes(); t2 = time.perf_counter()
    vol.keyframe_cache_device(1000 + g, dc[k0].data_ptr(), dd[k0].data_ptr(), stride=4, pose_inv16=pinv[k0]); t3 = time.perf_counter()
    vol.generate_patches(upd, np.full(len(upd), 1000 + g, np.int32)); t4 = time.perf_counter()
    vol.update_atlas(upd); t5 = time.perf_counter()
    if g >= 8: vol.keyframe_release(1000 + g - 8)
    if acc:
        T["unit"] += t1 - t0; T["compress"] += t2 - t1; T["cache"] += t3 - t2; T["gen"] += t4 - t3; T["atlas"] += t5 - t4
    return len(upd)
for g in range(28): one(g, False)
vol.sync(); n = 0
for g in range(28, 48): n += one(g, True)
print({k: round(1e6 * v / 20, 1) for k, v in T.items()}, "chunksToUpdate per kf", n / 20)
vol.close()
