"""tf_prepare's launches under HIP events: k_bbox / k_select / k_scan / k_acquire microseconds per call (room 640x480, hall 1280x960).
   PYTHONPATH=. python tools/scan_time.py   (TF_LIB=variants/x.so for an A/B)"""
import numpy as np
from texturefusion_amd import capi, synth

for name, cam, frame, kw in (("room", synth.Camera(), synth.room_frame, {}),
                             ("hall", synth.Camera.hires(), synth.room_frame, {"half": (4.0, 3.0, 4.0), "radius": 0.6})):
    vol = capi.Volume(np.float32(0.005), cam, max_chunks=1 << 19, max_list=1 << 20, max_coarse=1 << 22, mesh_blocks=1 << 14)
    frames = [frame(10 * k, cam, with_quality=False, **kw) for k in range(6)]
    for rep in range(2):
        if rep == 1:
            vol.profile_enable(("bbox", "select", "scan", "emit"))
        n = 0
        for _ in range(4):
            for d, c, _, P in frames:
                vol.frame_upload(d, c, None)
                ids, new = vol.prepare(P)
                n += 1
    pr = vol.profile_get(reset=True)
    print(name, "chunks listed %d" % len(ids), {k: round(1e3 * pr[k][0] / max(1, pr[k][1]), 2) for k in ("bbox", "select", "scan", "emit")}, "(us per launch, event pair included)")
    vol.close()
