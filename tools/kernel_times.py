#!/usr/bin/env python3
"""Per-kernel HIP-event timings of the call-by-call flow vs the fused unit (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from texturefusion_amd import capi, synth

def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    cam = synth.Camera(); res = np.float32(0.005)
    frames = [synth.room_frame(k, cam, with_quality=False) for k in range(nf)]
    for mode in ("split", "fused"):
        v = capi.Volume(res, cam, max_chunks=1 << 18)
        # warm-up pass so that chunks exist (steady state), then a timed pass
        for rep in range(2):
            if rep == 1:
                v.profile_enable(capi.PROF_NAMES)
            for (d, c, _, p) in frames:
                v.frame_upload(d, c, None)
                if mode == "split":
                    ids, new = v.prepare(p)
                    needs = np.zeros(len(ids), np.uint8)
                    v.integrate(p, ids, needs, 1, True, False)
                    v.finalize(ids, needs, new)
                else:
                    v.integrate_frame(p, True)
            v.sync()
        pr = v.profile_get()
        print(mode, "GP=%s" % os.environ.get("TF_KA_GP", "4"),
              {k: round(1e3 * ms / max(n, 1), 1) for k, (ms, n) in pr.items() if n})
        v.close()

if __name__ == "__main__":
    main()
