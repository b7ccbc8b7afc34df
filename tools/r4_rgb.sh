#!/bin/bash
# host frames as RGBA vs as Frame::rgb (tf_integrate_frame_host_rgb): textured and TSDF-only quick bench lines
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4_rgb
for m in textured tsdf; do
  a=""; [ "$m" = "tsdf" ] && a="--mode tsdf"
  python bench.py $a --steps 200 --warmup 20 --no-pmc --cpu-frames 0 --no-group --repeats 2 > gpurun_out/r4_rgb/$m.json 2> gpurun_out/r4_rgb/$m.err
  python - gpurun_out/r4_rgb/$m.json $m <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get("rgb_host_frames") or {}
print(sys.argv[2], "host rgba %.1f us | host rgb %s us (%s B/frame) | resident %.1f us" % (1e3*d["ms_per_step"], round(1e3*r.get("ms_per_step",0),1) if "ms_per_step" in r else r, r.get("bytes_uploaded_per_frame"), 1e3*d["resident"]["ms_per_step"]))
PY
done
