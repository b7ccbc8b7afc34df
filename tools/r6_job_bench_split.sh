#!/bin/bash
# bench.py after the split of main(): every way it is run
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/job_split; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/drv.json 2> $O/drv.err; echo "driver flags rc=$?"; tail -3 $O/drv.err
timeout 600 python bench.py --mode tsdf --steps 50 --warmup 10 --no-pmc --cpu-frames 0 --no-group > $O/tsdf.json 2> $O/tsdf.err; echo "tsdf rc=$?"; tail -3 $O/tsdf.err
timeout 600 python bench.py --gpus 1 --force-exchange --steps 50 --warmup 10 --no-pmc --cpu-frames 0 --no-group > $O/fx.json 2> $O/fx.err; echo "force-exchange rc=$?"; tail -3 $O/fx.err
timeout 600 python bench.py --gpus 1 --force-exchange --mode tsdf --steps 50 --warmup 10 --no-pmc --cpu-frames 0 --no-group > $O/fxt.json 2> $O/fxt.err; echo "force-exchange tsdf rc=$?"; tail -3 $O/fxt.err
timeout 600 python bench.py --staged-host-frames --steps 50 --warmup 10 --no-pmc --cpu-frames 0 --no-group --no-side > $O/st.json 2> $O/st.err; echo "staged rc=$?"; tail -3 $O/st.err
timeout 900 python -m pytest tests/test_gpu_bench_multi.py -m gpu -x -q > $O/multi.log 2>&1; echo "multi rc=$?"; tail -3 $O/multi.log
python - <<'PY'
import json
for n in ('drv','tsdf','fx','fxt','st'):
    try:
        d=json.loads(open('gpurun_out/job_split/%s.json'%n).read().strip().splitlines()[-1])
        print(n, round(d['value'],1), round(1e3*d['ms_per_step'],2), sorted(k for k in d.keys() if k not in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data','config')))
    except Exception as e: print(n,'FAILED',e)
PY
