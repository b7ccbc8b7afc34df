#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 120 python bench.py --steps 200 --warmup 20 --unique-frames 100 --cpu-frames 0 --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value']), round(1e3*d['ms_per_step'],1))"; }
TF_KA_DBG=256 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
for dbg in 0 256 1 128 136 64; do run TF_KA_GP=2 TF_KA_DBG=$dbg; done
for dbg in 0 256; do run TF_KA_GP=4 TF_KA_DBG=$dbg; done
