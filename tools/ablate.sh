#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 120 python bench.py --steps 120 --warmup 20 --unique-frames 70 --cpu-frames 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value']), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"; }
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for gp in 1 2 4 8; do run TF_KA_GP=$gp; done
run TF_KA_GP=2 TF_KA_DBG=128
run TF_KA_GP=2 TF_KA_DBG=64
