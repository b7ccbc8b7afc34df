#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 120 python bench.py --steps 200 --warmup 20 --unique-frames 100 --cpu-frames 0 --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value']), round(1e3*d['ms_per_step'],1))"; }
timeout 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for gp in 2 4 8; do run TF_KA_GP=$gp; done
