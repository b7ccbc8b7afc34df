#!/bin/bash
# SQ counters of k_integrate for one ablation setting: tools/pmc.sh <dbg> <gp> <tag>
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/pmc3
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --unique-frames 40 --cpu-frames 0 --no-roofline"
export TF_KA_DBG=$1 TF_KA_GP=$2
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O -o $3a -- $B >/dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $O -o $3b -- $B >/dev/null 2>&1
