#!/bin/bash
# On the GPU box: bench every variants/*.so (default + steady-state regime) for a few K-A grid sizes.
for f in variants/*.so; do
  cp $f texturefusion_amd/libtexfusion_hip.so
  for kb in ${KBS:-2048}; do
    d=$(TF_KA_BLOCKS=$kb python bench.py --steps 300 --warmup 300 --no-roofline --cpu-frames 0 2>&1 | tail -1 | grep -o "\"ms_per_step.: [0-9.]*" | grep -o "[0-9.]*$")
    s=$(TF_KA_BLOCKS=$kb python bench.py --steps 400 --warmup 400 --unique-frames 60 --no-roofline --cpu-frames 0 2>&1 | tail -1 | grep -o "\"ms_per_step.: [0-9.]*" | grep -o "[0-9.]*$")
    echo "$(basename $f .so) ka_blocks=$kb: default $d ms  steady $s ms"
  done
done
