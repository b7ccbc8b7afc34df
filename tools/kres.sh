#!/bin/bash
# register / scratch / occupancy of every kernel of one source file:  tools/kres.sh tf_kernels.hip [extra hipcc flags]
cd "$(dirname "$0")/../texturefusion_amd/csrc"
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \
  -Rpass-analysis=kernel-resource-usage "$@" -c $src -o /dev/null 2>&1 | \
  awk '/Function Name/{n=$NF} /remark: +VGPRs:/{v=$NF} /SGPRs:/{s=$NF} /ScratchSize/{sc=$NF} /Occupancy/{o=$NF} /LDS Size/{print n, "VGPR", v, "SGPR", s, "scratch", sc, "occ", o, "LDS", $NF}'
